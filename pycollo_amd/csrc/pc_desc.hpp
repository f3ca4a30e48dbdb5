// Host-side glue between the C ABI's problem descriptor (include/pycollo_amd.h) and the structure builder
// (pc_pattern.hpp): descriptor -> pcp::Problem, and the LDS sizing of a bulk workgroup.  Plain C++ (no HIP): the
// library's pc_create uses it, and tests/c/pattern_sanitize.cpp compiles the same code with
// -fsanitize=address,undefined on the CPU (sanitizer builds are for the host side only).
#pragma once

#include <algorithm>
#include <stdexcept>

#include "../../include/pycollo_amd.h"
#include "pc_args.h"
#include "pc_pattern.hpp"

namespace pcp {

// replaces the argument checking and copying at the top of Casadi.generate_nlp_function_callables' inputs
// (pycollo/backend.py:632-816: counts, slices and masks of every phase)
inline void from_desc(const pc_problem_desc& d, Problem& Q) {
  if (d.n_phases < 1 || d.n_phases > PC_MAX_PHASES) throw std::runtime_error("n_phases must be in [1, 8]");
  if (!d.phases) throw std::runtime_error("phase descriptors missing");
  Q.n_s = d.n_s;
  Q.n_b = d.n_b;
  Q.ph.resize(d.n_phases);
  for (int ip = 0; ip < d.n_phases; ++ip) {
    const pc_phase_desc& s = d.phases[ip];
    auto& P = Q.ph[ip];
    P.n_y = s.n_y; P.n_u = s.n_u; P.n_q = s.n_q; P.n_p = s.n_p;
    P.t_free[0] = s.t0_free != 0; P.t_free[1] = s.tF_free != 0;
    P.t_fixed[0] = s.t0_fixed; P.t_fixed[1] = s.tF_fixed;
    P.K = s.K;
    if (s.K < 1 || !s.n_k || !s.h_k) throw std::runtime_error("phase mesh arrays missing");
    P.n_k.assign(s.n_k, s.n_k + s.K);
    P.h_k.assign(s.h_k, s.h_k + s.K);
    P.jac_row.assign(s.jac_row, s.jac_row + s.n_jac);
    P.jac_col.assign(s.jac_col, s.jac_col + s.n_jac);
    P.hess_row.assign(s.hess_row, s.hess_row + s.n_hess);
    P.hess_col.assign(s.hess_col, s.hess_col + s.n_hess);
    if (s.n_w > 0) {
      if (!s.w_kind || !s.w_idx) throw std::runtime_error("phase parameter arrays missing");
      P.wkind.assign(s.w_kind, s.w_kind + s.n_w);
      P.widx.assign(s.w_idx, s.w_idx + s.n_w);
    }
    P.bulk_kernel = s.bulk_kernel ? s.bulk_kernel : "";
    P.eval_ops = s.eval_ops;
    P.compiled_order = s.compiled_order;
    if (s.n_spec < 0 || s.n_spec > 4) throw std::runtime_error("n_spec must be in [0, 4]");
    if (s.n_spec > 0 && s.compiled_order != 0) throw std::runtime_error("a phase is either compiled for one order or mixed");
    P.spec_orders.assign(s.spec_orders, s.spec_orders + s.n_spec);
    for (int n : P.spec_orders)
      if (n < 2 || n > PC_MAX_ORDER) throw std::runtime_error("specialised section order outside [2, 20]");
    if (s.n_fixed_tiles > 0) {
      if (!s.fixed_tile_k0) throw std::runtime_error("fixed tile table missing");
      P.fixed_tile_k0.assign(s.fixed_tile_k0, s.fixed_tile_k0 + s.n_fixed_tiles + 1);
      if (s.fixed_tile_order) P.fixed_tile_order.assign(s.fixed_tile_order, s.fixed_tile_order + s.n_fixed_tiles);
    }
    for (int k = 0; k < s.K; ++k)
      if (s.compiled_order > 0 && s.n_k[k] != s.compiled_order)
        throw std::runtime_error("phase kernel was compiled for a fixed section order that the mesh does not have");
  }
  Q.point_phase.assign(d.point_phase, d.point_phase + d.n_point);
  Q.point_kind.assign(d.point_kind, d.point_kind + d.n_point);
  Q.point_idx.assign(d.point_idx, d.point_idx + d.n_point);
  Q.jgrad_col.assign(d.jgrad_col, d.jgrad_col + d.n_jgrad);
  Q.bjac_row.assign(d.bjac_row, d.bjac_row + d.n_bjac);
  Q.bjac_col.assign(d.bjac_col, d.bjac_col + d.n_bjac);
  Q.pthess_row.assign(d.pthess_row, d.pthess_row + d.n_pthess);
  Q.pthess_col.assign(d.pthess_col, d.pthess_col + d.n_pthess);
}

// LDS of a bulk workgroup: pc_args.h::lds_plan, the same function the kernels carve their LDS with.
// entries of the packed scal | goff | hoff table of the phase's model (pc_kernels.hpp: St::NSCAL + NFN + 3 NZ + NS NZ)
inline int phase_tab_doubles(const pcp::Phase& P) {
  const int NZ = P.n_z, NS = P.n_v - P.n_z, NFN = P.n_y + P.n_p + P.n_q;
  const int NSCAL = 2 * NZ + 3 * P.n_q + 4 + 2 * NS + P.n_y + P.n_p;
  return NSCAL + NFN + 3 * NZ + NS * NZ;
}
inline int phase_nfs(const pcp::Phase& P) {
  int nfs = 0;
  for (int a = 0; a < P.n_y; ++a)
    for (int l = 0; l < P.n_w; ++l) nfs += P.dep(a, P.n_z + l) ? 1 : 0;
  return nfs;
}
// compiled_order > 0: the phase's kernel is order-specialised and stages no section tables
inline int phase_lds_bytes(const pcp::Phase& P, int TB, int qa_total, int qw_total, int lds_out, bool mesh_tables,
                           bool mixed = false) {
  return 8 * lds_plan(TB, qa_total, qw_total, P.n_y, phase_nfs(P), P.nred, lds_out, phase_tab_doubles(P), mesh_tables, mixed).total;
}

// doubles of a defect row of state a in a section of n nodes: D n + C (pc_kernels.hpp S<M>::D / C)
inline int phase_row_len(const pcp::Phase& P, int a, int n) {
  int Da = 0, Ca = P.n_t + (P.dep(a, a) ? 0 : 2);
  for (int b = 0; b < P.n_z; ++b) Da += P.dep(a, b) ? 1 : 0;
  for (int l = 0; l < P.n_w; ++l) Ca += (!P.is_t(l) && P.dep(a, P.n_z + l)) ? 1 : 0;   // time parameters add to the t columns
  return Da * n + Ca;
}
inline int phase_max_row_len(const pcp::Phase& P, int n) {
  int m = 0;
  for (int a = 0; a < P.n_y; ++a) m = std::max(m, phase_row_len(P, a, n));
  return m;
}
// staging doubles of the defect-Jacobian block of `rows` rows of n-node sections in a body compiled for that order:
// the body stages it in row groups (pc_args.h::pc_row_passes), a piece of every section at a time
inline int order_body_defect_out(const pcp::Phase& P, int rows, int n, bool row_groups = true) {
  if (!row_groups) return rows * phase_max_row_len(P, n);
  const int nq = (rows + n - 2) / (n - 1);
  return nq * pc_row_group(n, phase_max_row_len(P, n)) * phase_max_row_len(P, n);
}
// doubles of the output staging buffer of one phase: the longest CSR run a tile emits in one piece, for tiles of at
// most `rows` defect rows per state and `nodes` nodes
// (row_groups = false: the four-wave per-replica kernels stage a state's block in one piece, pc::bulk NPASS)
inline int phase_lds_out(const pcp::Phase& P, int rows, int nodes, bool row_groups = true) {
  int nmax = 0;
  for (int k = 0; k < P.K; ++k) nmax = std::max(nmax, (int)P.n_k[k]);
  int out = 0;
  if (P.compiled_order > 0) {
    out = order_body_defect_out(P, rows, P.compiled_order, row_groups);
  } else {
    for (int a = 0; a < P.n_y; ++a) out = std::max(out, phase_row_len(P, a, nmax) * rows);
  }
  for (int m = 0; m < P.n_p; ++m) {
    int R = 0;
    for (int c = 0; c < P.n_v; ++c) R += P.dep(P.n_y + m, c) ? 1 : 0;
    out = std::max(out, R * nodes);
  }
  for (int b = 0; b < P.n_z; ++b) out = std::max(out, P.hrow_count(b) * nodes);
  return out;
}
// the same for one actual tile, sections [ka, kb): every row counted with its own section's order
inline int tile_lds_out(const pcp::Phase& P, int ka, int kb, int order = 0, bool row_groups = true) {
  int rows = 0;
  for (int k = ka; k < kb; ++k) rows += P.n_k[k] - 1;
  const int nodes = rows + 1;
  int out = 0;
  if (order > 0) {   // order-pure tile run by the body of its order: staged in row groups
    out = order_body_defect_out(P, rows, order, row_groups);
  } else {
    for (int a = 0; a < P.n_y; ++a) {
      int len = 0;
      for (int k = ka; k < kb; ++k) len += (P.n_k[k] - 1) * phase_row_len(P, a, P.n_k[k]);
      out = std::max(out, len);
    }
  }
  for (int m = 0; m < P.n_p; ++m) {
    int R = 0;
    for (int c = 0; c < P.n_v; ++c) R += P.dep(P.n_y + m, c) ? 1 : 0;
    out = std::max(out, R * nodes);
  }
  for (int b = 0; b < P.n_z; ++b) out = std::max(out, P.hrow_count(b) * nodes);
  return out;
}
// mixed build: staging doubles of the phase's largest tile, and which kinds of tile body the phase's tiles run
inline int phase_lds_out_tiles(const pcp::Phase& P, bool* any_pure = nullptr, bool* any_generic = nullptr, bool row_groups = true) {
  int out = 0;
  for (size_t i = 0; i + 1 < P.tile_k0.size(); ++i) {
    const bool pure = i < P.tile_order.size() && P.tile_order[i] > 0;
    out = std::max(out, tile_lds_out(P, P.tile_k0[i], P.tile_k0[i + 1], pure ? P.tile_order[i] : 0, row_groups));
    if (any_pure && pure) *any_pure = true;
    if (any_generic && !pure) *any_generic = true;
  }
  return out;
}
// mixed build: the row caps (Phase::cap_rows / mix_cap_rows) under which a workgroup of W staging regions needs at most
// `budget` bytes of LDS, for tiles of at most TB nodes
inline void phase_set_caps(pcp::Phase& P, int TB, int W, int qa_total, int qw_total, int budget) {
  int nmax = 2;
  for (int k = 0; k < P.K; ++k) nmax = std::max(nmax, (int)P.n_k[k]);
  auto out_rows = [&](int rows, int n, bool pure) {   // phase_lds_out with every section of order n
    int out = pure ? order_body_defect_out(P, rows, n) : phase_max_row_len(P, n) * rows;
    for (int m = 0; m < P.n_p; ++m) {
      int R = 0;
      for (int c = 0; c < P.n_v; ++c) R += P.dep(P.n_y + m, c) ? 1 : 0;
      out = std::max(out, R * (rows + 1));
    }
    for (int b = 0; b < P.n_z; ++b) out = std::max(out, P.hrow_count(b) * (rows + 1));
    return out;
  };
  for (int n = 2; n <= PC_MAX_ORDER; ++n) {
    int rows = ((TB - 1) / (n - 1)) * (n - 1);
    while (rows > n - 1 && phase_lds_bytes(P, TB, qa_total, qw_total, W * out_rows(rows, n, true), false, true) > budget) rows -= n - 1;
    P.cap_rows[n] = rows;
  }
  int rows = TB - 1;
  while (rows > nmax - 1 && phase_lds_bytes(P, TB, qa_total, qw_total, W * out_rows(rows, nmax, false), true, true) > budget) --rows;
  P.mix_cap_rows = rows;
}

// defect rows of the largest tile a phase is cut into (after pcp::build_all)
inline int phase_max_tile_rows(const pcp::Phase& P) {
  int rows = 0;
  for (size_t i = 0; i + 1 < P.tile_k0.size(); ++i) rows = std::max(rows, (int)(P.sec_s[P.tile_k0[i + 1]] - P.sec_s[P.tile_k0[i]]));
  return rows;
}


}  // namespace pcp
