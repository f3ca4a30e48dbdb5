// Host-side structure builder: mesh prefix tables, NLP layout, CSR patterns of G and H, and the
// producer-slot tables the kernels write through (kernels never search).
//
// Pattern rules (the reference obtains the same sets from
// ca.jacobian's sparsity, pycollo/backend.py:1747-1761, or by NaN-probing, pycollo/iteration.py:928-1055):
//   G defect row (a,k,j): for every z_b with df_a/dz_b != 0 the whole section run of n_k columns;
//                         otherwise for b == a the two D entries (s_k, s_k+j); q with df_a/dq != 0; free
//                         times; s with df_a/ds != 0.          (compiled.py:305-334)
//   G path row (m,i):     (b,i) for dp_m/dz_b != 0; the parameters p_m depends on.   (compiled.py:336-355)
//   G integral row m:     (b,0..N-1) for dg_m/dz_b != 0; q_m and the q it depends on; free times; s.
//                                                              (compiled.py:357-379)
//   G endpoint row:       the point variables b_r depends on.  (compiled.py:381-403)
//   H (lower triangle):   node bands (flag 1), parameter strips (flag 2), parameter-parameter sums (flag 3),
//                         endpoint block.                      (compiled.py:479-500, sparse.py:48-61)
// "Parameters" w of a phase's node functions: everything in f, p, g that is not a node variable, in x order
// [q | free t | s] (the live reference leaves q, t0, tF, s global inside them, backend.py:1526-1539).  The free
// times always have columns in the defect and integral rows and strips over the z that f or g depend on (the
// stretch factor); a dependence of f, p, g on them adds to those entries.
// Pure C++ (no HIP): unit-testable on a CPU-only machine.
#pragma once

#include <algorithm>
#include <cstdint>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "pc_args.h"

namespace pcp {

struct Phase {
  // description
  int n_y = 0, n_u = 0, n_q = 0, n_p = 0;
  int n_w = 0;                        // parameters of the node functions
  std::vector<int32_t> wkind, widx;   // [n_w] 0 static parameter / 1 integral / 2 free time; index within the kind
  bool t_free[2] = {false, false};
  double t_fixed[2] = {0, 0};
  int K = 0;
  std::vector<int32_t> n_k;
  std::vector<double> h_k;
  std::vector<int32_t> jac_row, jac_col, hess_row, hess_col;
  std::string bulk_kernel;
  int eval_ops = 0;             // launch-shape hint (pc_phase_desc::eval_ops)
  int compiled_order = 0;       // > 0: the phase's kernel is specialised for sections of exactly that many nodes
  // mixed build: the section orders the phase's kernels carry a body for, next to the any-order body (empty: not mixed)
  std::vector<int32_t> spec_orders;
  int cap_rows[PC_MAX_ORDER + 1] = {};   // defect rows an order-pure tile of order n may hold (0: tile capacity - 1)
  int mix_cap_rows = 0;                  // ... and a tile of several orders (0: tile capacity - 1)
  int min_run_rows = 24;                 // a run of equal sections gets tiles of its own from this many rows
  // derived
  int n_z = 0, n_t = 0, n_fn = 0, n_v = 0, N = 0;
  std::vector<int32_t> sec_s;   // [K+1]
  std::vector<int64_t> sec_E;   // [K+1]
  std::vector<int8_t> jmask;    // [n_fn][n_v]
  int64_t x_off = 0, q_off = 0, t_off = 0, c_off = 0, c_path_off = 0, c_int_off = 0;
  int ocp_x_off = 0, ocp_c_off = 0;
  // kernel tables
  std::vector<int64_t> goff, hoff, hslot0, hslotN, hsum_slot;
  std::vector<int32_t> hsum_local;   // hsum_slot as indices into Problem::tail_owned
  int64_t gq_base[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<int32_t> tile_k0;
  std::vector<int32_t> tile_order;   // [n_tiles] mixed build: the order whose body runs the tile, 0 = the any-order body
  std::vector<int32_t> fixed_tile_k0, fixed_tile_order;   // the caller's tile table (pc_phase_desc::fixed_tile_k0), else empty
  int nred = 0;

  bool dep(int r, int c) const { return jmask[(size_t)r * n_v + c] != 0; }
  // does sum_r mu_r dF_r/dv_c (r over f and g rows, the ones the stretch factor multiplies) have structure?
  bool tz(int c) const {
    for (int r = 0; r < n_fn; ++r)
      if (!(r >= n_y && r < n_y + n_p) && dep(r, c)) return true;
    return false;
  }
  bool hdep(int r, int c) const {
    for (size_t e = 0; e < hess_row.size(); ++e)
      if (hess_row[e] == r && hess_col[e] == c) return true;
    return false;
  }
  bool is_t(int l) const { return wkind[l] == 2; }
  // parameter index of free time jt, or -1 when the node functions do not depend on it
  int tpar(int jt) const {
    for (int l = 0; l < n_w; ++l)
      if (wkind[l] == 2 && widx[l] == jt) return l;
    return -1;
  }
  // (t, z_b) strip: through the stretch factor, or through a second partial d2F/dt dz_b
  bool tzx(int b) const {
    if (tz(b)) return true;
    for (int l = 0; l < n_w; ++l)
      if (wkind[l] == 2 && hdep(n_z + l, b)) return true;
    return false;
  }
  int64_t wcol(int l, int64_t s_off) const {
    return wkind[l] == 0 ? s_off + widx[l] : (wkind[l] == 1 ? q_off + widx[l] : t_off + widx[l]);
  }
  int wocp(int l, int ocp_s_off) const {
    return wkind[l] == 0 ? ocp_s_off + widx[l] : ocp_x_off + n_z + (wkind[l] == 1 ? widx[l] : n_q + widx[l]);
  }
  int hrow_count(int b) const {
    int n = 0;
    for (size_t e = 0; e < hess_row.size(); ++e) n += (hess_row[e] == b && hess_col[e] < n_z) ? 1 : 0;
    return n;
  }
};

struct Problem {
  std::vector<Phase> ph;
  int n_s = 0, n_b = 0;
  std::vector<int32_t> point_phase, point_kind, point_idx;
  std::vector<int32_t> jgrad_col, bjac_row, bjac_col, pthess_row, pthess_col;
  // layout
  int64_t num_x = 0, num_c = 0, s_off = 0, c_end_off = 0;
  int num_ocp_x = 0, num_ocp_c = 0, ocp_s_off = 0, ocp_c_end_off = 0;
  std::vector<int64_t> point_x;
  std::vector<int32_t> point_ocp;
  // patterns
  std::vector<int64_t> g_indptr, h_indptr;
  std::vector<int32_t> g_row, g_col, h_row, h_col;
  int64_t g_end_base = 0;
  std::vector<int64_t> tail_owned, pt_hslot;
  std::vector<int32_t> pt_hlocal;    // pt_hslot as indices into tail_owned, -1 for slots of the bulk's edge nodes
};

inline void fail(const std::string& msg) { throw std::runtime_error(msg); }

inline void finalize_phase_tables(Phase& P, int n_s) {
  P.n_z = P.n_y + P.n_u;
  P.n_t = (P.t_free[0] ? 1 : 0) + (P.t_free[1] ? 1 : 0);
  P.n_fn = P.n_y + P.n_p + P.n_q;
  if (P.wkind.empty()) {   // the problem's static parameters, nothing else
    P.wkind.assign(n_s, 0);
    P.widx.resize(n_s);
    for (int l = 0; l < n_s; ++l) P.widx[l] = l;
  }
  P.n_w = (int)P.wkind.size();
  if (P.widx.size() != P.wkind.size()) fail("parameter kind / index arrays differ in length");
  for (int l = 0; l < P.n_w; ++l) {
    const int k = P.wkind[l], i = P.widx[l];
    if (k < 0 || k > 2 || i < 0 || i >= (k == 0 ? n_s : (k == 1 ? P.n_q : P.n_t))) fail("parameter out of range");
    // x order: integrals, then times, then static parameters, ascending inside a kind
    if (l > 0) {
      const int k0 = P.wkind[l - 1], o0 = k0 == 0 ? 2 : k0 - 1, o1 = k == 0 ? 2 : k - 1;
      if (o0 > o1 || (o0 == o1 && P.widx[l - 1] >= i)) fail("parameters must be listed in x order");
    }
  }
  P.n_v = P.n_z + P.n_w;
  const int n_w = P.n_w;
  if (P.K < 1) fail("a phase needs at least one mesh section");
  if (P.n_q > 8) fail("at most 8 integrals per phase are supported");
  if ((int)P.n_k.size() != P.K || (int)P.h_k.size() != P.K) fail("mesh arrays must have K entries");
  P.sec_s.assign(P.K + 1, 0);
  P.sec_E.assign(P.K + 1, 0);
  for (int k = 0; k < P.K; ++k) {
    const int n = P.n_k[k];
    if (n < 2 || n > PC_MAX_ORDER) fail("section order outside [2, 20]");
    P.sec_s[k + 1] = P.sec_s[k] + (n - 1);
    P.sec_E[k + 1] = P.sec_E[k] + (int64_t)(n - 1) * n;
  }
  P.N = P.sec_s[P.K] + 1;
  P.jmask.assign((size_t)P.n_fn * P.n_v, 0);
  for (size_t e = 0; e < P.jac_row.size(); ++e) {
    const int r = P.jac_row[e], c = P.jac_col[e];
    if (r < 0 || r >= P.n_fn || c < 0 || c >= P.n_v) fail("jacobian mask entry out of range");
    if (e > 0 && !(P.jac_row[e - 1] < r || (P.jac_row[e - 1] == r && P.jac_col[e - 1] < c)))
      fail("jacobian mask entries must be sorted by (row, col)");
    P.jmask[(size_t)r * P.n_v + c] = 1;
  }
  for (size_t e = 0; e < P.hess_row.size(); ++e) {
    const int r = P.hess_row[e], c = P.hess_col[e];
    if (r < 0 || r >= P.n_v || c < 0 || c > r) fail("hessian mask entry must be lower-triangular and in range");
    if (e > 0 && !(P.hess_row[e - 1] < r || (P.hess_row[e - 1] == r && P.hess_col[e - 1] < c)))
      fail("hessian mask entries must be sorted by (row, col)");
  }
  // per-tile partial sums with structure, counted as pc::S<M>::rqs / rts / rss count them
  P.nred = P.n_q;
  for (int m = 0; m < P.n_q; ++m)
    for (int l = 0; l < n_w; ++l) P.nred += P.dep(P.n_y + P.n_p + m, P.n_z + l) ? 1 : 0;
  for (int l = 0; l < n_w; ++l) P.nred += (P.n_t > 0 && P.tz(P.n_z + l)) ? 1 : 0;
  for (size_t e = 0; e < P.hess_row.size(); ++e) P.nred += P.hess_col[e] >= P.n_z ? 1 : 0;
}

// Mixed build (Phase::spec_orders): tiles are cut at order changes where the orders come in runs, so that a tile of
// equal sections runs the body compiled for its order.  ph refinement leaves exactly such meshes: a section that is
// subdivided becomes k sections of the minimum order and a merged stretch likewise (pycollo/mesh_refinement.py:
// 252-321), the others keep individual orders.  A run of at least min_run_rows rows of an order with a body is cut into
// tiles of its own (nearly equal section counts, at most cap_rows[n] rows); what lies between such runs is cut greedily
// into tiles of at most mix_cap_rows rows -- any of those that happens to hold one order with a body is marked as such.
inline void build_tiles_mixed(Phase& P, int TB) {
  auto has_body = [&](int n) { return std::find(P.spec_orders.begin(), P.spec_orders.end(), n) != P.spec_orders.end(); };
  auto cap = [&](int n) {
    int c = P.cap_rows[n] > 0 ? std::min(P.cap_rows[n], TB - 1) : TB - 1;
    return std::max(n - 1, (c / (n - 1)) * (n - 1));
  };
  const int mix_cap = P.mix_cap_rows > 0 ? std::min(P.mix_cap_rows, TB - 1) : TB - 1;
  P.tile_k0.clear();
  P.tile_order.clear();
  P.tile_k0.push_back(0);
  auto close_tile = [&](int k_end) {   // the open tile ends before section k_end
    const int k_begin = P.tile_k0.back();
    if (k_end <= k_begin) return;
    int n = P.n_k[k_begin];
    for (int k = k_begin; k < k_end; ++k) n = (P.n_k[k] == n) ? n : 0;
    if (n > 0 && !(has_body(n) && (k_end - k_begin) * (n - 1) <= cap(n))) n = 0;
    P.tile_k0.push_back(k_end);
    P.tile_order.push_back(n);
  };
  auto greedy = [&](int ka, int kb) {   // sections [ka, kb) between two qualifying runs
    int rows = 0;
    for (int k = ka; k < kb; ++k) {
      const int r = P.n_k[k] - 1;
      if (rows + r > mix_cap && rows > 0) {
        close_tile(k);
        rows = 0;
      }
      rows += r;
    }
    close_tile(kb);
  };
  int pend = 0, k = 0;
  while (k < P.K) {
    const int n = P.n_k[k];
    int e = k;
    while (e < P.K && P.n_k[e] == n) ++e;
    if (has_body(n) && (e - k) * (n - 1) >= P.min_run_rows) {
      greedy(pend, k);
      const int spt = cap(n) / (n - 1), nt = (e - k + spt - 1) / spt;
      int at = k;
      for (int i = 0; i < nt; ++i) {
        at += (e - k) / nt + (i < (e - k) % nt ? 1 : 0);
        close_tile(at);
      }
      pend = e;
    }
    k = e;
  }
  greedy(pend, P.K);
}

inline void build_tiles(Phase& P, int TB) {
  if (!P.fixed_tile_k0.empty()) {   // the caller's tiles, checked
    const auto& t = P.fixed_tile_k0;
    if (t.size() < 2 || t.front() != 0 || t.back() != P.K) fail("fixed tile table must run from section 0 to K");
    auto has_body = [&](int n) { return std::find(P.spec_orders.begin(), P.spec_orders.end(), n) != P.spec_orders.end(); };
    for (size_t i = 0; i + 1 < t.size(); ++i) {
      if (t[i + 1] <= t[i]) fail("fixed tile table must be strictly increasing");
      int rows = 0;
      for (int k = t[i]; k < t[i + 1]; ++k) rows += P.n_k[k] - 1;
      if (rows > TB - 1) fail("a fixed tile holds more nodes than a workgroup has threads");
      const int o = i < P.fixed_tile_order.size() ? P.fixed_tile_order[i] : 0;
      if (o != 0) {
        if (!has_body(o)) fail("a fixed tile names an order the code object has no body for");
        for (int k = t[i]; k < t[i + 1]; ++k)
          if (P.n_k[k] != o) fail("a fixed tile of order n holds a section of another order");
      }
    }
    P.tile_k0 = t;
    P.tile_order.assign(t.size() - 1, 0);
    for (size_t i = 0; i + 1 < t.size() && i < P.fixed_tile_order.size(); ++i) P.tile_order[i] = P.fixed_tile_order[i];
    return;
  }
  if (!P.spec_orders.empty()) {
    build_tiles_mixed(P, TB);
    return;
  }
  P.tile_order.clear();
  P.tile_k0.clear();
  P.tile_k0.push_back(0);
  int rows = 0;
  for (int k = 0; k < P.K; ++k) {
    const int r = P.n_k[k] - 1;
    if (rows + r > TB - 1) {  // tile holds rows+1 nodes <= TB
      P.tile_k0.push_back(k);
      rows = 0;
    }
    rows += r;
  }
  P.tile_k0.push_back(P.K);
}

inline void build_layout(Problem& Q) {
  int64_t x = 0, c = 0;
  int ox = 0, oc = 0;
  for (auto& P : Q.ph) {
    P.x_off = x;
    P.q_off = x + (int64_t)P.n_z * P.N;
    P.t_off = P.q_off + P.n_q;
    P.c_off = c;
    P.c_path_off = c + (int64_t)P.n_y * (P.N - 1);
    P.c_int_off = P.c_path_off + (int64_t)P.n_p * P.N;
    P.ocp_x_off = ox;
    P.ocp_c_off = oc;
    x = P.t_off + P.n_t;
    c = P.c_int_off + P.n_q;
    ox += P.n_z + P.n_q + P.n_t;
    oc += P.n_y + P.n_p + P.n_q;
  }
  Q.s_off = x;
  Q.ocp_s_off = ox;
  Q.num_x = x + Q.n_s;
  Q.c_end_off = c;
  Q.ocp_c_end_off = oc;
  Q.num_c = c + Q.n_b;
  Q.num_ocp_x = ox + Q.n_s;
  Q.num_ocp_c = oc + Q.n_b;
  if (Q.num_x >= INT32_MAX || Q.num_c >= INT32_MAX) fail("problem too large for 32-bit IPOPT indices");
  const size_t np = Q.point_kind.size();
  Q.point_x.assign(np, 0);
  Q.point_ocp.assign(np, 0);
  for (size_t i = 0; i < np; ++i) {
    const int kind = Q.point_kind[i], idx = Q.point_idx[i], ip = Q.point_phase[i];
    if (kind == 5) {
      if (idx < 0 || idx >= Q.n_s) fail("point variable: parameter index out of range");
      Q.point_x[i] = Q.s_off + idx;
      Q.point_ocp[i] = Q.ocp_s_off + idx;
      continue;
    }
    if (ip < 0 || ip >= (int)Q.ph.size()) fail("point variable: phase out of range");
    const Phase& P = Q.ph[ip];
    switch (kind) {
      case 0:
      case 1:
        if (idx < 0 || idx >= P.n_y) fail("point variable: state index out of range");
        Q.point_x[i] = P.x_off + (int64_t)idx * P.N + (kind == 1 ? P.N - 1 : 0);
        Q.point_ocp[i] = P.ocp_x_off + idx;
        break;
      case 2:
        if (idx < 0 || idx >= P.n_q) fail("point variable: integral index out of range");
        Q.point_x[i] = P.q_off + idx;
        Q.point_ocp[i] = P.ocp_x_off + P.n_z + idx;
        break;
      case 3:
        if (!P.t_free[0]) fail("point variable: t0 is not free");
        Q.point_x[i] = P.t_off;
        Q.point_ocp[i] = P.ocp_x_off + P.n_z + P.n_q;
        break;
      case 4:
        if (!P.t_free[1]) fail("point variable: tF is not free");
        Q.point_x[i] = P.t_off + (P.t_free[0] ? 1 : 0);
        Q.point_ocp[i] = P.ocp_x_off + P.n_z + P.n_q + (P.t_free[0] ? 1 : 0);
        break;
      default:
        fail("point variable: unknown kind");
    }
    if (i > 0 && Q.point_x[i] <= Q.point_x[i - 1]) fail("point variables must be in ascending x order");
  }
}

// ---- Jacobian ---------------------------------------------------------------------------------
inline void build_G(Problem& Q) {
  // pass 1: count, pass 2: fill
  auto row_len_defect = [](const Phase& P, int a, int n) {
    int len = 0;
    for (int b = 0; b < P.n_z; ++b) len += P.dep(a, b) ? n : (b == a ? 2 : 0);
    len += P.n_t;
    for (int l = 0; l < P.n_w; ++l) len += (!P.is_t(l) && P.dep(a, P.n_z + l)) ? 1 : 0;
    return len;
  };
  // q columns of integral row m: its own q_m and every integral the integrand depends on
  auto qcols = [](const Phase& P, int m) {
    std::set<int> c{m};
    for (int l = 0; l < P.n_w; ++l)
      if (P.wkind[l] == 1 && P.dep(P.n_y + P.n_p + m, P.n_z + l)) c.insert(P.widx[l]);
    return c;
  };
  int64_t nnz = 0;
  for (auto& P : Q.ph) {
    for (int a = 0; a < P.n_y; ++a)
      for (int k = 0; k < P.K; ++k) nnz += (int64_t)(P.n_k[k] - 1) * row_len_defect(P, a, P.n_k[k]);
    for (int m = 0; m < P.n_p; ++m) {
      int len = 0;
      for (int c = 0; c < P.n_v; ++c) len += P.dep(P.n_y + m, c) ? 1 : 0;
      nnz += (int64_t)len * P.N;
    }
    for (int m = 0; m < P.n_q; ++m) {
      const int r = P.n_y + P.n_p + m;
      for (int b = 0; b < P.n_z; ++b) nnz += P.dep(r, b) ? P.N : 0;
      nnz += (int64_t)qcols(P, m).size() + P.n_t;
      for (int l = 0; l < P.n_w; ++l) nnz += (P.wkind[l] == 0 && P.dep(r, P.n_z + l)) ? 1 : 0;
    }
  }
  nnz += (int64_t)Q.bjac_row.size();
  if (nnz >= INT32_MAX) fail("Jacobian has too many non-zeros for 32-bit IPOPT indices");
  Q.g_row.resize(nnz);
  Q.g_col.resize(nnz);
  Q.g_indptr.assign(Q.num_c + 1, 0);
  int64_t p = 0;
  auto put = [&](int64_t r, int64_t c) {
    Q.g_row[p] = (int32_t)r;
    Q.g_col[p] = (int32_t)c;
    ++p;
  };
  for (auto& P : Q.ph) {
    P.goff.assign(P.n_y + P.n_p + P.n_q, 0);
    for (int a = 0; a < P.n_y; ++a) {
      P.goff[a] = p;
      for (int k = 0; k < P.K; ++k) {
        const int n = P.n_k[k], sk = P.sec_s[k];
        for (int j = 1; j < n; ++j) {
          const int64_t row = P.c_off + (int64_t)a * (P.N - 1) + sk + j - 1;
          Q.g_indptr[row] = p;
          for (int b = 0; b < P.n_z; ++b) {
            const int64_t cb = P.x_off + (int64_t)b * P.N + sk;
            if (P.dep(a, b)) {
              for (int i = 0; i < n; ++i) put(row, cb + i);
            } else if (b == a) {
              put(row, cb);
              put(row, cb + j);
            }
          }
          for (int l = 0; l < P.n_w; ++l)
            if (P.wkind[l] == 1 && P.dep(a, P.n_z + l)) put(row, P.wcol(l, Q.s_off));
          for (int jt = 0; jt < P.n_t; ++jt) put(row, P.t_off + jt);
          for (int l = 0; l < P.n_w; ++l)
            if (P.wkind[l] == 0 && P.dep(a, P.n_z + l)) put(row, P.wcol(l, Q.s_off));
        }
      }
    }
    for (int m = 0; m < P.n_p; ++m) {
      const int r = P.n_y + m;
      P.goff[P.n_y + m] = p;
      for (int i = 0; i < P.N; ++i) {
        const int64_t row = P.c_path_off + (int64_t)m * P.N + i;
        Q.g_indptr[row] = p;
        for (int b = 0; b < P.n_z; ++b)
          if (P.dep(r, b)) put(row, P.x_off + (int64_t)b * P.N + i);
        for (int l = 0; l < P.n_w; ++l)          // parameters are listed in x order
          if (P.dep(r, P.n_z + l)) put(row, P.wcol(l, Q.s_off));
      }
    }
    for (int m = 0; m < P.n_q; ++m) {
      const int r = P.n_y + P.n_p + m;
      const int64_t row = P.c_int_off + m;
      P.goff[P.n_y + P.n_p + m] = p;
      Q.g_indptr[row] = p;
      for (int b = 0; b < P.n_z; ++b)
        if (P.dep(r, b))
          for (int i = 0; i < P.N; ++i) put(row, P.x_off + (int64_t)b * P.N + i);
      P.gq_base[m] = p;
      for (int mq : qcols(P, m)) put(row, P.q_off + mq);
      for (int jt = 0; jt < P.n_t; ++jt) put(row, P.t_off + jt);
      for (int l = 0; l < P.n_w; ++l)
        if (P.wkind[l] == 0 && P.dep(r, P.n_z + l)) put(row, P.wcol(l, Q.s_off));
    }
  }
  Q.g_end_base = p;
  {
    size_t e = 0;
    for (int r = 0; r < Q.n_b; ++r) {
      Q.g_indptr[Q.c_end_off + r] = p;
      for (; e < Q.bjac_row.size() && Q.bjac_row[e] == r; ++e) {
        if (e > 0 && Q.bjac_row[e - 1] == r && Q.bjac_col[e - 1] >= Q.bjac_col[e]) fail("bjac must be sorted");
        put(Q.c_end_off + r, Q.point_x[Q.bjac_col[e]]);
      }
    }
    if (e != Q.bjac_row.size()) fail("bjac rows must be sorted and < n_b");
  }
  Q.g_indptr[Q.num_c] = p;
  if (p != nnz) fail("internal error: Jacobian fill count mismatch");
}

// ---- Hessian ----------------------------------------------------------------------------------
inline int64_t find_in_row(const Problem& Q, int64_t row, int64_t col) {
  const int32_t* b = Q.h_col.data() + Q.h_indptr[row];
  const int32_t* e = Q.h_col.data() + Q.h_indptr[row + 1];
  const int32_t* it = std::lower_bound(b, e, (int32_t)col);
  if (it == e || *it != (int32_t)col) fail("internal error: Hessian slot lookup failed");
  return (int64_t)(it - Q.h_col.data());
}

inline void build_H(Problem& Q) {
  // irregular entries: endpoint block + parameter-parameter scalars; row -> sorted columns
  std::map<int64_t, std::set<int64_t>> extra;
  auto extra_lower = [&](int64_t a, int64_t b) { extra[std::max(a, b)].insert(std::min(a, b)); };
  for (size_t e = 0; e < Q.pthess_row.size(); ++e) {
    const int r = Q.pthess_row[e], c = Q.pthess_col[e];
    if (r < 0 || r >= (int)Q.point_x.size() || c < 0 || c > r) fail("endpoint Hessian entry out of range");
    extra[Q.point_x[r]].insert(Q.point_x[c]);
  }
  for (auto& P : Q.ph) {
    for (int l = 0; l < P.n_w; ++l) {
      const int64_t cl = P.wcol(l, Q.s_off);
      if (P.n_t > 0 && P.tz(P.n_z + l))       // d2/dt dw of stretch(t) * (mu . F)
        for (int jt = 0; jt < P.n_t; ++jt) extra_lower(cl, P.t_off + jt);
      for (int l2 = 0; l2 <= l; ++l2)
        if (P.hdep(P.n_z + l, P.n_z + l2)) extra_lower(cl, P.wcol(l2, Q.s_off));
    }
  }
  std::vector<int32_t> rows, cols;
  Q.h_indptr.assign(Q.num_x + 1, 0);
  std::vector<int64_t> reg;  // regular columns of the current row (ascending)
  auto emit_row = [&](int64_t row) {
    Q.h_indptr[row] = (int64_t)rows.size();
    auto it = extra.find(row);
    if (it == extra.end()) {
      for (int64_t c : reg) {
        rows.push_back((int32_t)row);
        cols.push_back((int32_t)c);
      }
    } else {
      std::vector<int64_t> merged;
      merged.reserve(reg.size() + it->second.size());
      std::set_union(reg.begin(), reg.end(), it->second.begin(), it->second.end(), std::back_inserter(merged));
      for (int64_t c : merged) {
        if (c > row) fail("internal error: upper-triangular Hessian entry");
        rows.push_back((int32_t)row);
        cols.push_back((int32_t)c);
      }
    }
    reg.clear();
  };
  auto strip = [&](const Phase& P, int b) {
    for (int i = 0; i < P.N; ++i) reg.push_back(P.x_off + (int64_t)b * P.N + i);
  };
  for (auto& P : Q.ph) {
    const int N = P.N;
    for (int b = 0; b < P.n_z; ++b) {
      std::vector<int> cs;
      for (size_t e = 0; e < P.hess_row.size(); ++e)
        if (P.hess_row[e] == b) cs.push_back(P.hess_col[e]);
      for (int i = 0; i < N; ++i) {
        for (int c : cs) reg.push_back(P.x_off + (int64_t)c * N + i);
        emit_row(P.x_off + (int64_t)b * N + i);
      }
    }
    for (int m = 0; m < P.n_q; ++m) {
      for (int l = 0; l < P.n_w; ++l)
        if (P.wkind[l] == 1 && P.widx[l] == m)
          for (int b = 0; b < P.n_z; ++b)
            if (P.hdep(P.n_z + l, b)) strip(P, b);
      emit_row(P.q_off + m);
    }
    for (int jt = 0; jt < P.n_t; ++jt) {
      for (int b = 0; b < P.n_z; ++b)
        if (P.tzx(b)) strip(P, b);
      emit_row(P.t_off + jt);
    }
  }
  for (int ls = 0; ls < Q.n_s; ++ls) {
    for (auto& P : Q.ph)
      for (int l = 0; l < P.n_w; ++l)
        if (P.wkind[l] == 0 && P.widx[l] == ls)
          for (int b = 0; b < P.n_z; ++b)
            if (P.hdep(P.n_z + l, b)) strip(P, b);
    emit_row(Q.s_off + ls);
  }
  Q.h_indptr[Q.num_x] = (int64_t)rows.size();
  if (rows.size() >= (size_t)INT32_MAX) fail("Hessian has too many non-zeros for 32-bit IPOPT indices");
  Q.h_row.swap(rows);
  Q.h_col.swap(cols);

  // ---- producer-slot tables -------------------------------------------------------------------
  std::set<int64_t> bulk_edge;   // slots the bulk kernels write at nodes 0 / N-1 (or strips' ends)
  std::set<int64_t> owned;       // slots only the tail writes
  for (auto& P : Q.ph) {
    const int N = P.N, NZ = P.n_z, NW = P.n_w;
    P.hoff.assign(NZ + 2 * NZ + NW * NZ, -1);
    for (int b = 0; b < NZ; ++b) {
      const int mb = P.hrow_count(b);
      if (mb > 0 && N >= 3) {
        const int64_t r1 = P.x_off + (int64_t)b * N + 1;
        P.hoff[b] = Q.h_indptr[r1] - mb;
        for (int i = 1; i < N - 1; ++i) {
          const int64_t r = P.x_off + (int64_t)b * N + i;
          if (Q.h_indptr[r + 1] - Q.h_indptr[r] != mb || Q.h_indptr[r] != P.hoff[b] + (int64_t)i * mb)
            fail("internal error: irregular interior Hessian row");
        }
      }
    }
    P.hslot0.clear();
    P.hslotN.clear();
    for (size_t e = 0; e < P.hess_row.size(); ++e) {
      const int b = P.hess_row[e], c = P.hess_col[e];
      if (b >= NZ) continue;
      P.hslot0.push_back(find_in_row(Q, P.x_off + (int64_t)b * N, P.x_off + (int64_t)c * N));
      P.hslotN.push_back(find_in_row(Q, P.x_off + (int64_t)b * N + N - 1, P.x_off + (int64_t)c * N + N - 1));
      bulk_edge.insert(P.hslot0.back());
      bulk_edge.insert(P.hslotN.back());
    }
    auto strip_slot = [&](int64_t row, int b, const char* what) {
      const int64_t s0 = find_in_row(Q, row, P.x_off + (int64_t)b * N);
      if (find_in_row(Q, row, P.x_off + (int64_t)b * N + N - 1) != s0 + N - 1)
        fail(std::string("internal error: ") + what + " strip is not contiguous");
      bulk_edge.insert(s0);
      bulk_edge.insert(s0 + N - 1);
      return s0;
    };
    for (int jt = 0; jt < P.n_t; ++jt)
      for (int b = 0; b < NZ; ++b)
        if (P.tzx(b)) P.hoff[NZ + jt * NZ + b] = strip_slot(P.t_off + jt, b, "t");
    for (int l = 0; l < NW; ++l)          // strips of the time parameters are the t strips above
      for (int b = 0; b < NZ; ++b)
        if (!P.is_t(l) && P.hdep(NZ + l, b)) P.hoff[3 * NZ + l * NZ + b] = strip_slot(P.wcol(l, Q.s_off), b, "parameter");
    P.hsum_slot.assign(2 * NW + NW * (NW + 1) / 2, -1);
    auto lower_slot = [&](int64_t a, int64_t b) { return find_in_row(Q, std::max(a, b), std::min(a, b)); };
    for (int l = 0; l < NW; ++l) {
      const int64_t cl = P.wcol(l, Q.s_off);
      if (P.n_t > 0 && P.tz(NZ + l))
        for (int jt = 0; jt < P.n_t; ++jt) {
          P.hsum_slot[jt * NW + l] = lower_slot(cl, P.t_off + jt);
          owned.insert(P.hsum_slot[jt * NW + l]);
        }
      for (int l2 = 0; l2 <= l; ++l2)
        if (P.hdep(NZ + l, NZ + l2)) {
          P.hsum_slot[2 * NW + l * (l + 1) / 2 + l2] = lower_slot(cl, P.wcol(l2, Q.s_off));
          owned.insert(P.hsum_slot[2 * NW + l * (l + 1) / 2 + l2]);
        }
    }
  }
  Q.pt_hslot.clear();
  for (size_t e = 0; e < Q.pthess_row.size(); ++e) {
    const int64_t slot = find_in_row(Q, Q.point_x[Q.pthess_row[e]], Q.point_x[Q.pthess_col[e]]);
    Q.pt_hslot.push_back(slot);
    if (!bulk_edge.count(slot)) owned.insert(slot);
  }
  for (int64_t s : owned)
    if (bulk_edge.count(s)) fail("internal error: Hessian slot has two owners");
  Q.tail_owned.assign(owned.begin(), owned.end());
  auto local = [&](int64_t slot) -> int32_t {
    if (slot < 0) return -1;
    auto it = std::lower_bound(Q.tail_owned.begin(), Q.tail_owned.end(), slot);
    return (it != Q.tail_owned.end() && *it == slot) ? (int32_t)(it - Q.tail_owned.begin()) : -1;
  };
  for (auto& P : Q.ph) {
    P.hsum_local.clear();
    for (int64_t sl : P.hsum_slot) P.hsum_local.push_back(local(sl));
  }
  Q.pt_hlocal.clear();
  for (int64_t sl : Q.pt_hslot) Q.pt_hlocal.push_back(local(sl));
}

inline void build_all(Problem& Q, int TB, bool tiles_only = false) {
  for (auto& P : Q.ph) {
    if (P.sec_s.empty()) finalize_phase_tables(P, Q.n_s);
    build_tiles(P, TB);
  }
  if (tiles_only) return;
  build_layout(Q);
  build_G(Q);
  build_H(Q);
}

}  // namespace pcp
