// Hand-written HIP kernel templates for the collocation NLP callbacks (gfx950 / CDNA4, wave64).
//
// A generated translation unit defines one `Model` struct per phase (compile-time sizes, the
// structural non-zero lists of the first/second partials and one straight-line `eval`) plus a
// `Point` struct for the endpoint functions, includes this header and instantiates
//   pc::bulk<Model>      one workgroup per mesh tile, one collocation node per lane
//   pc::tail_*           one workgroup: finishes the cross-tile sums and the endpoint rows
//
// What the bulk kernel replaces (reference file:line):
//   unscale x = V x~ + r ................. pycollo/scaling.py:176-178, backend.py:263-280
//   f, p, g at every node ................ backend.py:1565-1570 (expand_eqn_to_vec), compiled.py:174-189
//   defect  A y + stretch I f ............ backend.py:1601-1603, compiled.py:139-140
//   path / integral rows ................. backend.py:1605-1647, compiled.py:142-146
//   Jacobian blocks ...................... compiled.py:305-379
//   multiplier contraction I^T (W lam) ... iteration.py:1078-1103
//   Lagrangian Hessian bands/strips/sums . compiled.py:484-500, numbafy_hessian.py:93-121
//
// Mapping: tile = contiguous run of mesh sections with at most blockDim.x nodes (shared end node
// included as a read-only halo).  Lane t evaluates node n0+t once; f and df/ds are staged in LDS
// so the section-local (n_k-1) x n_k contractions read neighbours from LDS; quadrature A / weight
// tables are staged in LDS per workgroup.  Jacobian values are written "column-wise": the lane that
// owns node i' writes, for every row j of its section, the entries in column i' -- consecutive lanes
// hit consecutive addresses inside each n_k-wide run.  All cross-tile sums go through per-tile
// partials and are finished in a fixed order by the tail kernel: results are bit-reproducible.
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>

#include "pc_args.h"

namespace pc {

template <int I>
using ic = std::integral_constant<int, I>;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(ic<B>{});
    static_for<B + 1, E>(f);
  }
}

// ---------------------------------------------------------------------------------------------
// compile-time structure derived from the model's non-zero lists
// ---------------------------------------------------------------------------------------------
template <class M>
struct S {
  static constexpr int NY = M::NY, NU = M::NU, NZ = M::NY + M::NU, NQ = M::NQ, NP = M::NP, NS = M::NS;
  static constexpr int NT = (M::T0_FREE ? 1 : 0) + (M::TF_FREE ? 1 : 0);
  static constexpr int NFN = NY + NP + NQ, NV = NZ + NS, NJ = M::NJ, NH = M::NH;

  // index of the jacobian entry (function r, variable c) or -1
  static constexpr int jidx(int r, int c) {
    for (int e = 0; e < NJ; ++e)
      if (M::jr(e) == r && M::jc(e) == c) return e;
    return -1;
  }
  static constexpr bool dep(int r, int c) { return jidx(r, c) >= 0; }
  // ---- defect rows of state a
  static constexpr int D(int a) {  // number of section-dense z blocks
    int n = 0;
    for (int b = 0; b < NZ; ++b) n += dep(a, b) ? 1 : 0;
    return n;
  }
  static constexpr int ndep_before(int a, int b) {
    int n = 0;
    for (int bb = 0; bb < b; ++bb) n += dep(a, bb) ? 1 : 0;
    return n;
  }
  static constexpr bool own_sparse(int a) { return !dep(a, a); }  // D-only block of 2 entries
  static constexpr int nsdep(int r) {
    int n = 0;
    for (int l = 0; l < NS; ++l) n += dep(r, NZ + l) ? 1 : 0;
    return n;
  }
  static constexpr int srank(int r, int l) {
    int n = 0;
    for (int ll = 0; ll < l; ++ll) n += dep(r, NZ + ll) ? 1 : 0;
    return n;
  }
  static constexpr int C(int a) { return (own_sparse(a) ? 2 : 0) + NT + nsdep(a); }
  // number of df/ds entries that must be staged for the section contraction, and their slot
  static constexpr int NFS = [] {
    int n = 0;
    for (int a = 0; a < NY; ++a) n += nsdep(a);
    return n;
  }();
  static constexpr int fs_slot(int a, int l) {
    int n = 0;
    for (int aa = 0; aa < a; ++aa) n += nsdep(aa);
    return n + srank(a, l);
  }
  // ---- path / integral rows
  static constexpr int nzdep(int r) {
    int n = 0;
    for (int b = 0; b < NZ; ++b) n += dep(r, b) ? 1 : 0;
    return n;
  }
  static constexpr int zrank(int r, int b) {
    int n = 0;
    for (int bb = 0; bb < b; ++bb) n += dep(r, bb) ? 1 : 0;
    return n;
  }
  // ---- hessian (entries are lower-triangular in v = [z | s], sorted by (row, col))
  static constexpr int hrow_count(int b) {  // z-z entries in row b
    int n = 0;
    for (int e = 0; e < NH; ++e) n += (M::hr(e) == b && M::hc(e) < NZ) ? 1 : 0;
    return n;
  }
  static constexpr int hpos(int e) {  // rank of entry e inside its row (z-z part)
    int n = 0;
    for (int ee = 0; ee < e; ++ee) n += (M::hr(ee) == M::hr(e)) ? 1 : 0;
    return n;
  }
  static constexpr int hzz_index(int e) {  // running index among z-z entries
    int n = 0;
    for (int ee = 0; ee < e; ++ee) n += (M::hr(ee) < NZ) ? 1 : 0;
    return n;
  }
  static constexpr int NHZZ = [] {
    int n = 0;
    for (int e = 0; e < NH; ++e) n += (M::hr(e) < NZ) ? 1 : 0;
    return n;
  }();
  // t-strip mask: does sum_r mu_r dF_r/dz_b (r over f and g rows) have structure?
  static constexpr bool tz(int c) {
    for (int r = 0; r < NFN; ++r)
      if (!(r >= NY && r < NY + NP) && dep(r, c)) return true;
    return false;
  }
  // ---- reductions (per tile): [NQ] sum w g | [NQ*NS] sum w dg/ds | [NS] t-s | [NS*(NS+1)/2] s-s
  static constexpr int R_Q = 0, R_QS = NQ, R_TS = NQ + NQ * NS, R_SS = NQ + NQ * NS + NS;
  static constexpr int NRED = NQ + NQ * NS + NS + NS * (NS + 1) / 2;
  // ---- packed scaling offsets
  static constexpr int O_VZ = 0, O_RZ = NZ, O_VQ = 2 * NZ, O_RQ = 2 * NZ + NQ, O_VT = 2 * NZ + 2 * NQ,
                       O_RT = O_VT + 2, O_VS = O_RT + 2, O_RS = O_VS + NS, O_WD = O_RS + NS, O_WP = O_WD + NY,
                       O_WI = O_WP + NP, NSCAL = O_WI + NQ;
  // goff / hoff layout
  static constexpr int GO_D = 0, GO_P = NY, GO_Q = NY + NP;
  static constexpr int HO_Z = 0, HO_T = NZ, HO_S = 3 * NZ;
};

// LDS carve-up, shared by host (size query) and device.  All offsets in doubles.
struct LdsPlan {
  int qa, qw, h, E, s, kr, f, yu, fs, lam, red, total;
};
__host__ __device__ inline LdsPlan lds_plan(int TB, int qa_total, int qw_total, int NY, int NFS, int NRED) {
  LdsPlan p;
  int o = 0;
  p.qa = o; o += qa_total;
  p.qw = o; o += qw_total;
  p.h = o; o += TB + 2;
  p.E = o; o += TB + 2;                 // int64 entries
  p.s = o; o += (TB + 4) / 2 + 1;       // int32 entries, (TB+3) of them
  p.kr = o; o += (TB + 1) / 2 + 1;      // int32 entries
  p.f = o; o += NY * TB;
  p.yu = o; o += NY * TB;
  p.fs = o; o += NFS * TB;
  p.lam = o; o += NY * (TB + PC_MAX_ORDER);
  p.red = o; o += (NRED > 0 ? NRED : 1) * 16;
  p.total = o;
  return p;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// bulk kernel
// ---------------------------------------------------------------------------------------------
template <class M>
__device__ __forceinline__ void bulk(const PcPhaseArgs& A) {
  using St = S<M>;
  constexpr int NY = St::NY, NZ = St::NZ, NQ = St::NQ, NP = St::NP, NS = St::NS, NT = St::NT;
  constexpr int NFN = St::NFN, NV = St::NV, NJ = St::NJ, NH = St::NH, NFS = St::NFS, NRED = St::NRED;

  extern __shared__ double smem[];
  const int tid = threadIdx.x, TB = blockDim.x;
  const LdsPlan lp = lds_plan(TB, A.qa_total, A.qw_total, NY, NFS, NRED);
  double* s_qa = smem + lp.qa;
  double* s_qw = smem + lp.qw;
  double* s_h = smem + lp.h;
  long long* s_E = reinterpret_cast<long long*>(smem + lp.E);
  int* s_s = reinterpret_cast<int*>(smem + lp.s);
  int* s_kr = reinterpret_cast<int*>(smem + lp.kr);
  double* s_f = smem + lp.f;
  double* s_yu = smem + lp.yu;
  double* s_fs = smem + lp.fs;
  double* s_lam = smem + lp.lam;
  double* s_red = smem + lp.red;

  const bool wantC = A.flags & PC_FLAG_C, wantG = A.flags & PC_FLAG_G, wantH = A.flags & PC_FLAG_H;
  const int N = A.N;
  const int tile = blockIdx.x + A.tile_begin;
  const int k0 = A.tile_k0[tile], k1 = A.tile_k0[tile + 1];
  const bool has_prev = k0 > 0;
  const int kp = has_prev ? k0 - 1 : 0;  // first staged section
  const int nsec = k1 - kp;              // staged sections (previous one included)
  const bool last_tile = (k1 == A.K);

  for (int i = tid; i < A.qa_total; i += TB) s_qa[i] = A.qa[i];
  for (int i = tid; i < A.qw_total; i += TB) s_qw[i] = A.qw[i];
  for (int i = tid; i <= nsec; i += TB) s_s[i] = A.sec_s[kp + i];
  for (int i = tid; i < nsec; i += TB) {
    s_h[i] = A.sec_h[kp + i];
    s_E[i] = A.sec_E[kp + i];
  }
  __syncthreads();
  const int n0 = s_s[has_prev ? 1 : 0], n1 = s_s[nsec];
  const int T = n1 - n0;  // defect rows per state in this tile; nodes n0 .. n0+T
  const int lam0 = s_s[0];  // first staged defect row
  for (int ls = tid; ls < nsec; ls += TB) {
    const int sb = s_s[ls], se = s_s[ls + 1];
    for (int node = sb + 1; node <= se; ++node) {
      const int tt = node - n0;
      if (tt >= 0 && tt <= T) s_kr[tt] = ls;
    }
  }
  if (tid == 0 && n0 == 0) s_kr[0] = -1;
  if (wantH) {
    const int cnt = n1 - lam0;
    static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      for (int i = tid; i < cnt; i += TB)
        s_lam[a * (TB + PC_MAX_ORDER) + i] = A.lam[A.c_off + (int64_t)a * (N - 1) + lam0 + i];
    });
  }
  __syncthreads();

  // ---- uniform scalars ------------------------------------------------------------------------
  const double* sc = A.scal;
  double t0 = A.t_fixed[0], tF = A.t_fixed[1];
  double dst[2] = {0.0, 0.0};  // d stretch / d t~_j for the free times, in x order
  {
    const int64_t t_off = A.x_off + (int64_t)NZ * N + NQ;
    int j = 0;
    if constexpr (M::T0_FREE) {
      t0 = sc[St::O_VT + j] * A.x[t_off + j] + sc[St::O_RT + j];
      dst[j] = -0.5 * sc[St::O_VT + j];
      ++j;
    }
    if constexpr (M::TF_FREE) {
      tF = sc[St::O_VT + j] * A.x[t_off + j] + sc[St::O_RT + j];
      dst[j] = 0.5 * sc[St::O_VT + j];
    }
  }
  const double stretch = 0.5 * (tF - t0);

  // ---- per-node state -------------------------------------------------------------------------
  const int t = tid;
  const bool active = t <= T;
  const int node = n0 + t;
  const bool owns = active && (t < T || last_tile);
  double v[NV > 0 ? NV : 1], F[NFN > 0 ? NFN : 1], Jv[NJ > 0 ? NJ : 1], Hv[NH > 0 ? NH : 1], mu[NFN > 0 ? NFN : 1];
  double red[NRED > 0 ? NRED : 1];
  static_for<0, NRED>([&](auto r_) { red[decltype(r_)::value] = 0.0; });

  int ls_r = -1, pos_r = 0, n_r = 2, ls_s = 0, n_s = 2;
  bool has_start = false;
  double w_node = 0.0;
  if (active) {
    static_for<0, NZ>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      v[b] = sc[St::O_VZ + b] * A.x[A.x_off + (int64_t)b * N + node] + sc[St::O_RZ + b];
    });
    static_for<0, NS>([&](auto l_) {
      constexpr int l = decltype(l_)::value;
      v[NZ + l] = sc[St::O_VS + l] * A.x[A.s_off + l] + sc[St::O_RS + l];
    });
    ls_r = s_kr[t];
    ls_s = ls_r + 1;
    // the node opens section ls_s only if it is that section's first node (staged in this tile)
    has_start = (node < N - 1) && (ls_s < nsec) && (s_s[ls_s] == node);
    if (ls_r >= 0) {
      n_r = s_s[ls_r + 1] - s_s[ls_r] + 1;
      pos_r = node - s_s[ls_r];
      w_node = s_h[ls_r] * s_qw[A.qw_off[n_r] + pos_r];
    }
    if (has_start) {
      n_s = s_s[ls_s + 1] - s_s[ls_s] + 1;
      w_node += s_h[ls_s] * s_qw[A.qw_off[n_s]];
    }
  }

  // ---- adjoint node weights mu (iteration.py:1078-1103) ------------------------------------------
  static_for<0, NFN>([&](auto r_) { mu[decltype(r_)::value] = 0.0; });
  if (owns && wantH) {
    static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      const double* la = s_lam + a * (TB + PC_MAX_ORDER);
      double acc = 0.0;
      if (ls_r >= 0) {
        const double* At = s_qa + A.qa_off[n_r];
        const int base = s_s[ls_r] - lam0;
        double a2 = 0.0;
        for (int j = 1; j < n_r; ++j) a2 += la[base + j - 1] * At[(j - 1) * n_r + pos_r];
        acc += s_h[ls_r] * a2;
      }
      if (has_start) {
        const double* At = s_qa + A.qa_off[n_s];
        const int base = s_s[ls_s] - lam0;
        double a2 = 0.0;
        for (int j = 1; j < n_s; ++j) a2 += la[base + j - 1] * At[(j - 1) * n_s];
        acc += s_h[ls_s] * a2;
      }
      mu[a] = sc[St::O_WD + a] * acc;
    });
    static_for<0, NP>([&](auto m_) {
      constexpr int m = decltype(m_)::value;
      mu[NY + m] = sc[St::O_WP + m] * A.lam[A.c_path_off + (int64_t)m * N + node];
    });
    static_for<0, NQ>([&](auto m_) {
      constexpr int m = decltype(m_)::value;
      mu[NY + NP + m] = -sc[St::O_WI + m] * A.lam[A.c_int_off + m] * w_node;
    });
  }

  // ---- model evaluation -----------------------------------------------------------------------
  if (active) {
    double mult[NFN > 0 ? NFN : 1];
    static_for<0, NFN>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      mult[r] = (r >= NY && r < NY + NP) ? mu[r] : stretch * mu[r];
    });
    M::eval(v, mult, F, Jv, Hv);
    static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      s_f[a * TB + t] = F[a];
      s_yu[a * TB + t] = v[a];
      static_for<0, NS>([&](auto l_) {
        constexpr int l = decltype(l_)::value;
        if constexpr (St::dep(a, NZ + l)) s_fs[St::fs_slot(a, l) * TB + t] = Jv[St::jidx(a, NZ + l)];
      });
    });
  }
  __syncthreads();

  // ---- node-owned outputs ---------------------------------------------------------------------
  if (owns) {
    // path rows (backend.py:1612-1614, compiled.py:336-355)
    static_for<0, NP>([&](auto m_) {
      constexpr int m = decltype(m_)::value;
      constexpr int r = NY + m;
      const double Wp = sc[St::O_WP + m];
      if (wantC) A.c[A.c_path_off + (int64_t)m * N + node] = Wp * F[r];
      if (wantG) {
        constexpr int R = St::nzdep(r) + St::nsdep(r);
        const int64_t rs = A.goff[St::GO_P + m] + (int64_t)node * R;
        static_for<0, NZ>([&](auto b_) {
          constexpr int b = decltype(b_)::value;
          if constexpr (St::dep(r, b)) A.G[rs + St::zrank(r, b)] = Wp * sc[St::O_VZ + b] * Jv[St::jidx(r, b)];
        });
        static_for<0, NS>([&](auto l_) {
          constexpr int l = decltype(l_)::value;
          if constexpr (St::dep(r, NZ + l))
            A.G[rs + St::nzdep(r) + St::srank(r, l)] = Wp * sc[St::O_VS + l] * Jv[St::jidx(r, NZ + l)];
        });
      }
    });
    // integral rows: z entries and partial sums (backend.py:1645-1647, compiled.py:357-379)
    static_for<0, NQ>([&](auto m_) {
      constexpr int m = decltype(m_)::value;
      constexpr int r = NY + NP + m;
      red[St::R_Q + m] = w_node * F[r];
      static_for<0, NS>([&](auto l_) {
        constexpr int l = decltype(l_)::value;
        if constexpr (St::dep(r, NZ + l)) red[St::R_QS + m * NS + l] = w_node * Jv[St::jidx(r, NZ + l)];
      });
      if (wantG) {
        const double k = -sc[St::O_WI + m] * stretch * w_node;
        static_for<0, NZ>([&](auto b_) {
          constexpr int b = decltype(b_)::value;
          if constexpr (St::dep(r, b))
            A.G[A.goff[St::GO_Q + m] + (int64_t)St::zrank(r, b) * N + node] = k * sc[St::O_VZ + b] * Jv[St::jidx(r, b)];
        });
      }
    });
    // Hessian (compiled.py:484-500): flag 1 bands, flag 2 strips, flag 3 sums
    if (wantH) {
      const bool edge0 = (node == 0), edgeN = (node == N - 1);
      static_for<0, NH>([&](auto e_) {
        constexpr int e = decltype(e_)::value;
        constexpr int rv = M::hr(e), cv = M::hc(e);
        if constexpr (rv < NZ) {
          const double val = sc[St::O_VZ + rv] * sc[St::O_VZ + cv] * Hv[e];
          int64_t dst_i;
          if (edge0) dst_i = A.hslot0[St::hzz_index(e)];
          else if (edgeN) dst_i = A.hslotN[St::hzz_index(e)];
          else dst_i = A.hoff[St::HO_Z + rv] + (int64_t)node * St::hrow_count(rv) + St::hpos(e);
          A.H[dst_i] = val;
        } else if constexpr (cv < NZ) {
          A.H[A.hoff[St::HO_S + (rv - NZ) * NZ + cv] + node] = sc[St::O_VS + rv - NZ] * sc[St::O_VZ + cv] * Hv[e];
        } else {
          constexpr int l = rv - NZ, l2 = cv - NZ;
          red[St::R_SS + l * (l + 1) / 2 + l2] = sc[St::O_VS + l] * sc[St::O_VS + l2] * Hv[e];
        }
      });
      // time coupling through stretch: d/dt~_j of stretch * (mu . dF/dv), f and g rows only
      if constexpr (NT > 0) {
        static_for<0, NV>([&](auto c_) {
          constexpr int cvar = decltype(c_)::value;
          if constexpr (St::tz(cvar)) {
            double acc = 0.0;
            static_for<0, NFN>([&](auto r_) {
              constexpr int r = decltype(r_)::value;
              if constexpr (!(r >= NY && r < NY + NP) && St::dep(r, cvar)) acc += mu[r] * Jv[St::jidx(r, cvar)];
            });
            if constexpr (cvar < NZ) {
              static_for<0, NT>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                A.H[A.hoff[St::HO_T + j * NZ + cvar] + node] = dst[j] * sc[St::O_VZ + cvar] * acc;
              });
            } else {
              red[St::R_TS + cvar - NZ] = acc;
            }
          }
        });
      }
    }
  }

  // ---- Jacobian of the defect rows, written column-wise (compiled.py:305-334) -----------------
  if (active && wantG) {
    auto write_cols = [&](int ls, int pos, int n) {
      const double h = s_h[ls];
      const int sk = s_s[ls];
      const long long E = s_E[ls];
      const double* At = s_qa + A.qa_off[n];
      for (int j = 1; j < n; ++j) {
        const double coef = stretch * h * At[(j - 1) * n + pos];
        static_for<0, NY>([&](auto a_) {
          constexpr int a = decltype(a_)::value;
          const double Wd = sc[St::O_WD + a];
          const int64_t rs = A.goff[St::GO_D + a] + (int64_t)St::D(a) * (E + (long long)(j - 1) * n) +
                             (int64_t)St::C(a) * (sk + j - 1);
          static_for<0, NZ>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            if constexpr (St::dep(a, b)) {
              constexpr int before = St::ndep_before(a, b);
              constexpr int extra = (St::own_sparse(a) && a < b) ? 2 : 0;
              double val = coef * Jv[St::jidx(a, b)] * sc[St::O_VZ + b];
              if constexpr (a == b) val += sc[St::O_VZ + a] * ((pos == 0 ? 1.0 : 0.0) - (pos == j ? 1.0 : 0.0));
              A.G[rs + (int64_t)before * n + extra + pos] = Wd * val;
            }
          });
          if constexpr (St::own_sparse(a)) {
            const int64_t o = rs + (int64_t)St::ndep_before(a, a) * n;
            if (pos == 0) A.G[o] = Wd * sc[St::O_VZ + a];
            if (pos == j) A.G[o + 1] = -(Wd * sc[St::O_VZ + a]);
          }
        });
      }
    };
    if (ls_r >= (has_prev ? 1 : 0)) write_cols(ls_r, pos_r, n_r);
    if (has_start) write_cols(ls_s, 0, n_s);
  }

  // ---- defect rows: value, t and s columns (row-wise; backend.py:1601-1603, compiled.py:324-334)
  if (active && t >= 1 && (wantC || wantG)) {
    const int n = n_r, j = pos_r, sk = s_s[ls_r] - n0;
    const double h = s_h[ls_r];
    const double* Arow = s_qa + A.qa_off[n] + (j - 1) * n;
    static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      double acc = 0.0;
      for (int i = 0; i < n; ++i) acc += Arow[i] * s_f[a * TB + sk + i];
      const double Wd = sc[St::O_WD + a];
      if (wantC)
        A.c[A.c_off + (int64_t)a * (N - 1) + node - 1] = Wd * ((s_yu[a * TB + sk] - v[a]) + stretch * (h * acc));
      if (wantG) {
        if constexpr (NT + St::nsdep(a) > 0) {
          const int64_t rs = A.goff[St::GO_D + a] + (int64_t)St::D(a) * (s_E[ls_r] + (long long)(j - 1) * n) +
                             (int64_t)St::C(a) * (node - 1) + (int64_t)St::D(a) * n + (St::own_sparse(a) ? 2 : 0);
          static_for<0, NT>([&](auto jt_) {
            constexpr int jt = decltype(jt_)::value;
            A.G[rs + jt] = Wd * dst[jt] * (h * acc);
          });
          static_for<0, NS>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            if constexpr (St::dep(a, NZ + l)) {
              double as = 0.0;
              for (int i = 0; i < n; ++i) as += Arow[i] * s_fs[St::fs_slot(a, l) * TB + sk + i];
              A.G[rs + NT + St::srank(a, l)] = Wd * stretch * sc[St::O_VS + l] * (h * as);
            }
          });
        }
      }
    });
  }

  // ---- per-tile partial sums (fixed order: lanes -> waves -> tile) ------------------------------
  if constexpr (NRED > 0) {
    const int wave = tid >> 6, lane = tid & 63, nw = (TB + 63) >> 6;
    static_for<0, NRED>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      const double s = wave_sum(red[r]);
      if (lane == 0) s_red[r * 16 + wave] = s;
    });
    __syncthreads();
    if (tid < NRED) {
      double s = 0.0;
      for (int w = 0; w < nw; ++w) s += s_red[tid * 16 + w];
      A.partials[(int64_t)tile * NRED + tid] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// tail kernel pieces (one workgroup)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void tail_begin(const PcTailArgs& A) {
  if (A.flags & PC_FLAG_H)
    for (int i = threadIdx.x; i < A.n_tail_owned; i += blockDim.x) A.H[A.tail_owned[i]] = 0.0;
  __syncthreads();
}

// Finish the cross-tile sums of one phase: integral rows of c and G, (t,s)/(s,s) Hessian sums.
template <class M>
__device__ __forceinline__ void tail_phase(const PcTailArgs& A, int ip) {
  using St = S<M>;
  constexpr int NZ = St::NZ, NQ = St::NQ, NP = St::NP, NY = St::NY, NS = St::NS, NT = St::NT, NRED = St::NRED;
  if constexpr (NRED > 0) {
    __shared__ double s_part[256];
    __shared__ double s_sum[NRED];
    const PcTailPhase& P = A.ph[ip];
    const int tid = threadIdx.x, TB = blockDim.x;  // TB <= 256
    for (int r = 0; r < NRED; ++r) {
      double acc = 0.0;
      for (int b = tid; b < P.n_tiles; b += TB) acc += P.partials[(int64_t)b * NRED + r];
      s_part[tid] = acc;
      __syncthreads();
      for (int off = TB >> 1; off > 0; off >>= 1) {
        if (tid < off) s_part[tid] += s_part[tid + off];
        __syncthreads();
      }
      if (tid == 0) s_sum[r] = s_part[0];
      __syncthreads();
    }
    if (tid == 0) {
      const double* sc = P.scal;
      const int N = P.N;
      double t0 = P.t_fixed[0], tF = P.t_fixed[1], dst[2] = {0.0, 0.0};
      const int64_t q_off = P.x_off + (int64_t)NZ * N, t_off = q_off + NQ;
      int j = 0;
      if constexpr (M::T0_FREE) {
        t0 = sc[St::O_VT + j] * A.x[t_off + j] + sc[St::O_RT + j];
        dst[j] = -0.5 * sc[St::O_VT + j];
        ++j;
      }
      if constexpr (M::TF_FREE) {
        tF = sc[St::O_VT + j] * A.x[t_off + j] + sc[St::O_RT + j];
        dst[j] = 0.5 * sc[St::O_VT + j];
      }
      const double stretch = 0.5 * (tF - t0);
      static_for<0, NQ>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        constexpr int r = NY + NP + m;
        const double Wi = sc[St::O_WI + m];
        if (A.flags & PC_FLAG_C) {
          const double q = sc[St::O_VQ + m] * A.x[q_off + m] + sc[St::O_RQ + m];
          A.c[P.c_int_off + m] = Wi * (q - stretch * s_sum[St::R_Q + m]);
        }
        if (A.flags & PC_FLAG_G) {
          int64_t o = P.gq_base[m];
          A.G[o++] = Wi * sc[St::O_VQ + m];
          static_for<0, NT>([&](auto jt_) { A.G[o++] = -Wi * dst[decltype(jt_)::value] * s_sum[St::R_Q + m]; });
          static_for<0, NS>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            if constexpr (St::dep(r, NZ + l)) A.G[o++] = -Wi * stretch * sc[St::O_VS + l] * s_sum[St::R_QS + m * NS + l];
          });
        }
      });
      if (A.flags & PC_FLAG_H) {
        static_for<0, NS>([&](auto l_) {
          constexpr int l = decltype(l_)::value;
          if constexpr (NT > 0 && St::tz(NZ + l)) {
            static_for<0, NT>([&](auto jt_) {
              constexpr int jt = decltype(jt_)::value;
              A.H[P.hsum_slot[jt * NS + l]] += dst[jt] * sc[St::O_VS + l] * s_sum[St::R_TS + l];
            });
          }
        });
        static_for<0, St::NH>([&](auto e_) {
          constexpr int e = decltype(e_)::value;
          if constexpr (M::hc(e) >= NZ) {
            constexpr int l = M::hr(e) - NZ, l2 = M::hc(e) - NZ;
            A.H[P.hsum_slot[2 * NS + l * (l + 1) / 2 + l2]] += s_sum[St::R_SS + l * (l + 1) / 2 + l2];
          }
        });
      }
    }
    __syncthreads();
  }
}

// Endpoint functions: objective, endpoint constraint rows, their Jacobian and Hessian.
template <class PT>
__device__ __forceinline__ void tail_point(const PcTailArgs& A) {
  if (threadIdx.x != 0) return;
  constexpr int NPV = PT::NPV, NB = PT::NB, NGJ = PT::NGJ, NBJ = PT::NBJ, NPH = PT::NPH;
  double xb[NPV > 0 ? NPV : 1], lb[NB > 0 ? NB : 1];
  static_for<0, NPV>([&](auto i_) {
    constexpr int i = decltype(i_)::value;
    xb[i] = A.point_V[i] * A.x[A.point_x[i]] + A.point_r[i];
  });
  const double sigma = A.params[0], wJ = A.params[1];
  const bool wantH = A.flags & PC_FLAG_H;
  static_for<0, NB>([&](auto r_) {
    constexpr int r = decltype(r_)::value;
    lb[r] = wantH ? A.lam[A.c_end_off + r] * A.W_end[r] : 0.0;
  });
  double Jval, gJ[NGJ > 0 ? NGJ : 1], b[NB > 0 ? NB : 1], jb[NBJ > 0 ? NBJ : 1], hb[NPH > 0 ? NPH : 1];
  PT::eval(xb, sigma * wJ, lb, Jval, gJ, b, jb, hb);
  if (A.fobj) A.fobj[0] = wJ * Jval;
  if (A.grad) {
    static_for<0, NGJ>([&](auto e_) {
      constexpr int e = decltype(e_)::value;
      A.grad[A.point_x[PT::gc(e)]] = wJ * gJ[e] * A.point_V[PT::gc(e)];
    });
  }
  if (A.flags & PC_FLAG_C)
    static_for<0, NB>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      A.c[A.c_end_off + r] = A.W_end[r] * b[r];
    });
  if (A.flags & PC_FLAG_G)
    static_for<0, NBJ>([&](auto e_) {
      constexpr int e = decltype(e_)::value;
      A.G[A.g_end_base + e] = A.W_end[PT::br(e)] * jb[e] * A.point_V[PT::bc(e)];
    });
  if (wantH)
    static_for<0, NPH>([&](auto e_) {
      constexpr int e = decltype(e_)::value;
      A.H[A.pt_hslot[e]] += hb[e] * A.point_V[PT::phr(e)] * A.point_V[PT::phc(e)];
    });
}

}  // namespace pc
