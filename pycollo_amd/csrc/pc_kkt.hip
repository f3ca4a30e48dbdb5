// KKT solve on the GPU: block L D L^T of the interior-point system over the collocation structure
// (leaves = mesh-section interiors, chain = section boundary nodes, border = global unknowns).
// Tables come from pycollo_amd/kkt.py (which explains the ordering); entry points: pc_kkt_* in include/pycollo_amd.h.
// Replaces the sparse symmetric-indefinite solver IPOPT calls (MUMPS; the reference only names it:
// pycollo/backend.py:1703-1711, pycollo/settings.py:49-59).
//
// Everything a block needs lives in one value buffer: a leaf's [A | C] (its own matrix and its coupling to
// [left node | right node | border]) is overwritten by [L, D | X = A^-1 C], its Schur block S = -C^T A^-1 C goes to a
// slot of its own; a chain node's [D | E | F] likewise.  No atomics, no pivot search: the order of every sum is fixed
// by the tables, two factorisations of the same matrix give the same bits.
#include <hip/hip_runtime.h>

#include <chrono>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <utility>
#include <map>
#include <vector>

#include "../../include/pycollo_amd.h"
#include "pc_kkt_cr.hpp"

namespace {

thread_local std::string k_err;

#define KHIP(expr)                                                                               \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_));              \
  } while (0)

template <class T>
struct Dev {
  T* p = nullptr;
  size_t n = 0;
  void alloc(size_t count) {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = count;
    if (count) KHIP(hipMalloc(&p, count * sizeof(T)));
  }
  void upload(const T* src, size_t count) {
    alloc(count);
    if (count) KHIP(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
  }
  ~Dev() {
    if (p) (void)hipFree(p);
  }
};

// pinned host staging: the vectors of a call cross the bus from / to page-locked memory (a pageable hipMemcpyAsync is
// staged by the runtime through its own bounce buffer, synchronously: ~0.1 ms per 240 KB vector at config 2, more than
// the solve's kernels)
template <class T>
struct Pin {
  T* p = nullptr;
  size_t n = 0;
  void alloc(size_t count) {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = count;
    if (count) KHIP(hipHostMalloc(&p, count * sizeof(T), hipHostMallocDefault));
  }
  ~Pin() {
    if (p) (void)hipHostFree(p);
  }
};

enum { SRC_G = 0, SRC_H = 1, SRC_ONE = 2 };

// ---- assembly -------------------------------------------------------------------------------------------------
// one thread per destination: the sources of one matrix position, summed in table order
__global__ void kkt_scatter(double* __restrict__ vals, const int64_t* __restrict__ dst, const int64_t* __restrict__ run_ptr,
                            const int32_t* __restrict__ kind, const int32_t* __restrict__ idx, const double* __restrict__ coef,
                            const double* __restrict__ G, const double* __restrict__ H, int use_H, int64_t n_dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_dst) return;
  double acc = 0.0;
  for (int64_t e = run_ptr[i]; e < run_ptr[i + 1]; ++e) {
    const int k = kind[e];
    const double v = k == SRC_G ? G[idx[e]] : (k == SRC_H ? (use_H ? H[idx[e]] : 0.0) : 1.0);
    acc += v * coef[e];
  }
  vals[dst[i]] = acc;
}

// diagonal: Sigma + dw on primal unknowns, -dc on multipliers (dvec); a fixed unknown is a unit pivot
__global__ void kkt_diag(double* __restrict__ vals, const int64_t* __restrict__ diag_pos, const uint8_t* __restrict__ fixed,
                         const double* __restrict__ dvec, int64_t nu) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= nu) return;
  double* p = vals + diag_pos[u];
  *p = fixed[u] ? 1.0 : *p + dvec[u];
}

// y = K x over natural unknowns (rows of the symmetric recipe).  A KKT row of a collocation NLP has 7-14 entries (a
// node variable: its Hessian band + the defect rows of its section; a multiplier: its Jacobian row), so a wave per row
// (round 2: 178 us at 30 k rows, < 20 % of the lanes active, four dependent loads per entry in a lane of its own)
// wastes the machine.  Here 8 lanes share a row: consecutive lanes read consecutive table entries (rows are stored
// back to back, so a wave's 8 rows are one contiguous stretch of the tables), kind and index travel in ONE word
// (kind << 30 | index), and the 8 partial sums meet through DPP row shifts.  Rows longer than MV_LONG entries -- the
// integral / parameter rows with one entry per node -- would serialise 8 lanes over thousands of entries: the host
// lists them, they are skipped here and done by one workgroup each (kkt_matvec_long).
constexpr int MV_LANES = 8, MV_LONG = 256;
__device__ __forceinline__ double mv_entry(unsigned src, double coef, const double* __restrict__ G, const double* __restrict__ H, int use_H) {
  const unsigned k = src >> 30, idx = src & 0x3fffffffu;
  const double v = k == SRC_G ? G[idx] : (k == SRC_H ? (use_H ? H[idx] : 0.0) : 1.0);
  return v * coef;
}
// MODE 0: y = K x; MODE 1: y = b - K x (the residual of the iterative refinement)
template <int MODE>
__global__ void __launch_bounds__(256) kkt_matvec(const int64_t* __restrict__ ptr, const int32_t* __restrict__ col, const uint32_t* __restrict__ src,
                           const double* __restrict__ coef, const double* __restrict__ G,
                           const double* __restrict__ H, int use_H, const uint8_t* __restrict__ fixed,
                           const double* __restrict__ dvec, const double* __restrict__ x, const double* __restrict__ b,
                           double* __restrict__ y, int64_t nu) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MV_LANES;
  const int lane = threadIdx.x & (MV_LANES - 1);
  const bool live = row < nu;
  const int64_t e0 = live ? ptr[row] : 0, e1 = live ? ptr[row + 1] : 0;
  const bool is_long = e1 - e0 > MV_LONG;
  double acc = 0.0;
  if (!is_long)
    for (int64_t e = e0 + lane; e < e1; e += MV_LANES) acc += mv_entry(src[e], coef[e], G, H, use_H) * x[col[e]];
  // fixed order: lane i += lane i+1, i+2, i+4 inside its group of 8 (all 64 lanes take part: no divergence here)
  acc += __shfl_down(acc, 1, MV_LANES);
  acc += __shfl_down(acc, 2, MV_LANES);
  acc += __shfl_down(acc, 4, MV_LANES);
  if (live && lane == 0 && !is_long) {
    const double v = fixed[row] ? x[row] : acc + dvec[row] * x[row];
    y[row] = MODE == 0 ? v : b[row] - v;
  }
}
// A long row is cut into chunks of MV_CHUNK entries, one workgroup per chunk (a single workgroup per row walked
// 10 k entries in 40 dependent rounds of four loads each: 55 us); the chunks' sums are added in chunk order by
// kkt_matvec_long_finish, so the result does not depend on the launch geometry.
constexpr int MV_CHUNK = 1024;
__global__ void __launch_bounds__(256) kkt_matvec_long(const int64_t* __restrict__ chunk_e0, const int64_t* __restrict__ chunk_e1,
                                const int32_t* __restrict__ col, const uint32_t* __restrict__ src, const double* __restrict__ coef,
                                const double* __restrict__ G, const double* __restrict__ H, int use_H,
                                const double* __restrict__ x, double* __restrict__ part_out) {
  __shared__ double part[4];
  double acc = 0.0;
  for (int64_t e = chunk_e0[blockIdx.x] + threadIdx.x; e < chunk_e1[blockIdx.x]; e += 256) acc += mv_entry(src[e], coef[e], G, H, use_H) * x[col[e]];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part_out[blockIdx.x] = ((part[0] + part[1]) + part[2]) + part[3];
}
template <int MODE>
__global__ void kkt_matvec_long_finish(const int64_t* __restrict__ rows, const int64_t* __restrict__ row_chunk0, int64_t n_rows,
                                       const double* __restrict__ part, const uint8_t* __restrict__ fixed,
                                       const double* __restrict__ dvec, const double* __restrict__ x,
                                       const double* __restrict__ b, double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const int64_t row = rows[i];
  double s = 0.0;
  for (int64_t c = row_chunk0[i]; c < row_chunk0[i + 1]; ++c) s += part[c];
  const double v = fixed[row] ? x[row] : s + dvec[row] * x[row];
  y[row] = MODE == 0 ? v : b[row] - v;
}

// small vector kernels of the on-device iterative refinement (pc_kkt_solve_refined)
__global__ void kkt_add(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}
// out[0] = sum r^2, out[1] = number of non-finite entries of t: per-block partial sums, then one block in block order
__global__ void __launch_bounds__(256) kkt_norm_partial(const double* __restrict__ r, const double* __restrict__ t, double* __restrict__ part, int64_t n) {
  __shared__ double s0[4], s1[4];
  double a = 0.0, bad = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    a += r[i] * r[i];
    bad += isfinite(t[i]) ? 0.0 : 1.0;
  }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); bad += __shfl_down(bad, off, 64); }
  if ((threadIdx.x & 63) == 0) { s0[threadIdx.x >> 6] = a; s1[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = ((s0[0] + s0[1]) + s0[2]) + s0[3];
    part[2 * blockIdx.x + 1] = ((s1[0] + s1[1]) + s1[2]) + s1[3];
  }
}
// (one wave: lane l adds the partial sums l, l + 64, ... in turn, then the lanes meet in a fixed order)
__global__ void kkt_norm_final(const double* __restrict__ part, int nblocks, double* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  double a = 0.0, bad = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) { a += part[2 * i]; bad += part[2 * i + 1]; }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); bad += __shfl_down(bad, off, 64); }
  if (threadIdx.x == 0) {
    out[0] = a;
    out[1] = bad;
  }
}

// ---- dense block elimination ------------------------------------------------------------------------------------
// M = [A | C], row stride ld = m + w, A's lower triangle valid.  On return: strict lower triangle of A = L, its
// diagonal = D, the C part = X = A^-1 C; S (w x w, may be null) = -C^T A^-1 C; cnt = (positive, negative) pivots.
// One workgroup; lds holds 2 m + w doubles.
// WAVE: the workgroup is one wave and M, S, lds are all in LDS -- a step then only needs the wave's own LDS traffic to
// have landed (s_waitcnt lgkmcnt(0)); __syncthreads() would also wait for every global store still in flight.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <bool WAVE>
__device__ __forceinline__ void blk_sync() {
  if constexpr (WAVE) wave_lds_sync();
  else __syncthreads();
}
// (tid / nt: the calling thread's index among the nt threads that work on this block -- the workgroup by default; a wave
//  of a workgroup that handles several blocks at once passes its lane and 64, kkt_cr_top)
template <bool WAVE = false>
__device__ void block_eliminate(double* M, int m, int w, double* S, int* cnt, double* lds, int tid, int nt) {
  const int ld = m + w;
  double* colL = lds;
  double* lcol = lds + m;
  double* rowC = lds + 2 * m;
  int npos = 0, nneg = 0;
  for (int j = 0; j < m; ++j) {
    blk_sync<WAVE>();
    const double d = M[(size_t)j * ld + j];
    npos += d > 0.0;
    nneg += d < 0.0;
    for (int i = j + 1 + tid; i < m; i += nt) {
      const double c = M[(size_t)i * ld + j];
      colL[i] = c;
      lcol[i] = c / d;
    }
    for (int c = tid; c < w; c += nt) rowC[c] = M[(size_t)j * ld + m + c];
    blk_sync<WAVE>();
    // trailing update, two-dimensional: 16 lanes walk a row (its lower-triangle part, then its C part), nt / 16 rows at
    // a time -- no division per element (the flat e / width, e % width form of round 2 spent most of a leaf's 296 us in
    // integer division) and neighbouring lanes touch neighbouring LDS words
    {
      const int ti = tid >> 4, tk = tid & 15, nrow = nt >> 4;
      for (int i = j + 1 + ti; i < m; i += nrow) {
        const double li = lcol[i];
        double* Mi = M + (size_t)i * ld;
        for (int kk = j + 1 + tk; kk <= i; kk += 16) Mi[kk] -= li * colL[kk];
        for (int c = tk; c < w; c += 16) Mi[m + c] -= li * rowC[c];
      }
    }
    for (int i = j + 1 + tid; i < m; i += nt) M[(size_t)i * ld + j] = lcol[i];
  }
  blk_sync<WAVE>();
  // Z = L^-1 C now sits in the C part.  1 / D once (the scratch vectors are free again), then S = -Z^T D^-1 Z,
  // Y = D^-1 Z and the back-substitution L^T X = Y column-oriented: once row j is final every row above it loses
  // L[j][i] X[j][:] -- all (i, c) pairs of a step in parallel, one barrier per row.  (A thread per column walking all
  // of L serially, as this was first written, left 8 of a leaf's 128 threads busy for ~100 us of a 300 us leaf.)
  double* dinv = lds;
  for (int j = tid; j < m; j += nt) dinv[j] = 1.0 / M[(size_t)j * ld + j];
  blk_sync<WAVE>();
  if (S)
    for (int e = tid; e < w * w; e += nt) {
      const int a = e / w, b = e % w;
      double acc = 0.0;
      for (int j = 0; j < m; ++j) acc += M[(size_t)j * ld + m + a] * (M[(size_t)j * ld + m + b] * dinv[j]);
      S[e] = -acc;
    }
  blk_sync<WAVE>();
  // (rows x columns of the C part, two-dimensional again: wc = the power of two that holds w columns)
  const int wsh = w <= 4 ? 2 : (w <= 8 ? 3 : (w <= 16 ? 4 : (w <= 32 ? 5 : 6)));
  const int wc = 1 << wsh, ri = tid >> wsh, ci = tid & (wc - 1), rstep = nt >> wsh > 0 ? nt >> wsh : 1;
  if (w <= 64) {
    if (ci < w && ri < rstep)
      for (int j = ri; j < m; j += rstep) M[(size_t)j * ld + m + ci] *= dinv[j];
    for (int j = m - 1; j > 0; --j) {
      blk_sync<WAVE>();
      if (ci < w && ri < rstep) {
        const double xj = M[(size_t)j * ld + m + ci];
        for (int i = ri; i < j; i += rstep) M[(size_t)i * ld + m + ci] -= M[(size_t)j * ld + i] * xj;
      }
    }
  } else {   // (wider than a wave: the flat form)
    for (int e = tid; e < m * w; e += nt) {
      const int j = e / w, c = e - j * w;
      M[(size_t)j * ld + m + c] *= dinv[j];
    }
    for (int j = m - 1; j > 0; --j) {
      blk_sync<WAVE>();
      for (int e = tid; e < j * w; e += nt) {
        const int i = e / w, c = e - i * w;
        M[(size_t)i * ld + m + c] -= M[(size_t)j * ld + i] * M[(size_t)j * ld + m + c];
      }
    }
  }
  blk_sync<WAVE>();
  if (tid == 0 && cnt) {
    cnt[0] = npos;
    cnt[1] = nneg;
  }
}

// t = A^-1 r for a factored block (r in lds, length m; result left there); one workgroup
template <bool WAVE = false>
__device__ void block_solve(const double* M, int m, int ld, double* r, int tid, int nt) {
  for (int j = 0; j < m; ++j) {
    blk_sync<WAVE>();
    const double rj = r[j];
    for (int i = j + 1 + tid; i < m; i += nt) r[i] -= M[(size_t)i * ld + j] * rj;
  }
  blk_sync<WAVE>();
  for (int i = tid; i < m; i += nt) r[i] /= M[(size_t)i * ld + i];
  for (int j = m - 1; j > 0; --j) {   // L^T x = r, column-oriented: x_j is final, rows above it lose L[j][i] x_j
    blk_sync<WAVE>();
    const double rj = r[j];
    for (int i = tid; i < j; i += nt) r[i] -= M[(size_t)j * ld + i] * rj;
  }
  blk_sync<WAVE>();
}

template <bool WAVE = false>
__device__ __forceinline__ void block_eliminate(double* M, int m, int w, double* S, int* cnt, double* lds) {
  block_eliminate<WAVE>(M, m, w, S, cnt, lds, (int)threadIdx.x, (int)blockDim.x);
}
template <bool WAVE = false>
__device__ __forceinline__ void block_solve(const double* M, int m, int ld, double* r) {
  block_solve<WAVE>(M, m, ld, r, (int)threadIdx.x, (int)blockDim.x);
}

struct KArgs {
  double* vals;
  const int64_t *leaf_ptr, *chain_ptr, *chain_phase_ptr, *leaf_left, *leafA_off, *leafS_off, *chainD_off, *chainS_off;
  const int64_t* leaf_of_left;   // [n_chain] leaf whose left node is this chain node, -1 for the last node of a phase
  const uint8_t* chain_last;     // [n_chain]
  int64_t border_off, base_chain, base_border;
  int32_t nb, n_leaf, n_chain;
  int* counts;                   // [n_leaf + n_chain + 1][2]
  int32_t lds_doubles;           // doubles of dynamic LDS the launch provides
  int32_t chain_lds, nzmax, wcmax;   // chain factorisation in LDS: largest node block and coupling width
  // solve
  double* r;                     // [nu] right-hand side, then the solution, in block order
  double* leafG;                 // per leaf: X_C^T r_l, w doubles at leafG_off
  double* chainG;                // per chain node: Y^T r_c, wc doubles at chainG_off
  const int64_t *leafG_off, *chainG_off;
  // cyclic reduction of the chain (kkt_cr_*): per chain node its separators at the level it is eliminated at
  // (global chain ids, -1 = none), its panel / Schur block / X^T r in the cr buffer
  int32_t cr;                    // 1: the chain was eliminated by cyclic reduction (border kernels read its layout)
  double* crbuf;
  const int64_t *cr_a, *cr_b, *crP_off, *crS_off, *crG_off;
  const int64_t* cr_nodes;       // nodes by level, concatenated
  const int64_t *cr_mid_a, *cr_mid_b;   // the node whose Schur block holds a node's coupling to its separator a / b (-1: the
                                        // assembled entries: they are neighbours in the chain)
  const int64_t* pull_ptr;       // per chain node: the eliminated nodes it is a separator of, level by level --
  const int32_t* pull_e;         //   node << 1 | 1 if it is that node's LEFT separator
  const uint8_t* chain_first;    // [n_chain] first node of its segment
  const uint8_t* chain_export;   // [n_chain] or null: not eliminated, its assembled panel is handed out
  // wide borders: the border terms pre-summed by a grid (kkt_border_terms); 0 blocks = the border kernels sum them
  const double* border_part;
  int32_t border_part_blocks;
};

__device__ __forceinline__ int nzb_of(const KArgs& a, int64_t c) { return (int)(a.chain_ptr[c + 1] - a.chain_ptr[c]); }

// a leaf whose [A | C] fits the workgroup's LDS is eliminated there (every step of the elimination is a round trip
// through the block: in LDS instead of L2) and written back once
__global__ void kkt_leaf_factor(KArgs a) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t l = blockIdx.x;
  const int m = (int)(a.leaf_ptr[l + 1] - a.leaf_ptr[l]);
  const int64_t left = a.leaf_left[l];
  const int w = nzb_of(a, left) + nzb_of(a, left + 1) + a.nb;
  double* M = a.vals + a.leafA_off[l];
  const int sz = m * (m + w);
  if (sz + 2 * m + w <= a.lds_doubles) {
    double* Ml = lds + 2 * m + w;
    for (int e = tid; e < sz; e += nt) Ml[e] = M[e];
    block_eliminate(Ml, m, w, a.vals + a.leafS_off[l], a.counts + 2 * l, lds);
    for (int e = tid; e < sz; e += nt) M[e] = Ml[e];
  } else {
    block_eliminate(M, m, w, a.vals + a.leafS_off[l], a.counts + 2 * l, lds);
  }
}

// One workgroup per phase walks its chain of boundary nodes in order -- the sequential part of the factorisation.
// When the largest node fits (a.chain_lds), a node's [D | E | F], its Schur block and the previous node's are kept in
// LDS: the node's own entries and the two leaves' contributions arrive in one batch of global loads and every
// elimination step stays on chip, so a step costs one global round trip instead of one per access; the results are
// written back for the solves and the border.  Otherwise the same steps run on the value buffer itself.
__global__ void kkt_chain_factor(KArgs a) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x, nb = a.nb;
  const int64_t c0 = a.chain_phase_ptr[blockIdx.x], c1 = a.chain_phase_ptr[blockIdx.x + 1];
  const bool in_lds = a.chain_lds != 0;
  // LDS regions of fixed size: staging | carried Schur block | M | S
  double* carry = lds + (2 * a.nzmax + a.wcmax);
  double* Ml = carry + a.wcmax * a.wcmax;
  double* Sl = Ml + a.nzmax * (a.nzmax + a.wcmax);
  for (int64_t c = c0; c < c1; ++c) {
    const int nz = nzb_of(a, c), nx = a.chain_last[c] ? 0 : nzb_of(a, c + 1), wc = nx + nb, ld = nz + wc;
    double* Mg = a.vals + a.chainD_off[c];
    double* Sg = a.vals + a.chainS_off[c];
    double* M = in_lds ? Ml : Mg;
    double* S = in_lds ? Sl : Sg;
    const double* Cr = in_lds ? carry : a.vals + a.chainS_off[c > c0 ? c - 1 : c];   // previous node's Schur block
    __syncthreads();
    // own entries: the diagonal block was assembled as a lower triangle, the contributions below are full blocks
    if (in_lds) {
      for (int e = tid; e < nz * ld; e += nt) {
        const int i = e / ld, k = e % ld;
        M[e] = (k < nz && k > i) ? Mg[(size_t)k * ld + i] : Mg[e];
      }
    } else {
      for (int e = tid; e < nz * nz; e += nt) {
        const int i = e / nz, k = e % nz;
        if (k > i) M[(size_t)i * ld + k] = M[(size_t)k * ld + i];
      }
    }
    __syncthreads();
    if (c > c0) {   // leaf on the left (its R and border rows) and the previous node's Schur block
      const int nl = nzb_of(a, c - 1);
      const double* SL = a.vals + a.leafS_off[a.leaf_of_left[c - 1]];
      const int ws = nl + nz + nb, wr = nz + nb;
      for (int e = tid; e < nz * wr; e += nt) {
        const int i = e / wr, k = e % wr;
        M[(size_t)i * ld + (k < nz ? k : nx + k)] += SL[(size_t)(nl + i) * ws + nl + k] + Cr[(size_t)i * wr + k];
      }
    }
    __syncthreads();
    if (!a.chain_last[c]) {   // leaf on the right: its L rows reach this node, the next one and the border
      const double* SR = a.vals + a.leafS_off[a.leaf_of_left[c]];
      for (int e = tid; e < nz * ld; e += nt) M[e] += SR[e];
    }
    __syncthreads();
    block_eliminate(M, nz, wc, S, a.counts + 2 * (a.n_leaf + c), lds);
    if (in_lds) {
      for (int e = tid; e < nz * ld; e += nt) Mg[e] = M[e];
      for (int e = tid; e < wc * wc; e += nt) {
        const double v = S[e];
        Sg[e] = v;
        carry[e] = v;
      }
    }
    __threadfence_block();
  }
}

// ---- the chain by cyclic reduction -------------------------------------------------------------------------------------
// The chain of a phase is block-tridiagonal with a border.  Walking it node by node is hundreds to thousands of
// dependent steps of a few microseconds each (4 us per 3-unknown node even with everything in LDS and registers: the
// step is a string of LDS round trips and fp64 divisions in one wave) -- most of the factorisation and of both solves.
// Odd-even elimination has depth log2(n): at level l the nodes at positions (2k+1) 2^(l-1) are eliminated, all at once,
// each against its two neighbours at distance 2^(l-1); what survives is again block-tridiagonal.  Position 0 outlives
// every level and is eliminated last, against the border alone.  A node is a small "leaf": panel
// [D | K(c,a) | K(c,b) | F] -> L D L^T, X = D^-1 [K F], S = -[K F]^T D^-1 [K F] over [a | b | border].  Nothing is
// accumulated in place: a node *pulls* what earlier levels owe it -- the leaves' Schur blocks, and for every level
// below its own the Schur blocks of the two nodes eliminated next to it -- in a fixed order, so the factorisation is
// bit-reproducible and needs one launch per level and no atomics.
struct CrNode {
  int nz, na, nb_, w;          // own unknowns, separators' (0 if none), border; w = na + nbr + nb
  int nbr;
  int64_t a, b;
};
__device__ __forceinline__ CrNode cr_node(const KArgs& k, int64_t c) {
  CrNode n;
  n.nz = nzb_of(k, c);
  n.a = k.cr_a[c];
  n.b = k.cr_b[c];
  n.na = n.a >= 0 ? nzb_of(k, n.a) : 0;
  n.nbr = n.b >= 0 ? nzb_of(k, n.b) : 0;
  n.nb_ = k.nb;
  n.w = n.na + n.nbr + k.nb;
  return n;
}
// One chain node's turn in the factorisation, by one wave (tid = its lane): lds = the wave's scratch of the level kernels'
// dynamic LDS size, pl_* = its five tables of CR_MAX_PULL entries.
__device__ void cr_factor_node(const KArgs& k, int64_t c, double* lds, int64_t* pl_off, int* pl_row, int* pl_ew, int* pl_cD, int* pl_cF, int tid) {
  const int nb = k.nb;
  const CrNode me = cr_node(k, c);
  const int nz = me.nz, na = me.na, nr = me.nbr, w = me.w, ld = nz + w;
  double* M = lds + (2 * nz + w);                 // [nz][ld] panel; block_eliminate's scratch in front
  double* S = M + (size_t)nz * ld;                // [w][w]
  const bool first_in_phase = k.chain_first[c] != 0, last_in_phase = k.chain_last[c] != 0;
  // What the levels below owe this node, as a table built once per node: lane q looks up the q-th eliminated node c is a
  // separator of (host list, level by level) -- where its Schur block starts, the row and column offsets of c's part in
  // it.  The element loop below then issues one load per entry and pulled node; with these look-ups inside it (four
  // dependent index loads per entry, level and side) the level kernels of a 9-unknown node took 35 us each at config 3.
  const int64_t q0 = k.pull_ptr[c];
  const int npull = (int)(k.pull_ptr[c + 1] - q0);
  if (tid < npull) {
    const int32_t code = k.pull_e[q0 + tid];
    const int64_t e2 = code >> 1;
    int row, ew, cD, cF;
    if (code & 1) {                               // c is the left separator of e2: [c | b | B]
      const int64_t eb = k.cr_b[e2];
      const int ebn = eb >= 0 ? nzb_of(k, eb) : 0;
      row = 0; ew = nz + ebn + nb; cD = 0; cF = nz + ebn;
    } else {                                      // c is the right separator of e2: [a | c | B]
      const int64_t ea_ = k.cr_a[e2];
      const int ea = ea_ >= 0 ? nzb_of(k, ea_) : 0;
      row = ea; ew = ea + nz + nb; cD = ea; cF = ea + nz;
    }
    pl_off[tid] = k.crS_off[e2]; pl_row[tid] = row; pl_ew[tid] = ew; pl_cD[tid] = cD; pl_cF[tid] = cF;
  }
  wave_lds_sync();
  const int64_t mida = k.cr_mid_a[c], midb = k.cr_mid_b[c];
  // ---- level-0 values: own entries (assembled as a lower triangle + couplings) and the two leaves' Schur blocks
  const int nx0 = last_in_phase ? 0 : nzb_of(k, c + 1), ld0 = nz + nx0 + nb;
  const double* Mg = k.vals + k.chainD_off[c];
  const int nl = first_in_phase ? 0 : nzb_of(k, c - 1), wsl = nl + nz + nb;
  const double* SL = first_in_phase ? nullptr : k.vals + k.leafS_off[k.leaf_of_left[c - 1]];
  const double* SR = last_in_phase ? nullptr : k.vals + k.leafS_off[k.leaf_of_left[c]];
  for (int e = tid; e < nz * ld; e += 64) {
    const int i = e / ld, col = e - i * ld;
    double v = 0.0;
    if (col < nz || col >= nz + na + nr) {        // D and F: own entries, the leaves, and what the levels below owe
      const bool isF = col >= nz;
      const int kk = isF ? col - nz - na - nr : col;
      if (!isF) {
        v = kk > i ? Mg[(size_t)kk * ld0 + i] : Mg[(size_t)i * ld0 + kk];
        if (SL) v += SL[(size_t)(nl + i) * wsl + nl + kk];
        if (SR) v += SR[(size_t)i * ld0 + kk];
      } else {
        v = Mg[(size_t)i * ld0 + nz + nx0 + kk];
        if (SL) v += SL[(size_t)(nl + i) * wsl + nl + nz + kk];
        if (SR) v += SR[(size_t)i * ld0 + nz + nx0 + kk];
      }
      for (int q = 0; q < npull; ++q)             // (level by level, right separator's term before the left's)
        v += k.crbuf[pl_off[q] + (size_t)(pl_row[q] + i) * pl_ew[q] + (isF ? pl_cF[q] : pl_cD[q]) + kk];
    } else if (col < nz + na) {                   // K(c, a)
      const int kk = col - nz;
      if (mida < 0) {                             // a = c - 1: transposed E of a, and the left leaf's R-L block
        const int lda = na + nz + nb;             // a's own panel: [na | nz | nb]
        v = k.vals[k.chainD_off[me.a] + (size_t)kk * lda + na + i] + SL[(size_t)(nl + i) * wsl + kk];
      } else {                                    // created by the node eliminated between a and c: [a | c | B], S.ba
        const int ew = na + nz + nb;
        v = k.crbuf[k.crS_off[mida] + (size_t)(na + i) * ew + kk];
      }
    } else {                                      // K(c, b)
      const int kk = col - nz - na;
      if (midb < 0) {                             // b = c + 1: own E block and the right leaf's L-R block
        v = Mg[(size_t)i * ld0 + nz + kk] + SR[(size_t)i * ld0 + nz + kk];
      } else {                                    // created by the node eliminated between c and b: [c | b | B], S.ab
        const int ew = nz + nr + nb;
        v = k.crbuf[k.crS_off[midb] + (size_t)i * ew + nz + kk];
      }
    }
    M[e] = v;
  }
  wave_lds_sync();
  double* Pg = k.crbuf + k.crP_off[c];
  if (k.chain_export && k.chain_export[c]) {      // not eliminated: the assembled panel is this rank's term of the reduced system
    for (int e = tid; e < nz * ld; e += 64) Pg[e] = M[e];
    if (tid < 2) k.counts[2 * (k.n_leaf + c) + tid] = 0;
    return;
  }
  block_eliminate<true>(M, nz, w, S, k.counts + 2 * (k.n_leaf + c), lds, tid, 64);
  double* Sg = k.crbuf + k.crS_off[c];
  for (int e = tid; e < nz * ld; e += 64) Pg[e] = M[e];
  for (int e = tid; e < w * w; e += 64) Sg[e] = S[e];
}
// launch: one workgroup (one wave) per node of the level; `first` = offset of the level in cr_nodes
__global__ void __launch_bounds__(64) kkt_cr_factor(KArgs k, int64_t first) {
  extern __shared__ double lds[];
  __shared__ int64_t pl_off[CR_MAX_PULL];
  __shared__ int pl_row[CR_MAX_PULL], pl_ew[CR_MAX_PULL], pl_cD[CR_MAX_PULL], pl_cF[CR_MAX_PULL];
  cr_factor_node(k, k.cr_nodes[first + blockIdx.x], lds, pl_off, pl_row, pl_ew, pl_cD, pl_cF, (int)threadIdx.x);
}

__device__ void cr_forward_node(const KArgs& k, int64_t c, double* lds, int64_t* pg_off, int tid) {
  const CrNode me = cr_node(k, c);
  const int nz = me.nz, w = me.w, ld = nz + w;
  const bool first_in_phase = k.chain_first[c] != 0, last_in_phase = k.chain_last[c] != 0;
  const bool exported = k.chain_export && k.chain_export[c];
  double* rr = lds;
  double* M = lds + nz;
  const double* Pg = k.crbuf + k.crP_off[c];
  if (!exported)
    for (int e = tid; e < nz * ld; e += 64) M[e] = Pg[e];
  // (where the levels below left what they owe this node's right-hand side: looked up once, as in kkt_cr_factor)
  const int64_t q0 = k.pull_ptr[c];
  const int npull = (int)(k.pull_ptr[c + 1] - q0);
  if (tid < npull) {
    const int32_t code = k.pull_e[q0 + tid];
    const int64_t e2 = code >> 1;
    int64_t off = k.crG_off[e2];
    if (!(code & 1)) {                            // as the right separator: behind e2's left separator's part
      const int64_t ea_ = k.cr_a[e2];
      off += ea_ >= 0 ? nzb_of(k, ea_) : 0;
    }
    pg_off[tid] = off;
  }
  wave_lds_sync();
  if (tid < nz) {
    double v = k.r[k.base_chain + k.chain_ptr[c] + tid];
    if (!first_in_phase) v -= k.leafG[k.leafG_off[k.leaf_of_left[c - 1]] + nzb_of(k, c - 1) + tid];   // left leaf, R part
    if (!last_in_phase) v -= k.leafG[k.leafG_off[k.leaf_of_left[c]] + tid];                            // right leaf, L part
    for (int q = 0; q < npull; ++q) v -= k.crbuf[pg_off[q] + tid];
    rr[tid] = v;
    if (exported) k.r[k.base_chain + k.chain_ptr[c] + tid] = v;   // the rank's term of the reduced right-hand side
  }
  if (exported) return;
  wave_lds_sync();
  if (tid < w) {
    double g = 0.0;
    for (int i = 0; i < nz; ++i) g += M[(size_t)i * ld + nz + tid] * rr[i];
    k.crbuf[k.crG_off[c] + tid] = g;
  }
  block_solve<true>(M, nz, ld, rr, tid, 64);
  if (tid < nz) k.r[k.base_chain + k.chain_ptr[c] + tid] = rr[tid];
}
__global__ void __launch_bounds__(64) kkt_cr_forward(KArgs k, int64_t first) {
  extern __shared__ double lds[];
  __shared__ int64_t pg_off[CR_MAX_PULL];
  cr_forward_node(k, k.cr_nodes[first + blockIdx.x], lds, pg_off, (int)threadIdx.x);
}

__device__ void cr_backward_node(const KArgs& k, int64_t c, int tid) {
  const int nb = k.nb;
  if (k.chain_export && k.chain_export[c]) return;   // (its solution comes from the reduced system)
  const CrNode me = cr_node(k, c);
  const int nz = me.nz, na = me.na, nr = me.nbr, ld = nz + me.w;
  const double* Pg = k.crbuf + k.crP_off[c];
  if (tid < nz) {
    double v = k.r[k.base_chain + k.chain_ptr[c] + tid];
    const double* row = Pg + (size_t)tid * ld + nz;
    if (me.a >= 0) { const double* xa = k.r + k.base_chain + k.chain_ptr[me.a]; for (int q = 0; q < na; ++q) v -= row[q] * xa[q]; }
    if (me.b >= 0) { const double* xb = k.r + k.base_chain + k.chain_ptr[me.b]; for (int q = 0; q < nr; ++q) v -= row[na + q] * xb[q]; }
    const double* xB = k.r + k.base_border;
    for (int q = 0; q < nb; ++q) v -= row[na + nr + q] * xB[q];
    k.r[k.base_chain + k.chain_ptr[c] + tid] = v;
  }
}
__global__ void __launch_bounds__(64) kkt_cr_backward(KArgs k, int64_t first) {
  cr_backward_node(k, k.cr_nodes[first + blockIdx.x], (int)threadIdx.x);
}

// The last levels of the reduction hold a handful of nodes each (16, 8, 4, 2, 1, 1 of config 2's 501): as launches of
// their own each pays a kernel's fixed cost for a few microseconds of work.  Here ONE workgroup runs them all: a wave per
// node, the level's nodes dealt over the waves, a workgroup barrier between levels (what a level leaves in device memory
// is read by other waves of the same workgroup on the same compute unit).  Same per-node code, same bits.
// KIND 0: factorisation, 1: forward elimination, 2: back-substitution (levels walked downwards).
struct CrTop {
  int64_t ptr[18];     // level offsets into cr_nodes: levels [0, n) of the fused range
  int32_t n;
  int32_t wave_bytes;  // LDS per wave: the level kernels' dynamic size + the pull tables
};
template <int KIND>
__global__ void __launch_bounds__(1024) kkt_cr_top(KArgs k, CrTop t) {
  extern __shared__ double lds[];
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63, nw = (int)blockDim.x >> 6;
  char* mine = reinterpret_cast<char*>(lds) + (size_t)wave * t.wave_bytes;
  int64_t* tab64 = reinterpret_cast<int64_t*>(mine);                       // [CR_MAX_PULL]
  int* tab32 = reinterpret_cast<int*>(mine + 8 * CR_MAX_PULL);             // 4 x [CR_MAX_PULL]
  double* scratch = reinterpret_cast<double*>(mine + 24 * CR_MAX_PULL);
  for (int s = 0; s < t.n; ++s) {
    const int l = KIND == 2 ? t.n - 1 - s : s;
    const int64_t first = t.ptr[l], cnt = t.ptr[l + 1] - first;
    for (int64_t i = wave; i < cnt; i += nw) {
      const int64_t c = k.cr_nodes[first + i];
      if constexpr (KIND == 0) cr_factor_node(k, c, scratch, tab64, tab32, tab32 + CR_MAX_PULL, tab32 + 2 * CR_MAX_PULL, tab32 + 3 * CR_MAX_PULL, lane);
      else if constexpr (KIND == 1) cr_forward_node(k, c, scratch, tab64, lane);
      else cr_backward_node(k, c, lane);
    }
    __threadfence_block();
    __syncthreads();
  }
}

// border = its own entries + the border corner of every Schur block, then its factorisation (one workgroup)
// Every leaf and every chain node owes the border a term: 2 x (number of sections / group) of them.  A thread per border
// entry walking all of them is a thousand dependent-address loads in a row (0.27 ms at 10 k nodes); the walk is cut
// into `ns` interleaved slices per entry, one thread each, whose partial sums are added in slice order -- fixed order,
// same bits every run.  term(j, i, k) = contribution of leaf j (j < n_leaf) or chain node j - n_leaf.
template <class Term>
__device__ __forceinline__ void border_accumulate(const KArgs& a, int n_entries, double* part, Term term) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ns = n_entries >= nt ? 1 : nt / (n_entries > 0 ? n_entries : 1);
  const int64_t n_terms = (int64_t)a.n_leaf + a.n_chain;
  for (int base = 0; base < n_entries; base += nt / ns) {   // entries in rounds of nt / ns
    const int e = base + tid / ns, sl = tid % ns;
    double acc = 0.0;
    if (e < n_entries)
      for (int64_t j = sl; j < n_terms; j += ns) acc += term(j, e);
    part[tid] = acc;
    __syncthreads();
    if (sl == 0 && e < n_entries) {
      double tot = 0.0;
      for (int q = 0; q < ns; ++q) tot += part[tid + q];
      part[tid] = tot;
    }
    __syncthreads();
    // the caller's consume step reads part[(e - base) * ns] -- done through the callback below
    if (sl == 0 && e < n_entries) term(-1 - (int64_t)e, tid);   // hand the total over: term(-1 - e, index into part)
    __syncthreads();
  }
}

// term j (leaf j < n_leaf, else chain node j - n_leaf) of border matrix entry e = i * nb + k (lower triangle, k <= i)
__device__ __forceinline__ double border_matrix_term(const KArgs& a, int64_t j, int e) {
  const int nb = a.nb, i = e / nb, k = e % nb;
  if (k > i) return 0.0;
  if (j < a.n_leaf) {
    const int64_t left = a.leaf_left[j];
    const int ws = nzb_of(a, left) + nzb_of(a, left + 1) + nb, o = ws - nb;
    return a.vals[a.leafS_off[j] + (size_t)(o + i) * ws + o + k];
  }
  const int64_t c = j - a.n_leaf;
  if (a.cr) {
    if (a.chain_export && a.chain_export[c]) return 0.0;   // (not eliminated: no Schur block)
    const int o = (a.cr_a[c] >= 0 ? nzb_of(a, a.cr_a[c]) : 0) + (a.cr_b[c] >= 0 ? nzb_of(a, a.cr_b[c]) : 0), wc = o + nb;
    return a.crbuf[a.crS_off[c] + (size_t)(o + i) * wc + o + k];
  }
  const int nx = a.chain_last[c] ? 0 : nzb_of(a, c + 1), wc = nx + nb;
  return a.vals[a.chainS_off[c] + (size_t)(nx + i) * wc + nx + k];
}
// term j of entry e of the border's right-hand side
__device__ __forceinline__ double border_rhs_term(const KArgs& a, int64_t j, int e) {
  if (j < a.n_leaf) {
    const int64_t left = a.leaf_left[j];
    return a.leafG[a.leafG_off[j] + nzb_of(a, left) + nzb_of(a, left + 1) + e];
  }
  const int64_t c = j - a.n_leaf;
  if (a.cr && a.chain_export && a.chain_export[c]) return 0.0;
  if (a.cr) return a.crbuf[a.crG_off[c] + (a.cr_a[c] >= 0 ? nzb_of(a, a.cr_a[c]) : 0) + (a.cr_b[c] >= 0 ? nzb_of(a, a.cr_b[c]) : 0) + e];
  return a.chainG[a.chainG_off[c] + (a.chain_last[c] ? 0 : nzb_of(a, c + 1)) + e];
}
// A wide border (a rank's local border in a factorisation cut across ranks carries its cut nodes: tens of unknowns
// where the NLP's own has one or two) times thousands of leaves is millions of terms -- too many for the one workgroup
// of the border kernels (5 ms of a 6 ms factorisation at 6 000 shuttle sections cut in two).  Then the terms are summed
// first by a grid: workgroup b adds the terms b, b + gridDim.x, ... of every entry into part[b][entry]; the border kernel
// adds the workgroups' sums in workgroup order (a.border_part / a.border_part_blocks) -- fixed order again.
template <bool RHS>
__global__ void __launch_bounds__(256) kkt_border_terms(KArgs a, double* __restrict__ part) {
  const int n_entries = RHS ? a.nb : a.nb * a.nb;
  const int64_t n_terms = (int64_t)a.n_leaf + a.n_chain;
  for (int e = threadIdx.x; e < n_entries; e += blockDim.x) {
    double acc = 0.0;
    for (int64_t j = blockIdx.x; j < n_terms; j += gridDim.x) acc += RHS ? border_rhs_term(a, j, e) : border_matrix_term(a, j, e);
    part[(size_t)blockIdx.x * n_entries + e] = acc;
  }
}

// MODE 0: add every Schur term and factorise; 1: add the terms only (a rank's part of a factorisation cut across ranks:
// the block goes to the reduced system, pc_kkt_factor_partial); 2: factorise the block as it stands (the reduced system,
// pc_kkt_border_load_factor)
template <int MODE>
__global__ void kkt_border_factor(KArgs a) {
  extern __shared__ double lds[];
  const int nb = a.nb;
  double* B = a.vals + a.border_off;
  double* part = lds + (2 * nb + 2);               // [blockDim.x] partial sums, behind block_eliminate's scratch
  if constexpr (MODE == 2) {
    block_eliminate(B, nb, 0, nullptr, a.counts + 2 * (a.n_leaf + a.n_chain), lds);
    return;
  }
  auto term = [&](int64_t j, int e) -> double {
    if (j < 0) {                                    // total of entry -1 - j sits in part[e]
      const int ent = (int)(-1 - j);
      const int i = ent / nb, k = ent % nb;
      if (k <= i) B[ent] += part[e];
      return 0.0;
    }
    return border_matrix_term(a, j, e);
  };
  if (a.border_part_blocks) {                       // the terms were summed by kkt_border_terms
    for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) {
      if (e % nb > e / nb) continue;
      double tot = 0.0;
      for (int b = 0; b < a.border_part_blocks; ++b) tot += a.border_part[(size_t)b * nb * nb + e];
      B[e] += tot;
    }
  } else
    border_accumulate(a, nb * nb, part, term);
  __syncthreads();
  if constexpr (MODE == 0) block_eliminate(B, nb, 0, nullptr, a.counts + 2 * (a.n_leaf + a.n_chain), lds);
}

// second stage of kkt_border_terms: entry e of workgroup 0's slot = the sum over workgroups, in workgroup order
// (one wave per entry: lane i adds the workgroups i, i + 64, ..., then a fixed tree over the lanes -- a thread per entry
//  walking all 256 sums in a row took 60 us)
__global__ void __launch_bounds__(64) kkt_border_terms_sum(double* __restrict__ part, int n_entries, int n_blocks) {
  const int e = blockIdx.x, lane = threadIdx.x;
  double acc = 0.0;
  for (int b = lane; b < n_blocks; b += 64) acc += part[(size_t)b * n_entries + e];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) part[e] = acc;   // (slot 0 of entry e: read above by this wave's lane 0 only)
}

// ---- solve ----------------------------------------------------------------------------------------------------
__global__ void kkt_perm_in(const double* __restrict__ rhs, const int64_t* __restrict__ perm, const uint8_t* __restrict__ fixed,
                            double* __restrict__ r, int64_t nu) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nu) r[i] = fixed[perm[i]] ? 0.0 : rhs[perm[i]];
}
__global__ void kkt_perm_out(const double* __restrict__ r, const int64_t* __restrict__ perm, double* __restrict__ x, int64_t nu) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nu) x[perm[i]] = r[i];
}

__global__ void kkt_leaf_forward(KArgs a) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t l = blockIdx.x;
  const int m = (int)(a.leaf_ptr[l + 1] - a.leaf_ptr[l]);
  const int64_t left = a.leaf_left[l];
  const int w = nzb_of(a, left) + nzb_of(a, left + 1) + a.nb, ld = m + w;
  const double* M = a.vals + a.leafA_off[l];
  double* rl = a.r + a.leaf_ptr[l];
  for (int i = tid; i < m; i += nt) lds[i] = rl[i];
  // a factored leaf that fits the workgroup's LDS is solved from there: the substitution is 2 m dependent steps, each a
  // column of L -- from device memory a strided, uncoalesced read per step (27 us a leaf pass at config 2), from LDS not
  if (m * ld + m <= a.lds_doubles) {
    double* Ml = lds + m;
    for (int e = tid; e < m * ld; e += nt) Ml[e] = M[e];
    M = Ml;
  }
  __syncthreads();
  double* g = a.leafG + a.leafG_off[l];
  for (int c = tid; c < w; c += nt) {   // X_C^T r_l
    double acc = 0.0;
    for (int i = 0; i < m; ++i) acc += M[(size_t)i * ld + m + c] * lds[i];
    g[c] = acc;
  }
  block_solve(M, m, ld, lds);
  for (int i = tid; i < m; i += nt) rl[i] = lds[i];
}

__global__ void kkt_chain_forward(KArgs a) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x, nb = a.nb;
  const int64_t c0 = a.chain_phase_ptr[blockIdx.x], c1 = a.chain_phase_ptr[blockIdx.x + 1];
  for (int64_t c = c0; c < c1; ++c) {
    const int nz = nzb_of(a, c), nx = a.chain_last[c] ? 0 : nzb_of(a, c + 1), wc = nx + nb, ld = nz + wc;
    const double* M = a.vals + a.chainD_off[c];
    double* rc = a.r + a.base_chain + a.chain_ptr[c];
    __syncthreads();
    for (int i = tid; i < nz; i += nt) {
      double v = rc[i];
      if (c > c0) {
        const int nl = nzb_of(a, c - 1);
        v -= a.leafG[a.leafG_off[a.leaf_of_left[c - 1]] + nl + i];   // left leaf, R part
        v -= a.chainG[a.chainG_off[c - 1] + i];                       // previous node, next-node part
      }
      if (!a.chain_last[c]) v -= a.leafG[a.leafG_off[a.leaf_of_left[c]] + i];   // right leaf, L part
      lds[i] = v;
    }
    __syncthreads();
    double* g = a.chainG + a.chainG_off[c];
    for (int k = tid; k < wc; k += nt) {
      double acc = 0.0;
      for (int i = 0; i < nz; ++i) acc += M[(size_t)i * ld + nz + k] * lds[i];
      g[k] = acc;
    }
    block_solve(M, nz, ld, lds);
    for (int i = tid; i < nz; i += nt) rc[i] = lds[i];
    __threadfence_block();
  }
}

// SOLVE false: stop once the border's right-hand side has lost what leaves and chain owe it (pc_kkt_forward_partial)
template <bool SOLVE>
__global__ void kkt_border_solve(KArgs a) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x, nb = a.nb;
  double* rb = a.r + a.base_border;
  double* part = lds + (nb + 2);
  for (int i = tid; i < nb; i += nt) lds[i] = rb[i];
  __syncthreads();
  auto term = [&](int64_t j, int e) -> double {
    if (j < 0) {
      lds[(int)(-1 - j)] -= part[e];
      return 0.0;
    }
    return border_rhs_term(a, j, e);
  };
  if (a.border_part_blocks) {
    for (int e = tid; e < nb; e += nt) {
      double tot = 0.0;
      for (int b = 0; b < a.border_part_blocks; ++b) tot += a.border_part[(size_t)b * nb + e];
      lds[e] -= tot;
    }
  } else
    border_accumulate(a, nb, part, term);
  __syncthreads();
  if constexpr (SOLVE) block_solve(a.vals + a.border_off, nb, nb, lds);
  for (int i = tid; i < nb; i += nt) rb[i] = lds[i];
}

__global__ void kkt_chain_backward(KArgs a) {
  const int tid = threadIdx.x, nt = blockDim.x, nb = a.nb;
  const int64_t c0 = a.chain_phase_ptr[blockIdx.x], c1 = a.chain_phase_ptr[blockIdx.x + 1];
  const double* xb = a.r + a.base_border;
  for (int64_t c = c1 - 1; c >= c0; --c) {
    const int nz = nzb_of(a, c), nx = a.chain_last[c] ? 0 : nzb_of(a, c + 1), wc = nx + nb, ld = nz + wc;
    const double* M = a.vals + a.chainD_off[c];
    double* xc = a.r + a.base_chain + a.chain_ptr[c];
    const double* xn = a.r + a.base_chain + a.chain_ptr[c] + nz;   // the next node's unknowns follow in block order
    __syncthreads();
    for (int i = tid; i < nz; i += nt) {
      double v = xc[i];
      for (int k = 0; k < nx; ++k) v -= M[(size_t)i * ld + nz + k] * xn[k];
      for (int k = 0; k < nb; ++k) v -= M[(size_t)i * ld + nz + nx + k] * xb[k];
      xc[i] = v;
    }
    __threadfence_block();
  }
}

__global__ void kkt_leaf_backward(KArgs a) {
  const int tid = threadIdx.x, nt = blockDim.x, nb = a.nb;
  const int64_t l = blockIdx.x;
  const int m = (int)(a.leaf_ptr[l + 1] - a.leaf_ptr[l]);
  const int64_t left = a.leaf_left[l];
  const int nl = nzb_of(a, left), nr = nzb_of(a, left + 1), w = nl + nr + nb, ld = m + w;
  const double* M = a.vals + a.leafA_off[l];
  double* xl = a.r + a.leaf_ptr[l];
  const double* xL = a.r + a.base_chain + a.chain_ptr[left];
  const double* xR = a.r + a.base_chain + a.chain_ptr[left + 1];
  const double* xb = a.r + a.base_border;
  for (int i = tid; i < m; i += nt) {
    double v = xl[i];
    const double* row = M + (size_t)i * ld + m;
    for (int k = 0; k < nl; ++k) v -= row[k] * xL[k];
    for (int k = 0; k < nr; ++k) v -= row[nl + k] * xR[k];
    for (int k = 0; k < nb; ++k) v -= row[nl + nr + k] * xb[k];
    xl[i] = v;
  }
}

template <class F>
int guarded(F&& f) {
  try {
    f();
    return 1;
  } catch (const std::exception& e) {
    k_err = e.what();
    return 0;
  }
}

}  // namespace

struct pc_kkt {
  int device = 0;
  hipStream_t stream = nullptr;        // the stream the work is queued on: own_stream, or a caller's (pc_kkt_set_stream)
  hipStream_t own_stream = nullptr;    // created and destroyed with the handle
  const double* d_G = nullptr;
  const double* d_H = nullptr;
  int64_t nu = 0, n_dst = 0, total = 0;
  int n_leaf = 0, n_chain = 0, n_phase = 0, nb = 0;
  int lds_leaf = 0, lds_chain = 0, lds_border = 0;
  int lds_chain_factor = 0;
  bool chain_cr = false;     // the chain by cyclic reduction: one launch per level
  std::vector<int64_t> cr_lvl_ptr;   // nodes of level l: cr_nodes[cr_lvl_ptr[l-1] .. cr_lvl_ptr[l])
  size_t cr_top_from = 0;            // levels [cr_top_from, cr_lvl_ptr.size()) run in one launch (kkt_cr_top); = size(): none
  int cr_top_waves = 0, cr_top_lds = 0;
  CrTop cr_top{};
  int lds_cr = 0;
  Dev<double> crbuf;
  Dev<int64_t> cr_a, cr_b, crP_off, crS_off, crG_off, cr_nodes, cr_mid_a, cr_mid_b, pull_ptr;
  Dev<int32_t> pull_e;
  Dev<uint8_t> chain_first, chain_export;
  bool any_export = false;
  std::vector<int64_t> export_nodes;    // chain nodes that are not eliminated (a rank's shared nodes), ascending
  std::vector<int64_t> h_crP_off, h_chain_ptr, h_cr_b;      // host copies for the export calls
  Dev<double> border_part;   // [border_blocks][nb * nb] partial sums of the border terms (wide borders only)
  int border_blocks = 0;
  int lds_leaf_full = 0;   // LDS of the leaf factorisation: the largest leaf block that fits, plus its staging vectors
  Dev<double> vals, r, leafG, chainG, dvec, vin, vout, src_coef, mv_coef;
  Dev<int64_t> perm, leaf_ptr, chain_ptr, chain_phase_ptr, leaf_left, leafA_off, leafS_off, chainD_off, chainS_off,
      leaf_of_left, leafG_off, chainG_off, dst, run_ptr, diag_pos, mv_ptr;
  Dev<int32_t> src_kind, src_idx, mv_col;
  Dev<uint32_t> mv_src;            // kind << 30 | index of every matvec entry
  Dev<int64_t> mv_long, mv_long_c0, mv_chunk_e0, mv_chunk_e1;   // rows with more than MV_LONG entries, cut into chunks
  Dev<double> mv_long_part;                                     // one partial sum per chunk
  int64_t n_mv_long = 0, n_mv_chunks = 0;
  Dev<double> w_rhs, w_sol, w_res, w_trial, w_dx, w_dvec, w_part, w_norm;   // on-device iterative refinement
  Pin<double> h_norm;
  Dev<uint8_t> fixed, chain_last;
  Dev<int> counts;
  Pin<int> h_counts;
  Pin<double> h_a, h_b;      // two vectors in, or one in and one out
  KArgs args{};
  bool factored = false;
  // a refined solve stops once ||rhs - K x||_2 <= resid_tol ||rhs||_2 (IPOPT: residual_ratio_max = 1e-10 on its own ratio).
  // Config 2, per interior-point iteration: no test 1.40 ms and 3.9 back-substitutions, 1e-12 1.03 ms with the objective
  // unchanged in all ten printed digits, 1e-11 1.06 ms (objective moves in the 8th digit), 1e-9 0.95 ms
  // (profiles/r04_ipm_iter_time.txt)
  double resid_tol = 1e-12;
};

// Wait for the stream by polling first: a blocking wait costs an interrupt and a wake-up (10-20 us on the MI355X host),
// and an interior-point iteration makes four or five of them (pivot counts, residual norms)
static void kwait(hipStream_t st) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipStreamQuery(st);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) KHIP(e);
    if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > 2000) break;
  }
  KHIP(hipStreamSynchronize(st));
}

// the solve chain on the handle's stream, device vectors in natural order (d_rhs is not modified; d_x may alias it)
// (A HIP graph of this string of ~25 launches -- and of the factorisation's ~15 -- was built and measured in round 4:
//  the interior-point iteration got slower, 1.29 -> 1.39 ms at config 2: the launches are not what it waits for, the
//  level kernels' own latency is; profiles/r04_ipm_iter_time.txt.  Removed.)
// wide borders: the border terms summed by a grid, then across its workgroups; the border kernels read one slot
template <bool RHS>
static void border_terms_device(pc_kkt* k) {
  if (!k->border_blocks) return;
  const int n_entries = RHS ? k->nb : k->nb * k->nb;
  hipLaunchKernelGGL(kkt_border_terms<RHS>, dim3(k->border_blocks), dim3(256), 0, k->stream, k->args, k->border_part.p);
  hipLaunchKernelGGL(kkt_border_terms_sum, dim3(n_entries), dim3(64), 0, k->stream, k->border_part.p, n_entries, k->border_blocks);
}

// the chain's levels on the handle's stream: one launch per level with many nodes, one launch for all the last ones
// KIND 0: factorisation, 1: forward elimination, 2: back-substitution (downwards)
template <int KIND>
static void cr_levels_device(pc_kkt* k) {
  hipStream_t st = k->stream;
  // (the factorisation keeps a launch per level: its node kernel under a 1 024-thread workgroup's register budget measured
  //  9 % slower at config 3 -- 0.65 -> 0.71 ms -- and no faster at config 2; the two substitutions gain 5-8 % at config 2 and
  //  lose nothing at config 3, profiles/r04_ipm_iter_time.txt)
  const size_t L = k->cr_lvl_ptr.size(), from = KIND == 0 ? L : k->cr_top_from;
  auto level = [&](size_t l) {
    const int64_t first = k->cr_lvl_ptr[l - 1], cnt = k->cr_lvl_ptr[l] - first;
    if (cnt <= 0) return;
    if (KIND == 0) hipLaunchKernelGGL(kkt_cr_factor, dim3((unsigned)cnt), dim3(64), k->lds_cr, st, k->args, first);
    else if (KIND == 1) hipLaunchKernelGGL(kkt_cr_forward, dim3((unsigned)cnt), dim3(64), k->lds_cr, st, k->args, first);
    else hipLaunchKernelGGL(kkt_cr_backward, dim3((unsigned)cnt), dim3(64), 0, st, k->args, first);
  };
  auto top = [&] {
    if (from < L) hipLaunchKernelGGL(kkt_cr_top<KIND>, dim3(1), dim3(64 * k->cr_top_waves), k->cr_top_lds, st, k->args, k->cr_top);
  };
  if (KIND == 2) {
    top();
    for (size_t l = from - 1; l >= 1; --l) level(l);
  } else {
    for (size_t l = 1; l < from; ++l) level(l);
    top();
  }
}

static void forward_device(pc_kkt* k, const double* d_rhs) {
  hipStream_t st = k->stream;
  const unsigned nbk = (unsigned)((k->nu + 255) / 256);
  hipLaunchKernelGGL(kkt_perm_in, dim3(nbk), dim3(256), 0, st, d_rhs, k->perm.p, k->fixed.p, k->r.p, k->nu);
  if (k->n_leaf) {
    // (two waves and the leaf in LDS: pc_kkt_solve 0.199 -> 0.183 ms at config 2, 0.425 -> 0.410 at config 3; with the shuttle's
    //  6 000 leaves the LDS costs occupancy and the solve got slower, 0.63 -> 0.65-0.71: staged up to 4 096 leaves)
    static const int stage_env = std::getenv("PYCOLLO_AMD_KKT_LEAF_FWD_LDS") ? std::atoi(std::getenv("PYCOLLO_AMD_KKT_LEAF_FWD_LDS")) : -1;
    const int stage = stage_env >= 0 ? stage_env : (k->n_leaf <= 4096 ? 2 : 0);
    KArgs la = k->args;
    la.lds_doubles = stage ? k->lds_leaf_full / 8 : 0;     // (0: solve from device memory, the A/B switch)
    hipLaunchKernelGGL(kkt_leaf_forward, dim3(k->n_leaf), dim3(stage > 1 ? 64 * stage : 64), stage ? k->lds_leaf_full : k->lds_leaf, st, la);
  }
  if (k->chain_cr) cr_levels_device<1>(k);
  else if (k->n_phase) hipLaunchKernelGGL(kkt_chain_forward, dim3(k->n_phase), dim3(64), k->lds_chain, st, k->args);
}
static void backward_device(pc_kkt* k, double* d_x);
static void solve_device(pc_kkt* k, const double* d_rhs, double* d_x) {
  forward_device(k, d_rhs);
  border_terms_device<true>(k);
  hipLaunchKernelGGL(kkt_border_solve<true>, dim3(1), dim3(256), k->lds_border, k->stream, k->args);
  backward_device(k, d_x);
}
static void backward_device(pc_kkt* k, double* d_x) {
  hipStream_t st = k->stream;
  const unsigned nbk = (unsigned)((k->nu + 255) / 256);
  if (k->chain_cr) cr_levels_device<2>(k);
  else if (k->n_phase) hipLaunchKernelGGL(kkt_chain_backward, dim3(k->n_phase), dim3(64), 0, st, k->args);
  if (k->n_leaf) hipLaunchKernelGGL(kkt_leaf_backward, dim3(k->n_leaf), dim3(64), 0, st, k->args);
  hipLaunchKernelGGL(kkt_perm_out, dim3(nbk), dim3(256), 0, st, k->r.p, k->perm.p, d_x, k->nu);
  KHIP(hipGetLastError());
}

// y = K x (MODE 0) or y = b - K x (MODE 1) on the handle's stream, device vectors
template <int MODE>
static void matvec_device(pc_kkt* k, int use_hess, const double* d_dvec, const double* d_x, const double* d_b, double* d_y) {
  hipStream_t st = k->stream;
  const int64_t threads = k->nu * MV_LANES;
  hipLaunchKernelGGL(kkt_matvec<MODE>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, k->mv_ptr.p, k->mv_col.p,
                     k->mv_src.p, k->mv_coef.p, k->d_G, k->d_H, use_hess, k->fixed.p, d_dvec, d_x, d_b, d_y, k->nu);
  if (k->n_mv_long) {
    hipLaunchKernelGGL(kkt_matvec_long, dim3((unsigned)k->n_mv_chunks), dim3(256), 0, st, k->mv_chunk_e0.p, k->mv_chunk_e1.p,
                       k->mv_col.p, k->mv_src.p, k->mv_coef.p, k->d_G, k->d_H, use_hess, d_x, k->mv_long_part.p);
    hipLaunchKernelGGL(kkt_matvec_long_finish<MODE>, dim3((unsigned)((k->n_mv_long + 63) / 64)), dim3(64), 0, st, k->mv_long.p,
                       k->mv_long_c0.p, k->n_mv_long, k->mv_long_part.p, k->fixed.p, d_dvec, d_x, d_b, d_y);
  }
  KHIP(hipGetLastError());
}

extern "C" {

const char* pc_kkt_last_error(void) { return k_err.c_str(); }

int pc_kkt_create(const pc_kkt_desc* d, const double* d_jac, const double* d_hess, int device, pc_kkt** out) {
  if (out) *out = nullptr;
  pc_kkt* k = nullptr;
  const int ok = guarded([&] {
    if (!d || !out || !d_jac || !d_hess) throw std::runtime_error("null argument");
    KHIP(hipSetDevice(device));
    k = new pc_kkt();
    k->device = device;
    KHIP(hipStreamCreateWithFlags(&k->own_stream, hipStreamNonBlocking));
    k->stream = k->own_stream;
    k->d_G = d_jac;
    k->d_H = d_hess;
    k->nu = d->nu;
    k->n_leaf = (int)d->n_leaf;
    k->n_chain = (int)d->n_chain;
    k->n_phase = (int)d->n_phase;
    k->nb = (int)d->nb;
    k->n_dst = d->n_dst;
    k->total = d->total_vals;
    k->perm.upload(d->perm, d->nu);
    k->leaf_ptr.upload(d->leaf_ptr, d->n_leaf + 1);
    k->chain_ptr.upload(d->chain_ptr, d->n_chain + 1);
    k->chain_phase_ptr.upload(d->chain_phase_ptr, d->n_phase + 1);
    k->leaf_left.upload(d->leaf_left, d->n_leaf);
    k->leafA_off.upload(d->leafA_off, d->n_leaf);
    k->leafS_off.upload(d->leafS_off, d->n_leaf);
    k->chainD_off.upload(d->chainD_off, d->n_chain);
    k->chainS_off.upload(d->chainS_off, d->n_chain);
    k->dst.upload(d->dst, d->n_dst);
    k->run_ptr.upload(d->run_ptr, d->n_dst + 1);
    k->src_kind.upload(d->src_kind, d->n_src);
    k->src_idx.upload(d->src_idx, d->n_src);
    k->src_coef.upload(d->src_coef, d->n_src);
    k->diag_pos.upload(d->diag_pos, d->nu);
    k->fixed.upload(d->fixed, d->nu);
    k->mv_ptr.upload(d->mv_ptr, d->nu + 1);
    k->mv_col.upload(d->mv_col, d->n_mv);
    {
      std::vector<uint32_t> srcw((size_t)d->n_mv);
      for (int64_t e = 0; e < d->n_mv; ++e) {
        if (d->mv_idx[e] < 0 || d->mv_idx[e] >= (1 << 30) || d->mv_kind[e] < 0 || d->mv_kind[e] > 2)
          throw std::runtime_error("matvec table entry out of range");
        srcw[e] = ((uint32_t)d->mv_kind[e] << 30) | (uint32_t)d->mv_idx[e];
      }
      k->mv_src.upload(srcw.data(), srcw.size());
      std::vector<int64_t> longs, c0{0}, ce0, ce1;
      for (int64_t r = 0; r < d->nu; ++r)
        if (d->mv_ptr[r + 1] - d->mv_ptr[r] > MV_LONG) {
          longs.push_back(r);
          for (int64_t e = d->mv_ptr[r]; e < d->mv_ptr[r + 1]; e += MV_CHUNK) {
            ce0.push_back(e);
            ce1.push_back(std::min<int64_t>(e + MV_CHUNK, d->mv_ptr[r + 1]));
          }
          c0.push_back((int64_t)ce0.size());
        }
      k->n_mv_long = (int64_t)longs.size();
      k->n_mv_chunks = (int64_t)ce0.size();
      k->mv_long.upload(longs.data(), longs.size());
      k->mv_long_c0.upload(c0.data(), c0.size());
      k->mv_chunk_e0.upload(ce0.data(), ce0.size());
      k->mv_chunk_e1.upload(ce1.data(), ce1.size());
      k->mv_long_part.alloc((size_t)std::max<int64_t>(1, k->n_mv_chunks));
    }
    k->mv_coef.upload(d->mv_coef, d->n_mv);
    for (auto* w : {&k->w_rhs, &k->w_sol, &k->w_res, &k->w_trial, &k->w_dx, &k->w_dvec}) w->alloc((size_t)d->nu);
    k->w_part.alloc(4 * 256);
    k->w_norm.alloc(4);
    k->h_norm.alloc(4);
    if (const char* env = std::getenv("PYCOLLO_AMD_KKT_RESID_TOL")) k->resid_tol = std::atof(env);
    // derived tables
    std::vector<uint8_t> last(d->n_chain, 0);
    for (int64_t p = 0; p < d->n_phase; ++p) last[d->chain_phase_ptr[p + 1] - 1] = 1;
    std::vector<int64_t> leaf_of_left(d->n_chain, -1), leafG_off(d->n_leaf), chainG_off(d->n_chain);
    int64_t og = 0, mmax = 1, wmax = 1, nzmax = 1, wcmax = 1;
    for (int64_t l = 0; l < d->n_leaf; ++l) {
      const int64_t left = d->leaf_left[l];
      leaf_of_left[left] = l;
      const int64_t m = d->leaf_ptr[l + 1] - d->leaf_ptr[l];
      const int64_t w = (d->chain_ptr[left + 1] - d->chain_ptr[left]) + (d->chain_ptr[left + 2] - d->chain_ptr[left + 1]) + d->nb;
      leafG_off[l] = og;
      og += w;
      mmax = std::max(mmax, m);
      wmax = std::max(wmax, w);
    }
    k->leafG.alloc((size_t)std::max<int64_t>(1, og));
    og = 0;
    for (int64_t c = 0; c < d->n_chain; ++c) {
      const int64_t nz = d->chain_ptr[c + 1] - d->chain_ptr[c];
      const int64_t nx = last[c] ? 0 : d->chain_ptr[c + 2] - d->chain_ptr[c + 1];
      if (!last[c] && leaf_of_left[c] < 0) throw std::runtime_error("chain node without a leaf on its right");
      chainG_off[c] = og;
      og += nx + d->nb;
      nzmax = std::max(nzmax, nz);
      wcmax = std::max(wcmax, nx + d->nb);
    }
    k->chainG.alloc((size_t)std::max<int64_t>(1, og));
    k->chain_last.upload(last.data(), last.size());
    k->leaf_of_left.upload(leaf_of_left.data(), leaf_of_left.size());
    k->leafG_off.upload(leafG_off.data(), leafG_off.size());
    k->chainG_off.upload(chainG_off.data(), chainG_off.size());
    k->lds_leaf = (int)(8 * (2 * mmax + wmax + 2));
    {
      int64_t want = 0;
      for (int64_t l = 0; l < d->n_leaf; ++l) {
        const int64_t left = d->leaf_left[l], m = d->leaf_ptr[l + 1] - d->leaf_ptr[l];
        const int64_t w = (d->chain_ptr[left + 1] - d->chain_ptr[left]) + (d->chain_ptr[left + 2] - d->chain_ptr[left + 1]) + d->nb;
        const int64_t need = 8 * (m * (m + w) + 2 * m + w + 2);
        if (need <= 64000) want = std::max(want, need);
      }
      k->lds_leaf_full = (int)std::max<int64_t>(want, k->lds_leaf);
    }
    k->lds_chain = (int)(8 * (2 * nzmax + wcmax + 2));
    {
      const int64_t full = 8 * (2 * nzmax + wcmax + 2 * wcmax * wcmax + nzmax * (nzmax + wcmax) + 2);
      k->args.nzmax = (int32_t)nzmax;
      k->args.wcmax = (int32_t)wcmax;
      k->args.chain_lds = full <= 64000 ? 1 : 0;
      k->lds_chain_factor = (int)(k->args.chain_lds ? full : k->lds_chain);
    }
    {   // cyclic reduction of the chain: levels, separators, what every node pulls from the levels below (pc_kkt_cr.hpp)
      const int64_t nc = d->n_chain;
      CrPlan P;
      cr_build(nc, d->n_phase, d->chain_phase_ptr, d->chain_ptr, d->nb, d->chain_export, P);
      k->chain_cr = P.ldsmax <= 64000 && P.max_pull <= CR_MAX_PULL;
      if (const char* env = std::getenv("PYCOLLO_AMD_KKT_CR")) k->chain_cr = k->chain_cr && std::atoi(env) != 0;
      if (P.any_export && !k->chain_cr) throw std::runtime_error("exported chain nodes need the cyclic-reduction kernels (blocks too large for their LDS)");
      k->any_export = P.any_export;
      if (k->chain_cr) {
        std::vector<int64_t> nodes;
        k->cr_lvl_ptr.assign(1, 0);
        for (int l = 1; l <= P.lmax; ++l) {
          for (int64_t c = 0; c < nc; ++c)
            if (P.lvl[c] == l) nodes.push_back(c);
          k->cr_lvl_ptr.push_back((int64_t)nodes.size());
        }
        k->lds_cr = (int)P.ldsmax;
        {   // the last levels in one launch (kkt_cr_top): as many trailing levels as hold at most two nodes per wave
          const size_t L = k->cr_lvl_ptr.size();
          const int wave_bytes = (int)(((int64_t)P.ldsmax + 24 * CR_MAX_PULL + 15) & ~(int64_t)15);
          int nw = (int)std::min<int64_t>(16, 65536 / wave_bytes);
          if (const char* env = std::getenv("PYCOLLO_AMD_KKT_CR_TOP")) nw = std::min(nw, std::atoi(env));   // (0 / 1: off)
          size_t from = L;
          if (nw >= 2) {
            while (from > 1 && k->cr_lvl_ptr[from - 1] - k->cr_lvl_ptr[from - 2] <= nw && L - (from - 1) <= 17) --from;
            if (L - from < 2) from = L;             // (a single level gains nothing over its own launch)
          }
          k->cr_top_from = from;
          k->cr_top_waves = nw;
          k->cr_top_lds = nw * wave_bytes;
          k->cr_top.n = (int32_t)(L - from);
          k->cr_top.wave_bytes = wave_bytes;
          for (size_t i = 0; from + i <= L && i < 18; ++i) k->cr_top.ptr[i] = k->cr_lvl_ptr[from - 1 + i];
        }
        k->crbuf.alloc((size_t)std::max<int64_t>(1, P.buf_len));
        k->cr_a.upload(P.ca.data(), P.ca.size()); k->cr_b.upload(P.cb.data(), P.cb.size());
        k->cr_mid_a.upload(P.mida.data(), P.mida.size()); k->cr_mid_b.upload(P.midb.data(), P.midb.size());
        k->pull_ptr.upload(P.pull_ptr.data(), P.pull_ptr.size()); k->pull_e.upload(P.pull_e.data(), P.pull_e.size());
        k->chain_first.upload(P.first.data(), P.first.size());
        if (P.any_export) {
          k->chain_export.upload(P.exported.data(), P.exported.size());
          for (int64_t c = 0; c < nc; ++c)
            if (P.exported[c]) k->export_nodes.push_back(c);
        }
        k->crP_off.upload(P.oP.data(), P.oP.size()); k->crS_off.upload(P.oS.data(), P.oS.size()); k->crG_off.upload(P.oG.data(), P.oG.size());
        k->h_crP_off = P.oP;
        k->h_cr_b = P.cb;
        k->h_chain_ptr.assign(d->chain_ptr, d->chain_ptr + nc + 1);
        k->cr_nodes.upload(nodes.data(), nodes.size());
      }
    }
    k->lds_border = (int)(8 * (2 * (int64_t)d->nb + 2 + 256));
    // (also the NLP's own narrow border once there are thousands of terms: the one workgroup's walk took 93 us per
    //  factorisation and 40 us per back-substitution at config 3, 2 x 2 entries x 5 001 terms; config 2's 1 001 terms stay)
    if (d->nb > 0 && ((int64_t)d->nb * d->nb * (d->n_leaf + d->n_chain) > ((int64_t)1 << 18) || d->n_leaf + d->n_chain >= 2048)) {
      k->border_blocks = (int)std::min<int64_t>(256, d->n_leaf + d->n_chain);
      k->border_part.alloc((size_t)k->border_blocks * d->nb * d->nb);
    }
    if (k->lds_leaf > 60000 || k->lds_chain > 60000 || k->lds_border > 60000)
      throw std::runtime_error("KKT block too large for the solver's LDS staging");
    k->vals.alloc((size_t)d->total_vals);
    k->r.alloc((size_t)d->nu);
    k->dvec.alloc((size_t)d->nu);
    k->vin.alloc((size_t)d->nu);
    k->vout.alloc((size_t)d->nu);
    k->counts.alloc((size_t)2 * (d->n_leaf + d->n_chain + 1));
    k->h_counts.alloc((size_t)2 * (d->n_leaf + d->n_chain + 1));
    k->h_a.alloc((size_t)d->nu);
    k->h_b.alloc((size_t)d->nu);
    KArgs& a = k->args;
    a.vals = k->vals.p;
    a.leaf_ptr = k->leaf_ptr.p; a.chain_ptr = k->chain_ptr.p; a.chain_phase_ptr = k->chain_phase_ptr.p;
    a.leaf_left = k->leaf_left.p; a.leafA_off = k->leafA_off.p; a.leafS_off = k->leafS_off.p;
    a.chainD_off = k->chainD_off.p; a.chainS_off = k->chainS_off.p;
    a.leaf_of_left = k->leaf_of_left.p; a.chain_last = k->chain_last.p;
    a.border_off = d->border_off;
    a.base_chain = d->leaf_ptr[d->n_leaf];
    a.base_border = a.base_chain + d->chain_ptr[d->n_chain];
    a.nb = (int32_t)d->nb; a.n_leaf = (int32_t)d->n_leaf; a.n_chain = (int32_t)d->n_chain;
    a.counts = k->counts.p;
    a.r = k->r.p; a.leafG = k->leafG.p; a.chainG = k->chainG.p;
    a.leafG_off = k->leafG_off.p; a.chainG_off = k->chainG_off.p;
    a.border_part = k->border_part.p;
    a.border_part_blocks = k->border_blocks ? 1 : 0;   // (the border kernels read the slot kkt_border_terms_sum leaves)
    a.cr = k->chain_cr ? 1 : 0;
    if (k->chain_cr) {
      a.crbuf = k->crbuf.p;
      a.cr_a = k->cr_a.p; a.cr_b = k->cr_b.p;
      a.crP_off = k->crP_off.p; a.crS_off = k->crS_off.p; a.crG_off = k->crG_off.p;
      a.cr_nodes = k->cr_nodes.p;
      a.cr_mid_a = k->cr_mid_a.p; a.cr_mid_b = k->cr_mid_b.p;
      a.pull_ptr = k->pull_ptr.p; a.pull_e = k->pull_e.p;
      a.chain_first = k->chain_first.p;
      a.chain_export = k->any_export ? k->chain_export.p : nullptr;
      // (every table the level kernels dereference must be in device memory before the first launch: a forgotten upload is a
      //  null read on the GPU, not an error code)
      if (d->n_chain > 0) {
        const void* need[] = {a.crbuf, a.cr_a, a.cr_b, a.crP_off, a.crS_off, a.crG_off, a.cr_nodes, a.cr_mid_a, a.cr_mid_b,
                              a.pull_ptr, a.pull_e, a.chain_first, a.chain_last, a.chain_ptr, a.chainD_off};
        for (const void* q : need)
          if (!q) throw std::runtime_error("internal: a cyclic-reduction table was not uploaded");
        if ((int64_t)k->cr_nodes.n != d->n_chain || k->cr_lvl_ptr.back() != d->n_chain)
          throw std::runtime_error("internal: the level lists do not cover the chain");
      }
    }
  });
  if (!ok) {
    delete k;
    return 0;
  }
  *out = k;
  return 1;
}

// Host integer work (no device): where in the value buffer the entry K[u, v] of the blocked matrix lives, for n pairs
// of natural unknowns -- the rule pycollo_amd/kkt.py documents (leaf < chain < border; inside a class the lower block
// first; inside a block the larger local index is the row).  -1 where the elimination order keeps the pair apart.
static inline int64_t plan_position(const pc_kkt_plan* P, int64_t a, int64_t b) {
  constexpr int LEAF = 0, CHAIN = 1, BORDER = 2;
  if (a < 0 || a >= P->nu || b < 0 || b >= P->nu) throw std::runtime_error("unknown index out of range");
  const int cu = P->cls[a], cv = P->cls[b];
  const bool swap = cu > cv || (cu == cv && (P->blk[a] > P->blk[b] || (P->blk[a] == P->blk[b] && P->local[a] < P->local[b])));
  if (swap) std::swap(a, b);
  const int ca = P->cls[a], cb = P->cls[b];
  const int64_t ba = P->blk[a], bb = P->blk[b], la = P->local[a], lb = P->local[b];
  int64_t pos = -1;
  if (ca == LEAF) {
    const int64_t row = P->leafA_off[ba] + la * (P->m_l[ba] + P->w_l[ba]), left = P->leaf_left[ba];
    if (cb == LEAF) {
      if (ba == bb) pos = row + lb;
    } else if (cb == CHAIN) {
      if (bb == left) pos = row + P->m_l[ba] + lb;
      else if (bb == left + 1) pos = row + P->m_l[ba] + P->nzb[left] + lb;
    } else {
      pos = row + P->m_l[ba] + P->nzb[left] + P->nzb[left + 1] + lb;
    }
  } else if (ca == CHAIN) {
    const int64_t row = P->chainD_off[ba] + la * (P->nzb[ba] + P->wc[ba]);
    if (cb == CHAIN) {
      if (ba == bb) pos = row + lb;
      else if (bb == ba + 1 && !P->last_of_phase[ba]) pos = row + P->nzb[ba] + lb;
    } else if (cb == BORDER) {
      pos = row + P->nzb[ba] + P->nzb_next[ba] + lb;
    }
  } else if (cb == BORDER) {
    pos = P->border_off + la * P->nb + lb;
  }
  return pos;
}

// Host-only (no device): the cyclic reduction's tables for a chain as pc_kkt_create builds them (pc_kkt_cr.hpp) -- what the
// CPU tests hold against a symbolic elimination of the chain graph.
int pc_kkt_cr_plan(int64_t n_chain, int64_t n_phase, const int64_t* chain_phase_ptr, const int64_t* chain_ptr, int64_t nb,
                   const uint8_t* chain_export, int64_t* cr_a, int64_t* cr_b, int64_t* mid_a, int64_t* mid_b, int32_t* level,
                   int64_t* pull_ptr, int32_t* pull_e, int64_t pull_cap) {
  return guarded([&] {
    if (!chain_phase_ptr || !chain_ptr || !cr_a || !cr_b || !mid_a || !mid_b || !level || !pull_ptr || !pull_e)
      throw std::runtime_error("null argument");
    CrPlan P;
    cr_build(n_chain, n_phase, chain_phase_ptr, chain_ptr, nb, chain_export, P);
    if ((int64_t)P.n_pull > pull_cap) throw std::runtime_error("pull_e too short");
    for (int64_t c = 0; c < n_chain; ++c) {
      cr_a[c] = P.ca[c]; cr_b[c] = P.cb[c]; mid_a[c] = P.mida[c]; mid_b[c] = P.midb[c]; level[c] = P.lvl[c];
    }
    for (int64_t c = 0; c <= n_chain; ++c) pull_ptr[c] = P.pull_ptr[c];
    for (int64_t e = 0; e < P.n_pull; ++e) pull_e[e] = P.pull_e[e];
  });
}

int pc_kkt_plan_positions(const pc_kkt_plan* P, int64_t n, const int64_t* u, const int64_t* v, int64_t* out) {
  return guarded([&] {
    if (!P || !u || !v || !out || n < 0) throw std::runtime_error("null argument");
    for (int64_t e = 0; e < n; ++e) out[e] = plan_position(P, u[e], v[e]);
  });
}

// The entry tables of pc_kkt_desc in one pass of host C++ (pycollo_amd/kkt.py::build_tables states the rule in NumPy
// and the CPU tests hold this against it): the lower-triangle entries of K -- H~ (hr, hc), the scaled G~ (row nv + jr,
// column jc), the slack columns (row nv + ineq_rows[i], column n + i, value -1) -- minus those that touch a fixed unknown;
// sorted by their position in the value buffer into runs (dst, run_ptr, src_*), and expanded symmetrically into a CSR
// over the unknowns with ascending columns (mv_*).  Two calls: with dst == NULL the three counts are returned
// (n_src, n_dst, n_mv), then the caller allocates and calls again.
int pc_kkt_plan_entries(const pc_kkt_plan* P, int64_t n, int64_t nv, int64_t nH, const int64_t* hr, const int64_t* hc,
                        int64_t nG, const int64_t* jr, const int64_t* jc, const double* row_scale, int64_t ns,
                        const int64_t* ineq_rows, const uint8_t* fixed, int64_t* counts, int64_t* dst, int64_t* run_ptr,
                        int32_t* src_kind, int32_t* src_idx, double* src_coef, int64_t* mv_ptr, int32_t* mv_col,
                        int32_t* mv_kind, int32_t* mv_idx, double* mv_coef) {
  return guarded([&] {
    if (!P || !counts || !fixed || (nH && (!hr || !hc)) || (nG && (!jr || !jc || !row_scale)) || (ns && !ineq_rows))
      throw std::runtime_error("null argument");
    struct Ent { int64_t d, u, v; int32_t kind, idx; double coef; };
    std::vector<Ent> E;
    E.reserve((size_t)(nH + nG + ns));
    auto add = [&](int64_t u, int64_t v, int kind, int64_t idx, double coef) {
      if (u < 0 || u >= P->nu || v < 0 || v >= P->nu) throw std::runtime_error("KKT entry outside the unknowns");
      if (fixed[u] || fixed[v]) return;
      const int64_t d = plan_position(P, u, v);
      if (d < 0) throw std::runtime_error("KKT entry (" + std::to_string(u) + ", " + std::to_string(v) +
                                          ") couples two blocks the elimination order keeps apart");
      if (idx >= (int64_t)1 << 30) throw std::runtime_error("source index out of range");
      E.push_back(Ent{d, u, v, kind, (int32_t)idx, coef});
    };
    for (int64_t e = 0; e < nH; ++e) add(hr[e], hc[e], 1 /* SRC_H */, e, 1.0);
    for (int64_t e = 0; e < nG; ++e) add(nv + jr[e], jc[e], 0 /* SRC_G */, e, row_scale[jr[e]]);
    for (int64_t i = 0; i < ns; ++i) add(nv + ineq_rows[i], n + i, 2 /* SRC_ONE */, 0, -1.0);
    const int64_t ne = (int64_t)E.size();
    // order by destination, entries of one destination in the order they were listed (what a stable argsort gives)
    // (as pairs (destination, entry number): the keys travel with the elements, no indirect comparisons)
    // (a stable least-significant-digit radix sort of the entry numbers by destination, 11 bits a pass: the comparison sort of
    //  (destination, entry) pairs it replaces took a third longer at config 3's 1.1 M entries: 118 -> 105 ms for the whole build)
    std::vector<int64_t> so((size_t)ne);
    {
      std::vector<int64_t> key((size_t)ne), key2((size_t)ne), tmp((size_t)ne);
      int64_t dmax = 0;
      for (int64_t e = 0; e < ne; ++e) { so[e] = e; key[e] = E[e].d; dmax = std::max(dmax, key[e]); }
      constexpr int BITS = 11;
      std::vector<int64_t> bucket((size_t)1 << BITS);
      for (int shift = 0; shift < 63 && (dmax >> shift) != 0; shift += BITS) {
        std::fill(bucket.begin(), bucket.end(), 0);
        for (int64_t e = 0; e < ne; ++e) ++bucket[(size_t)((key[e] >> shift) & ((1 << BITS) - 1))];
        int64_t run = 0;
        for (auto& b : bucket) { const int64_t c = b; b = run; run += c; }
        for (int64_t e = 0; e < ne; ++e) {
          const int64_t p = bucket[(size_t)((key[e] >> shift) & ((1 << BITS) - 1))]++;
          key2[(size_t)p] = key[e];
          tmp[(size_t)p] = so[e];
        }
        key.swap(key2);
        so.swap(tmp);
      }
    }
    int64_t n_dst = 0, n_mv = 0;
    for (int64_t k = 0; k < ne; ++k) {
      if (k == 0 || E[so[k]].d != E[so[k - 1]].d) ++n_dst;
      n_mv += E[k].u != E[k].v ? 2 : 1;
    }
    counts[0] = ne; counts[1] = n_dst; counts[2] = n_mv;
    if (!dst) return;
    if (!run_ptr || !src_kind || !src_idx || !src_coef || !mv_ptr || !mv_col || !mv_kind || !mv_idx || !mv_coef)
      throw std::runtime_error("null output");
    int64_t r = 0;
    for (int64_t k = 0; k < ne; ++k) {
      const Ent& x = E[so[k]];
      if (k == 0 || x.d != E[so[k - 1]].d) { dst[r] = x.d; run_ptr[r] = k; ++r; }
      src_kind[k] = x.kind; src_idx[k] = x.idx; src_coef[k] = x.coef;
    }
    run_ptr[r] = ne;
    // symmetric CSR: rows by counting, columns ascending inside a row (pairs are unique -- checked)
    const int64_t nu = P->nu;
    std::vector<int64_t> cnt((size_t)nu + 1, 0);
    for (const Ent& x : E) { ++cnt[x.u + 1]; if (x.u != x.v) ++cnt[x.v + 1]; }
    for (int64_t i = 0; i < nu; ++i) cnt[i + 1] += cnt[i];
    for (int64_t i = 0; i <= nu; ++i) mv_ptr[i] = cnt[i];
    std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
    std::vector<int64_t> who((size_t)n_mv);     // entry number, bit 62 set for the transposed copy
    auto put = [&](int64_t row, int64_t col, int64_t e) { const int64_t p = fill[row]++; mv_col[p] = (int32_t)col; who[p] = e; };
    for (int64_t e = 0; e < ne; ++e) { put(E[e].u, E[e].v, e); if (E[e].u != E[e].v) put(E[e].v, E[e].u, e); }
    std::vector<std::pair<int32_t, int64_t>> rowbuf;
    for (int64_t i = 0; i < nu; ++i) {
      const int64_t a = mv_ptr[i], b = mv_ptr[i + 1];
      rowbuf.clear();
      for (int64_t p = a; p < b; ++p) rowbuf.emplace_back(mv_col[p], who[p]);
      std::sort(rowbuf.begin(), rowbuf.end());
      for (int64_t p = a; p < b; ++p) {
        if (p > a && rowbuf[p - a].first == rowbuf[p - a - 1].first) throw std::runtime_error("a KKT entry occurs twice in the symmetric expansion");
        const Ent& x = E[rowbuf[p - a].second];
        mv_col[p] = rowbuf[p - a].first; mv_kind[p] = x.kind; mv_idx[p] = x.idx; mv_coef[p] = x.coef;
      }
    }
  });
}

void pc_kkt_destroy(pc_kkt* k) {
  if (!k) return;
  (void)hipSetDevice(k->device);

  if (k->stream) {
    (void)hipStreamSynchronize(k->stream);
    (void)hipStreamSynchronize(k->stream);
    (void)hipStreamDestroy(k->own_stream);
  }
  delete k;
}

// assembly + factorisation on the handle's stream with the diagonal in device memory; the pivot signs come back
// border_mode: kkt_border_factor's MODE (0: the whole factorisation; 1: leaves and chain only, the border block left
// unfactorised with every Schur term added)
static void factor_device(pc_kkt* k, int use_hess, const double* d_dvec, int32_t* n_pos, int32_t* n_neg, int border_mode = 0) {
  {
    hipStream_t st = k->stream;
    KHIP(hipMemsetAsync(k->vals.p, 0, (size_t)k->total * sizeof(double), st));
    if (k->n_dst)
      hipLaunchKernelGGL(kkt_scatter, dim3((unsigned)((k->n_dst + 255) / 256)), dim3(256), 0, st, k->vals.p, k->dst.p, k->run_ptr.p,
                         k->src_kind.p, k->src_idx.p, k->src_coef.p, k->d_G, k->d_H, use_hess, k->n_dst);
    hipLaunchKernelGGL(kkt_diag, dim3((unsigned)((k->nu + 255) / 256)), dim3(256), 0, st, k->vals.p, k->diag_pos.p, k->fixed.p,
                       d_dvec, k->nu);
    if (k->n_leaf) {
      KArgs la = k->args;
      la.lds_doubles = k->lds_leaf_full / 8;
      // four waves per leaf: 0.246 -> 0.221 ms per factorisation at config 2, 0.644 -> 0.594 at config 3 against two (the
      // trailing update of a pivot step is the parallel part); one wave with LDS-only ordering 0.30 / 0.75, six or eight waves
      // no better than four and 4 % worse on the shuttle's blocks (profiles/r04_ipm_iter_time.txt)
      static const int leaf_waves = std::getenv("PYCOLLO_AMD_KKT_LEAF_WAVES") ? std::max(1, std::min(16, std::atoi(std::getenv("PYCOLLO_AMD_KKT_LEAF_WAVES")))) : 4;
      hipLaunchKernelGGL(kkt_leaf_factor, dim3(k->n_leaf), dim3(64 * leaf_waves), k->lds_leaf_full, st, la);
    }
    if (k->chain_cr) cr_levels_device<0>(k);
    else if (k->n_phase) hipLaunchKernelGGL(kkt_chain_factor, dim3(k->n_phase), dim3(64), k->lds_chain_factor, st, k->args);
    border_terms_device<false>(k);
    if (border_mode == 0) hipLaunchKernelGGL(kkt_border_factor<0>, dim3(1), dim3(256), k->lds_border, st, k->args);
    else hipLaunchKernelGGL(kkt_border_factor<1>, dim3(1), dim3(256), k->lds_border, st, k->args);
    KHIP(hipGetLastError());
    KHIP(hipMemcpyAsync(k->h_counts.p, k->counts.p, k->h_counts.n * sizeof(int), hipMemcpyDeviceToHost, st));
    kwait(st);
    int64_t p = 0, q = 0;
    const size_t n_counts = border_mode == 0 ? k->h_counts.n : k->h_counts.n - 2;   // (the last pair is the border's)
    for (size_t i = 0; i < n_counts; i += 2) {
      p += k->h_counts.p[i];
      q += k->h_counts.p[i + 1];
    }
    if (n_pos) *n_pos = (int32_t)p;
    if (n_neg) *n_neg = (int32_t)q;
    k->factored = true;
  }
}

int pc_kkt_factor(pc_kkt* k, int use_hess, const double* dvec, int32_t* n_pos, int32_t* n_neg) {
  return guarded([&] {
    if (!k || !dvec) throw std::runtime_error("null argument");
    KHIP(hipSetDevice(k->device));
    std::memcpy(k->h_a.p, dvec, k->nu * sizeof(double));
    KHIP(hipMemcpyAsync(k->dvec.p, k->h_a.p, k->nu * sizeof(double), hipMemcpyHostToDevice, k->stream));
    factor_device(k, use_hess, k->dvec.p, n_pos, n_neg);
  });
}

// ---- a rank's part of a factorisation cut across ranks (pycollo_amd/kkt_sharded.py) --------------------------------
// The handle holds a rank's leaves, chain segments and local border.  pc_kkt_factor_partial eliminates leaves and chain
// and hands out the border block with every Schur complement added ([nb][nb] doubles, lower triangle valid) for the
// reduction over ranks; pc_kkt_border_load_factor factorises a border block given from outside (the reduced system's
// handle has nothing else); pc_kkt_forward_partial / pc_kkt_backward_partial are the two halves of pc_kkt_solve around
// the reduced solve.  Host vectors; the matrices stay on the device.
int pc_kkt_factor_partial(pc_kkt* k, int use_hess, const double* dvec, double* border_out, int32_t* n_pos, int32_t* n_neg) {
  return guarded([&] {
    if (!k || !dvec || !border_out) throw std::runtime_error("null argument");
    KHIP(hipSetDevice(k->device));
    std::memcpy(k->h_a.p, dvec, k->nu * sizeof(double));
    KHIP(hipMemcpyAsync(k->dvec.p, k->h_a.p, k->nu * sizeof(double), hipMemcpyHostToDevice, k->stream));
    factor_device(k, use_hess, k->dvec.p, n_pos, n_neg, 1);
    if (k->nb) {
      KHIP(hipMemcpyAsync(border_out, k->vals.p + k->args.border_off, (size_t)k->nb * k->nb * sizeof(double),
                          hipMemcpyDeviceToHost, k->stream));
      KHIP(hipStreamSynchronize(k->stream));
    }
  });
}

int pc_kkt_border_load_factor(pc_kkt* k, const double* border, int32_t* n_pos, int32_t* n_neg) {
  return guarded([&] {
    if (!k || !border) throw std::runtime_error("null argument");
    KHIP(hipSetDevice(k->device));
    hipStream_t st = k->stream;
    if (k->nb)
      KHIP(hipMemcpyAsync(k->vals.p + k->args.border_off, border, (size_t)k->nb * k->nb * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(kkt_border_factor<2>, dim3(1), dim3(256), k->lds_border, st, k->args);
    KHIP(hipGetLastError());
    int* cnt = k->counts.p + 2 * ((size_t)k->n_leaf + k->n_chain);
    KHIP(hipMemcpyAsync(k->h_counts.p, cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    KHIP(hipStreamSynchronize(st));
    if (n_pos) *n_pos = k->h_counts.p[0];
    if (n_neg) *n_neg = k->h_counts.p[1];
    k->factored = true;
  });
}

int pc_kkt_forward_partial(pc_kkt* k, const double* rhs, double* border_rhs_out) {
  return guarded([&] {
    if (!k || !rhs || !border_rhs_out) throw std::runtime_error("null argument");
    if (!k->factored) throw std::runtime_error("pc_kkt_forward_partial before a factorisation");
    KHIP(hipSetDevice(k->device));
    hipStream_t st = k->stream;
    std::memcpy(k->h_a.p, rhs, k->nu * sizeof(double));
    KHIP(hipMemcpyAsync(k->vin.p, k->h_a.p, k->nu * sizeof(double), hipMemcpyHostToDevice, st));
    forward_device(k, k->vin.p);
    border_terms_device<true>(k);
    hipLaunchKernelGGL(kkt_border_solve<false>, dim3(1), dim3(256), k->lds_border, st, k->args);
    KHIP(hipGetLastError());
    if (k->nb) KHIP(hipMemcpyAsync(border_rhs_out, k->r.p + k->args.base_border, (size_t)k->nb * sizeof(double), hipMemcpyDeviceToHost, st));
    KHIP(hipStreamSynchronize(st));
  });
}

int pc_kkt_backward_partial(pc_kkt* k, const double* border_x, double* x) {
  return guarded([&] {
    if (!k || !border_x || !x) throw std::runtime_error("null argument");
    if (!k->factored) throw std::runtime_error("pc_kkt_backward_partial before a factorisation");
    KHIP(hipSetDevice(k->device));
    hipStream_t st = k->stream;
    if (k->nb) KHIP(hipMemcpyAsync(k->r.p + k->args.base_border, border_x, (size_t)k->nb * sizeof(double), hipMemcpyHostToDevice, st));
    backward_device(k, k->vout.p);
    KHIP(hipGetLastError());
    KHIP(hipMemcpyAsync(k->h_b.p, k->vout.p, k->nu * sizeof(double), hipMemcpyDeviceToHost, st));
    KHIP(hipStreamSynchronize(st));
    std::memcpy(x, k->h_b.p, k->nu * sizeof(double));
  });
}

// Exported chain nodes (pc_kkt_desc::chain_export): after pc_kkt_factor_partial their assembled panels, one after the
// other in ascending node order, each [nz][nz + nr + nb] row-major = [D | K(node, exported last node of its segment) | F]
// (nr = 0 for a last node or when the segment's last node is not exported); after pc_kkt_forward_partial their right-hand
// sides minus what the eliminated blocks owe them; before pc_kkt_backward_partial their solution from the reduced system.
static void export_copy(pc_kkt* k, double* host, int what) {   // 0: panels out, 1: right-hand sides out, 2: solution in
  KHIP(hipSetDevice(k->device));
  hipStream_t st = k->stream;
  size_t o = 0;
  for (int64_t c : k->export_nodes) {
    const int64_t nz = k->h_chain_ptr[c + 1] - k->h_chain_ptr[c];
    const int64_t b = k->h_cr_b[c], nr = b >= 0 ? k->h_chain_ptr[b + 1] - k->h_chain_ptr[b] : 0;
    if (what == 0) {
      const size_t cnt = (size_t)(nz * (nz + nr + k->nb));
      if (cnt) KHIP(hipMemcpyAsync(host + o, k->crbuf.p + k->h_crP_off[c], cnt * sizeof(double), hipMemcpyDeviceToHost, st));
      o += cnt;
    } else {
      double* dev = k->r.p + k->args.base_chain + k->h_chain_ptr[c];
      if (nz) KHIP(hipMemcpyAsync(what == 1 ? host + o : dev, what == 1 ? dev : host + o, (size_t)nz * sizeof(double),
                                  what == 1 ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, st));
      o += (size_t)nz;
    }
  }
  KHIP(hipStreamSynchronize(st));
}
int pc_kkt_export_panels(pc_kkt* k, double* out) {
  return guarded([&] {
    if (!k || !out) throw std::runtime_error("null argument");
    if (!k->factored) throw std::runtime_error("pc_kkt_export_panels before a factorisation");
    export_copy(k, out, 0);
  });
}
int pc_kkt_export_rhs(pc_kkt* k, double* out) {
  return guarded([&] {
    if (!k || !out) throw std::runtime_error("null argument");
    export_copy(k, out, 1);
  });
}
int pc_kkt_import_solution(pc_kkt* k, const double* in) {
  return guarded([&] {
    if (!k || !in) throw std::runtime_error("null argument");
    export_copy(k, const_cast<double*>(in), 2);
  });
}

// ---- the same with every vector in device memory (the device-resident interior-point iteration, pc_ipm.hip) ----
int pc_kkt_set_stream(pc_kkt* k, void* stream) {
  return guarded([&] {
    if (!k) throw std::runtime_error("null argument");
    KHIP(hipSetDevice(k->device));
    KHIP(hipStreamSynchronize(k->stream));
    if (stream) k->stream = (hipStream_t)stream;   // (the handle's own stream stays allocated; it is destroyed with the handle)
  });
}

int pc_kkt_factor_device(pc_kkt* k, int use_hess, const double* d_dvec, int32_t* n_pos, int32_t* n_neg) {
  return guarded([&] {
    if (!k || !d_dvec) throw std::runtime_error("null argument");
    KHIP(hipSetDevice(k->device));
    factor_device(k, use_hess, d_dvec, n_pos, n_neg);
  });
}

int pc_kkt_matvec_device(pc_kkt* k, int use_hess, const double* d_dvec, const double* d_x, double* d_y) {
  return guarded([&] {
    if (!k || !d_dvec || !d_x || !d_y) throw std::runtime_error("null argument");
    KHIP(hipSetDevice(k->device));
    matvec_device<0>(k, use_hess, d_dvec, d_x, nullptr, d_y);
  });
}

int pc_kkt_solve(pc_kkt* k, const double* rhs, double* x) {
  return guarded([&] {
    if (!k || !rhs || !x) throw std::runtime_error("null argument");
    if (!k->factored) throw std::runtime_error("pc_kkt_solve before pc_kkt_factor");
    KHIP(hipSetDevice(k->device));
    hipStream_t st = k->stream;
    std::memcpy(k->h_a.p, rhs, k->nu * sizeof(double));
    KHIP(hipMemcpyAsync(k->vin.p, k->h_a.p, k->nu * sizeof(double), hipMemcpyHostToDevice, st));
    solve_device(k, k->vin.p, k->vout.p);
    KHIP(hipMemcpyAsync(k->h_b.p, k->vout.p, k->nu * sizeof(double), hipMemcpyDeviceToHost, st));
    KHIP(hipStreamSynchronize(st));
    std::memcpy(x, k->h_b.p, k->nu * sizeof(double));
  });
}

int pc_kkt_matvec(pc_kkt* k, int use_hess, const double* dvec, const double* x, double* y) {
  return guarded([&] {
    if (!k || !dvec || !x || !y) throw std::runtime_error("null argument");
    KHIP(hipSetDevice(k->device));
    hipStream_t st = k->stream;
    std::memcpy(k->h_a.p, dvec, k->nu * sizeof(double));
    std::memcpy(k->h_b.p, x, k->nu * sizeof(double));
    KHIP(hipMemcpyAsync(k->w_dvec.p, k->h_a.p, k->nu * sizeof(double), hipMemcpyHostToDevice, st));
    KHIP(hipMemcpyAsync(k->vin.p, k->h_b.p, k->nu * sizeof(double), hipMemcpyHostToDevice, st));
    matvec_device<0>(k, use_hess, k->w_dvec.p, k->vin.p, nullptr, k->vout.p);
    KHIP(hipMemcpyAsync(k->h_a.p, k->vout.p, k->nu * sizeof(double), hipMemcpyDeviceToHost, st));   // (queued behind the kernel that read dvec)
    KHIP(hipStreamSynchronize(st));
    std::memcpy(y, k->h_a.p, k->nu * sizeof(double));
  });
}

// Solve K x = rhs with the current factors, refined against the system that has `dvec_true` on its diagonal -- the
// whole loop of the interior-point method's linear step (solve, residual, up to `max_steps` corrections, each kept
// only while it at least halves the residual's 2-norm and stays finite) in ONE call: rhs and dvec_true go up once, x
// comes down once, and per correction only two doubles (the residual norm and a non-finite count) cross the bus.
// Round 2 made eight host calls of this (k.solve / k.matvec), each with its own vector copies and stream wait.
// the refined solve with right-hand side and true diagonal in device memory; returns the device vector that holds x
static double* solve_refined_device(pc_kkt* k, int use_hess, const double* d_dvec, const double* d_rhs, int max_steps, int* n_solves) {
  {
    hipStream_t st = k->stream;
    const int64_t nu = k->nu;
    const unsigned nbk = (unsigned)((nu + 255) / 256);
    const int nred = (int)std::min<int64_t>(256, (nu + 255) / 256);
    auto norms = [&](const double* d_r, const double* d_t, double& sumsq, double& bad) {
      hipLaunchKernelGGL(kkt_norm_partial, dim3(nred), dim3(256), 0, st, d_r, d_t, k->w_part.p, nu);
      hipLaunchKernelGGL(kkt_norm_final, dim3(1), dim3(64), 0, st, k->w_part.p, nred, k->w_norm.p);
      KHIP(hipMemcpyAsync(k->h_norm.p, k->w_norm.p, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
      kwait(st);
      sumsq = k->h_norm.p[0];
      bad = k->h_norm.p[1];
    };
    double* sol = k->w_sol.p;
    double* res = k->w_res.p;
    double* trial = k->w_trial.p;
    double* res_t = k->w_dx.p;      // (the correction is formed in vout, the trial's residual lands here)
    solve_device(k, d_rhs, sol);
    matvec_device<1>(k, use_hess, d_dvec, sol, d_rhs, res);
    int solves = 1;
    double nres = 0.0, bad = 0.0, nrhs = 0.0;
    if (max_steps > 0) {   // ||rhs||^2 rides in the same wait as the first residual's norm
      hipLaunchKernelGGL(kkt_norm_partial, dim3(nred), dim3(256), 0, st, d_rhs, d_rhs, k->w_part.p + 2 * 256, nu);
      hipLaunchKernelGGL(kkt_norm_final, dim3(1), dim3(64), 0, st, k->w_part.p + 2 * 256, nred, k->w_norm.p + 2);
      hipLaunchKernelGGL(kkt_norm_partial, dim3(nred), dim3(256), 0, st, res, sol, k->w_part.p, nu);
      hipLaunchKernelGGL(kkt_norm_final, dim3(1), dim3(64), 0, st, k->w_part.p, nred, k->w_norm.p);
      KHIP(hipMemcpyAsync(k->h_norm.p, k->w_norm.p, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
      kwait(st);
      nres = k->h_norm.p[0];
      bad = k->h_norm.p[1];
      nrhs = k->h_norm.p[2];
    }
    const double stop2 = k->resid_tol * k->resid_tol * nrhs;   // (NaN or 0 right-hand side: never true, the loop below decides)
    for (int it = 0; it < max_steps; ++it) {
      if (nres <= stop2) break;                                 // already as accurate as a correction could make it matter
      solve_device(k, res, k->vout.p);
      hipLaunchKernelGGL(kkt_add, dim3(nbk), dim3(256), 0, st, sol, k->vout.p, trial, nu);
      matvec_device<1>(k, use_hess, d_dvec, trial, d_rhs, res_t);
      ++solves;
      double nt = 0.0, bad_t = 0.0;
      norms(res_t, trial, nt, bad_t);
      // numpy semantics of the loop this replaces: a NaN norm compares false, i.e. the trial is kept unless it is
      // non-finite itself or fails to halve the residual
      if (bad_t > 0.0 || std::sqrt(nt) >= 0.5 * std::sqrt(nres)) break;
      std::swap(sol, trial);
      std::swap(res, res_t);
      nres = nt;
    }
    if (n_solves) *n_solves = solves;
    return sol;
  }
}

int pc_kkt_solve_refined(pc_kkt* k, int use_hess, const double* dvec_true, const double* rhs, int max_steps, double* x,
                         int32_t* n_solves) {
  return guarded([&] {
    if (!k || !dvec_true || !rhs || !x) throw std::runtime_error("null argument");
    if (!k->factored) throw std::runtime_error("pc_kkt_solve_refined before pc_kkt_factor");
    KHIP(hipSetDevice(k->device));
    hipStream_t st = k->stream;
    const int64_t nu = k->nu;
    std::memcpy(k->h_a.p, rhs, nu * sizeof(double));
    std::memcpy(k->h_b.p, dvec_true, nu * sizeof(double));
    KHIP(hipMemcpyAsync(k->w_rhs.p, k->h_a.p, nu * sizeof(double), hipMemcpyHostToDevice, st));
    KHIP(hipMemcpyAsync(k->w_dvec.p, k->h_b.p, nu * sizeof(double), hipMemcpyHostToDevice, st));
    int solves = 0;
    const double* sol = solve_refined_device(k, use_hess, k->w_dvec.p, k->w_rhs.p, max_steps, &solves);
    KHIP(hipMemcpyAsync(k->h_b.p, sol, nu * sizeof(double), hipMemcpyDeviceToHost, st));
    KHIP(hipStreamSynchronize(st));
    std::memcpy(x, k->h_b.p, nu * sizeof(double));
    if (n_solves) *n_solves = solves;
  });
}

int pc_kkt_solve_refined_device(pc_kkt* k, int use_hess, const double* d_dvec_true, const double* d_rhs, int max_steps,
                                double* d_x, int32_t* n_solves) {
  return guarded([&] {
    if (!k || !d_dvec_true || !d_rhs || !d_x) throw std::runtime_error("null argument");
    if (!k->factored) throw std::runtime_error("pc_kkt_solve_refined_device before a factorisation");
    KHIP(hipSetDevice(k->device));
    int solves = 0;
    const double* sol = solve_refined_device(k, use_hess, d_dvec_true, d_rhs, max_steps, &solves);
    KHIP(hipMemcpyAsync(d_x, sol, k->nu * sizeof(double), hipMemcpyDeviceToDevice, k->stream));
    if (n_solves) *n_solves = solves;
  });
}

}  // extern "C"
