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
#include <utility>

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
// threshold of the two-pass build (see pc::bulk)
#ifndef PC_SPLIT_MIN
#define PC_SPLIT_MIN 90
#endif

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
  // ---- parameters of the node functions, v[NZ + l]: everything in f, p, g that is not a node variable, in x
  //      order [integral variables | free times | static parameters] (M::wk = 1 / 2 / 0, M::wi = index within the
  //      kind).  The live reference keeps q, t0, tF, s global inside f, p, g (backend.py:1526-1539).  A model
  //      without q / t dependence has static parameters only and every helper below reduces to its old form.
  static constexpr bool is_t(int l) { return M::wk(l) == 2; }
  static constexpr bool is_q(int l) { return M::wk(l) == 1; }
  static constexpr int NWT = [] {
    int n = 0;
    for (int l = 0; l < NS; ++l) n += is_t(l) ? 1 : 0;
    return n;
  }();
  static constexpr int tpar(int jt) {   // parameter index of free time jt, or -1
    for (int l = 0; l < NS; ++l)
      if (is_t(l) && M::wi(l) == jt) return l;
    return -1;
  }
  static constexpr int qpar(int m) {    // parameter index of integral variable m, or -1
    for (int l = 0; l < NS; ++l)
      if (is_q(l) && M::wi(l) == m) return l;
    return -1;
  }
  // all parameters a row depends on (path rows: every one of them is a column of its own)
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
  // defect rows end with [q the state equation depends on | the NT free times, always | s it depends on]; a
  // dependence on a time parameter adds to that time's column instead of opening one
  static constexpr int nqdep(int r) {
    int n = 0;
    for (int l = 0; l < NS; ++l) n += (is_q(l) && dep(r, NZ + l)) ? 1 : 0;
    return n;
  }
  static constexpr int nxdep(int r) {   // columns besides the times
    int n = 0;
    for (int l = 0; l < NS; ++l) n += (!is_t(l) && dep(r, NZ + l)) ? 1 : 0;
    return n;
  }
  static constexpr int xpos(int r, int l) {   // position of a non-time parameter's column in that tail of the row
    int n = 0;
    for (int ll = 0; ll < l; ++ll) n += (!is_t(ll) && dep(r, NZ + ll)) ? 1 : 0;
    return n + (is_q(l) ? 0 : NT);
  }
  static constexpr int C(int a) { return (own_sparse(a) ? 2 : 0) + NT + nxdep(a); }
  // number of df/dw entries that must be staged for the section contraction, and their slot
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
  // index of the Hessian entry (row, col) in v = [z | w], or -1
  static constexpr int hidx(int r, int c) {
    for (int e = 0; e < NH; ++e)
      if (M::hr(e) == r && M::hc(e) == c) return e;
    return -1;
  }
  // (t, z_c) strip: through the stretch factor, or through a second partial of a time parameter
  static constexpr bool tzx(int c) {
    if (tz(c)) return true;
    for (int l = 0; l < NS; ++l)
      if (is_t(l) && hidx(NZ + l, c) >= 0) return true;
    return false;
  }
  // rank of an edge-node entry site among the flagged ones (M::efl), node 0's sites first
  static constexpr int erank(int i) {
    int n = 0;
    for (int k = 0; k < i; ++k) n += M::efl(k) ? 1 : 0;
    return n;
  }
  // ---- reductions (per tile): [NQ] sum w g | [NQ*NS] sum w dg/dw | [NS] t-w | [NS*(NS+1)/2] w-w   (w: parameters)
  //      Only the sums with structure exist (a phase with five parameters would otherwise carry 32 of them, most
  //      identically zero, through every tile and through the tail's registers): rqs / rts / rss give a sum's index,
  //      counted in this order; pcp::finalize_phase_tables counts the same way on the host.
  static constexpr int R_Q = 0;
  static constexpr bool has_rts(int l) { return NT > 0 && tz(NZ + l); }
  static constexpr int rqs(int m, int l) {   // sum_i w_i dg_m/dw_l
    int n = NQ;
    for (int mm = 0; mm < NQ; ++mm)
      for (int ll = 0; ll < NS; ++ll) {
        if (mm == m && ll == l) return n;
        n += dep(NY + NP + mm, NZ + ll) ? 1 : 0;
      }
    return n;   // (m, l) = (NQ, 0): one past the last
  }
  static constexpr int rts(int l) {          // sum_i (mu . dF/dw_l)(z_i), f and g rows
    int n = rqs(NQ, 0);
    for (int ll = 0; ll < l; ++ll) n += has_rts(ll) ? 1 : 0;
    return n;
  }
  static constexpr int rss(int l, int l2) {  // sum_i d2(node Lagrangian)/dw_l dw_l2
    int n = rts(NS);
    for (int e = 0; e < NH; ++e)
      if (M::hc(e) >= NZ) {
        if (M::hr(e) == NZ + l && M::hc(e) == NZ + l2) return n;
        ++n;
      }
    return n;   // not present: one past the last
  }
  static constexpr int NRED = rss(NS, NS);
  // ---- packed scaling offsets
  static constexpr int O_VZ = 0, O_RZ = NZ, O_VQ = 2 * NZ, O_RQ = 2 * NZ + NQ, O_VT = 2 * NZ + 2 * NQ,
                       O_RT = O_VT + 2, O_VS = O_RT + 2, O_RS = O_VS + NS, O_WD = O_RS + NS, O_WP = O_WD + NY,
                       O_WI = O_WP + NP, NSCAL = O_WI + NQ;
  // goff / hoff layout
  static constexpr int GO_D = 0, GO_P = NY, GO_Q = NY + NP;
  static constexpr int HO_Z = 0, HO_T = NZ, HO_S = 3 * NZ;

  // ---- output items of a tile, dealt to the W waves ("replicas") that share it.  One item = one group of
  //      CSR runs a single replica produces: a state's defect rows (c~ and its G~ block), a path row, an
  //      integral row (its G~ entries and partial sums), a Hessian row block, the s-/t-strip of one z column,
  //      the (t,s)/(s,s) partial sums.  Items are weighed by the entries they emit per node and assigned
  //      largest-first to the least loaded replica; Hessian items break ties towards the last replica and the
  //      others towards the first, so that replicas without a Hessian item exist and skip the adjoint weights.
  static constexpr int IT_D = 0, IT_P = NY, IT_Q = NY + NP, IT_HB = NY + NP + NQ, IT_HS = IT_HB + NZ,
                       IT_HT = IT_HS + NZ, IT_HSUM = IT_HT + NZ, NITEMS = IT_HSUM + 1;
  struct Deal {
    int own[NITEMS];
    unsigned hmask;   // bit w: replica w owns a Hessian item
  };
  template <int W, int NN>
  static constexpr Deal deal() {
    Deal d{};
    int wt[NITEMS] = {};
    bool done[NITEMS] = {};
    for (int a = 0; a < NY; ++a) wt[IT_D + a] = D(a) * NN + C(a) + 1;
    for (int m = 0; m < NP; ++m) wt[IT_P + m] = nzdep(NY + m) + nsdep(NY + m) + 1;
    for (int m = 0; m < NQ; ++m) wt[IT_Q + m] = nzdep(NY + NP + m) + 1;
    for (int b = 0; b < NZ; ++b) {
      wt[IT_HB + b] = hrow_count(b) > 0 ? hrow_count(b) + 1 : 0;
      int ns = 0;
      for (int e = 0; e < NH; ++e) ns += (M::hr(e) >= NZ && !is_t(M::hr(e) - NZ) && M::hc(e) == b) ? 1 : 0;
      wt[IT_HS + b] = ns;
      wt[IT_HT + b] = tzx(b) ? NT : 0;
    }
    wt[IT_HSUM] = NS > 0 ? NS + NS * (NS + 1) / 2 : 0;
    int load[W] = {};
    d.hmask = 0;
    for (int it = 0; it < NITEMS; ++it) {
      int best = -1;
      for (int i = 0; i < NITEMS; ++i)
        if (!done[i] && (best < 0 || wt[i] > wt[best])) best = i;
      done[best] = true;
      const bool hess = best >= IT_HB;
      int bin = hess ? W - 1 : 0;
      for (int k = 0; k < W; ++k) {
        const int b = hess ? W - 1 - k : k;
        if (load[b] < load[bin]) bin = b;
      }
      d.own[best] = wt[best] > 0 ? bin : 0;
      load[bin] += wt[best];
      if (hess && wt[best] > 0) d.hmask |= 1u << bin;
    }
    return d;
  }
  // which item a partial sum belongs to
  static constexpr int red_item(int r) {
    if (r < NQ) return IT_Q + r;
    for (int m = 0; m < NQ; ++m)
      for (int l = 0; l < NS; ++l)
        if (dep(NY + NP + m, NZ + l) && rqs(m, l) == r) return IT_Q + m;
    return IT_HSUM;
  }
};

// (LDS carve-up: LdsPlan / lds_plan live in pc_args.h, shared with the host's size query)

// Sum over the 64 lanes of a wave in a fixed order; the result is valid in lane 0.
// Steps 1..16 use DPP row shifts / a row mirror-free butterfly inside 16-lane rows (VALU speed), the last
// two steps cross rows with v_readlane.  (A chain of six ds_bpermute shuffles costs ~130 cycles a step.)
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)(unsigned)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  // row_shr:n (0x110 + n) with bound_ctrl: lanes shifted in from outside the row read 0
  v += dpp_mov<0x111>(v);   // lane i += lane i-1
  v += dpp_mov<0x112>(v);   // += lane i-2 (of the partial sums)
  v += dpp_mov<0x114>(v);   // += i-4
  v += dpp_mov<0x118>(v);   // += i-8 : lane 15 of every row holds the row total
  const long long b = __builtin_bit_cast(long long, v);
  double tot = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 16 * r + 15);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 16 * r + 15);
    tot += __builtin_bit_cast(double, ((long long)(unsigned)hi << 32) | (long long)(unsigned)lo);
  }
  return tot;   // uniform: every lane holds the wave total
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits for
// every global store the wave has in flight -- a micro-second stall per use once the kernel has started
// writing its outputs.  Here only lgkmcnt (LDS) is waited for; global stores keep streaming.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Cooperative, fully coalesced copy of a staged CSR run from LDS to HBM: 16 B per lane and instruction (1 KiB per
// wave instruction), counted from the run's FIRST element.  A run starts wherever the CSR layout puts it (8-byte
// granularity): the 16-byte stores are simply issued at that address -- global memory takes them unaligned, a wave's
// 64 stores still cover one contiguous KiB -- and the LDS side is read as two 8-byte halves.  Every batch of
// PC_FLUSH_DEPTH wave-instructions issues ALL its LDS reads before the first store: a run costs one LDS round trip per
// batch and nothing else.  (Measured and dropped, records in profiles/r03_ab_*.txt: an alignment peel, a
// software-pipelined batch loop, non-temporal stores, a predicate on every store; superseded in round 4,
// profiles/r04_flush_buffer_ab.txt: lane predicates on the partial chunk and a separate odd last element.)
typedef double pc_d2_a8 __attribute__((ext_vector_type(2), aligned(8)));
#ifndef PC_FLUSH_DEPTH
#define PC_FLUSH_DEPTH 4
#endif
typedef int pc_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void flush_run(double* __restrict__ dst, const double* __restrict__ src, int len, int tid,
                                          int TB) {
  // The length is the same in every lane, but on a mesh of mixed orders it is computed from LDS tables, i.e. in a vector
  // register: the compiler then treats the batch loop and the chunk tests below as divergent (exec-mask loops).
  len = __builtin_amdgcn_readfirstlane(len);
  if (len <= 0) return;
  // The stores go through a buffer descriptor of exactly `len` doubles: the hardware's range check is per dword
  // (tools/bufstore_probe.hip: every length and both alignments on gfx950), so the partial last store instruction, its odd
  // last element and the lanes beyond the run's end need no predicate -- no compares, no exec-mask save / restore, no
  // separate 8-byte tail.  The check covers the vector offset and the immediate, not the scalar offset: everything that
  // varies is in the vector offset.  (The predicated version was 35 % of the instructions of a Delta III tile body.)
  const unsigned long long d64 = (unsigned long long)dst;
  double* base = (double*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(d64 >> 32)) << 32) |
                           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)d64));
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, len * 8, 0x00020000);
  const int pairs = (len + 1) >> 1;
  const pc_d2_a8* sp = reinterpret_cast<const pc_d2_a8*>(src) + tid;
  int voff = 16 * tid;
  int b0 = 0;
  for (; b0 + PC_FLUSH_DEPTH * TB <= pairs; b0 += PC_FLUSH_DEPTH * TB, voff += PC_FLUSH_DEPTH * TB * 16) {
    pc_d2_a8 a[PC_FLUSH_DEPTH];
#pragma unroll
    for (int q = 0; q < PC_FLUSH_DEPTH; ++q) a[q] = sp[b0 + q * TB];
#pragma unroll
    for (int q = 0; q < PC_FLUSH_DEPTH; ++q)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pc_i4, a[q]), rs, voff + q * TB * 16, 0, 0);
  }
  if (b0 < pairs) {   // the last, partial batch: all its LDS reads first (past the run's end they fetch what is never stored)
    pc_d2_a8 a[PC_FLUSH_DEPTH];
#pragma unroll
    for (int q = 0; q < PC_FLUSH_DEPTH; ++q) a[q] = sp[b0 + q * TB];
#pragma unroll
    for (int q = 0; q < PC_FLUSH_DEPTH; ++q)
      if (b0 + q * TB < pairs)   // (wave-uniform: a scalar branch around a chunk that lies wholly past the end)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pc_i4, a[q]), rs, voff + q * TB * 16, 0, 0);
  }
}

// The same copy for a run staged in pieces: `nq` chunks of L doubles lie back to back in LDS and go to dst, dst + GS,
// dst + 2 GS, ... (pc::bulk, "row groups": chunk = the rows of one pass of one section, GS = all rows of a section).
// One chunk at a time, in a scalar loop: a wave moves a chunk with ceil(L / 128) 16-byte store instructions whose lane
// offsets never change (the chunk's base advances in scalar registers) and whose last one is cut by the descriptor's
// range check -- two instructions per KiB, like flush_run.  (A first version mapped staging element e to chunk e / L per lane:
// eight address instructions per store, a third more instructions in the whole tile body -- Delta III order 5, two
// passes: 23.3 -> 34.2 us.)  The next chunk's LDS reads are issued before this chunk's stores.
template <int L, int GS>
__device__ __forceinline__ void flush_chunks(double* __restrict__ dst, const double* __restrict__ src, int nq, int t, int TN) {
  nq = __builtin_amdgcn_readfirstlane(nq);
  if (nq <= 0) return;
  if (TN == 64) {
    // as flush_run: a descriptor of exactly one chunk (its base advances in scalar registers), the last store instruction
    // of a chunk cut by the range check instead of a lane mask
    constexpr int NR = (L + 127) / 128;
    const unsigned long long d64 = (unsigned long long)dst;
    double* base = (double*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(d64 >> 32)) << 32) |
                             (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)d64));
    const double* sp = src + 2 * t;
    const int voff = 16 * t;
    pc_d2_a8 a[NR], nx[NR];
    auto fetch = [&](pc_d2_a8* r, const double* p) {
#pragma unroll
      for (int i = 0; i < NR; ++i) r[i] = *reinterpret_cast<const pc_d2_a8*>(p + 128 * i);
    };
    auto store = [&](const pc_d2_a8* r, double* b) {
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(b, 0, L * 8, 0x00020000);
#pragma unroll
      for (int i = 0; i < NR; ++i) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pc_i4, r[i]), rs, voff + 1024 * i, 0, 0);
    };
    // two chunks per trip, the two register sets taking turns (no copies): the next chunk's LDS reads are issued before
    // this chunk's stores
    fetch(a, sp);
    int q = 0;
    for (; q + 2 <= nq; q += 2) {
      fetch(nx, sp + L);
      store(a, base);
      if (q + 2 < nq) fetch(a, sp + 2 * L);
      store(nx, base + GS);
      sp += 2 * L;
      base += 2 * GS;
    }
    if (q < nq) store(a, base);
    return;
  }
  // wider workgroups (tiles of 128 / 256 nodes): the same per chunk, a runtime stride
  for (int q = 0; q < nq; ++q) {
    const double* sp = src + (long long)q * L;
    double* dp = dst + (long long)q * GS;
    for (int o = 2 * t; o < L; o += 2 * TN) {
      if (o + 1 < L) *reinterpret_cast<pc_d2_a8*>(dp + o) = *reinterpret_cast<const pc_d2_a8*>(sp + o);
      else dp[o] = sp[o];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bulk kernel
// ---------------------------------------------------------------------------------------------
// Every uniform kernel input, copied out of the kernarg segment in the entry block.  hipcc otherwise
// loads each field lazily (s_load + s_waitcnt right before its first use, behind whatever branch that is
// in): dozens of serial scalar-cache round trips, which is most of the run time of a one-wave tile.
template <class St>
struct BulkIn {
  const double* x; const double* lam; double* c; double* G; double* H;
  const double* xz; const double* lamd;   // x + x_off, lam + c_off
  const int32_t* tile_k0; const int32_t* tile_n0; const int32_t* sec_s; const double* sec_h; const int64_t* sec_E;
  const double* qa; const double* qw; const int64_t* hslot0; const int64_t* hslotN; double* partials;
  unsigned long long* gran; unsigned long long* erec; const double* tab;
  int64_t x_off, s_off, c_off, c_path_off, c_int_off;
  double t_fixed[2];
  int32_t N, K, flags, qa_total, qw_total, tile_begin, uni_n, spt, lds_out, wpt, n_blocks, block_threads, qa0, qw0, qwabs;
  uint32_t epoch;
  int32_t erec0;
  double scal[St::NSCAL > 0 ? St::NSCAL : 1];
  int64_t goff[St::NFN > 0 ? St::NFN : 1];
  int64_t hoff[3 * St::NZ + St::NS * St::NZ > 0 ? 3 * St::NZ + St::NS * St::NZ : 1];
};

#define PC_PIN(v) asm volatile("" ::"s"(v))
#ifndef PC_PIN_BUDGET
#define PC_PIN_BUDGET 40   // SGPRs the per-variable constants may take
#endif
// One empty asm per value would do the forcing too, but every inline asm is a scheduling boundary: the loads
// then come in several dependent batches (measured: 6 loads, wait, 2 loads, wait, 2 loads, wait ...).  Folding all
// values into one word that a single asm consumes leaves the scheduler free to issue every load at once.
struct PinAcc {
  unsigned long long h = 0;
  template <class T>
  __device__ __forceinline__ void operator()(T* v) { h ^= (unsigned long long)reinterpret_cast<uintptr_t>(v); }
  __device__ __forceinline__ void operator()(double v) { h ^= __builtin_bit_cast(unsigned long long, v); }
  __device__ __forceinline__ void operator()(long long v) { h ^= (unsigned long long)v; }
  __device__ __forceinline__ void operator()(long v) { h ^= (unsigned long long)v; }
  __device__ __forceinline__ void operator()(int v) { h ^= (unsigned long long)(unsigned)v; }
  __device__ __forceinline__ void done() const { asm volatile("" ::"s"(h)); }
};
template <class T, int N, int... I>
__device__ __forceinline__ void pin_array_impl(PinAcc& p, const T (&a)[N], std::integer_sequence<int, I...>) {
  (p(a[I]), ...);
}
template <int CNT, class T, int N>
__device__ __forceinline__ void pin_array(PinAcc& p, const T (&a)[N]) {
  pin_array_impl(p, a, std::make_integer_sequence<int, CNT>{});
}

// UN > 0: the phase's mesh has UN nodes in every section and the kernel is compiled for exactly that
// order (loops over section rows / nodes unroll, their LDS reads issue back to back, index arithmetic
// divides by constants).  UN == 0: any mesh (orders may differ section by section).
//
// RES = false: the per-phase bulk kernel; a separate `pc_tail` launch finishes the evaluation (what a rank of the
// section-sharded evaluation runs: its partial sums travel through the all-gather first).
// RES = true: the "resident tail" build -- the launch carries one more workgroup (block 0) that runs the tail
// concurrently with the tiles, so an evaluation is ONE launch and the tail's own work overlaps the bulk's.  The
// tail needs two kinds of values other workgroups of the same launch produce: the per-tile partial sums and the
// Hessian entries of the edge nodes 0 / N-1 that endpoint terms are added to.  Both are handed over as *granules*
// (cdna_hip_programming.md G16, recipe R2 "the data IS the flag"): an aligned 8-byte word {epoch : 32, half of the
// double : 32}, written with ONE agent-scope (sc1, write-through) store and polled with agent-scope loads until the
// tag equals this launch's epoch.  No release fence, no store drain, no atomics, no counter on the producer side:
// a tile publishes as soon as the values exist (before its bulky c~ / G~ stores) and never waits.  The tail
// workgroup waits for producers that themselves wait for nothing, so no co-residency is required; its spin is
// bounded and reports through PcTailArgs::timeout.
__device__ __forceinline__ void publish_granules(unsigned long long* g, unsigned epoch, double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v), e = (unsigned long long)epoch << 32;
  __hip_atomic_store(g, e | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(g + 1, e | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one attempt, in two steps so that a caller can request many granules before it looks at the first:
// load_granule = one agent-scope load; join_granules = true when both halves carry this launch's tag
__device__ __forceinline__ unsigned long long load_granule(const unsigned long long* g) {
  return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool join_granules(unsigned long long lo, unsigned long long hi, unsigned epoch, double& v) {
  v = __builtin_bit_cast(double, (hi << 32) | (lo & 0xffffffffull));
  return (unsigned)(lo >> 32) == epoch && (unsigned)(hi >> 32) == epoch;
}
#ifndef PC_SPIN_LIMIT
#define PC_SPIN_LIMIT (1 << 21)   // passes of ~1 us each before the tail gives up (an evaluation takes < 1 ms)
#endif

// Force compile-time evaluation of a structure helper: `constexpr` alone lets the compiler emit the helper's search
// loop at run time (it did, 134 times in the Delta III kernel), and an array subscripted by such a run-time value
// is demoted from registers to scratch memory.
#define PC_CE(expr) (std::integral_constant<int, (expr)>::value)

// Workgroups are handed to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and every XCD has its own L2.
// Neighbouring tiles share a halo node on the read side and, on the write side, the cache line in which one tile's
// CSR run ends and the next one's begins; mapping consecutive *tiles* to the same XCD keeps both in one L2
// (the shared lines are merged before they are written back).  b -> position in an XCD-major order.
#ifndef PC_XCD_SWIZZLE
#define PC_XCD_SWIZZLE 1
#endif
__device__ __forceinline__ int xcd_major(int b, int nb) {
#if PC_XCD_SWIZZLE
  constexpr int NX = 8;
  const int q = nb / NX, r = nb % NX, x = b % NX, j = b / NX;
  return x * q + (x < r ? x : r) + j;
#else
  (void)nb;
  return b;
#endif
}

// Diagnostic build only (-DPC_STAMPS, tools/stamps.py): lane 0 of every wave writes the shader clock at a handful of
// points of the tile body into a device array of the code object -- memory nothing else reads.  Never in a shipped
// object: the stamps cost a scalar-memory round trip each.
#ifdef PC_STAMPS
#ifndef PC_STAMPS_WAVES
#define PC_STAMPS_WAVES 16384
#endif
extern "C" __device__ unsigned long long pc_stamps[PC_STAMPS_WAVES * 24];
#define PC_STAMP(i)                                                                                         \
  do {                                                                                                      \
    if ((threadIdx.x & 63) == 0 && pc_stamp_slot < PC_STAMPS_WAVES)                                         \
      pc_stamps[(size_t)pc_stamp_slot * 24 + (i)] = (i) == 9 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define PC_STAMP(i) do { } while (0)
#endif

// MULTI: one launch covers every phase (pc_bulk_all); KA then lives in device memory and the per-call pointers, the
// flags and the granule tag arrive as plain values (mx .. mepoch: the members of PcMultiArgs the kernel needs --
// handed over as a pointer to the struct, a large kernel keeps the whole struct in scratch memory), the workgroup's
// tile from its index relative to the phase's first block.
//
// MIX: the phase's sections differ in order and this code object carries one tile body per frequent order next to the
// any-order one (bulk_mix below picks per tile).  The host cuts tiles at order changes where the orders come in runs --
// which is what ph refinement leaves: subdivided and merged stretches are runs of the minimum order,
// pycollo/mesh_refinement.py:252-321 -- so such a tile runs the body compiled for its order (UN > 0), its place in the
// mesh arriving as scalars from the tile's record (r_*: PcTileRec) instead of section tables.  The one thing an
// order-pure tile still owes a neighbour of another order is its first node, which closes the section before the
// tile: that lane's adjoint weight and quadrature weight take the previous section's own last A column / weight.
template <class M, int UN, bool RES = false, int WN = 0, int WIDX = 0, bool MIX = false>
__device__ __forceinline__ void bulk(const PcPhaseArgs& KA, bool MULTI = false, int first_block = 0, int block = -1,
                                     const PcLead* LD = nullptr, const double* mx = nullptr, const double* mlam = nullptr,
                                     double* mc = nullptr, double* mG = nullptr, double* mH = nullptr, int mflags = 0,
                                     unsigned mepoch = 0, int r_k0 = 0, int r_nsec = 0, int r_n0 = 0, int r_nprev = 0,
                                     int r_qaprev = 0, long long r_E0 = 0, double r_wprev = 0.0) {
  using St = S<M>;
  constexpr int NY = St::NY, NZ = St::NZ, NQ = St::NQ, NP = St::NP, NS = St::NS, NT = St::NT;
  constexpr int NFN = St::NFN, NV = St::NV, NJ = St::NJ, NH = St::NH, NFS = St::NFS, NRED = St::NRED;
  constexpr bool PURE = MIX && UN > 0;   // order-pure tile of a mixed mesh: geometry relative to the tile's record
  BulkIn<St> A;
  // part 1: what the node loads are addressed with.  pc_bulk_p<i> gets these as leading scalar arguments, which the
  // command processor preloads into SGPRs (LD); the other launches read them from their argument block like the rest.
  if (LD) {
    const int wa = LD->wa, wb = LD->wb;
    A.xz = LD->xz; A.lamd = LD->lamd; A.qa = LD->qa; A.sec_h = LD->sec_h; A.N = LD->N; A.K = LD->K;
    A.tile_begin = LD->tile_begin; A.n_blocks = LD->n_blocks;
    A.flags = wa & 0xff; A.wpt = (wa >> 8) & 0xf; A.block_threads = ((wa >> 12) & 0xf) << 6; A.spt = (wa >> 16) & 0xfff;
    A.qa0 = wb & 0xffff; A.qwabs = wb >> 16;
  } else {
    A.x = MULTI ? mx : KA.x; A.lam = MULTI ? mlam : KA.lam; A.x_off = KA.x_off; A.c_off = KA.c_off; A.N = KA.N; A.K = KA.K;
    A.xz = A.x + A.x_off; A.lamd = A.lam + A.c_off; A.qa = KA.qa; A.sec_h = KA.sec_h;
    A.tile_begin = KA.tile_begin; A.spt = KA.spt; A.n_blocks = KA.n_blocks;
    A.flags = MULTI ? mflags : KA.flags; A.wpt = KA.wpt; A.block_threads = KA.block_threads;
    A.qa0 = KA.qa_off[UN > 0 ? UN : 0]; A.qwabs = KA.qa_total + KA.qw_off[UN > 0 ? UN : 0];
  }
  if constexpr (UN == 0) { A.uni_n = KA.uni_n; A.tile_k0 = KA.tile_k0; A.tile_n0 = KA.tile_n0; A.sec_s = KA.sec_s; }
  // ---- this workgroup's tile and this lane's node, then the node loads, before anything else: with the lead
  //      scalars preloaded (pc_bulk_p<i>) their addresses need no scalar load, so the longest latency of the
  //      prologue starts at the wave's first instructions and everything below overlaps it
#ifdef PC_STAMPS
  const int pc_stamp_slot = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
#endif
  PC_STAMP(0);
  PC_STAMP(9);   // constant-rate clock at the start (the pair 0 / 9 of two waves gives the shader clock rate)
  const int N = A.N;
  // (a single-phase launch is swizzled here; pc_bulk_all swizzles before it picks the phase and passes `first_block`
  //  relative to the swizzled index, together with that index)
  const int blk = block >= 0 ? block : xcd_major((int)blockIdx.x, A.n_blocks);
  const int tile = blk - first_block + A.tile_begin;
  // tile geometry: index arithmetic on a uniform mesh, two small tables otherwise
  const int un = UN > 0 ? UN : A.uni_n;
  const bool uni = UN > 0 || un > 0;
  int k0, k1, n0, n1;
  if constexpr (PURE) {
    k0 = r_k0;
    k1 = r_k0 + r_nsec;
    n0 = r_n0;
    n1 = r_n0 + r_nsec * (UN - 1);
  } else if (uni) {
    k0 = tile * A.spt;
    k1 = min(k0 + A.spt, A.K);
    n0 = k0 * (un - 1);
    n1 = k1 * (un - 1);
  } else {
    k0 = A.tile_k0[tile];
    k1 = A.tile_k0[tile + 1];
    n0 = A.tile_n0[tile];
    n1 = A.tile_n0[tile + 1];
  }
  const int T = n1 - n0;                 // defect rows per state in this tile; nodes n0 .. n0+T
  const int tid = threadIdx.x;
  const int W = WN > 0 ? WN : A.wpt;                          // 1, 2 or 4
  const int t = W > 1 ? (tid & 63) : tid;                     // node slot of this lane inside its replica
  const bool active = t <= T;
  const int node = n0 + t;
  double v[NV > 0 ? NV : 1];
  if (active) {
    static_for<0, NZ>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      v[b] = A.xz[(int64_t)b * N + node];
    });
  }
  // part 2: everything else, fetched while the node loads are in flight.  With preloaded lead scalars the argument
  // block is read through the kernarg segment pointer passed through an empty asm, so that the scalar loads below
  // (and the wait their first use needs) depend on a point after the node loads and cannot be hoisted above them.
  auto part2 = [&](const auto& KB) {
    if (MULTI) {
      A.c = mc; A.G = mG; A.H = mH;
    } else {
      A.c = KB.c; A.G = KB.G; A.H = KB.H;
    }
    if (LD) { A.x = KB.x; A.lam = KB.lam; A.x_off = KB.x_off; A.c_off = KB.c_off; }
    if constexpr (UN > 0) { A.uni_n = KB.uni_n; A.tile_k0 = KB.tile_k0; A.tile_n0 = KB.tile_n0; A.sec_s = KB.sec_s; }
    A.sec_E = KB.sec_E;
    A.qw = KB.qw; A.hslot0 = KB.hslot0; A.hslotN = KB.hslotN; A.partials = KB.partials;
    A.gran = KB.gran; A.erec = KB.erec; A.tab = KB.tab;
    A.erec0 = KB.erec0;
    A.epoch = MULTI ? mepoch : KB.epoch;
    A.s_off = KB.s_off; A.c_path_off = KB.c_path_off; A.c_int_off = KB.c_int_off;
    A.t_fixed[0] = KB.t_fixed[0]; A.t_fixed[1] = KB.t_fixed[1];
    A.qa_total = KB.qa_total; A.qw_total = KB.qw_total;
    A.lds_out = KB.lds_out;
    A.qw0 = KB.qw_off[UN > 0 ? UN : 0];
    if constexpr (PURE) {   // (the lead words of a mixed launch describe no order)
      A.qa0 = KB.qa_off[UN];
      A.qwabs = KB.qa_total + KB.qw_off[UN];
    }
    static_for<0, St::NSCAL>([&](auto i_) { A.scal[decltype(i_)::value] = KB.scal[decltype(i_)::value]; });
    static_for<0, NFN>([&](auto i_) { A.goff[decltype(i_)::value] = KB.goff[decltype(i_)::value]; });
    static_for<0, 3 * NZ + NS * NZ>([&](auto i_) { A.hoff[decltype(i_)::value] = KB.hoff[decltype(i_)::value]; });
  };
  if (LD) {
    typedef const __attribute__((address_space(4))) char* kseg_t;
    kseg_t kseg = (kseg_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kseg)::"memory");
    part2(*(const __attribute__((address_space(4))) PcPhaseArgs*)(kseg + sizeof(PcLead)));
  } else {
    part2(KA);
  }
  // the multipliers of the defect rows, the section widths and (compile-time order) the order's A and weight tables:
  // first chunks into registers.  Issued here -- behind the scalar loads of part 2, ahead of the wait for them -- when
  // their addresses are preloaded scalars (pc_bulk_p<i>, compile-time order); otherwise after that wait, where the
  // pointers have arrived anyway and fewer values are live across it (Delta III and the any-order kernels lose 2 %
  // with the early form).
  const bool wantH = A.flags & PC_FLAG_H;
  const int TB = A.block_threads;   // (blockDim.x is a separate, late scalar load)
  const bool has_prev = k0 > 0;
  const int kp = has_prev ? k0 - 1 : 0;  // first staged section
  const int nsec = k1 - kp;              // staged sections (previous one included)
  constexpr int QA_N = UN > 0 ? (UN - 1) * UN : 0;   // a compile-time order stages just its own A_n and w_n, at the
  double r_h = 0.0, r_qa = 0.0, r_qw = 0.0;          // start of the LDS table areas (QAO = QWO = 0)
  const bool odd_prev = PURE && has_prev && r_nprev != UN;   // (wave-uniform)
  const int lsA = has_prev ? 1 : 0;                  // first section of the tile (local index)
  double r_cp = 0.0;
  double r_lam[2 * (NY > 0 ? NY : 1)];
  int lam0 = 0, lam_cnt = 0;   // first staged defect row; rows staged (<= TB + PC_MAX_ORDER - 2: two chunks)
  auto aux_loads = [&]() {
    r_h = tid < nsec ? A.sec_h[kp + tid] : 0.0;   // widths are data even on a uniform-order mesh
    if constexpr (PURE) {
      lam0 = has_prev ? n0 - (r_nprev - 1) : n0;
    } else if (uni) {
      lam0 = kp * (un - 1);
    } else {
      lam0 = A.sec_s[kp];
    }
    lam_cnt = n1 - lam0;
    if (wantH) {
      // one predicated block per chunk for all states (a select per load is an exec-mask save / restore and a branch
      // around each one: 2 NY of them here and again at the LDS stores below)
      static_for<0, 2 * NY>([&](auto i_) { r_lam[decltype(i_)::value] = 0.0; });
      const double* src0 = A.lamd + lam0;
      if (tid < lam_cnt) static_for<0, NY>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        r_lam[2 * a] = src0[(int64_t)a * (N - 1) + tid];
      });
      if (tid + TB < lam_cnt) static_for<0, NY>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        r_lam[2 * a + 1] = src0[(int64_t)a * (N - 1) + tid + TB];
      });
    }
    if constexpr (UN > 0) {
      r_qa = tid < QA_N ? A.qa[A.qa0 + tid] : 0.0;
      r_qw = tid < UN ? A.qa[A.qwabs + tid] : 0.0;
    }
    if constexpr (PURE) {   // last column of the previous section's A table when that section has another order
      if (odd_prev) r_cp = tid < r_nprev - 1 ? A.qa[r_qaprev + tid * r_nprev + r_nprev - 1] : 0.0;
    }
  };
  const bool early_aux = LD != nullptr && UN > 0 && !MIX;
  if (early_aux) aux_loads();
  // (only what this build of the kernel can use: everything pinned is live in SGPRs from here on, and the file has
  //  ~100 of them -- an over-full pin list is loaded in several dependent batches and partly spilled to VGPR lanes)
  // Two groups.  The first is what the per-node loads and the table staging need (pointers, offsets, geometry);
  // the second -- output pointers, scaling constants, run offsets -- is consumed by a second asm placed after those
  // loads have been issued, so that its scalar-load round trip overlaps their latency instead of preceding it.
  PinAcc pin;
  pin(A.x); pin(A.lam); pin(A.sec_h); pin(A.qa); pin(A.qw);
  pin(A.x_off); pin(A.c_off); pin(A.N); pin(A.K); pin(A.flags); pin(A.qa_total); pin(A.qw_total); pin(A.tile_begin);
  pin(A.uni_n); pin(A.spt); pin(A.lds_out); pin(A.wpt); pin(A.n_blocks); pin(A.block_threads);
  if constexpr (UN > 0) { pin(A.qa0); pin(A.qwabs); }
  if constexpr (NP > 0) pin(A.c_path_off);
  if constexpr (NQ > 0) pin(A.c_int_off);
  if constexpr (NS > 0) pin(A.s_off);
  if constexpr (UN == 0) { pin(A.tile_k0); pin(A.tile_n0); pin(A.sec_s); pin(A.sec_E); }   // any-mesh tables
  pin.done();
  // (hslot0 / hslotN are left lazy: only the two edge tiles read them)
  constexpr int NHO = 3 * NZ + NS * NZ;
  constexpr bool PINNED = 2 * (St::NSCAL + NFN + NHO) <= PC_PIN_BUDGET;
  auto pin_second_group = [&]() {
    PinAcc pin2;
    pin2(A.c); pin2(A.G); pin2(A.H);
    if constexpr (NRED > 0) { if constexpr (RES) pin2(A.gran); else pin2(A.partials); }
    if constexpr (!M::T0_FREE) pin2(A.t_fixed[0]);
    if constexpr (!M::TF_FREE) pin2(A.t_fixed[1]);
    // The per-variable constants (scaling, run offsets) are hoisted too while they fit the scalar register file
    // next to the above.  A model with many variables would have them spilled to VGPR lanes (the shuttle kernel
    // carried 2000 v_readlane, whole 16-register tuples reloaded per use): such a model reads them from an LDS
    // copy instead (uniform-address ds_read, staged with the quadrature tables).
    if constexpr (PINNED) {
      // the scaling entries the bulk kernel reads: V, r of z, of the free times and of s; the row weights.  (V, r
      // of the integral variables belong to the tail kernel.)
      static_for<0, St::NSCAL>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr bool used = (i < St::O_VQ) || (i >= St::O_VT && i < St::O_VT + NT) || (i >= St::O_RT && i < St::O_RT + NT) ||
                              (i >= St::O_VS);
        if constexpr (used) pin2(A.scal[i]);
      });
      pin_array<NFN>(pin2, A.goff);
      static_for<0, NHO>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        if constexpr (i < NZ || (NT > 0 && i < 3 * NZ) || i >= 3 * NZ) pin2(A.hoff[i]);   // t strips only with free times
      });
    } else {
      pin2(A.tab);
    }
    pin2.done();
  };

  extern __shared__ double smem[];
  // W replicas ("waves per tile") share one tile of TN nodes: every replica evaluates all TN nodes (the node
  // functions are cheap next to a SIMD that would otherwise idle) and produces a disjoint subset of the output
  // runs -- states, Hessian row blocks, path rows are dealt round-robin.  W > 1 only with TN = 64 (a replica is
  // exactly one wave, so its private staging region needs no workgroup barrier).
  const int TN = W > 1 ? 64 : TB;
  // (WN > 0: the replica index is a template argument -- see the kernel entry -- and everything a replica does not
  //  own, node-function outputs included, is dead code in its instantiation)
  const int w = WN > 0 ? WIDX : (W > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0);   // wave-uniform: branches on it are scalar
  // (tables only some kernels stage -- the scal | goff | hoff copy sized by the model, the section tables of the
  //  any-order kernels -- take no LDS in the others: LDS per tile is what bounds the waves a CU holds)
  const LdsPlan lp = lds_plan(TN, A.qa_total, A.qw_total, NY, NFS, NRED, A.lds_out * W, St::NSCAL + NFN + 3 * NZ + NS * NZ, UN == 0, MIX);
  double* s_cp = smem + lp.cp;
  double* s_qa = smem + lp.qa;
  double* s_qw = smem + lp.qw;
  int* s_off = reinterpret_cast<int*>(smem + lp.off);   // [0..20] qa_off, [21..41] qw_off (any-mesh kernels)
  auto QAO = [&](int n) -> int { return UN > 0 ? 0 : s_off[n]; };                       // start of A_n in s_qa
  auto QWO = [&](int n) -> int { return UN > 0 ? 0 : s_off[PC_MAX_ORDER + 1 + n]; };    // start of w_n in s_qw
  double* s_h = smem + lp.h;
  long long* s_E = reinterpret_cast<long long*>(smem + lp.E);
  int* s_s = reinterpret_cast<int*>(smem + lp.s);
  int* s_kr = reinterpret_cast<int*>(smem + lp.kr);
  double* s_f = smem + lp.f;
  double* s_yu = smem + lp.yu;
  double* s_fs = smem + lp.fs;
  double* s_lam = smem + lp.lam;
  double* s_red = smem + lp.red;
  double* s_out = smem + lp.out + w * A.lds_out;   // this replica's staging region
  // static deal of the tile's output items to the replicas (S<M>::deal); W == 1 owns everything
  constexpr auto DEAL2 = St::template deal<2, (UN > 0 ? UN : 4)>();
  constexpr auto DEAL4 = St::template deal<4, (UN > 0 ? UN : 4)>();
  auto mine = [&](auto item_) -> bool {
    constexpr int item = decltype(item_)::value;
    return W == 1 || (W == 2 ? DEAL2.own[item] : DEAL4.own[item]) == w;
  };
  // does this replica produce any Hessian output (and hence need the adjoint node weights)?
  const bool hess_replica = W == 1 || (((W == 2 ? DEAL2.hmask : DEAL4.hmask) >> w) & 1u);
#define PC_ITEM(i) std::integral_constant<int, (i)>{}
  // Synchronisation by who shares the data.  LDS operations of ONE wave complete in program order, so a wave that
  // hands data to itself through LDS (a one-wave workgroup, or a replica's private staging region) needs no wait
  // and no barrier at all -- only the compiler must not reorder the accesses (PC_WAVE_FENCE emits nothing).
  // `one_wave`: the workgroup is a single wave.  `wave_private`: additionally true for replicas (W > 1), for data
  // every replica writes in full for itself (node values) or keeps to itself (staging).
#ifndef PC_WAVE_FENCE
#define PC_WAVE_FENCE() asm volatile("" ::: "memory")
#endif
  const bool one_wave = (TB == 64);
  const bool wave_private = one_wave || W > 1;
  auto stage_sync = [&]() {       // staging buffer of this tile / replica
    if (wave_private) PC_WAVE_FENCE();
    else lds_barrier();
  };
  auto block_sync = [&]() {       // data of the whole workgroup (tables staged cooperatively, the overlay hand-over)
    if (one_wave) PC_WAVE_FENCE();
    else lds_barrier();
  };
  auto node_sync = [&]() {        // per-node values: every replica wrote all of them itself
    if (wave_private) PC_WAVE_FENCE();
    else lds_barrier();
  };

  const bool wantC = A.flags & PC_FLAG_C, wantG = A.flags & PC_FLAG_G;

  const bool last_tile = (k1 == A.K);

  // ---- the rest of the per-node loads are issued before any staging so that their latency overlaps it ------
  // kernarg-resident (scalar registers) or the LDS copy, see PINNED above
  double* s_tab = smem + lp.tab;
  const double* sc;
  const int64_t* goff;
  const int64_t* hoff;
  if constexpr (PINNED) {
    sc = A.scal;
    goff = A.goff;
    hoff = A.hoff;
  } else {
    sc = s_tab;
    goff = reinterpret_cast<const int64_t*>(s_tab + St::NSCAL);
    hoff = goff + NFN;
  }
  const bool owns = active && (t < T || last_tile);
  double F[NFN > 0 ? NFN : 1], Jv[NJ > 0 ? NJ : 1], Hv[NH > 0 ? NH : 1], mu[NFN > 0 ? NFN : 1];
  double red[NRED > 0 ? NRED : 1];
  static_for<0, NRED>([&](auto r_) { red[decltype(r_)::value] = 0.0; });
  if (active) {
    static_for<0, NS>([&](auto l_) {   // parameters: static (after every phase), or this phase's q / free t
      constexpr int l = decltype(l_)::value;
      constexpr int kind = M::wk(l), idx = M::wi(l);
      v[NZ + l] = A.x[kind == 0 ? A.s_off + idx : A.x_off + (int64_t)NZ * N + (kind == 1 ? idx : NQ + idx)];
    });
  }
  double lam_p[NP > 0 ? NP : 1], lam_q[NQ > 0 ? NQ : 1];
  if (owns && wantH && hess_replica) {
    static_for<0, NP>([&](auto m_) { lam_p[decltype(m_)::value] = A.lam[A.c_path_off + (int64_t) decltype(m_)::value * N + node]; });
    static_for<0, NQ>([&](auto m_) { lam_q[decltype(m_)::value] = A.lam[A.c_int_off + decltype(m_)::value]; });
  }

  // ---- staging: every table's first chunk is loaded into registers before anything is written to LDS,
  //      so the global-load latencies overlap instead of queueing behind one loop after another ----------
  // table offsets by order: a compile-time order needs just its own two (scalar kernel arguments, see QAO / QWO)
  const int r_off = (UN == 0 && tid < 2 * (PC_MAX_ORDER + 1)) ? reinterpret_cast<const int32_t*>(&KA.qa_off[0])[tid] : 0;   // qa_off, qw_off adjacent
  if (!early_aux) aux_loads();
  if constexpr (UN == 0) {   // any order: the whole packed tables
    r_qa = tid < A.qa_total ? A.qa[tid] : 0.0;
    r_qw = tid < A.qw_total ? A.qw[tid] : 0.0;
  }
  pin_second_group();
  if (!uni) {
    for (int i = tid; i <= nsec; i += TB) {
      s_s[i] = A.sec_s[kp + i];
      s_E[i] = A.sec_E[kp + i];
    }
  }
  if (UN == 0 && tid < 2 * (PC_MAX_ORDER + 1)) s_off[tid] = r_off;
  if constexpr (UN > 0) {
    if (tid < QA_N) s_qa[tid] = r_qa;
    if (tid < UN) s_qw[tid] = r_qw;
  } else {
    if (tid < A.qa_total) s_qa[tid] = r_qa;
    if (tid < A.qw_total) s_qw[tid] = r_qw;
  }
  if (tid < nsec) s_h[tid] = r_h;
  if constexpr (PURE) {
    if (odd_prev && tid < r_nprev - 1) s_cp[tid] = r_cp;
  }
  if constexpr (!PINNED) {   // scal | goff | hoff, packed by the host in exactly this order
    for (int i = tid; i < St::NSCAL + NFN + NHO; i += TB) s_tab[i] = A.tab[i];
  }
  if constexpr (UN > 0) {
    for (int i = tid + TB; i < QA_N; i += TB) s_qa[i] = A.qa[A.qa0 + i];
  } else {
    for (int i = tid + TB; i < A.qa_total; i += TB) s_qa[i] = A.qa[i];
    for (int i = tid + TB; i < A.qw_total; i += TB) s_qw[i] = A.qw[i];
  }
  for (int i = tid + TB; i < nsec; i += TB) s_h[i] = A.sec_h[kp + i];
  if (wantH) {
    if (tid < lam_cnt) static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      s_lam[a * (TN + PC_MAX_ORDER) + tid] = r_lam[2 * a];
    });
    if (tid + TB < lam_cnt) static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      s_lam[a * (TN + PC_MAX_ORDER) + tid + TB] = r_lam[2 * a + 1];
    });
  }
  if (!uni) {
    block_sync();
    for (int ls = tid; ls < nsec; ls += TB) {
      const int sb = s_s[ls], se = s_s[ls + 1];
      for (int nd = sb + 1; nd <= se; ++nd) {
        const int tt = nd - n0;
        if (tt >= 0 && tt <= T) s_kr[tt] = ls;
      }
    }
    if (tid == 0 && n0 == 0) s_kr[0] = -1;
  }
  block_sync();

  // section accessors (local index ls counts from section kp)
  auto S_s = [&](int ls) -> int {
    if constexpr (PURE) return ls >= lsA ? n0 + (ls - lsA) * (UN - 1) : n0 - (r_nprev - 1);
    return uni ? (kp + ls) * (un - 1) : s_s[ls];
  };
  // (UN > 0 spelled out: the loops over a section's rows must see a compile-time trip count even when this
  //  lambda is inlined late -- otherwise they stay rolled and every array they index lands in scratch memory)
  auto S_n = [&](int ls) -> int { return UN > 0 ? UN : (uni ? un : s_s[ls + 1] - s_s[ls] + 1); };
  auto S_h = [&](int ls) -> double { return s_h[ls]; };
  auto S_E = [&](int ls) -> long long {
    if constexpr (PURE) return r_E0 + (long long)((ls - lsA) * (UN - 1) * UN);
    return uni ? (long long)(kp + ls) * (un - 1) * un : s_E[ls];
  };

  // ---- uniform scalars ------------------------------------------------------------------------
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

  // ---- where the node sits in the mesh ----------------------------------------------------------
  int ls_r = -1, pos_r = 0, n_r = UN > 0 ? UN : 2, ls_s = 0, n_s = UN > 0 ? UN : 2;
  bool has_start = false;
  double w_node = 0.0;
  // the first lane of an order-pure tile whose previous section has another order (see MIX above)
  const bool prev_lane = odd_prev && t == 0;
  if (active) {
    static_for<0, NZ>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      v[b] = sc[St::O_VZ + b] * v[b] + sc[St::O_RZ + b];
    });
    static_for<0, NS>([&](auto l_) {
      constexpr int l = decltype(l_)::value;
      v[NZ + l] = sc[St::O_VS + l] * v[NZ + l] + sc[St::O_RS + l];
    });
    PC_STAMP(1);   // node values have arrived
    if constexpr (PURE) {
      ls_r = t == 0 ? lsA - 1 : lsA + (t - 1) / (UN - 1);
    } else if (uni) {
      const int g = node - kp * (un - 1);           // node index relative to the first staged section
      ls_r = (node == 0) ? -1 : (g - 1) / (un - 1);
    } else {
      ls_r = s_kr[t];
    }
    ls_s = ls_r + 1;
    // the node opens section ls_s only if it is that section's first node (staged in this tile)
    has_start = (node < N - 1) && (ls_s < nsec) && (S_s(ls_s) == node);
    if (ls_r >= 0) {
      n_r = S_n(ls_r);
      pos_r = node - S_s(ls_r);
      w_node = prev_lane ? S_h(ls_r) * r_wprev : S_h(ls_r) * s_qw[QWO(n_r) + (prev_lane ? 0 : pos_r)];
    }
    if (has_start) {
      n_s = S_n(ls_s);
      w_node += S_h(ls_s) * s_qw[QWO(n_s)];
    }
  }

  // ---- h_k A[j][pos]: the node's column of the integration matrix in the section it closes (cr) and in the
  //      section it opens (cs).  Both the adjoint weights and the defect Jacobian are built from these; with a
  //      compile-time order they are read from LDS once, here, instead of once per state and use
#ifndef PC_HOIST_MAX
#define PC_HOIST_MAX 8
#endif
  constexpr bool HOIST = (UN > 0 && UN <= PC_HOIST_MAX);
  constexpr int NC = HOIST ? UN - 1 : 1;
  double cr[NC], cs[NC];
  if constexpr (HOIST) {
#pragma unroll
    for (int j = 0; j < NC; ++j) cr[j] = cs[j] = 0.0;
    if (active) {
      const double* At = s_qa + QAO(UN);
      if (ls_r >= 0 && !prev_lane) {
        const double h = S_h(ls_r);
#pragma unroll
        for (int j = 1; j < UN; ++j) cr[j - 1] = h * At[(j - 1) * UN + pos_r];
      }
      if (has_start) {
        const double h = S_h(ls_s);
#pragma unroll
        for (int j = 1; j < UN; ++j) cs[j - 1] = h * At[(j - 1) * UN];
      }
    }
  }

  // ---- adjoint node weights mu (iteration.py:1078-1103) ------------------------------------------
  static_for<0, NFN>([&](auto r_) { mu[decltype(r_)::value] = 0.0; });
  if (owns && wantH && hess_replica) {
    static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      const double* la = s_lam + a * (TN + PC_MAX_ORDER);
      double acc = 0.0;
      if constexpr (PURE) {
        if (prev_lane) {   // the section before the tile: its own order's last A column, rows lam0 .. lam0 + nprev - 2
          double a2 = 0.0;
          for (int j = 0; j < r_nprev - 1; ++j) a2 += la[j] * s_cp[j];
          acc += S_h(0) * a2;
        }
      }
      if constexpr (HOIST) {
        if (ls_r >= 0 && !prev_lane) {
          const int base = S_s(ls_r) - lam0;
#pragma unroll
          for (int j = 1; j < UN; ++j) acc += la[base + j - 1] * cr[j - 1];
        }
        if (has_start) {
          const int base = S_s(ls_s) - lam0;
          double a2 = 0.0;
#pragma unroll
          for (int j = 1; j < UN; ++j) a2 += la[base + j - 1] * cs[j - 1];
          acc += a2;
        }
      } else {
        if (ls_r >= 0 && !prev_lane) {
          const double* At = s_qa + QAO(n_r);
          const int base = S_s(ls_r) - lam0;
          double a2 = 0.0;
#pragma unroll
          for (int j = 1; j < n_r; ++j) a2 += la[base + j - 1] * At[(j - 1) * n_r + pos_r];
          acc += S_h(ls_r) * a2;
        }
        if (has_start) {
          const double* At = s_qa + QAO(n_s);
          const int base = S_s(ls_s) - lam0;
          double a2 = 0.0;
#pragma unroll
          for (int j = 1; j < n_s; ++j) a2 += la[base + j - 1] * At[(j - 1) * n_s];
          acc += S_h(ls_s) * a2;
        }
      }
      mu[a] = sc[St::O_WD + a] * acc;
    });
    static_for<0, NP>([&](auto m_) {
      constexpr int m = decltype(m_)::value;
      mu[NY + m] = sc[St::O_WP + m] * lam_p[m];
    });
    static_for<0, NQ>([&](auto m_) {
      constexpr int m = decltype(m_)::value;
      mu[NY + NP + m] = -sc[St::O_WI + m] * lam_q[m] * w_node;
    });
  }

  // ---- per-tile partial sums (fixed order: lanes -> waves -> tile): every wave deposits its sums here.  The
  //      two-launch build combines the waves' sums at the very end (no replica waits for another mid-kernel); the
  //      resident-tail build publishes them at once as granules, ahead of the bulky c~ / G~ runs, so that the tail
  //      workgroup finishes while this tile is still storing.
  auto deposit_partials = [&]() {
  if constexpr (NRED > 0) {
    static_for<0, NRED>([&](auto r_) {   // replicas hold copies: the owner of the sum's item contributes
      constexpr int r = decltype(r_)::value;
      if (!mine(PC_ITEM(PC_CE(St::red_item(r))))) red[r] = 0.0;
    });
    const int wave = tid >> 6, lane = tid & 63, nw = (TB + 63) >> 6;
    if (RES && wave_private) {
      // one wave holds every node of the tile: the sum of the item's owner IS the tile's sum (the other replicas
      // would add exact zeros), and the owner publishes it itself
      static_for<0, NRED>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        const double sr = wave_sum(red[r]);
        if (lane == 0 && mine(PC_ITEM(PC_CE(St::red_item(r)))))
          publish_granules(A.gran + 2 * ((int64_t)tile * NRED + r), A.epoch, sr);
      });
      return;
    }
    static_for<0, NRED>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      const double sr = wave_sum(red[r]);
      if (lane == 0) s_red[r * 16 + wave] = sr;
    });
    if constexpr (RES) {   // several waves share the tile's nodes: their sums meet in LDS, in wave order
      lds_barrier();
      if (tid < NRED) {
        double sr = 0.0;
        for (int ww = 0; ww < nw; ++ww) sr += s_red[tid * 16 + ww];
        publish_granules(A.gran + 2 * ((int64_t)tile * NRED + tid), A.epoch, sr);
      }
    }
  }
  };
  // ---- model evaluation -----------------------------------------------------------------------
  // The heaviest models evaluate in two passes -- values and first partials here, second partials after the
  // Jacobian has been written -- so that the two derivative sets are never live together.  It pays only where
  // the one-pass kernel spills (space station, 96 partials: 512 VGPRs + 584 B scratch -> 472, no scratch, 58.6 ->
  // 49.2 us at 60k nodes); where it merely lowers the register count the recomputed subexpressions cost more
  // than the occupancy returns (shuttle 194 -> 171 VGPRs: 6 % slower; Delta III 302 -> 218: 3 % slower).
  // A heavy model (M::HEAVY, codegen) is split as well, in every build of its tile body: there the point is not the
  // spill but the register count itself -- at <= 256 VGPRs a SIMD holds two of its waves and one issues while the other
  // waits (Delta III: 308 -> 256 VGPRs; 4 x 12.5 k nodes, two waves per tile: 30.6 -> 23.2 us; 4 x 50 k: 108 -> 82 us).
  // Every build, not only the two-wave one: the two passes round differently from the fused evaluation in the last
  // bits, and the sharded evaluation (two-launch kernels) must return the bits of the unsharded one.
  constexpr bool SPLIT = ((NJ + NH >= PC_SPLIT_MIN) || M::HEAVY) && St::NWT == 0;   // (time parameters: the t strips need both passes' values)
  constexpr bool RED_EARLY = RES && NS == 0 && NRED > 0;
  double mult[NFN > 0 ? NFN : 1];
  if (active) {
    static_for<0, NFN>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      mult[r] = (r >= NY && r < NY + NP) ? mu[r] : stretch * mu[r];
    });
  }
  // number of nodes this tile owns, and the first one (for the staged per-node runs)
  const int n_own = T + (last_tile ? 1 : 0);
  // ---- Hessian (compiled.py:484-500): flag 1 bands, flag 2 strips, flag 3 sums -------------------
  const bool edge0 = (node == 0), edgeN = (node == N - 1);
  // A Hessian entry of an edge node (0 or N-1) is stored -- unless an endpoint term lands on the same slot: in the
  // resident-tail build such an entry goes to the tail workgroup as a record (the tail adds the term and stores it).
  // Which sites those are is a property of the model (M::efl, compile time); the record of a flagged site is
  // erec0 + its rank among the flagged sites (node 0's first).  A model without such terms pays nothing.
  constexpr int NEDGE = St::NHZZ + 2 * NZ + NS * NZ;
  auto edge_store = [&](auto site_, double* dstp, double val) {
    constexpr int site = decltype(site_)::value;
    constexpr bool f0 = M::efl(site) != 0, fN = M::efl(NEDGE + site) != 0;
    if constexpr (RES && (f0 || fN)) {
      constexpr int r0 = PC_CE(St::erank(site)), rN = PC_CE(St::erank(NEDGE + site));
      if (edge0 ? f0 : fN) {
        publish_granules(A.erec + 2 * (int64_t)(A.erec0 + (edge0 ? r0 : rN)), A.epoch, val);
        return;
      }
    }
    *dstp = val;
  };
  auto hess_second = [&]() {     // everything built from the second partials
    // bands: one variable block row at a time; rows with several entries go through the staging buffer
    static_for<0, NZ>([&](auto rv_) {
      constexpr int rv = decltype(rv_)::value;
      constexpr int MB = PC_CE(St::hrow_count(rv));
      if constexpr (MB > 0) if (mine(PC_ITEM(St::IT_HB + rv))) {
        double vals[MB];
        static_for<0, NH>([&](auto e_) {
          constexpr int e = decltype(e_)::value;
          if constexpr (M::hr(e) == rv)
            vals[PC_CE(St::hpos(e))] = sc[St::O_VZ + rv] * sc[St::O_VZ + M::hc(e)] * Hv[e];
        });
        if (owns && (edge0 || edgeN)) {   // edge rows may interleave endpoint entries: explicit slots
          static_for<0, NH>([&](auto e_) {
            constexpr int e = decltype(e_)::value;
            if constexpr (M::hr(e) == rv)
              edge_store(ic<St::hzz_index(e)>{}, A.H + (edge0 ? A.hslot0 : A.hslotN)[PC_CE(St::hzz_index(e))],
                         vals[PC_CE(St::hpos(e))]);
          });
        }
        if constexpr (MB == 1) {
          if (owns && !edge0 && !edgeN) A.H[hoff[St::HO_Z + rv] + (int64_t)node] = vals[0];
        } else {
          if (owns) static_for<0, MB>([&](auto q_) { s_out[t * MB + decltype(q_)::value] = vals[decltype(q_)::value]; });
          stage_sync();
          // interior nodes of the tile: [lo, hi)
          const int lo = (n0 == 0) ? 1 : 0, hi = (last_tile ? n_own - 1 : n_own);
          if (hi > lo)
            flush_run(A.H + hoff[St::HO_Z + rv] + (int64_t)(n0 + lo) * MB, s_out + lo * MB, (hi - lo) * MB, t, TN);
          stage_sync();
        }
      }
    });
    if (owns) {
      static_for<0, NH>([&](auto e_) {
        constexpr int e = decltype(e_)::value;
        constexpr int rv = M::hr(e), cv = M::hc(e);
        if constexpr (rv >= NZ && cv < NZ && St::is_t(rv >= NZ ? rv - NZ : 0)) {
          // a time parameter's strip is the t strip (hess_tstrips adds this entry to the stretch term)
        } else if constexpr (rv >= NZ && cv < NZ) {
          if (mine(PC_ITEM(St::IT_HS + cv))) {
            double* dstp = A.H + hoff[St::HO_S + (rv - NZ) * NZ + cv] + node;
            const double val = sc[St::O_VS + rv - NZ] * sc[St::O_VZ + cv] * Hv[e];
            if (edge0 || edgeN) edge_store(ic<St::NHZZ + 2 * NZ + (rv - NZ) * NZ + cv>{}, dstp, val); else *dstp = val;
          }
        } else if constexpr (rv >= NZ) {
          constexpr int l = rv - NZ, l2 = cv - NZ;
          red[PC_CE(St::rss(l, l2))] = sc[St::O_VS + l] * sc[St::O_VS + l2] * Hv[e];
        }
      });
    }
  };
  // The second pass of a split build: second partials, then the Hessian runs built from them.
  auto second_pass = [&]() {
    if (wantH && hess_replica) {
      if (active) M::eval_h(v, mult, Hv);
      PC_STAMP(6);   // second partials evaluated
      hess_second();
    }
  };
  if (active) {
    if constexpr (SPLIT) M::eval_fj(v, F, Jv);
    else M::eval(v, mult, F, Jv, Hv);
    if (owns) {   // this node's terms of the integral rows' sums (backend.py:1645-1647)
      static_for<0, NQ>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        constexpr int r = NY + NP + m;
        red[St::R_Q + m] = w_node * F[r];
        static_for<0, NS>([&](auto l_) {
          constexpr int l = decltype(l_)::value;
          if constexpr (PC_CE(St::dep(r, NZ + l))) red[PC_CE(St::rqs(m, l))] = w_node * Jv[PC_CE(St::jidx(r, NZ + l))];
        });
      });
    }
    static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      s_f[a * TN + t] = F[a];
      s_yu[a * TN + t] = v[a];
      static_for<0, NS>([&](auto l_) {
        constexpr int l = decltype(l_)::value;
        if constexpr (PC_CE(St::dep(a, NZ + l))) s_fs[PC_CE(St::fs_slot(a, l)) * TN + t] = Jv[PC_CE(St::jidx(a, NZ + l))];
      });
    });
  }
  // without static parameters every partial sum is an integrand sum and complete here: the resident-tail build
  // publishes them now, a kernel's length ahead of the tail's need for them
  if constexpr (RED_EARLY) deposit_partials();
  PC_STAMP(2);   // node functions (first pass) evaluated
  node_sync();

  // ---- defect rows: value (row-wise; backend.py:1601-1603) ---------------------------------------
  double accf[NY > 0 ? NY : 1];
  const bool rowthr = active && t >= 1;
#ifdef PC_MFMA_DEFECT
  // Optional build (-DPC_MFMA_DEFECT, off in the product; SURVEY row X1, backend.py:1601-1603): the contraction with the
  // integration matrix, sum_i A[j][i] f_a(z_i), on the matrix cores -- v_mfma_f64_16x16x4_f64, sixteen consecutive
  // defect rows of the tile (columns of the result) against up to sixteen states (its rows), four nodes per step:
  //   lane l, step s:  A-operand = f_(l & 15)(node kb + 4 s + (l >> 4))          [state x node]
  //                    B-operand = A[j(q) - 1][that node - section start of row q], q = l & 15, 0 outside the section
  //   result:          lane l, register i holds state (l >> 4) + 4 i of row q = l & 15
  // (cdna_hip_programming.md, "f64 MFMA does NOT use these maps").  The row's lane then collects its NY sums with one
  // cross-lane read per state.  Uniform-order tiles of 64 nodes only; anything else takes the vector form below.
  // Not equal to it in the last bits (fused accumulation, zero terms of neighbouring sections -- a NaN in one section's
  // f reaches the rows of the sections sharing its group of 16), which is why it is an A/B build and not the default.
  constexpr bool MFMA_DEF = UN > 0 && NY > 0 && NY <= 16;
  double accm[NY > 0 ? NY : 1];
  bool mfma_done = false;
  if constexpr (MFMA_DEF) {
    if (TN == 64 && (wantC || wantG)) {   // wave-uniform
      typedef double pc_d4 __attribute__((ext_vector_type(4)));
      constexpr int n = UN > 0 ? UN : 2;
      constexpr int KS = (16 + 2 * (n - 1) + 1 + 3) / 4;    // node span of 16 consecutive rows, in steps of four
      const int lane = t & 63, q = lane & 15, hq = lane >> 4;
      const double* Atab = s_qa + QAO(n);
      static_for<0, NY>([&](auto a_) { accm[decltype(a_)::value] = 0.0; });
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (16 * g >= T) break;                            // (T is wave-uniform)
        const int rr = 16 * g + q;                          // this lane's column of the result: tile row rr
        const int sec = rr / (n - 1), jm1 = rr - sec * (n - 1), sk = sec * (n - 1);
        const int kb = ((16 * g) / (n - 1)) * (n - 1);     // first node any row of the group touches
        pc_d4 acc4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4) {
          const int kk = kb + 4 * s4 + hq;                  // node (tile-local) of this lane's k
          const int i = kk - sk;
          const bool in = rr < T && i >= 0 && i < n;
          const double coef = Atab[in ? jm1 * n + i : 0];
          const double fval = s_f[min(q, NY - 1) * TN + min(kk, T)];
          acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(q < NY ? fval : 0.0, in ? coef : 0.0, acc4, 0, 0, 0);
        }
        // row 16 g + q' lives in lanes q', 16 + q', 32 + q', 48 + q': state a in lane 16 (a & 3) + q', register a >> 2
        const int mine_q = (t - 1) & 15;
        const bool here = rowthr && ((t - 1) >> 4) == g;
        static_for<0, NY>([&](auto a_) {
          constexpr int a = decltype(a_)::value;
          const double got = __shfl(acc4[a >> 2], 16 * (a & 3) + mine_q, 64);
          if (here) accm[a] = got;
        });
      }
      mfma_done = true;
    }
  }
#endif
  if (rowthr && (wantC || wantG)) {
    const int n = UN > 0 ? UN : n_r, j = pos_r, sk = S_s(ls_r) - n0;
    const double h = S_h(ls_r);
    const double* Arow = s_qa + QAO(n) + (j - 1) * n;
    static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      accf[a] = 0.0;
      if (!mine(PC_ITEM(St::IT_D + a))) return;
      double acc = 0.0;
#ifdef PC_MFMA_DEFECT
      if (mfma_done) {
        acc = accm[a];
      } else
#endif
      {
#pragma unroll
        for (int i = 0; i < n; ++i) acc += Arow[i] * s_f[a * TN + sk + i];
      }
      accf[a] = h * acc;
      if (wantC)
        A.c[A.c_off + (int64_t)a * (N - 1) + node - 1] = sc[St::O_WD + a] * ((s_yu[a * TN + sk] - v[a]) + stretch * accf[a]);
    });
  }
  PC_STAMP(3);   // defect values formed, c~ stores issued
  block_sync();   // every replica is done with f / y / lambda: the staging buffer may overwrite them

  // ---- path rows (backend.py:1612-1614, compiled.py:336-355) ------------------------------------
  static_for<0, NP>([&](auto m_) {
    constexpr int m = decltype(m_)::value;
    constexpr int r = NY + m;
    constexpr int R = PC_CE(St::nzdep(r)) + PC_CE(St::nsdep(r));
    const double Wp = sc[St::O_WP + m];
    if (owns && wantC && mine(PC_ITEM(St::IT_P + m))) A.c[A.c_path_off + (int64_t)m * N + node] = Wp * F[r];
    if (wantG && R > 0 && mine(PC_ITEM(St::IT_P + m))) {
      if (owns) {
        static_for<0, NZ>([&](auto b_) {
          constexpr int b = decltype(b_)::value;
          if constexpr (PC_CE(St::dep(r, b))) s_out[t * R + PC_CE(St::zrank(r, b))] = Wp * sc[St::O_VZ + b] * Jv[PC_CE(St::jidx(r, b))];
        });
        static_for<0, NS>([&](auto l_) {
          constexpr int l = decltype(l_)::value;
          if constexpr (PC_CE(St::dep(r, NZ + l)))
            s_out[t * R + PC_CE(St::nzdep(r)) + PC_CE(St::srank(r, l))] = Wp * sc[St::O_VS + l] * Jv[PC_CE(St::jidx(r, NZ + l))];
        });
      }
      stage_sync();
      flush_run(A.G + goff[St::GO_P + m] + (int64_t)n0 * R, s_out, n_own * R, t, TN);
      stage_sync();
    }
  });

  // ---- integral rows: z entries and partial sums (backend.py:1645-1647, compiled.py:357-379) -----
  if (owns) {
    static_for<0, NQ>([&](auto m_) {
      constexpr int m = decltype(m_)::value;
      constexpr int r = NY + NP + m;
      if (wantG && mine(PC_ITEM(St::IT_Q + m))) {
        const double k = -sc[St::O_WI + m] * stretch * w_node;
        static_for<0, NZ>([&](auto b_) {
          constexpr int b = decltype(b_)::value;
          if constexpr (PC_CE(St::dep(r, b)))
            A.G[goff[St::GO_Q + m] + (int64_t)PC_CE(St::zrank(r, b)) * N + node] = k * sc[St::O_VZ + b] * Jv[PC_CE(St::jidx(r, b))];
        });
      }
    });
  }

  // ---- Hessian (compiled.py:484-500): the pieces built from the second partials are defined ahead of the first pass
  //      (edge_store, hess_second); here the t strips, built from the adjoint weights and the first partials
  auto hess_tstrips = [&]() {    // built from the adjoint weights and the first partials
    if (owns) {
      // time coupling through stretch: d/dt~_j of stretch * (mu . dF/dv), f and g rows only
      if constexpr (NT > 0) {
        static_for<0, NV>([&](auto c_) {
          constexpr int cvar = decltype(c_)::value;
          if constexpr (cvar < NZ ? PC_CE(St::tzx(cvar)) : PC_CE(St::tz(cvar))) {
            double acc = 0.0;
            static_for<0, NFN>([&](auto r_) {
              constexpr int r = decltype(r_)::value;
              if constexpr (!(r >= NY && r < NY + NP) && PC_CE(St::dep(r, cvar))) acc += mu[r] * Jv[PC_CE(St::jidx(r, cvar))];
            });
            if constexpr (cvar < NZ) {
              if (mine(PC_ITEM(St::IT_HT + cvar))) static_for<0, NT>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                double* dstp = A.H + hoff[St::HO_T + j * NZ + cvar] + node;
                double val = dst[j] * sc[St::O_VZ + cvar] * acc;
                // f, p or g depends on this time itself: the node Lagrangian's second partial (t_j, z) joins the strip
                constexpr int lt = PC_CE(St::tpar(j));
                if constexpr (lt >= 0) {
                  constexpr int eh = PC_CE(St::hidx(NZ + (lt >= 0 ? lt : 0), cvar));
                  if constexpr (eh >= 0) val += sc[St::O_VS + lt] * sc[St::O_VZ + cvar] * Hv[eh];
                }
                if (edge0 || edgeN) edge_store(ic<St::NHZZ + j * NZ + cvar>{}, dstp, val); else *dstp = val;
              });
            } else {
              red[PC_CE(St::rts(cvar - NZ))] = acc;
            }
          }
        });
      }
    }
  };
  if (wantH && hess_replica) {
    if constexpr (!SPLIT) hess_second();
    hess_tstrips();
  }

  if constexpr (!SPLIT && !RED_EARLY) deposit_partials();
  PC_STAMP(4);   // path / integral rows (and, fused build, the Hessian) done
  // ---- Jacobian of the defect rows (compiled.py:305-334), one state at a time:
  //      entries are produced column-wise / row-wise into the staging buffer, then the tile's
  //      contiguous CSR run of that state is written to HBM fully coalesced
  if (wantG) {
    // (wave-uniform values read from an LDS table on a mixed-order mesh: as scalars, so that the run's start address,
    //  its length and the row offsets below are scalar arithmetic)
    auto uniform64 = [](long long v) -> long long {
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)v);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
      return (long long)(((unsigned long long)hi << 32) | lo);
    };
    const long long E0 = uni ? S_E(lsA) : uniform64(S_E(lsA)), E1 = uni ? S_E(nsec) : uniform64(S_E(nsec));
    // Row groups (order-specialised bodies): a state's block is staged and flushed in NPASS passes over the section
    // rows -- pass g holds rows j in [1 + g RG, 1 + (g + 1) RG) of EVERY section of the tile, section after section, so
    // every lane still produces its column in every pass and the staging buffer is RG / (n - 1) of the whole block.
    // The staged piece of one section is contiguous in the CSR run too (rows of a section follow each other), so the
    // flush copies chunks of RG rows (flush_chunks).  NPASS is pc_row_passes' choice (pc_args.h; the host sizes the
    // LDS with the same function): 1 unless a full tile's block is too large for two staging regions per workgroup --
    // the high orders ph refinement ends on (7 .. 9 nodes: 82 doubles a row for Delta III) need 41 KB in one piece.
    constexpr int RSTRIDE_MAX = [] {
      int m = 0;
      for (int a = 0; a < NY; ++a) m = St::D(a) * (UN > 0 ? UN : 0) + St::C(a) > m ? St::D(a) * (UN > 0 ? UN : 0) + St::C(a) : m;
      return m;
    }();
    // (not in the four-wave per-replica kernels: four waves share a tile only on meshes of a few hundred tiles, where
    //  every wave has a SIMD to itself whatever the LDS footprint, and the passes only cost -- space station 6 k nodes,
    //  96 tiles x 4 waves: 17.5 us in one pass, 22.1 us in two)
    constexpr int NPASS = (UN > 0 && WN != 4) ? pc_row_passes(UN, RSTRIDE_MAX) : 1;
    constexpr int RG = UN > 0 ? (UN - 1 + NPASS - 1) / NPASS : 0;
    auto defect_jacobian_of_state = [&](auto a_) {
      constexpr int a = decltype(a_)::value;
      if (!mine(PC_ITEM(St::IT_D + a))) return;
      constexpr int Da = St::D(a), Ca = St::C(a);
      const double Wd = sc[St::O_WD + a];
      // local offset of row (section ls, row j) inside the run: rows before it in the tile
      // all offsets inside a tile fit 32 bits: section base + (j-1) * row length
      auto row_off = [&](int ls, int j, int n) -> int {
        const int eloc = uni ? (ls - lsA) * (n - 1) * n : (int)(S_E(ls) - E0);
        return Da * eloc + Ca * (S_s(ls) - n0) + (j - 1) * (Da * n + Ca);
      };
      // row a of dF/dz with every factor that does not depend on the CSR row folded in once:
      // W_a * stretch * (df_a/dz_b * V_b); an entry is then one multiply by h_k A[j][pos]
      double js[Da > 0 ? Da : 1];
      const double WS = Wd * stretch, WV = Wd * sc[St::O_VZ + a];
      if (active) {
        static_for<0, NZ>([&](auto b_) {
          constexpr int b = decltype(b_)::value;
          if constexpr (PC_CE(St::dep(a, b))) js[PC_CE(St::ndep_before(a, b))] = WS * (Jv[PC_CE(St::jidx(a, b))] * sc[St::O_VZ + b]);
        });
      }
      // q, t and s columns of a lane's own row, written at staging offset rs (behind the row's z blocks)
      auto param_cols = [&](int rs) {
        if constexpr (NT + PC_CE(St::nxdep(a)) > 0) {
          const int n = UN > 0 ? UN : n_r, jr = pos_r, sk = S_s(ls_r) - n0;
          auto dfdw = [&](auto l_) -> double {   // W stretch V_l h_k sum_i A[j][i] df_a/dw_l(z_i)
            constexpr int l = decltype(l_)::value;
            const double* Arow = s_qa + QAO(n) + (jr - 1) * n;
            double as = 0.0;
#pragma unroll
            for (int i = 0; i < n; ++i) as += Arow[i] * s_fs[PC_CE(St::fs_slot(a, l)) * TN + sk + i];
            return Wd * stretch * sc[St::O_VS + l] * (S_h(ls_r) * as);
          };
          static_for<0, NT>([&](auto jt_) {
            constexpr int jt = decltype(jt_)::value;
            double val = Wd * dst[jt] * accf[a];
            constexpr int lt = PC_CE(St::tpar(jt));   // f_a depends on this time itself
            if constexpr (lt >= 0) {
              if constexpr (PC_CE(St::dep(a, NZ + (lt >= 0 ? lt : 0)))) val += dfdw(ic<(lt >= 0 ? lt : 0)>{});
            }
            s_out[rs + PC_CE(St::nqdep(a)) + jt] = val;
          });
          static_for<0, NS>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            if constexpr (!St::is_t(l) && PC_CE(St::dep(a, NZ + l))) s_out[rs + PC_CE(St::xpos(a, l))] = dfdw(l_);
          });
        }
      };
      const int64_t g0 = goff[St::GO_D + a] + (int64_t)Da * E0 + (int64_t)Ca * n0;
      if constexpr (NPASS > 1) {
        constexpr int RS = Da * UN + Ca;   // doubles per row
        static_for<0, NPASS>([&](auto g_) {
          constexpr int g = decltype(g_)::value;
          constexpr int JLO = 1 + g * RG, JHI = (JLO + RG < UN) ? JLO + RG : UN, RGG = JHI - JLO;
          if constexpr (RGG > 0) {
            if (active) {
              // this lane's column in the rows JLO .. JHI-1 of section ls (staging: section q's piece at q * RGG rows)
              auto write_cols = [&](int ls, int pos, const double* cc) {
                const double h = HOIST ? 0.0 : S_h(ls);
                const double* At = s_qa;
                const int rs0 = (ls - lsA) * (RGG * RS);
                static_for<JLO, JHI>([&](auto j_) {
                  constexpr int jj = decltype(j_)::value;
                  const double coef = HOIST ? cc[jj - 1] : h * At[(jj - 1) * UN + pos];
                  static_for<0, NZ>([&](auto b_) {
                    constexpr int b = decltype(b_)::value;
                    if constexpr (PC_CE(St::dep(a, b))) {
                      constexpr int before = PC_CE(St::ndep_before(a, b));
                      constexpr int extra = (PC_CE(St::own_sparse(a)) && a < b) ? 2 : 0;
                      double val = coef * js[before];
                      if constexpr (a == b) val += WV * ((pos == 0 ? 1.0 : 0.0) - (pos == jj ? 1.0 : 0.0));
                      s_out[rs0 + (jj - JLO) * RS + before * UN + extra + pos] = val;
                    }
                  });
                });
              };
              if (ls_r >= lsA) write_cols(ls_r, pos_r, cr);
              if (has_start) write_cols(ls_s, 0, cs);
              if (t >= 1 && pos_r >= JLO && pos_r < JHI) {   // this lane's own row belongs to the group
                const int rs = (ls_r - lsA) * (RGG * RS) + (pos_r - JLO) * RS;
                if constexpr (PC_CE(St::own_sparse(a))) {
                  s_out[rs + PC_CE(St::ndep_before(a, a)) * UN] = WV;
                  s_out[rs + PC_CE(St::ndep_before(a, a)) * UN + 1] = -WV;
                }
                param_cols(rs + Da * UN + (PC_CE(St::own_sparse(a)) ? 2 : 0));
              }
            }
            stage_sync();
            flush_chunks<RGG * RS, (UN - 1) * RS>(A.G + g0 + (JLO - 1) * RS, s_out, k1 - k0, t, TN);
            stage_sync();
          }
        });
        if constexpr (a < 7) PC_STAMP(11 + 2 * a);
        return;
      }
      if (active) {
        auto write_cols = [&](int ls, int pos, int n_in, const double* cc) {
          const int n = UN > 0 ? UN : n_in;
          const double h = HOIST ? 0.0 : S_h(ls);
          const double* At = s_qa + (HOIST ? 0 : QAO(n));
          // (first row's offset once: inside the loop the section tables would be re-read from LDS for every row --
          //  the stores to the staging buffer may alias them as far as the compiler knows)
          const int rs1 = row_off(ls, 1, n), rstride = Da * n + Ca;
#pragma unroll
          for (int j = 1; j < n; ++j) {
            const double coef = HOIST ? cc[j - 1] : h * At[(j - 1) * n + pos];
            const int rs = rs1 + (j - 1) * rstride;
            static_for<0, NZ>([&](auto b_) {
              constexpr int b = decltype(b_)::value;
              if constexpr (PC_CE(St::dep(a, b))) {
                constexpr int before = PC_CE(St::ndep_before(a, b));
                constexpr int extra = (PC_CE(St::own_sparse(a)) && a < b) ? 2 : 0;
                double val = coef * js[before];
                if constexpr (a == b) val += WV * ((pos == 0 ? 1.0 : 0.0) - (pos == j ? 1.0 : 0.0));
                s_out[rs + before * n + extra + pos] = val;
              }
            });
          }
        };
        if (ls_r >= lsA) write_cols(ls_r, pos_r, n_r, cr);
        if (has_start) write_cols(ls_s, 0, n_s, cs);
        // a state whose own derivative does not depend on it has just the two D entries in its own columns (section
        // start: +W V, node j: -W V): both written by the lane that owns the ROW -- one predicated block per state
        // instead of two predicated stores per row and column pass
        if constexpr (PC_CE(St::own_sparse(a))) {
          if (t >= 1) {
            const int n = UN > 0 ? UN : n_r;
            const int o = row_off(ls_r, pos_r, n) + PC_CE(St::ndep_before(a, a)) * n;
            s_out[o] = WV;
            s_out[o + 1] = -WV;
          }
        }
        if (rowthr) {   // q, t and s columns of this lane's own row
          const int n = UN > 0 ? UN : n_r;
          param_cols(row_off(ls_r, pos_r, n) + Da * n + (PC_CE(St::own_sparse(a)) ? 2 : 0));
        }
      }
      stage_sync();
      if constexpr (a < 7) PC_STAMP(10 + 2 * a);   // state a's block produced into the staging buffer
      const int len = (int)((long long)Da * (E1 - E0)) + Ca * T;
      flush_run(A.G + g0, s_out, len, t, TN);
      stage_sync();
      if constexpr (a < 7) PC_STAMP(11 + 2 * a);   // ... read back and its stores issued
    };
    static_for<0, NY>(defect_jacobian_of_state);
  }

  PC_STAMP(5);   // Jacobian of the defect rows staged and stored
  if constexpr (SPLIT) {         // second pass: second partials, the Hessian runs built from them, then the sums
    second_pass();
    if constexpr (!RED_EARLY) deposit_partials();
  }
  // two-launch build: the waves' sums meet here, at the end, so that no replica waits for another mid-kernel
  if constexpr (NRED > 0 && !RES) {
    block_sync();
    if (tid < NRED) {
      const int nw = (TB + 63) >> 6;
      double sr = 0.0;
      for (int ww = 0; ww < nw; ++ww) sr += s_red[tid * 16 + ww];
      A.partials[(int64_t)tile * NRED + tid] = sr;
    }
  }
  PC_STAMP(7);   // instruction stream done, stores in flight
#ifdef PC_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  PC_STAMP(8);   // this wave's stores have left
}
#undef PC_ITEM

// The by-value PcPhaseArgs of pc_bulk_p<i>, read where it lies in the kernarg segment (behind the lead scalars).  A
// kernel that names the parameter itself gets a private copy of the 2.2 KB block once enough tile bodies read it (the
// compiler stops splitting the aggregate: seen with the twelve bodies of a mixed four-wave kernel, 2 240 B of scratch).
__device__ __forceinline__ const PcPhaseArgs& kernarg_phase_args() {
  typedef const __attribute__((address_space(4))) char* kseg_t;
  return *(const PcPhaseArgs*)((kseg_t)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(PcLead));
}

// Mixed build: run the tile of workgroup `block` with the body compiled for its order (ORD...: the orders this code
// object specialises for the phase) or, for a tile that spans sections of several orders, with the any-order body.
// The record is read with scalar loads (the workgroup index is uniform) and the choice is a chain of scalar branches.
template <class M, bool RES, int WN, int WIDX, int... ORD>
__device__ __forceinline__ void bulk_mix(std::integer_sequence<int, ORD...>, const PcTileRec* rec, const PcPhaseArgs& KA,
                                         bool MULTI, int first_block, int block, const PcLead* LD = nullptr,
                                         const double* mx = nullptr, const double* mlam = nullptr, double* mc = nullptr,
                                         double* mG = nullptr, double* mH = nullptr, int mflags = 0, unsigned mepoch = 0) {
  const int order = rec->order, k0 = rec->k0, nsec = rec->nsec, n0 = rec->n0, nprev = rec->nprev, qaprev = rec->qa_prev;
  const long long E0 = rec->E0;
  const double wprev = rec->w_prev;
  const bool done = (... || (order == ORD ? (bulk<M, ORD, RES, WN, WIDX, true>(KA, MULTI, first_block, block, LD, mx, mlam, mc, mG, mH,
                                                                              mflags, mepoch, k0, nsec, n0, nprev, qaprev, E0, wprev), true)
                                          : false));
  if (!done) bulk<M, 0, RES, WN, WIDX, true>(KA, MULTI, first_block, block, LD, mx, mlam, mc, mG, mH, mflags, mepoch);
}

// ---------------------------------------------------------------------------------------------
// tail pieces (one workgroup): the separate `pc_tail` launch, or block 0 of a resident-tail bulk launch
// ---------------------------------------------------------------------------------------------
// All of the tail's LDS is carved from the dynamic region (the resident build shares the bulk kernel's allocation;
// a static array would be added to every tile's footprint):
//   acc[n_tail_owned]  Hessian entries only the tail writes (sums over tiles, endpoint terms that meet no node
//                      block) are accumulated here and stored once: no zero-fill of global memory, no RMW round trips
//   part[16 * nred] sum[nred]   cross-wave partial sums of one phase
//   xb[NPV] lb[NB] hold[NPH] hb[NPH]   endpoint inputs, current edge values, endpoint Hessian terms
struct TailLds {
  double *acc, *part, *sum, *xb, *lb, *hold, *hb;
};
template <class PT>
__device__ __forceinline__ TailLds tail_lds(const PcTailArgs& A, double* base) {
  TailLds L;
  L.acc = base;
  L.part = L.acc + A.n_tail_owned;
  L.sum = L.part + 16 * A.lds_nred;
  L.xb = L.sum + A.lds_nred;
  L.lb = L.xb + PT::NPV;
  L.hold = L.lb + PT::NB;
  L.hb = L.hold + PT::NPH;
  return L;
}
__device__ __forceinline__ void tail_begin(const PcTailArgs& A, const TailLds& L) {
  if (A.flags & PC_FLAG_H)
    for (int i = threadIdx.x; i < A.n_tail_owned; i += A.block_threads) L.acc[i] = 0.0;
  lds_barrier();
}
__device__ __forceinline__ void tail_end(const PcTailArgs& A, const TailLds& L) {
  lds_barrier();
  if (A.flags & PC_FLAG_H)
    for (int i = threadIdx.x; i < A.n_tail_owned; i += A.block_threads) A.H[A.tail_owned[i]] = L.acc[i];
}
// The resident tail reads its argument block field by field, lazily: every first touch of a 64-byte line of the
// kernarg segment is a scalar-cache miss served from device memory, one after the other along the tail's critical
// path (measured: 1.7 us between the arrival of the last partial sum and the tail's last store, for a dozen
// instructions of arithmetic).  `issue` touches every line of the block with one scalar load at the tail's first
// instructions -- all misses overlap each other and the endpoint inputs' vector loads -- and `settle`, placed where
// those have returned anyway, keeps the loads alive.
struct KernargWarm {
  unsigned acc = 0;
  template <int BYTES>
  __device__ __forceinline__ void issue(int byte_off) {
    typedef const __attribute__((address_space(4))) unsigned* kp;
    kp p = (kp)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + byte_off);
#pragma unroll
    for (int i = 0; i < BYTES / 64; ++i) acc ^= p[16 * i];
  }
  __device__ __forceinline__ void settle() const { asm volatile("" ::"s"(acc)); }
};
// bounded spin of the resident tail: a pass that found a granule missing sleeps a little; PC_SPIN_LIMIT passes mean
// the producer will never come (a launch bug): the timeout word is set and the caller's wait loop ends
__device__ __forceinline__ bool spin_again(unsigned& spins, const PcTailArgs& A, unsigned code) {
  __builtin_amdgcn_s_sleep(2);
  if (++spins < (unsigned)PC_SPIN_LIMIT) return true;
  if (A.timeout) __hip_atomic_store(A.timeout, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return false;
}

// Finish the cross-tile sums of one phase: integral rows of c and G, (t,s)/(s,s) Hessian sums.  The order of the
// additions is fixed by 256 *virtual* lanes whatever the workgroup size (64, 128 or 256 threads: a lane plays
// 4, 2 or 1 of them): virtual lane v adds the tiles v, v + 256, ... in turn, the 64 lanes of a virtual wave are
// summed by wave_sum, the four wave totals are added in order.  Every build of the tail -- separate launch, resident
// block of any size -- therefore returns the same bits.
// Two steps, so that the separate launch can issue the first phase's loads at the very start of the kernel from
// preloaded scalars (PcTailLead).
template <class M>
struct TailPhaseRegs {
  using St = S<M>;
  double xt[2], vt[2], rt[2];
  double xq[St::NQ > 0 ? St::NQ : 1], wi[St::NQ > 0 ? St::NQ : 1], vq[St::NQ > 0 ? St::NQ : 1], rq[St::NQ > 0 ? St::NQ : 1];
  double vs[St::NS > 0 ? St::NS : 1], acc[4][St::NRED > 0 ? St::NRED : 1];
};
template <class M, bool RES = false, bool BIG = false>
__device__ __forceinline__ void tail_phase_issue(const PcTailArgs& A, int ip, TailPhaseRegs<M>& R, const PcTailLead* L = nullptr) {
  using St = S<M>;
  constexpr int NZ = St::NZ, NQ = St::NQ, NS = St::NS, NT = St::NT, NRED = St::NRED;
  if constexpr (NRED > 0) {
    const PcTailPhase& P = A.ph[ip];
    const int tid = threadIdx.x, TB = L ? L->block_threads : A.block_threads;  // 64, 128 or 256
    const int NG = 256 / TB;                                                    // virtual lanes per lane
    const double* xv = L ? L->x : A.x;
    // lane 0's own inputs are requested before the partial sums so that the two round trips overlap
    const double* sc = L ? L->scal0 : P.scal;
    const int N = L ? L->N0 : P.N;
    const int64_t q_off = (L ? L->x_off0 : P.x_off) + (int64_t)NZ * N, t_off = q_off + NQ;
    double (&xt)[2] = R.xt, (&vt)[2] = R.vt, (&rt)[2] = R.rt;
    auto& xq = R.xq; auto& wi = R.wi; auto& vq = R.vq; auto& rq = R.rq; auto& vs = R.vs; auto& acc = R.acc;
    xt[0] = xt[1] = vt[0] = vt[1] = rt[0] = rt[1] = 0.0;
    if (tid == 0) {
      static_for<0, NT>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        xt[j] = xv[t_off + j];
        vt[j] = sc[St::O_VT + j];
        rt[j] = sc[St::O_RT + j];
      });
      static_for<0, NQ>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        xq[m] = xv[q_off + m];
        wi[m] = sc[St::O_WI + m];
        vq[m] = sc[St::O_VQ + m];
        rq[m] = sc[St::O_RQ + m];
      });
      static_for<0, NS>([&](auto l_) { vs[decltype(l_)::value] = sc[St::O_VS + decltype(l_)::value]; });
    }
    static_for<0, 4>([&](auto g_) { static_for<0, NRED>([&](auto r_) { acc[decltype(g_)::value][decltype(r_)::value] = 0.0; }); });
    const int nt = L ? L->n_tiles0 : P.n_tiles;
    if constexpr (RES) {
      // the tiles publish their sums as granules while this workgroup runs: one stride of 256 tiles at a time, every
      // wave re-reading its granules of the stride until all carry this launch's tag
      const unsigned long long* gr = P.gran;
      unsigned spins = 0;
      // The strides are walked in the order in which a launch DISPATCHES its tiles -- position b stands for tile
      // xcd_major(b, nt), the tile workgroup b of the launch runs -- not in tile order: tiles 0 .. nt/8 all run on XCD 0
      // and the last of them is dispatched at the very end of the launch, so a walk in tile order stood still at the
      // second stride until the launch was over and then had every other stride left to do (39 k tiles: 153 strides,
      // 42 us after the last tile).  The separate tail kernel sums in the same order, so that the builds stay bitwise
      // equal; any fixed order is a correct sum.
      // SL tiles per lane and pass (tiles b0 + tid + TB s, s < SL): a pass is one memory round trip however many
      // tiles it covers, and behind a long bulk kernel the tail must not fall a round trip per stride behind; SL is
      // what the register budget allows (the kernel's VGPR count is the tiles' occupancy too).  Tile b belongs to
      // virtual lane group (b / TB) % NG and a group's tiles are added in rising order, as the plain loop does.
      constexpr int SL = NRED <= 2 ? 4 : (NRED <= 4 ? 2 : 1);
      for (int b0 = 0; b0 < nt; b0 += SL * TB) {
        double val[SL][NRED];
        for (;;) {
          // every granule this lane needs of the pass is requested before the first is looked at
          unsigned long long raw[SL][2 * NRED];
          static_for<0, SL>([&](auto s_) {
            constexpr int sl = decltype(s_)::value;
            const int b = b0 + tid + TB * sl;
            if (b < nt) {
              const int64_t tile_b = xcd_major(b, nt);   // the b-th tile in DISPATCH order (see below)
              static_for<0, 2 * NRED>([&](auto i_) { raw[sl][decltype(i_)::value] = load_granule(gr + 2 * tile_b * NRED + decltype(i_)::value); });
            }
          });
          bool ok = true;
          static_for<0, SL>([&](auto s_) {
            constexpr int sl = decltype(s_)::value;
            const int b = b0 + tid + TB * sl;
            if (b < nt)
              static_for<0, NRED>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                ok &= join_granules(raw[sl][2 * r], raw[sl][2 * r + 1], A.epoch, val[sl][r]);
              });
          });
          if (__all(ok)) break;
          if (!spin_again(spins, A, 1u + (unsigned)ip)) break;
        }
        const int gb = (b0 / TB) % NG;   // uniform
        static_for<0, SL>([&](auto s_) {
          constexpr int sl = decltype(s_)::value;
          const int b = b0 + tid + TB * sl;
          // (every group adds, the others an exact +0.0: a branch per group is folded by the compiler into a computed
          //  subscript, and that puts the accumulators -- and the dispatch -- in scratch memory)
          const int g = (gb + sl) % NG;
          static_for<0, 4>([&](auto g_) {
            constexpr int gg = decltype(g_)::value;
            const bool hit = b < nt && g == gg;
            static_for<0, NRED>([&](auto r_) { acc[gg][decltype(r_)::value] += hit ? val[sl][decltype(r_)::value] : 0.0; });
          });
        });
      }
    } else {
      const double* part = L ? L->partials0 : P.partials;
      // BIG (pc_tail_big, the build the host picks for many tiles): U strides of tiles are requested before the first
      // is added, in the same order as the plain loop adds them (a wait per stride made the tail 6 us behind a 27 us
      // bulk kernel at 4 k tiles, 58 us at 39 k); the plain loop stays the code of pc_tail, whose few strides leave
      // nothing to overlap
      constexpr int U = BIG ? (NRED <= 4 ? 8 : 4) : 1;
      static_for<0, 4>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        if (g >= NG) return;
        for (int b0 = tid + TB * g; b0 < nt; b0 += U * 256) {
          double tmp[U][NRED];
          static_for<0, U>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            const int b = b0 + u * 256;
            static_for<0, NRED>([&](auto r_) {
              constexpr int r = decltype(r_)::value;
              tmp[u][r] = b < nt ? part[(int64_t)xcd_major(b, nt) * NRED + r] : 0.0;   // (same order as the resident walk)
            });
          });
          static_for<0, U>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            if (b0 + u * 256 < nt) static_for<0, NRED>([&](auto r_) { acc[g][decltype(r_)::value] += tmp[u][decltype(r_)::value]; });
          });
        }
      });
    }
  }
}
template <class M>
__device__ __forceinline__ void tail_phase_finish(const PcTailArgs& A, int ip, TailPhaseRegs<M>& R, const TailLds& L) {
  using St = S<M>;
  constexpr int NZ = St::NZ, NQ = St::NQ, NP = St::NP, NY = St::NY, NS = St::NS, NT = St::NT, NRED = St::NRED;
  if constexpr (NRED > 0) {
    double* s_part = L.part;
    double* s_sum = L.sum;
    const PcTailPhase& P = A.ph[ip];
    const int tid = threadIdx.x, TB = A.block_threads, NG = 256 / TB;
    double (&xt)[2] = R.xt, (&vt)[2] = R.vt, (&rt)[2] = R.rt;
    auto& xq = R.xq; auto& wi = R.wi; auto& vq = R.vq; auto& rq = R.rq; auto& vs = R.vs; auto& acc = R.acc;
    lds_barrier();   // the previous phase's totals have been consumed
    static_for<0, 4>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      if (g >= NG) return;
      static_for<0, NRED>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        const double w = wave_sum(acc[g][r]);
        if ((tid & 63) == 0) s_part[r * 16 + (tid >> 6) + (TB >> 6) * g] = w;   // virtual wave of (lane, g)
      });
    });
    lds_barrier();
    if (tid == 0) {
      static_for<0, NRED>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        double tot = s_part[r * 16];
        for (int w = 1; w < 4; ++w) tot += s_part[r * 16 + w];
        s_sum[r] = tot;
      });
      double* hacc = L.acc;
      double t0 = P.t_fixed[0], tF = P.t_fixed[1], dst[2] = {0.0, 0.0};
      int j = 0;
      if constexpr (M::T0_FREE) {
        t0 = vt[j] * xt[j] + rt[j];
        dst[j] = -0.5 * vt[j];
        ++j;
      }
      if constexpr (M::TF_FREE) {
        tF = vt[j] * xt[j] + rt[j];
        dst[j] = 0.5 * vt[j];
      }
      const double stretch = 0.5 * (tF - t0);
      static_for<0, NQ>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        constexpr int r = NY + NP + m;
        const double Wi = wi[m];
        if (A.flags & PC_FLAG_C) {
          const double q = vq[m] * xq[m] + rq[m];
          A.c[P.c_int_off + m] = Wi * (q - stretch * s_sum[St::R_Q + m]);
        }
        if (A.flags & PC_FLAG_G) {
          int64_t o = P.gq_base[m];
          auto dgdw = [&](auto l_) -> double {   // -W stretch V_l sum_i w_i dg_m/dw_l(z_i)
            constexpr int l = decltype(l_)::value;
            return -Wi * stretch * vs[l] * s_sum[PC_CE(St::rqs(m, l))];
          };
          // q columns, ascending: the integrals the integrand depends on, with q_m's own 1 among them
          static_for<0, NS>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            if constexpr (St::is_q(l) && M::wi(l) < m && PC_CE(St::dep(r, NZ + l))) A.G[o++] = dgdw(l_);
          });
          {
            double val = Wi * vq[m];
            constexpr int lq = PC_CE(St::qpar(m));
            if constexpr (lq >= 0) {
              if constexpr (PC_CE(St::dep(r, NZ + (lq >= 0 ? lq : 0)))) val += dgdw(ic<(lq >= 0 ? lq : 0)>{});
            }
            A.G[o++] = val;
          }
          static_for<0, NS>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            if constexpr (St::is_q(l) && M::wi(l) > m && PC_CE(St::dep(r, NZ + l))) A.G[o++] = dgdw(l_);
          });
          static_for<0, NT>([&](auto jt_) {
            constexpr int jt = decltype(jt_)::value;
            double val = -Wi * dst[jt] * s_sum[St::R_Q + m];
            constexpr int lt = PC_CE(St::tpar(jt));
            if constexpr (lt >= 0) {
              if constexpr (PC_CE(St::dep(r, NZ + (lt >= 0 ? lt : 0)))) val += dgdw(ic<(lt >= 0 ? lt : 0)>{});
            }
            A.G[o++] = val;
          });
          static_for<0, NS>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            if constexpr (M::wk(l) == 0 && PC_CE(St::dep(r, NZ + l))) A.G[o++] = dgdw(l_);
          });
        }
      });
      if (A.flags & PC_FLAG_H) {
        static_for<0, NS>([&](auto l_) {
          constexpr int l = decltype(l_)::value;
          if constexpr (NT > 0 && PC_CE(St::tz(NZ + l))) {
            static_for<0, NT>([&](auto jt_) {
              constexpr int jt = decltype(jt_)::value;
              // d2/dt_j dw_l of stretch(t) (mu . F): dstretch/dt_j d(mu . F)/dw_l -- twice when w_l is t_j itself
              constexpr double twice = (St::is_t(l) && M::wi(l) == jt) ? 2.0 : 1.0;
              hacc[P.hsum_local[jt * NS + l]] += twice * (dst[jt] * vs[l] * s_sum[PC_CE(St::rts(l))]);
            });
          }
        });
        static_for<0, St::NH>([&](auto e_) {
          constexpr int e = decltype(e_)::value;
          if constexpr (M::hc(e) >= NZ) {
            constexpr int l = M::hr(e) - NZ, l2 = M::hc(e) - NZ;
            hacc[P.hsum_local[2 * NS + l * (l + 1) / 2 + l2]] += s_sum[PC_CE(St::rss(l, l2))];
          }
        });
      }
    }
  }
}
template <class M, bool RES = false, bool BIG = false>
__device__ __forceinline__ void tail_phase(const PcTailArgs& A, int ip, const TailLds& L) {
  TailPhaseRegs<M> R;
  tail_phase_issue<M, RES, BIG>(A, ip, R);
  tail_phase_finish<M>(A, ip, R, L);
}

// Endpoint functions: objective, endpoint constraint rows, their Jacobian and Hessian.  The inputs -- the point
// variables scattered over x~, the endpoint multipliers, (separate launch) the edge-node Hessian entries the endpoint
// terms are added to -- are fetched by as many lanes as there are values, all at once, into LDS (fetched by the
// evaluating lane alone they were scalar loads issued a register-file-full at a time: ~9 us of an 11 us tail for
// Delta III's 56 point variables).  The block is generated in PT::NPARTS parts of similar cost (codegen.py); lane 0
// of wave w evaluates the parts g with g % waves == w and stores their rows of c~ and G~; the Hessian terms go to LDS
// and are applied by all lanes once the phases' sums are in (tail_point_apply).
template <class PT, bool RES = false>
__device__ __forceinline__ void tail_point_load(const PcTailArgs& A, const TailLds& L) {
  constexpr int NPV = PT::NPV, NB = PT::NB, NPH = PT::NPH;
  const bool wantH = A.flags & PC_FLAG_H;
  const int tid = threadIdx.x, TB = A.block_threads;
  for (int i = tid; i < NPV; i += TB) L.xb[i] = A.pt_V[i] * A.x[A.pt_x[i]] + A.pt_r[i];
  for (int r = tid; r < NB; r += TB) L.lb[r] = wantH ? A.lam[A.c_end_off + r] * A.pt_W[r] : 0.0;
  if constexpr (!RES) {   // (resident build: those entries arrive as granules, see tail_point_apply)
    if (wantH)
      for (int e = tid; e < NPH; e += TB) L.hold[e] = A.pt_hlocal[e] < 0 ? A.H[A.pt_hslot[e]] : 0.0;
  }
  // (the workgroup barrier of tail_begin, which follows, publishes the arrays)
}
// (tb, ntb): this workgroup is tail block tb of ntb -- a heavy endpoint block whose tiles run as single waves gets
// one single-wave tail block per part instead of one four-wave block (a wider launch would cost every tile three
// idle waves' registers); block tb evaluates the parts g with g % ntb == tb, one per wave.
template <class PT>
__device__ __forceinline__ void tail_point_eval(const PcTailArgs& A, const TailLds& L, int tb = 0, int ntb = 1) {
  constexpr int NPV = PT::NPV, NB = PT::NB, NGJ = PT::NGJ, NBJ = PT::NBJ, NPH = PT::NPH;
  const int tid = threadIdx.x, wave = tid >> 6, nw = (A.block_threads + 63) >> 6;
  if ((tid & 63) != 0) return;
  const double sigma = A.sigma, wJ = A.wJ;
  const bool wantH = A.flags & PC_FLAG_H;
  static_for<0, PT::NPARTS>([&](auto g_) {
    constexpr int g = decltype(g_)::value;
    if (g % ntb != tb || (g / ntb) % nw != wave) return;
    double xb[NPV > 0 ? NPV : 1], lb[NB > 0 ? NB : 1];
    static_for<0, NPV>([&](auto i_) { xb[decltype(i_)::value] = L.xb[decltype(i_)::value]; });
    static_for<0, NB>([&](auto r_) { lb[decltype(r_)::value] = L.lb[decltype(r_)::value]; });
    double Jval = 0.0, gJ[NGJ > 0 ? NGJ : 1], b[NB > 0 ? NB : 1], jb[NBJ > 0 ? NBJ : 1], hb[NPH > 0 ? NPH : 1];
    PT::template eval_part<g>(xb, sigma * wJ, lb, Jval, gJ, b, jb, hb);
    if constexpr (g == 0) {
      if (A.fobj) A.fobj[0] = wJ * Jval;
      if (A.grad_nz) {
        static_for<0, NGJ>([&](auto e_) {
          constexpr int e = decltype(e_)::value;
          A.grad_nz[e] = wJ * gJ[e] * A.pt_V[PT::gc(e)];
        });
      }
    }
    if (A.flags & PC_FLAG_C)
      static_for<0, NB>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        if constexpr (PT::part_b(r) == g) A.c[A.c_end_off + r] = A.pt_W[r] * b[r];
      });
    if (A.flags & PC_FLAG_G)
      static_for<0, NBJ>([&](auto e_) {
        constexpr int e = decltype(e_)::value;
        if constexpr (PT::part_jb(e) == g) A.G[A.g_end_base + e] = A.pt_W[PT::br(e)] * jb[e] * A.pt_V[PT::bc(e)];
      });
    if (wantH)
      static_for<0, NPH>([&](auto e_) {
        constexpr int e = decltype(e_)::value;
        if constexpr (PT::part_hb(e) == g) L.hb[e] = hb[e] * A.pt_V[PT::phr(e)] * A.pt_V[PT::phc(e)];
      });
  });
}
// helper tail blocks hand the Hessian terms of their parts to block 0 as granules; block 0 collects them
template <class PT>
__device__ __forceinline__ void tail_point_publish(const PcTailArgs& A, const TailLds& L, int tb, int ntb) {
  constexpr int NPH = PT::NPH;
  lds_barrier();
  if constexpr (NPH > 0) {
    if (!(A.flags & PC_FLAG_H)) return;
    for (int e = threadIdx.x; e < NPH; e += A.block_threads)
      if (PT::part_hb(e) % ntb == tb) publish_granules(A.hb_gran + 2 * e, A.epoch, L.hb[e]);
  }
}
template <class PT>
__device__ __forceinline__ void tail_point_collect(const PcTailArgs& A, const TailLds& L, int ntb) {
  constexpr int NPH = PT::NPH;
  if constexpr (NPH > 0) {
    if (ntb <= 1 || !(A.flags & PC_FLAG_H)) return;
    unsigned spins = 0;
    for (int e0 = 0; e0 < NPH; e0 += A.block_threads) {
      const int e = e0 + threadIdx.x;
      const bool mine = e < NPH && PT::part_hb(e) % ntb != 0;
      double v = 0.0;
      for (;;) {
        bool ok = true;
        if (mine) {
          const unsigned long long lo = load_granule(A.hb_gran + 2 * e), hi = load_granule(A.hb_gran + 2 * e + 1);
          ok = join_granules(lo, hi, A.epoch, v);
        }
        if (__all(ok)) break;
        if (!spin_again(spins, A, 200u)) break;
      }
      if (mine) L.hb[e] = v;
    }
  }
}

// every endpoint Hessian term meets its slot: a tail-owned slot in LDS (each slot receives one term, after the
// phases' sums), or an edge-node entry the bulk produced -- read back from H (separate launch) or received as
// granules from the edge tile (resident build) -- plus the term
template <class PT, bool RES = false>
__device__ __forceinline__ void tail_point_apply(const PcTailArgs& A, const TailLds& L) {
  constexpr int NPH = PT::NPH;
  lds_barrier();   // the terms (lane 0 of every wave) and the phases' sums (lane 0 of wave 0) are in LDS
  if (!(A.flags & PC_FLAG_H)) return;
  const int tid = threadIdx.x, TB = A.block_threads;
  if constexpr (NPH > 0) {
    for (int e = tid; e < NPH; e += TB) {
      const int hl = A.pt_hlocal[e];
      if (hl >= 0) L.acc[hl] += L.hb[e];
      else if constexpr (!RES) A.H[A.pt_hslot[e]] = L.hold[e] + L.hb[e];
    }
  }
  if constexpr (RES) {
    // every Hessian entry of the edge nodes arrives here as a record; stored as it is, or with its endpoint term
    unsigned spins = 0;
    for (int r0 = 0; r0 < A.n_rec; r0 += TB) {
      const int r = r0 + tid;
      const int64_t slot = r < A.n_rec ? A.rec_slot[r] : -1;
      const bool live = slot >= 0;
      double cur = 0.0;
      for (;;) {
        bool ok = true;
        if (live) {
          const unsigned long long* g = A.erec + 2 * (int64_t)r;
          const unsigned long long lo = load_granule(g), hi = load_granule(g + 1);
          ok = join_granules(lo, hi, A.epoch, cur);
        }
        if (__all(ok)) break;
        if (!spin_again(spins, A, 100u)) break;
      }
      if (live) {
        const int term = A.rec_term[r];
        if constexpr (NPH > 0) A.H[slot] = term >= 0 ? cur + L.hb[term] : cur;
        else A.H[slot] = cur;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ph mesh-error estimate (pycollo/mesh_refinement.py:63-240 + solution/solution_abc.py:60-107)
// One workgroup per run of sections; section k owns n_k + 1 consecutive lanes = its nodes on the "ph mesh"
// (one node more per section).  Lane j < n_k also evaluates solution node j of the section.
//   1. f at the solution nodes                      (casadi_solution.py:71)
//   2. states / controls at the interior ph nodes from the section's interpolants, as two small
//      table contractions:  y_ph = y_start + stretch h_k B f,   u_ph = E u   (solution_abc.py:70-100)
//   3. f at the ph nodes, Y_ph = y_start + stretch h_k A_(n+1) f_ph          (mesh_refinement.py:199-210)
//   4. |Y_ph - y_ph| relative to 1 + (1 + max|y_ph|) per state, section maximum (mesh_refinement.py:211-233)
// ---------------------------------------------------------------------------------------------
template <class M>
__device__ __forceinline__ void mesh_error(const PcRefineArgs& A) {
  using St = S<M>;
  constexpr int NY = St::NY, NU = St::NU, NZ = St::NZ, NS = St::NS, NQ = St::NQ;
  extern __shared__ double smem[];
  const int tid = threadIdx.x, TB = blockDim.x;
  double* s_B = smem;
  double* s_E = s_B + A.tab_total_BE;
  double* s_A = s_E + A.tab_total_BE;
  double* s_fs = s_A + A.tab_total_A;          // [NY][TB] f at solution nodes
  double* s_ys = s_fs + NY * TB;               // [NY][TB] y at solution nodes
  double* s_us = s_ys + NY * TB;               // [NU][TB]
  double* s_yp = s_us + (NU > 0 ? NU : 1) * TB;  // [NY][TB] y on the ph mesh
  double* s_fp = s_yp + NY * TB;               // [NY][TB] f on the ph mesh
  double* s_re = s_fp + NY * TB;               // [TB] row maxima of the relative error
  double* s_ae = s_re + TB;                    // [NY][TB] absolute errors
  int* s_sec = reinterpret_cast<int*>(s_ae + NY * TB);   // [TB] section of every lane
  const int k0 = A.tile_k0[blockIdx.x], k1 = A.tile_k0[blockIdx.x + 1];
  for (int i = tid; i < A.tab_total_BE; i += TB) {
    s_B[i] = A.tabB[i];
    s_E[i] = A.tabE[i];
  }
  for (int i = tid; i < A.tab_total_A; i += TB) s_A[i] = A.tabA[i];
  s_sec[tid] = -1;
  __syncthreads();
  for (int k = k0 + tid; k < k1; k += TB) {
    const int n = A.sec_s[k + 1] - A.sec_s[k] + 1, l0 = A.lane0[k];
    for (int j = 0; j <= n; ++j) s_sec[l0 + j] = k;
  }
  __syncthreads();
  const int k = s_sec[tid];
  const bool active = k >= 0;
  const double* sc = A.scal;
  double t0 = A.t_fixed[0], tF = A.t_fixed[1];
  {
    const int64_t t_off = A.x_off + (int64_t)NZ * A.N + NQ;
    int j = 0;
    if constexpr (M::T0_FREE) { t0 = sc[St::O_VT + j] * A.x[t_off + j] + sc[St::O_RT + j]; ++j; }
    if constexpr (M::TF_FREE) { tF = sc[St::O_VT + j] * A.x[t_off + j] + sc[St::O_RT + j]; }
  }
  const double stretch = 0.5 * (tF - t0);
  int n = 2, j = 0, l0 = 0, sk = 0;
  double h = 0.0;
  double v[St::NV > 0 ? St::NV : 1], F[NY > 0 ? NY : 1];
  if (active) {
    sk = A.sec_s[k];
    n = A.sec_s[k + 1] - sk + 1;
    l0 = A.lane0[k];
    j = tid - l0;
    h = A.sec_h[k];
    static_for<0, NS>([&](auto l_) {
      constexpr int l = decltype(l_)::value;
      constexpr int kind = M::wk(l), idx = M::wi(l);   // static parameter, or this phase's q / free t
      const int64_t col = kind == 0 ? A.s_off + idx : A.x_off + (int64_t)NZ * A.N + (kind == 1 ? idx : St::NQ + idx);
      v[NZ + l] = sc[St::O_VS + l] * A.x[col] + sc[St::O_RS + l];
    });
    if (j < n) {   // this lane doubles as solution node j of its section
      static_for<0, NZ>([&](auto b_) {
        constexpr int b = decltype(b_)::value;
        v[b] = sc[St::O_VZ + b] * A.x[A.x_off + (int64_t)b * A.N + sk + j] + sc[St::O_RZ + b];
      });
      M::eval_f(v, F);
      static_for<0, NY>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        s_fs[a * TB + tid] = F[a];
        s_ys[a * TB + tid] = v[a];
      });
      static_for<0, NU>([&](auto b_) { s_us[decltype(b_)::value * TB + tid] = v[NY + decltype(b_)::value]; });
    }
  }
  __syncthreads();
  if (active) {
    const double* Bt = s_B + A.offBE[n];
    const double* Et = s_E + A.offBE[n];
    if (j == 0 || j == n) {   // section boundaries keep the solution's values (mesh_refinement.py:164-166,176-178)
      const int src = l0 + (j == 0 ? 0 : n - 1);
      static_for<0, NY>([&](auto a_) { v[decltype(a_)::value] = s_ys[decltype(a_)::value * TB + src]; });
      static_for<0, NU>([&](auto b_) { v[NY + decltype(b_)::value] = s_us[decltype(b_)::value * TB + src]; });
    } else {
      static_for<0, NY>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += Bt[(j - 1) * n + i] * s_fs[a * TB + l0 + i];
        v[a] = s_ys[a * TB + l0] + stretch * (h * acc);
      });
      static_for<0, NU>([&](auto b_) {
        constexpr int b = decltype(b_)::value;
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += Et[(j - 1) * n + i] * s_us[b * TB + l0 + i];
        v[NY + b] = acc;
      });
    }
    M::eval_f(v, F);
    static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      s_yp[a * TB + tid] = v[a];
      s_fp[a * TB + tid] = F[a];
    });
  }
  __syncthreads();
  double rel = 0.0;
  if (active && j >= 1) {
    const double* At = s_A + A.offA[n] + (j - 1) * (n + 1);
    static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      double acc = 0.0;
      for (int i = 0; i <= n; ++i) acc += At[i] * s_fp[a * TB + l0 + i];
      const double Yph = s_yp[a * TB + l0] + stretch * (h * acc);
      const double err = fabs(Yph - s_yp[a * TB + tid]);
      double mx = 0.0;
      for (int i = 1; i <= n; ++i) mx = fmax(mx, fabs(s_yp[a * TB + l0 + i]));
      rel = fmax(rel, err / (1.0 + (mx + 1.0)));
      s_ae[a * TB + tid] = err;
    });
    s_re[tid] = rel;
  }
  __syncthreads();
  if (active && j == 0) {
    double m = 0.0;
    for (int i = 1; i <= n; ++i) m = fmax(m, s_re[l0 + i]);
    A.max_rel[k] = m;
    static_for<0, NY>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      double ma = 0.0;
      for (int i = 1; i <= n; ++i) ma = fmax(ma, s_ae[a * TB + l0 + i]);
      A.max_abs[(int64_t)k * NY + a] = ma;
    });
  }
}

}  // namespace pc
