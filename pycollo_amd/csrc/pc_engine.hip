// C-ABI engine: owns device memory, the per-problem code object and the launch sequence.
// See include/pycollo_amd.h for the contract and the reference interfaces each entry point replaces.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pycollo_amd.h"
#include "pc_args.h"
#include "pc_pattern.hpp"

namespace {

thread_local std::string g_err;

void set_err(const std::string& s) { g_err = s; }

#define HIP_OK(expr)                                                                             \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_));              \
  } while (0)

// LDS plan must match pc_kernels.hpp::lds_plan (kept in one place there; mirrored here because this
// TU does not include device templates).
int lds_doubles(int TB, int qa_total, int qw_total, int NY, int NFS, int NRED, int lds_out) {
  int o = 0;
  o += qa_total;
  o += qw_total;
  o += PC_MAX_ORDER + 1;
  o += PC_MAX_SCAL + PC_MAX_GOFF + PC_MAX_HOFF;
  o += TB + 2;
  o += TB + 2;
  o += (TB + 4) / 2 + 1;
  o += (TB + 1) / 2 + 1;
  o += NFS * TB;
  o += (NRED > 0 ? NRED : 1) * 16;
  const int node_arrays = 2 * NY * TB + NY * (TB + PC_MAX_ORDER);
  o += std::max(node_arrays, lds_out);   // the staging buffer overlays f / y / lambda
  return o;
}

// doubles of the output staging buffer of one phase: the longest CSR run a tile emits in one piece
int phase_lds_out(const pcp::Phase& P, int n_s, int TB) {
  int nmax = 0;
  for (int k = 0; k < P.K; ++k) nmax = std::max(nmax, (int)P.n_k[k]);
  int out = 0;
  for (int a = 0; a < P.n_y; ++a) {
    int Da = 0, Ca = P.n_t + (P.dep(a, a) ? 0 : 2);
    for (int b = 0; b < P.n_z; ++b) Da += P.dep(a, b) ? 1 : 0;
    for (int l = 0; l < n_s; ++l) Ca += P.dep(a, P.n_z + l) ? 1 : 0;
    out = std::max(out, (Da * nmax + Ca) * (TB - 1));
  }
  for (int m = 0; m < P.n_p; ++m) {
    int R = 0;
    for (int c = 0; c < P.n_v; ++c) R += P.dep(P.n_y + m, c) ? 1 : 0;
    out = std::max(out, R * TB);
  }
  for (int b = 0; b < P.n_z; ++b) out = std::max(out, P.hrow_count(b) * TB);
  return out;
}

int phase_lds_bytes(const pcp::Phase& P, int n_s, int TB, int qa_total, int qw_total) {
  int nfs = 0;
  for (int a = 0; a < P.n_y; ++a)
    for (int l = 0; l < n_s; ++l) nfs += P.dep(a, P.n_z + l) ? 1 : 0;
  return 8 * lds_doubles(TB, qa_total, qw_total, P.n_y, nfs, P.nred, phase_lds_out(P, n_s, TB));
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  void alloc(size_t count) {
    free();
    n = count;
    if (count) HIP_OK(hipMalloc(&p, count * sizeof(T)));
  }
  void upload(const std::vector<T>& v) {
    alloc(v.size());
    if (!v.empty()) HIP_OK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  void free() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { free(); }
};

template <class T>
struct PinBuf {
  T* p = nullptr;
  size_t n = 0;
  void alloc(size_t count) {
    free();
    n = count;
    if (count) HIP_OK(hipHostMalloc(&p, count * sizeof(T), hipHostMallocDefault));
  }
  void free() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = 0;
  }
  ~PinBuf() { free(); }
};

struct PhaseDev {
  DevBuf<int32_t> tile_k0, tile_n0, sec_s;
  DevBuf<double> sec_h, scal, partials, tab;
  DevBuf<int64_t> sec_E, hslot0, hslotN, hsum_slot;
  DevBuf<int32_t> hsum_local;
  DevBuf<long long> dbg;
  int uni_n = 0, spt = 0, lds_out = 0;
  int wpt = 1;                       // waves (replicas) per 64-node tile, see pc::bulk
  hipFunction_t fn = nullptr;
  hipFunction_t fn_fused = nullptr;  // last phase only: bulk kernel with the tail folded in
  int lds_bytes = 0, n_tiles = 0, nfs = 0;
  int tile_begin = 0, tile_end = 0;  // launched tile range (whole phase unless sharded)
  double* partials_ext = nullptr;    // caller-owned partial-sum buffer (sharded exchange), else `partials`
  std::vector<double> scal_host;
};

// generic kernel: 2-norm of every CSR row (scaling.py:392-395 without densifying)
__global__ void row_norms_kernel(const int64_t* __restrict__ indptr, const double* __restrict__ val,
                                 double* __restrict__ out, int64_t m) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= m) return;
  double acc = 0.0;
  for (int64_t i = indptr[wave] + lane; i < indptr[wave + 1]; i += 64) acc += val[i] * val[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) out[wave] = sqrt(acc);
}

// generic kernel: piecewise-linear interpolation of every row of vals_prev[n_vars][n_prev] from the
// abscissae tau_prev onto tau_new, extrapolating with the end segments -- the arithmetic of
// scipy.interpolate.interp1d(kind="linear", fill_value="extrapolate") that the reference uses to carry a
// guess / solution to the next mesh (pycollo/iteration.py:96-137): slope * (x - x_lo) + y_lo.
__global__ void interp_linear_kernel(const double* __restrict__ tau_prev, int n_prev, const double* __restrict__ vals_prev,
                                     int n_vars, const double* __restrict__ tau_new, int n_new, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const double x = tau_new[i];
  // searchsorted(tau_prev, x, side="left") clipped to [1, n_prev-1]
  int lo = 0, hi = n_prev;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tau_prev[mid] < x) lo = mid + 1; else hi = mid;
  }
  int idx = lo < 1 ? 1 : (lo > n_prev - 1 ? n_prev - 1 : lo);
  const double x_lo = tau_prev[idx - 1], x_hi = tau_prev[idx];
  for (int v = 0; v < n_vars; ++v) {
    const double y_lo = vals_prev[(size_t)v * n_prev + idx - 1], y_hi = vals_prev[(size_t)v * n_prev + idx];
    const double slope = (y_hi - y_lo) / (x_hi - x_lo);
    out[(size_t)v * n_new + i] = slope * (x - x_lo) + y_lo;
  }
}

// Copy a list of contiguous runs between two device buffers: the pack / unpack step around the all-gather of the
// section-sharded evaluation (one rank's share of c~, G~, H~ is a handful of CSR runs per state / variable).
// One workgroup per chunk of at most PC_RUN_CHUNK doubles; chunks[3*i .. 3*i+2] = (src offset, dst offset, length).
#define PC_RUN_CHUNK 2048
__global__ void copy_runs_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                 const int64_t* __restrict__ chunks) {
  const int64_t so = chunks[3 * (int64_t)blockIdx.x], d0 = chunks[3 * (int64_t)blockIdx.x + 1];
  const int len = (int)chunks[3 * (int64_t)blockIdx.x + 2];
  for (int i = threadIdx.x; i < len; i += blockDim.x) dst[d0 + i] = src[so + i];
}

}  // namespace

struct pc_handle {
  pcp::Problem Q;
  int device = -1;
  int TB = 64;
  int TC = 64;   // nodes a tile may hold (<= TB)
  bool scaling_set = false;
  double w_J = 1.0;
  // quadrature
  std::vector<double> qa, qw;
  int32_t qa_off[PC_MAX_ORDER + 1], qw_off[PC_MAX_ORDER + 1];
  // device
  hipStream_t stream = nullptr;
  hipModule_t module = nullptr;
  hipFunction_t tail_fn = nullptr;
  hipFunction_t tail_big_fn = nullptr;   // same kernel with the partial-sum loads of several strides in flight
  hipFunction_t bulk_all_fn = nullptr;   // multi-phase problems: every phase's bulk kernel in one launch
  DevBuf<char> d_phase_args;             // [n_phases] PcPhaseArgs read by pc_bulk_all
  bool args_dirty = true;                // scaling / tile range / partials buffer changed since the last upload
  // host-side argument blocks, filled once per change of scaling / tile range / partials buffer; a call only
  // patches the caller's pointers, the flags and sigma into them
  std::vector<PcBulkArgs> host_bulk_args;   // per phase: lead scalars + argument block, kept filled between calls
  PcTailLaunch host_tail_launch;   // lead scalars + argument block of pc_tail, kept filled between calls
  bool host_args_dirty = true;
  int wpt_all = 1, lds_all = 0;          // launch shape of pc_bulk_all
  std::vector<std::unique_ptr<PhaseDev>> pd;
  DevBuf<double> d_qa, d_x, d_lam, d_c, d_G, d_H, d_fobj, d_grad, d_pointV, d_pointr, d_Wend, d_norms;
  DevBuf<int64_t> d_point_x, d_tail_owned, d_pt_hslot, d_g_indptr;
  DevBuf<int32_t> d_pt_hlocal;
  std::vector<double> h_pointV, h_pointr, h_Wend;   // host copies: travel by value in PcTailArgs
  DevBuf<unsigned> d_sync;   // arrival counters of the fused tail (zero between launches)
  bool allow_fuse = false;   // PYCOLLO_AMD_FUSE=1 folds the tail into the last bulk launch (experimental:
                             // measured no faster than two launches on MI355X)
  PinBuf<double> h_x, h_lam, h_c, h_G, h_H, h_fobj, h_grad, h_norms;
  std::vector<double> V_ocp, r_ocp, W_ocp;
  // cache for new_x == 0
  bool have_cG = false;
  int n_launches = 0;
  int dbg_stage = 0;  // PYCOLLO_AMD_DBG_STAGE: diagnostic timeline build (profiling only)
  int lds_max = 0;
  int lds_limit = 64 * 1024;  // dynamic LDS a workgroup may request (queried from the device)
};

namespace {

void fill_point_tables(pc_handle* h, PcTailArgs& t) {
  auto& Q = h->Q;
  for (size_t i = 0; i < Q.point_x.size(); ++i) {
    t.pt_x[i] = Q.point_x[i];
    t.pt_V[i] = h->h_pointV[i];
    t.pt_r[i] = h->h_pointr[i];
  }
  for (size_t r = 0; r < h->h_Wend.size(); ++r) t.pt_W[r] = h->h_Wend[r];
}

void fill_tail_args(pc_handle* h, PcTailArgs& t, const double* d_x, const double* d_lam, double* d_c, double* d_G,
                    double* d_H, double* d_fobj, double* d_grad, int flags, double sigma) {
  auto& Q = h->Q;
  std::memset(&t, 0, sizeof(t));
  t.x = d_x;
  t.lam = d_lam;
  t.c = d_c;
  t.G = d_G;
  t.H = d_H;
  t.fobj = d_fobj;
  t.grad = d_grad;
  t.sigma = sigma;
  t.wJ = h->w_J;
  t.point_x = h->d_point_x.p;
  t.point_V = h->d_pointV.p;
  t.point_r = h->d_pointr.p;
  t.W_end = h->d_Wend.p;
  fill_point_tables(h, t);
  t.tail_owned = h->d_tail_owned.p;
  t.pt_hslot = h->d_pt_hslot.p;
  t.pt_hlocal = h->d_pt_hlocal.p;
  t.c_end_off = Q.c_end_off;
  t.g_end_base = Q.g_end_base;
  t.n_tail_owned = (int32_t)Q.tail_owned.size();
  t.flags = flags;
  t.block_threads = PC_TAIL_THREADS;
  for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
    auto& P = Q.ph[ip];
    auto& D = *h->pd[ip];
    PcTailPhase& tp = t.ph[ip];
    tp.partials = D.partials_ext ? D.partials_ext : D.partials.p;
    tp.scal = D.scal.p;
    tp.x_off = P.x_off;
    tp.s_off = Q.s_off;
    tp.c_int_off = P.c_int_off;
    for (int m = 0; m < 8; ++m) tp.gq_base[m] = P.gq_base[m];
    tp.hsum_slot = D.hsum_slot.p;
    tp.hsum_local = D.hsum_local.p;
    tp.t_fixed[0] = P.t_fixed[0];
    tp.t_fixed[1] = P.t_fixed[1];
    tp.n_tiles = D.n_tiles;
    tp.N = P.N;
  }
}

// The argument block of one phase's bulk kernel (everything but the per-call pointers and flags when `d_x` is null).
void fill_phase_args(pc_handle* h, size_t ip, PcPhaseArgs& a, const double* d_x, const double* d_lam, double* d_c,
                     double* d_G, double* d_H, int flags, int wpt) {
  auto& Q = h->Q;
  auto& P = Q.ph[ip];
  auto& D = *h->pd[ip];
  std::memset(&a, 0, sizeof(a));
  a.x = d_x;
  a.lam = d_lam;
  a.c = d_c;
  a.G = d_G;
  a.H = d_H;
  a.tile_k0 = D.tile_k0.p;
  a.tile_n0 = D.tile_n0.p;
  a.sec_s = D.sec_s.p;
  a.sec_h = D.sec_h.p;
  a.sec_E = D.sec_E.p;
  a.qa = h->d_qa.p;
  a.qw = h->d_qa.p + h->qa.size();
  if (D.scal_host.size() > PC_MAX_SCAL) throw std::runtime_error("too many scaling constants for the kernel argument block");
  for (size_t i = 0; i < D.scal_host.size(); ++i) a.scal[i] = D.scal_host[i];
  for (size_t i = 0; i < P.goff.size(); ++i) a.goff[i] = P.goff[i];
  for (size_t i = 0; i < P.hoff.size(); ++i) a.hoff[i] = P.hoff[i];
  a.uni_n = D.uni_n;
  a.spt = D.spt;
  a.lds_out = D.lds_out;
  a.wpt = wpt;
  a.dbg_stage = h->dbg_stage;
  a.hslot0 = D.hslot0.p;
  a.hslotN = D.hslotN.p;
  a.partials = D.partials_ext ? D.partials_ext : D.partials.p;
  a.dbg = D.dbg.p;
  a.sync = h->d_sync.p;
  a.tab = D.tab.p;
  a.x_off = P.x_off;
  a.s_off = Q.s_off;
  a.c_off = P.c_off;
  a.c_path_off = P.c_path_off;
  a.c_int_off = P.c_int_off;
  a.t_fixed[0] = P.t_fixed[0];
  a.t_fixed[1] = P.t_fixed[1];
  a.N = P.N;
  a.K = P.K;
  a.n_tiles = D.n_tiles;
  a.flags = flags;
  a.tile_begin = D.tile_begin;
  a.n_blocks = std::max(0, D.tile_end - D.tile_begin);
  a.block_threads = h->TB * wpt;
  a.qa_total = (int32_t)h->qa.size();
  a.qw_total = (int32_t)h->qw.size();
  std::memcpy(a.qa_off, h->qa_off, sizeof(a.qa_off));
  std::memcpy(a.qw_off, h->qw_off, sizeof(a.qw_off));
}

void launch_all(pc_handle* h, const double* d_x, const double* d_lam, double* d_c, double* d_G, double* d_H,
                double* d_fobj, double* d_grad, int flags, hipStream_t st, double sigma, bool bulk = true,
                bool tail = true) {
  auto& Q = h->Q;
  const size_t last = Q.ph.size() - 1;
  // one launch per evaluation when the last phase runs whole: its last workgroup to arrive runs the tail
  bool fuse = false;
  if (bulk && tail && h->allow_fuse && (h->dbg_stage == 0 || h->dbg_stage == 9)) {
    auto& D = *h->pd[last];
    fuse = !h->bulk_all_fn && D.fn_fused && D.tile_begin == 0 && D.tile_end == D.n_tiles && D.n_tiles <= 8192;
  }
  struct Both {
    PcPhaseArgs a;
    PcTailArgs t;
  };
  if (bulk && h->bulk_all_fn) {
    // several phases, one launch: the phases' workgroups run side by side instead of one kernel after another
    if (h->args_dirty) {
      std::vector<PcPhaseArgs> blocks(Q.ph.size());
      for (size_t ip = 0; ip < Q.ph.size(); ++ip)
        fill_phase_args(h, ip, blocks[ip], nullptr, nullptr, nullptr, nullptr, nullptr, 0, h->wpt_all);
      HIP_OK(hipDeviceSynchronize());   // no launch in flight may still read the old blocks
      HIP_OK(hipMemcpy(h->d_phase_args.p, blocks.data(), blocks.size() * sizeof(PcPhaseArgs), hipMemcpyHostToDevice));
      h->args_dirty = false;
    }
    PcMultiArgs m;
    std::memset(&m, 0, sizeof(m));
    m.x = d_x;
    m.lam = d_lam;
    m.c = d_c;
    m.G = d_G;
    m.H = d_H;
    m.ph = reinterpret_cast<const PcPhaseArgs*>(h->d_phase_args.p);
    m.flags = flags;
    m.n_phases = (int32_t)Q.ph.size();
    int nb = 0;
    for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
      m.first_block[ip] = nb;
      nb += std::max(0, h->pd[ip]->tile_end - h->pd[ip]->tile_begin);
    }
    for (size_t ip = Q.ph.size(); ip <= PC_MAX_PHASES; ++ip) m.first_block[ip] = nb;
    if (nb > 0) {
      size_t sz = sizeof(m);
      void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &m, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
      HIP_OK(hipModuleLaunchKernel(h->bulk_all_fn, nb, 1, 1, h->TB * h->wpt_all, 1, 1, h->lds_all, st, nullptr, cfg));
    }
    bulk = false;
  }
  if (h->host_args_dirty) {
    h->host_bulk_args.resize(Q.ph.size());
    for (size_t ip = 0; ip < Q.ph.size(); ++ip)
      fill_phase_args(h, ip, h->host_bulk_args[ip].a, nullptr, nullptr, nullptr, nullptr, nullptr, 0, h->pd[ip]->wpt);
    fill_tail_args(h, h->host_tail_launch.t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1.0);
    h->host_args_dirty = false;
  }
  auto patch_tail = [&](PcTailArgs& t) {
    t.x = d_x; t.lam = d_lam; t.c = d_c; t.G = d_G; t.H = d_H;
    t.fobj = d_fobj; t.grad = d_grad; t.flags = flags; t.sigma = sigma;
    t.block_threads = PC_TAIL_THREADS;
  };
  for (size_t ip = 0; bulk && ip < Q.ph.size(); ++ip) {
    auto& D = *h->pd[ip];
    if (D.tile_end <= D.tile_begin) continue;
    PcPhaseArgs& a = h->host_bulk_args[ip].a;
    const int wpt = (fuse && ip == last) ? 1 : D.wpt;
    a.x = d_x; a.lam = d_lam; a.c = d_c; a.G = d_G; a.H = d_H;
    a.flags = flags;
    a.wpt = wpt;
    a.block_threads = h->TB * wpt;
    if (fuse && ip == last) {
      Both both;
      both.a = a;
      both.t = h->host_tail_launch.t;
      patch_tail(both.t);
      both.t.block_threads = h->TB;   // the tail runs inside the bulk workgroup
      size_t sz = sizeof(both);
      void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &both, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
      HIP_OK(hipModuleLaunchKernel(D.fn_fused, D.n_tiles, 1, 1, h->TB, 1, 1, D.lds_bytes, st, nullptr, cfg));
    } else {
      // (lead scalars..., PcPhaseArgs): the lead is what the command processor preloads into SGPRs (pc_args.h)
      PcBulkArgs& ba = h->host_bulk_args[ip];
      const int un = a.uni_n > 0 ? a.uni_n : 0;
      ba.lead = PcLead{a.x + a.x_off, a.lam ? a.lam + a.c_off : nullptr, a.qa, a.sec_h, a.N, a.K, a.tile_begin, a.n_blocks,
                       a.flags | (a.wpt << 8) | ((a.block_threads >> 6) << 12) | (a.spt << 16),
                       un ? (a.qa_off[un] | ((a.qa_total + a.qw_off[un]) << 16)) : 0};
      size_t sz = sizeof(ba);
      void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ba, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
      HIP_OK(hipModuleLaunchKernel(D.fn, D.tile_end - D.tile_begin, 1, 1, h->TB * D.wpt, 1, 1, D.lds_bytes, st, nullptr, cfg));
    }
  }
  if (!tail || fuse) return;
  PcTailLaunch& tl = h->host_tail_launch;
  PcTailArgs& t = tl.t;
  patch_tail(t);
  tl.lead = PcTailLead{t.x, t.ph[0].partials, t.ph[0].scal, t.ph[0].x_off, t.ph[0].n_tiles, t.ph[0].N, t.flags, t.block_threads};
  size_t sz = sizeof(tl);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &tl, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  // more than four strides of partial sums per lane in some phase: the build that overlaps their loads
  int max_tiles = 0;
  for (size_t ip = 0; ip < Q.ph.size(); ++ip) max_tiles = std::max(max_tiles, (int)t.ph[ip].n_tiles);
  hipFunction_t fn = (h->tail_big_fn && max_tiles > 4 * PC_TAIL_THREADS) ? h->tail_big_fn : h->tail_fn;
  HIP_OK(hipModuleLaunchKernel(fn, 1, 1, 1, PC_TAIL_THREADS, 1, 1, 0, st, nullptr, cfg));
}

void upload_scaling(pc_handle* h) {
  auto& Q = h->Q;
  for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
    auto& P = Q.ph[ip];
    auto& D = *h->pd[ip];
    const int NZ = P.n_z, NQ = P.n_q, NS = Q.n_s, NY = P.n_y, NP = P.n_p;
    std::vector<double> s(2 * NZ + 2 * NQ + 4 + 2 * NS + NY + NP + NQ, 0.0);
    const double* V = h->V_ocp.data() + P.ocp_x_off;
    const double* r = h->r_ocp.data() + P.ocp_x_off;
    int o = 0;
    for (int b = 0; b < NZ; ++b) s[o++] = V[b];
    for (int b = 0; b < NZ; ++b) s[o++] = r[b];
    for (int m = 0; m < NQ; ++m) s[o++] = V[NZ + m];
    for (int m = 0; m < NQ; ++m) s[o++] = r[NZ + m];
    for (int j = 0; j < 2; ++j) s[o++] = j < P.n_t ? V[NZ + NQ + j] : 1.0;
    for (int j = 0; j < 2; ++j) s[o++] = j < P.n_t ? r[NZ + NQ + j] : 0.0;
    for (int l = 0; l < NS; ++l) s[o++] = h->V_ocp[Q.ocp_s_off + l];
    for (int l = 0; l < NS; ++l) s[o++] = h->r_ocp[Q.ocp_s_off + l];
    const double* W = h->W_ocp.data() + P.ocp_c_off;
    for (int a = 0; a < NY + NP + NQ; ++a) s[o++] = W[a];
    D.scal_host = s;
    if (h->device >= 0) {
      if (D.scal.n != s.size()) D.scal.alloc(s.size());
      HIP_OK(hipMemcpy(D.scal.p, s.data(), s.size() * sizeof(double), hipMemcpyHostToDevice));
      // scal | goff | hoff as one table (pc::bulk stages it into LDS for models with many variables)
      std::vector<double> tab(s);
      auto append = [&](const std::vector<int64_t>& v) {
        for (int64_t e : v) {
          double d;
          std::memcpy(&d, &e, sizeof(d));
          tab.push_back(d);
        }
      };
      append(P.goff);
      append(P.hoff);
      if (D.tab.n != tab.size()) D.tab.alloc(tab.size());
      HIP_OK(hipMemcpy(D.tab.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  if (h->device >= 0) {
    const size_t np = Q.point_x.size();
    std::vector<double> pv(np), pr(np), we(Q.n_b);
    for (size_t i = 0; i < np; ++i) {
      pv[i] = h->V_ocp[Q.point_ocp[i]];
      pr[i] = h->r_ocp[Q.point_ocp[i]];
    }
    for (int r = 0; r < Q.n_b; ++r) we[r] = h->W_ocp[Q.ocp_c_end_off + r];
    h->d_pointV.upload(pv);
    h->d_pointr.upload(pr);
    h->d_Wend.upload(we);
    h->h_pointV = pv;
    h->h_pointr = pr;
    h->h_Wend = we;
  }
  h->have_cG = false;
}

void require_device(pc_handle* h) {
  if (!h) throw std::runtime_error("null handle");
  if (h->device < 0)
    throw std::runtime_error("handle was created with device = -1 (structure only): no evaluation is possible "
                             "without a GPU; this library has no CPU fallback");
  if (!h->scaling_set) throw std::runtime_error("pc_set_scaling must be called before evaluating");
  HIP_OK(hipSetDevice(h->device));
}

void copy_x_in(pc_handle* h, const double* x) {
  std::memcpy(h->h_x.p, x, h->Q.num_x * sizeof(double));
  HIP_OK(hipMemcpyAsync(h->d_x.p, h->h_x.p, h->Q.num_x * sizeof(double), hipMemcpyHostToDevice, h->stream));
}

template <class F>
int guarded(F&& f) {
  try {
    f();
    return 1;
  } catch (const std::exception& e) {
    set_err(e.what());
    return 0;
  }
}

}  // namespace

extern "C" {

const char* pc_last_error(void) { return g_err.c_str(); }

int pc_create(const pc_problem_desc* d, pc_handle** out) {
  if (out) *out = nullptr;
  std::unique_ptr<pc_handle> h;
  const int ok = guarded([&] {
    if (!d || !out) throw std::runtime_error("null descriptor or output pointer");
    if (d->n_phases < 1 || d->n_phases > PC_MAX_PHASES) throw std::runtime_error("n_phases must be in [1, 8]");
    h.reset(new pc_handle());
    auto& Q = h->Q;
    Q.n_s = d->n_s;
    Q.n_b = d->n_b;
    Q.ph.resize(d->n_phases);
    for (int ip = 0; ip < d->n_phases; ++ip) {
      const pc_phase_desc& s = d->phases[ip];
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
      P.bulk_kernel = s.bulk_kernel ? s.bulk_kernel : "";
      P.eval_ops = s.eval_ops;
      for (int k = 0; k < s.K; ++k)
        if (s.compiled_order > 0 && s.n_k[k] != s.compiled_order)
          throw std::runtime_error("phase kernel was compiled for a fixed section order that the mesh does not have");
    }
    Q.point_phase.assign(d->point_phase, d->point_phase + d->n_point);
    Q.point_kind.assign(d->point_kind, d->point_kind + d->n_point);
    Q.point_idx.assign(d->point_idx, d->point_idx + d->n_point);
    Q.jgrad_col.assign(d->jgrad_col, d->jgrad_col + d->n_jgrad);
    Q.bjac_row.assign(d->bjac_row, d->bjac_row + d->n_bjac);
    Q.bjac_col.assign(d->bjac_col, d->bjac_col + d->n_bjac);
    Q.pthess_row.assign(d->pthess_row, d->pthess_row + d->n_pthess);
    Q.pthess_col.assign(d->pthess_col, d->pthess_col + d->n_pthess);
    // quadrature tables
    for (int i = 0; i <= PC_MAX_ORDER; ++i) h->qa_off[i] = h->qw_off[i] = -1;
    {
      size_t oa = 0, ow = 0;
      for (int i = 0; i < d->n_orders; ++i) {
        const int n = d->orders[i];
        if (n < 2 || n > PC_MAX_ORDER) throw std::runtime_error("quadrature order outside [2, 20]");
        h->qa_off[n] = (int32_t)oa;
        h->qw_off[n] = (int32_t)ow;
        oa += (size_t)(n - 1) * n;
        ow += n;
      }
      h->qa.assign(d->quad_A, d->quad_A + oa);
      h->qw.assign(d->quad_w, d->quad_w + ow);
    }
    // threads per block
    int64_t Nmax = 0;
    for (int ip = 0; ip < d->n_phases; ++ip) {
      int64_t N = 1;
      for (int k = 0; k < d->phases[ip].K; ++k) N += d->phases[ip].n_k[k] - 1;
      Nmax = std::max(Nmax, N);
    }
    int TB = d->threads_per_block;
    if (const char* env = std::getenv("PYCOLLO_AMD_TB")) TB = std::atoi(env);
    if (d->device >= 0) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, d->device) == hipSuccess && v > 0)
        h->lds_limit = v;
    }
    const bool auto_tb = (TB == 0);
    if (auto_tb) TB = Nmax >= 262144 ? 256 : (Nmax >= 65536 ? 128 : 64);
    if (TB != 64 && TB != 128 && TB != 256) throw std::runtime_error("threads_per_block must be 64, 128 or 256");
    for (auto& P : Q.ph) pcp::finalize_phase_tables(P, Q.n_s);
    if (auto_tb) {  // largest tile whose staging fits the 64 KiB of dynamic LDS a module kernel may request
      auto fits = [&](int tb) {
        for (auto& P : Q.ph)
          if (phase_lds_bytes(P, Q.n_s, tb, (int)h->qa.size(), (int)h->qw.size()) > h->lds_limit) return false;
        return true;
      };
      while (TB > 64 && !fits(TB)) TB /= 2;
    }
    h->TB = TB;
    // Tile capacity in nodes (<= TB): a tile smaller than its workgroup leaves lanes idle but shortens every
    // wave's store phase and puts more waves on the chip -- what a problem with far fewer tiles than SIMDs wants.
    int TC = TB, max_nk = 2;
    for (auto& P : Q.ph)
      for (int k = 0; k < P.K; ++k) max_nk = std::max(max_nk, P.n_k[k]);
    if (const char* env = std::getenv("PYCOLLO_AMD_TILE_NODES")) {
      const int v = std::atoi(env);
      if (v >= max_nk && v <= TB) TC = v;
    }
    h->TC = TC;
    if (const char* env = std::getenv("PYCOLLO_AMD_DBG_STAGE")) h->dbg_stage = std::atoi(env);
    pcp::build_all(Q, TC);
    if (Q.point_x.size() > PC_MAX_POINT || Q.n_b > PC_MAX_ENDPOINT_ROWS)
      throw std::runtime_error("too many endpoint variables / endpoint constraints for the tail kernel's argument block");
    if (Q.tail_owned.size() > PC_TAIL_OWNED_MAX)
      throw std::runtime_error("too many Hessian entries owned by the tail kernel (static parameters / endpoint terms)");
    for (auto& P : Q.ph)
      for (int k = 0; k < P.K; ++k)
        if (h->qa_off[P.n_k[k]] < 0) throw std::runtime_error("no quadrature table for a section order in use");
    h->device = d->device;
    h->V_ocp.assign(Q.num_ocp_x, 1.0);
    h->r_ocp.assign(Q.num_ocp_x, 0.0);
    h->W_ocp.assign(Q.num_ocp_c, 1.0);
    bool fuse_env = false;
    if (const char* env = std::getenv("PYCOLLO_AMD_FUSE")) fuse_env = std::atoi(env) != 0;
    h->pd.resize(Q.ph.size());
    for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
      h->pd[ip].reset(new PhaseDev());
      auto& P = Q.ph[ip];
      auto& D = *h->pd[ip];
      D.n_tiles = (int)P.tile_k0.size() - 1;
      D.tile_begin = 0;
      D.tile_end = D.n_tiles;
      int nfs = 0;
      for (int a = 0; a < P.n_y; ++a)
        for (int l = 0; l < Q.n_s; ++l) nfs += P.dep(a, P.n_z + l) ? 1 : 0;
      D.nfs = nfs;
      // uniform section order: index arithmetic replaces the section tables
      bool same = true;
      for (int k = 0; k < P.K; ++k) same = same && P.n_k[k] == P.n_k[0];
      D.uni_n = same ? P.n_k[0] : 0;
      D.spt = same ? (TC - 1) / (P.n_k[0] - 1) : 0;
      if (same && P.tile_k0.size() > 1 && P.tile_k0[1] != std::min(D.spt, P.K))
        throw std::runtime_error("internal error: uniform tiling mismatch");
      D.lds_out = phase_lds_out(P, Q.n_s, TB);
      // Few tiles and several states: W waves share a tile and split its output runs, so that the chip's 1024
      // SIMDs each hold a wave (or two) instead of a fraction of them holding one long-running wave.  Beyond that
      // the replicas only add redundant node evaluations (measured on 64-node tiles, W = 1 / 2 / 4: shuttle
      // 381 tiles 22.7 / 19.2 / 17.6 us, 953 tiles 25.2 / 23.3 / 31.3 us, 2858 tiles 54 / 60 / 81 us).
      D.wpt = 1;
      if (TB == 64 && !fuse_env) {
        // (one state: W = 2 measured no faster, 5.58 vs 5.49 us.  A heavy model -- Delta III, ~10k operations --
        //  loses at 834 tiles, 358 / 410 / 522 us: every sharing wave re-evaluates the node functions)
        const bool heavy = P.eval_ops > 4000;
        if (P.n_y >= 2 && D.n_tiles <= (heavy ? 512 : 1024)) D.wpt = 2;
        if (P.n_y >= 3 && D.n_tiles <= 400) D.wpt = 4;
        if (const char* env = std::getenv("PYCOLLO_AMD_WPT")) {
          const int v = std::atoi(env);
          if (v == 1 || v == 2 || v == 4) D.wpt = v;
        }
        while (D.wpt > 1 && 8 * lds_doubles(TB, (int)h->qa.size(), (int)h->qw.size(), P.n_y, nfs, P.nred,
                                            D.lds_out * D.wpt) > h->lds_limit)
          D.wpt /= 2;
      }
      D.lds_bytes = 8 * lds_doubles(TB, (int)h->qa.size(), (int)h->qw.size(), P.n_y, nfs, P.nred, D.lds_out * D.wpt);
      h->lds_max = std::max(h->lds_max, D.lds_bytes);
      if (D.lds_bytes > h->lds_limit)
        throw std::runtime_error("tile needs more dynamic LDS than a workgroup may request; use a smaller "
                                 "threads_per_block");
      if ((int)P.goff.size() > PC_MAX_GOFF || (int)P.hoff.size() > PC_MAX_HOFF)
        throw std::runtime_error("too many variables/constraints per phase for the kernel argument block");
    }
    h->n_launches = (int)Q.ph.size() + 1;   // refined after the module is loaded (fused tail: one fewer)
    // launch shape of the all-phases kernel: the phases share one workgroup size, chosen from the total tile count
    if (Q.ph.size() > 1) {
      int total = 0, min_ny = 1 << 30;
      bool heavy = false;
      for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
        total += h->pd[ip]->n_tiles;
        min_ny = std::min(min_ny, Q.ph[ip].n_y);
        heavy = heavy || Q.ph[ip].eval_ops > 4000;
      }
      h->wpt_all = 1;
      auto lds_for = [&](int w) {
        int mx = 0;
        for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
          auto& P = Q.ph[ip];
          auto& D = *h->pd[ip];
          mx = std::max(mx, 8 * lds_doubles(TB, (int)h->qa.size(), (int)h->qw.size(), P.n_y, D.nfs, P.nred, D.lds_out * w));
        }
        return mx;
      };
      if (TB == 64 && !fuse_env) {
        if (min_ny >= 2 && total <= (heavy ? 512 : 1024)) h->wpt_all = 2;
        if (min_ny >= 3 && total <= 400) h->wpt_all = 4;
        if (const char* env = std::getenv("PYCOLLO_AMD_WPT")) {
          const int v = std::atoi(env);
          if (v == 1 || v == 2 || v == 4) h->wpt_all = v;
        }
        while (h->wpt_all > 1 && lds_for(h->wpt_all) > h->lds_limit) h->wpt_all /= 2;
      }
      h->lds_all = lds_for(h->wpt_all);
    }
    if (h->device < 0) return;  // structure-only handle

    // ---- device side ----------------------------------------------------------------------------
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= h->device)
      throw std::runtime_error("requested HIP device is not available (no GPU visible?)");
    HIP_OK(hipSetDevice(h->device));
    if (!d->code_object || !d->tail_kernel) throw std::runtime_error("code_object and tail_kernel are required");
    HIP_OK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_OK(hipModuleLoad(&h->module, d->code_object));
    HIP_OK(hipModuleGetFunction(&h->tail_fn, h->module, d->tail_kernel));
    if (hipModuleGetFunction(&h->tail_big_fn, h->module, (std::string(d->tail_kernel) + "_big").c_str()) != hipSuccess)
      h->tail_big_fn = nullptr;
    if (Q.ph.size() > 1) {
      bool merge = true;
      if (const char* env = std::getenv("PYCOLLO_AMD_MERGE")) merge = std::atoi(env) != 0;
      if (!merge || hipModuleGetFunction(&h->bulk_all_fn, h->module, "pc_bulk_all") != hipSuccess) h->bulk_all_fn = nullptr;
      if (h->bulk_all_fn) {
        h->d_phase_args.alloc(Q.ph.size() * sizeof(PcPhaseArgs));
        h->n_launches = 2;
      }
    }
    {   // one buffer: the weight tables right behind the A tables (the kernels address both from `qa`)
      std::vector<double> both(h->qa);
      both.insert(both.end(), h->qw.begin(), h->qw.end());
      h->d_qa.upload(both);
    }
    for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
      auto& P = Q.ph[ip];
      auto& D = *h->pd[ip];
      HIP_OK(hipModuleGetFunction(&D.fn, h->module, P.bulk_kernel.c_str()));
      if (ip + 1 == Q.ph.size()) {
        if (hipModuleGetFunction(&D.fn_fused, h->module, (P.bulk_kernel + "_f").c_str()) != hipSuccess)
          D.fn_fused = nullptr;   // code object without the fused variant: two launches per evaluation
      }
      D.tile_k0.upload(P.tile_k0);
      {
        std::vector<int32_t> tn(P.tile_k0.size());
        for (size_t i = 0; i < tn.size(); ++i) tn[i] = P.sec_s[P.tile_k0[i]];
        D.tile_n0.upload(tn);
      }
      D.sec_s.upload(P.sec_s);
      D.sec_h.upload(P.h_k);
      D.sec_E.upload(P.sec_E);
      D.hslot0.upload(P.hslot0);
      D.hslotN.upload(P.hslotN);
      D.hsum_slot.upload(P.hsum_slot);
      D.hsum_local.upload(P.hsum_local);
      D.partials.alloc((size_t)std::max(1, P.nred) * D.n_tiles);
      if (h->dbg_stage == 9) D.dbg.alloc((size_t)64 * D.n_tiles);   // 4 waves x 16 stamps per tile
    }
    h->d_sync.upload(std::vector<unsigned>((PC_SYNC_SHARDS + 1) * 16, 0u));
    if (const char* env = std::getenv("PYCOLLO_AMD_FUSE")) h->allow_fuse = std::atoi(env) != 0;
    h->d_point_x.upload(Q.point_x);
    h->d_tail_owned.upload(Q.tail_owned);
    h->d_pt_hslot.upload(Q.pt_hslot);
    h->d_pt_hlocal.upload(Q.pt_hlocal);
    h->d_g_indptr.upload(Q.g_indptr);
    const size_t nG = Q.g_row.size(), nH = Q.h_row.size();
    h->d_x.alloc(Q.num_x); h->d_lam.alloc(Q.num_c); h->d_c.alloc(Q.num_c);
    h->d_G.alloc(nG); h->d_H.alloc(nH); h->d_fobj.alloc(1);
    h->d_grad.alloc(Q.num_x); h->d_norms.alloc(Q.num_c);
    h->h_x.alloc(Q.num_x); h->h_lam.alloc(Q.num_c); h->h_c.alloc(Q.num_c);
    h->h_G.alloc(nG); h->h_H.alloc(nH); h->h_fobj.alloc(1);
    h->h_grad.alloc(Q.num_x); h->h_norms.alloc(Q.num_c);
    HIP_OK(hipMemset(h->d_lam.p, 0, Q.num_c * sizeof(double)));
    if (!h->bulk_all_fn && h->allow_fuse && h->pd.back()->fn_fused && h->pd.back()->n_tiles <= 8192)
      h->n_launches = (int)Q.ph.size();
  });
  if (!ok) return 0;
  *out = h.release();
  return 1;
}

void pc_destroy(pc_handle* h) {
  if (!h) return;
  if (h->device >= 0) {
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
  }
  h->pd.clear();
  if (h->module) (void)hipModuleUnload(h->module);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int pc_get_info(const pc_handle* h, pc_info* info) {
  return guarded([&] {
    if (!h || !info) throw std::runtime_error("null argument");
    const auto& Q = h->Q;
    info->n = (int32_t)Q.num_x;
    info->m = (int32_t)Q.num_c;
    info->nnz_jac = (int64_t)Q.g_row.size();
    info->nnz_hess = (int64_t)Q.h_row.size();
    info->algorithmic_bytes = 8 * (Q.num_x + Q.num_c) + 8 * (Q.num_c + info->nnz_jac + info->nnz_hess);
    int nt = 0;
    for (auto& D : h->pd) nt += D->n_tiles;
    info->n_tiles_total = nt;
    info->threads_per_block = h->TB;
    info->lds_bytes_max = h->lds_max;
    info->n_launches = h->n_launches;
    info->waves_per_tile = 1;
    for (auto& D : h->pd) info->waves_per_tile = std::max(info->waves_per_tile, (int32_t)D->wpt);
    if (h->bulk_all_fn) info->waves_per_tile = h->wpt_all;
    info->reserved = 0;
  });
}

int pc_sizes(const pc_handle* h, int32_t* n, int32_t* m, int64_t* nnz_jac, int64_t* nnz_hess) {
  return guarded([&] {
    if (!h) throw std::runtime_error("null handle");
    if (n) *n = (int32_t)h->Q.num_x;
    if (m) *m = (int32_t)h->Q.num_c;
    if (nnz_jac) *nnz_jac = (int64_t)h->Q.g_row.size();
    if (nnz_hess) *nnz_hess = (int64_t)h->Q.h_row.size();
  });
}

int pc_jac_structure(const pc_handle* h, int32_t* iRow, int32_t* jCol) {
  return guarded([&] {
    if (!h || !iRow || !jCol) throw std::runtime_error("null argument");
    std::memcpy(iRow, h->Q.g_row.data(), h->Q.g_row.size() * sizeof(int32_t));
    std::memcpy(jCol, h->Q.g_col.data(), h->Q.g_col.size() * sizeof(int32_t));
  });
}

int pc_hess_structure(const pc_handle* h, int32_t* iRow, int32_t* jCol) {
  return guarded([&] {
    if (!h || !iRow || !jCol) throw std::runtime_error("null argument");
    std::memcpy(iRow, h->Q.h_row.data(), h->Q.h_row.size() * sizeof(int32_t));
    std::memcpy(jCol, h->Q.h_col.data(), h->Q.h_col.size() * sizeof(int32_t));
  });
}

int pc_set_scaling(pc_handle* h, const double* V, const double* r, const double* W, double w_J) {
  return guarded([&] {
    if (!h || !V || !r || !W) throw std::runtime_error("null argument");
    if (h->device >= 0) {
      HIP_OK(hipSetDevice(h->device));
      HIP_OK(hipStreamSynchronize(h->stream));
    }
    h->V_ocp.assign(V, V + h->Q.num_ocp_x);
    h->r_ocp.assign(r, r + h->Q.num_ocp_x);
    h->W_ocp.assign(W, W + h->Q.num_ocp_c);
    h->w_J = w_J;
    upload_scaling(h);
    h->scaling_set = true;
    h->args_dirty = true;
    h->host_args_dirty = true;
  });
}

int pc_eval_all_device(pc_handle* h, const double* d_x, double obj_factor, const double* d_lambda, double* d_g,
                       double* d_jac, double* d_hess, void* stream) {
  return guarded([&] {
    require_device(h);
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    launch_all(h, d_x, d_lambda, d_g, d_jac, d_hess, h->d_fobj.p, nullptr, PC_FLAG_C | PC_FLAG_G | PC_FLAG_H, st,
               obj_factor);
  });
}

int pc_launch_bulk_device(pc_handle* h, const double* d_x, const double* d_lambda, double* d_g, double* d_jac,
                          double* d_hess, void* stream) {
  return guarded([&] {
    require_device(h);
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    launch_all(h, d_x, d_lambda, d_g, d_jac, d_hess, h->d_fobj.p, nullptr, PC_FLAG_C | PC_FLAG_G | PC_FLAG_H, st, 1.0,
               true, false);
  });
}

int pc_launch_tail_device(pc_handle* h, const double* d_x, double obj_factor, const double* d_lambda, double* d_g,
                          double* d_jac, double* d_hess, void* stream) {
  return guarded([&] {
    require_device(h);
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    launch_all(h, d_x, d_lambda, d_g, d_jac, d_hess, h->d_fobj.p, nullptr, PC_FLAG_C | PC_FLAG_G | PC_FLAG_H, st,
               obj_factor, false, true);
  });
}

int pc_set_tile_range(pc_handle* h, int phase, int tile_begin, int tile_end) {
  return guarded([&] {
    if (!h || phase < 0 || phase >= (int)h->pd.size()) throw std::runtime_error("phase out of range");
    auto& D = *h->pd[phase];
    if (tile_begin < 0 || tile_end < tile_begin || tile_end > D.n_tiles) throw std::runtime_error("tile range out of range");
    D.tile_begin = tile_begin;
    D.tile_end = tile_end;
    h->have_cG = false;
    h->args_dirty = true;
    h->host_args_dirty = true;
  });
}

int pc_phase_tiles(const pc_handle* h, int phase, int32_t* n_tiles, int32_t* nred, int32_t* tile_k0) {
  return guarded([&] {
    if (!h || phase < 0 || phase >= (int)h->pd.size()) throw std::runtime_error("phase out of range");
    const auto& P = h->Q.ph[phase];
    if (n_tiles) *n_tiles = (int32_t)P.tile_k0.size() - 1;
    if (nred) *nred = P.nred;
    if (tile_k0) std::memcpy(tile_k0, P.tile_k0.data(), P.tile_k0.size() * sizeof(int32_t));
  });
}

int pc_set_partials_buffer(pc_handle* h, int phase, double* d_partials) {
  return guarded([&] {
    if (!h || phase < 0 || phase >= (int)h->pd.size()) throw std::runtime_error("phase out of range");
    h->pd[phase]->partials_ext = d_partials;
    h->args_dirty = true;
    h->host_args_dirty = true;
  });
}

int pc_eval_all(pc_handle* h, const double* x, double obj_factor, const double* lambda, double* g, double* jac,
                double* hess) {
  return guarded([&] {
    require_device(h);
    auto& Q = h->Q;
    copy_x_in(h, x);
    std::memcpy(h->h_lam.p, lambda, Q.num_c * sizeof(double));
    HIP_OK(hipMemcpyAsync(h->d_lam.p, h->h_lam.p, Q.num_c * sizeof(double), hipMemcpyHostToDevice, h->stream));
    launch_all(h, h->d_x.p, h->d_lam.p, h->d_c.p, h->d_G.p, h->d_H.p, h->d_fobj.p, nullptr,
               PC_FLAG_C | PC_FLAG_G | PC_FLAG_H, h->stream, obj_factor);
    HIP_OK(hipMemcpyAsync(h->h_c.p, h->d_c.p, Q.num_c * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipMemcpyAsync(h->h_G.p, h->d_G.p, h->d_G.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipMemcpyAsync(h->h_H.p, h->d_H.p, h->d_H.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    std::memcpy(g, h->h_c.p, Q.num_c * sizeof(double));
    std::memcpy(jac, h->h_G.p, h->d_G.n * sizeof(double));
    std::memcpy(hess, h->h_H.p, h->d_H.n * sizeof(double));
    h->have_cG = true;
  });
}

// c and G are produced together on a new x and cached for the companion call (IPOPT evaluates
// g and jac_g at the same x; pycollo/nlp.py:53-57)
static void eval_cG(pc_handle* h, const double* x, int new_x) {
  require_device(h);
  if (!new_x && h->have_cG) return;
  auto& Q = h->Q;
  copy_x_in(h, x);
  launch_all(h, h->d_x.p, nullptr, h->d_c.p, h->d_G.p, nullptr, h->d_fobj.p, nullptr, PC_FLAG_C | PC_FLAG_G, h->stream,
             1.0);
  HIP_OK(hipMemcpyAsync(h->h_c.p, h->d_c.p, Q.num_c * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(h->h_G.p, h->d_G.p, h->d_G.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  h->have_cG = true;
}

int pc_eval_g(pc_handle* h, const double* x, int new_x, double* g) {
  return guarded([&] {
    eval_cG(h, x, new_x);
    std::memcpy(g, h->h_c.p, h->Q.num_c * sizeof(double));
  });
}

int pc_eval_jac_g(pc_handle* h, const double* x, int new_x, double* values) {
  return guarded([&] {
    eval_cG(h, x, new_x);
    std::memcpy(values, h->h_G.p, h->d_G.n * sizeof(double));
  });
}

int pc_eval_h(pc_handle* h, const double* x, int new_x, double obj_factor, const double* lambda, int new_lambda,
              double* values) {
  (void)new_lambda;
  return guarded([&] {
    require_device(h);
    auto& Q = h->Q;
    if (new_x) h->have_cG = false;
    copy_x_in(h, x);
    std::memcpy(h->h_lam.p, lambda, Q.num_c * sizeof(double));
    HIP_OK(hipMemcpyAsync(h->d_lam.p, h->h_lam.p, Q.num_c * sizeof(double), hipMemcpyHostToDevice, h->stream));
    launch_all(h, h->d_x.p, h->d_lam.p, nullptr, nullptr, h->d_H.p, h->d_fobj.p, nullptr, PC_FLAG_H, h->stream,
               obj_factor);
    HIP_OK(hipMemcpyAsync(h->h_H.p, h->d_H.p, h->d_H.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    std::memcpy(values, h->h_H.p, h->d_H.n * sizeof(double));
  });
}

static void eval_obj(pc_handle* h, const double* x, bool want_grad) {
  require_device(h);
  auto& Q = h->Q;
  copy_x_in(h, x);
  if (want_grad) HIP_OK(hipMemsetAsync(h->d_grad.p, 0, Q.num_x * sizeof(double), h->stream));
  // flags = 0: the bulk kernels are skipped entirely, only the endpoint block runs
  PcTailLaunch tl;
  std::memset(&tl, 0, sizeof(tl));
  PcTailArgs& t = tl.t;
  t.x = h->d_x.p;
  t.fobj = h->d_fobj.p;
  t.grad = want_grad ? h->d_grad.p : nullptr;
  t.sigma = 1.0;
  t.wJ = h->w_J;
  t.point_x = h->d_point_x.p;
  t.point_V = h->d_pointV.p;
  t.point_r = h->d_pointr.p;
  t.W_end = h->d_Wend.p;
  fill_point_tables(h, t);
  t.c_end_off = Q.c_end_off;
  t.g_end_base = Q.g_end_base;
  t.flags = 0;
  t.block_threads = PC_TAIL_THREADS;
  for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
    t.ph[ip].n_tiles = 0;
    t.ph[ip].partials = h->pd[ip]->partials.p;
    t.ph[ip].scal = h->pd[ip]->scal.p;
    t.ph[ip].x_off = Q.ph[ip].x_off;
    t.ph[ip].N = Q.ph[ip].N;
    t.ph[ip].t_fixed[0] = Q.ph[ip].t_fixed[0];
    t.ph[ip].t_fixed[1] = Q.ph[ip].t_fixed[1];
  }
  tl.lead = PcTailLead{t.x, t.ph[0].partials, t.ph[0].scal, t.ph[0].x_off, t.ph[0].n_tiles, t.ph[0].N, t.flags, t.block_threads};
  size_t sz = sizeof(tl);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &tl, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  HIP_OK(hipModuleLaunchKernel(h->tail_fn, 1, 1, 1, PC_TAIL_THREADS, 1, 1, 0, h->stream, nullptr, cfg));
  HIP_OK(hipMemcpyAsync(h->h_fobj.p, h->d_fobj.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (want_grad)
    HIP_OK(hipMemcpyAsync(h->h_grad.p, h->d_grad.p, Q.num_x * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
}

int pc_eval_f(pc_handle* h, const double* x, int new_x, double* f) {
  (void)new_x;
  return guarded([&] {
    eval_obj(h, x, false);
    *f = h->h_fobj.p[0];
  });
}

int pc_eval_grad_f(pc_handle* h, const double* x, int new_x, double* grad) {
  (void)new_x;
  return guarded([&] {
    eval_obj(h, x, true);
    std::memcpy(grad, h->h_grad.p, h->Q.num_x * sizeof(double));
  });
}

int pc_row_norms_jac(pc_handle* h, const double* x, double* norms) {
  return guarded([&] {
    eval_cG(h, x, 1);
    const int64_t m = h->Q.num_c;
    const int64_t threads = m * 64;
    const int blocks = (int)((threads + 255) / 256);
    hipLaunchKernelGGL(row_norms_kernel, dim3(blocks), dim3(256), 0, h->stream, h->d_g_indptr.p, h->d_G.p,
                       h->d_norms.p, m);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(h->h_norms.p, h->d_norms.p, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    std::memcpy(norms, h->h_norms.p, m * sizeof(double));
  });
}

int pc_debug_stamps(pc_handle* h, int phase, long long* out, int n_tiles) {
  return guarded([&] {
    require_device(h);
    auto& D = *h->pd.at(phase);
    if (!D.dbg.p) throw std::runtime_error("no stamps: create the handle with PYCOLLO_AMD_DBG_STAGE=9");
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out, D.dbg.p, sizeof(long long) * 64 * std::min(n_tiles, D.n_tiles), hipMemcpyDeviceToHost));
  });
}

int pc_interp_linear(int device, const double* tau_prev, int n_prev, const double* vals_prev, int n_vars,
                     const double* tau_new, int n_new, double* out) {
  return guarded([&] {
    if (n_prev < 2 || n_new < 1 || n_vars < 0) throw std::runtime_error("interpolation needs at least two abscissae");
    int ndev = 0;
    if (device < 0 || hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device)
      throw std::runtime_error("requested HIP device is not available (no GPU visible?); this library has no CPU fallback");
    HIP_OK(hipSetDevice(device));
    DevBuf<double> d_tp, d_vp, d_tn, d_out;
    d_tp.upload(std::vector<double>(tau_prev, tau_prev + n_prev));
    d_vp.upload(std::vector<double>(vals_prev, vals_prev + (size_t)n_vars * n_prev));
    d_tn.upload(std::vector<double>(tau_new, tau_new + n_new));
    d_out.alloc((size_t)std::max(1, n_vars) * n_new);
    hipLaunchKernelGGL(interp_linear_kernel, dim3((n_new + 255) / 256), dim3(256), 0, 0, d_tp.p, n_prev, d_vp.p, n_vars,
                       d_tn.p, n_new, d_out.p);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpy(out, d_out.p, sizeof(double) * (size_t)n_vars * n_new, hipMemcpyDeviceToHost));
  });
}

int pc_copy_runs(const double* d_src, double* d_dst, const int64_t* d_chunks, int64_t n_chunks, void* stream) {
  return guarded([&] {
    if (n_chunks <= 0) return;
    if (!d_src || !d_dst || !d_chunks) throw std::runtime_error("null argument");
    hipLaunchKernelGGL(copy_runs_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, d_src, d_dst, d_chunks);
    HIP_OK(hipGetLastError());
  });
}

int pc_run_chunk(void) { return PC_RUN_CHUNK; }

int pc_mesh_error(pc_handle* h, int phase, const double* x, int n_orders, const int32_t* orders, const double* tabB,
                  const double* tabE, const double* tabA, double* max_rel, double* max_abs) {
  return guarded([&] {
    require_device(h);
    if (phase < 0 || phase >= (int)h->pd.size()) throw std::runtime_error("phase out of range");
    auto& Q = h->Q;
    auto& P = Q.ph[phase];
    auto& D = *h->pd[phase];
    hipFunction_t fn = nullptr;
    const std::string name = "pc_mesh_err_p" + std::to_string(phase);
    HIP_OK(hipModuleGetFunction(&fn, h->module, name.c_str()));
    PcRefineArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int i = 0; i <= PC_MAX_ORDER; ++i) a.offBE[i] = a.offA[i] = -1;
    size_t oBE = 0, oA = 0;
    for (int i = 0; i < n_orders; ++i) {
      const int n = orders[i];
      if (n < 2 || n >= PC_MAX_ORDER) throw std::runtime_error("mesh-error tables: order outside [2, 19]");
      a.offBE[n] = (int32_t)oBE;
      a.offA[n] = (int32_t)oA;
      oBE += (size_t)(n - 1) * n;
      oA += (size_t)n * (n + 1);
    }
    for (int k = 0; k < P.K; ++k)
      if (a.offBE[P.n_k[k]] < 0) throw std::runtime_error("mesh-error tables: an order in use has no table");
    // tiles: every section occupies n_k + 1 lanes
    const int TB = 256;
    std::vector<int32_t> tile_k0{0}, lane0(P.K);
    int lanes = 0;
    for (int k = 0; k < P.K; ++k) {
      const int need = P.n_k[k] + 1;
      if (lanes + need > TB) {
        tile_k0.push_back(k);
        lanes = 0;
      }
      lane0[k] = lanes;
      lanes += need;
    }
    tile_k0.push_back(P.K);
    DevBuf<int32_t> d_tile, d_lane;
    DevBuf<double> d_B, d_E, d_A, d_rel, d_abs;
    d_tile.upload(tile_k0);
    d_lane.upload(lane0);
    d_B.upload(std::vector<double>(tabB, tabB + oBE));
    d_E.upload(std::vector<double>(tabE, tabE + oBE));
    d_A.upload(std::vector<double>(tabA, tabA + oA));
    d_rel.alloc(P.K);
    d_abs.alloc((size_t)P.K * std::max(1, P.n_y));
    copy_x_in(h, x);
    a.x = h->d_x.p;
    a.tile_k0 = d_tile.p;
    a.lane0 = d_lane.p;
    a.sec_s = D.sec_s.p;
    a.sec_h = D.sec_h.p;
    a.tabB = d_B.p;
    a.tabE = d_E.p;
    a.tabA = d_A.p;
    a.max_rel = d_rel.p;
    a.max_abs = d_abs.p;
    a.x_off = P.x_off;
    a.s_off = Q.s_off;
    a.t_fixed[0] = P.t_fixed[0];
    a.t_fixed[1] = P.t_fixed[1];
    a.N = P.N;
    a.K = P.K;
    a.tab_total_BE = (int32_t)oBE;
    a.tab_total_A = (int32_t)oA;
    for (size_t i = 0; i < D.scal_host.size(); ++i) a.scal[i] = D.scal_host[i];
    const int NU = P.n_u, NY = P.n_y;
    const size_t lds = 8 * (2 * oBE + oA + (size_t)TB * (5 * NY + std::max(1, NU) + 1)) + 4 * (size_t)TB;
    if ((int)lds > h->lds_limit) throw std::runtime_error("mesh-error kernel: tables do not fit in LDS");
    size_t sz = sizeof(a);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    HIP_OK(hipModuleLaunchKernel(fn, (int)tile_k0.size() - 1, 1, 1, TB, 1, 1, (unsigned)lds, h->stream, nullptr, cfg));
    HIP_OK(hipStreamSynchronize(h->stream));
    HIP_OK(hipMemcpy(max_rel, d_rel.p, sizeof(double) * P.K, hipMemcpyDeviceToHost));
    if (max_abs && NY > 0) HIP_OK(hipMemcpy(max_abs, d_abs.p, sizeof(double) * (size_t)P.K * NY, hipMemcpyDeviceToHost));
  });
}

int pc_synchronize(pc_handle* h) {
  return guarded([&] {
    require_device(h);
    HIP_OK(hipStreamSynchronize(h->stream));
  });
}

void* pc_stream(pc_handle* h) { return h ? (void*)h->stream : nullptr; }

}  // extern "C"
