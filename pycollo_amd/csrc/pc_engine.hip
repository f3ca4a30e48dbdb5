// C-ABI engine: owns device memory, the per-problem code object and the launch sequence.
// See include/pycollo_amd.h for the contract and the reference interfaces each entry point replaces.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pycollo_amd.h"
#include "pc_args.h"
#include "pc_pattern.hpp"
#include "pc_desc.hpp"

struct pc_kkt;   // (pc_kkt.hip; the interior-point state below drives it through the exported calls)

namespace {

thread_local std::string g_err;

void set_err(const std::string& s) { g_err = s; }

#define HIP_OK(expr)                                                                             \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_));              \
  } while (0)

using pcp::phase_nfs;
using pcp::phase_lds_bytes;
using pcp::phase_lds_out;
using pcp::phase_max_tile_rows;

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
    if (v.size() != n) alloc(v.size());   // same size: keep the allocation (pointers held in argument blocks stay valid)
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
  DevBuf<unsigned long long> gran;   // resident tail: the per-tile partial sums as granules, [n_tiles][nred][2]
  int32_t erec0 = 0;                 // resident tail: first record of this phase's edge-node Hessian entries
  int uni_n = 0, spt = 0, lds_out = 0;
  int lds_out4 = 0;                  // staging doubles per replica of a four-wave launch (one pass per state: pc::bulk NPASS)
  int wpt = 1;                       // waves (replicas) per 64-node tile, see pc::bulk
  hipFunction_t fn = nullptr;
  hipFunction_t fn_res = nullptr;    // single-phase problems: bulk kernel with the resident tail as block 0
  int lds_bytes = 0, n_tiles = 0, nfs = 0;
  std::function<int(int)> lds_for;   // LDS bytes of this phase's workgroups with w staging regions
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
  bool two_wave = false;   // heavy model in the two-wave build: W = 2, tiles sized for four workgroups per CU
  bool scaling_set = false;
  double w_J = 1.0;
  // quadrature
  std::vector<double> qa, qw;
  int32_t qa_off[PC_MAX_ORDER + 1], qw_off[PC_MAX_ORDER + 1];
  // device
  hipStream_t stream = nullptr;
  hipModule_t module = nullptr;
  std::vector<hipModule_t> more_modules;   // a heavy model's code object comes in parts (<base>.p<k>.hsaco, codegen.n_parts)
  hipFunction_t tail_fn = nullptr;
  hipFunction_t tail_big_fn = nullptr;   // same kernel with the partial-sum loads of several strides in flight
  hipFunction_t bulk_all_fn = nullptr;   // multi-phase problems: every phase's bulk kernel in one launch
  DevBuf<char> d_phase_args;             // [n_phases] PcPhaseArgs read by pc_bulk_all
  bool mixed = false;                    // some phase runs the mixed build: per-tile records pick the tile body
  std::vector<PcTileRec> trec_all;       // records of every tile, phase after phase
  std::vector<size_t> trec_base;         // [n_phases] first record of each phase in trec_all
  DevBuf<PcTileRec> d_trec;              // records of the launched tile ranges, in launch order
  std::vector<int> trec_first;           // [n_phases] first record of each phase in d_trec
  bool args_dirty = true;                // scaling / tile range / partials buffer changed since the last upload
  // host-side argument blocks, filled once per change of scaling / tile range / partials buffer; a call only
  // patches the caller's pointers, the flags and sigma into them
  std::vector<PcBulkArgs> host_bulk_args;   // per phase: lead scalars + argument block, kept filled between calls
  PcTailLaunch host_tail_launch;   // lead scalars + argument block of pc_tail, kept filled between calls
  bool host_args_dirty = true;
  int wpt_all = 1, lds_all = 0;          // launch shape of pc_bulk_all
  std::vector<std::unique_ptr<PhaseDev>> pd;
  DevBuf<double> d_qa, d_pointV, d_pointr, d_Wend, d_norms;
  // Host-pointer calls stage through ONE packed input block [x~ | lambda] and ONE packed output block
  // [J | grad J non-zeros | c~ | G~ | H~] (pinned on the host, mirrored on the device): one copy up, one copy down.
  DevBuf<double> d_in, d_out;
  PinBuf<double> h_in, h_out;
  size_t o_lam = 0, in_total = 0;                              // offsets in doubles
  size_t o_f = 0, o_gn = 0, o_c = 0, o_G = 0, o_H = 0, out_total = 0;
  // bit 0: the kernels read x~ / lambda straight from the pinned host block (no copy up);
  // bit 1: the kernels write their outputs straight into the pinned host block (no copy down)
  int host_mode = 0;
  hipEvent_t ev_small = nullptr, ev_G = nullptr;               // completion of [J | grad | c~] and of G~ at the cached x
  bool x_valid = false;      // the staged x~ is the caller's current point
  bool fc_valid = false;     // J, grad J, c~, G~ at that point have been launched (new_x == 0 reuses them)
  bool small_synced = false, G_synced = false;
  bool prefetch_jac = true;  // copy G~ down with every new point (IPOPT asks for it next); off: only when eval_jac_g asks
  bool G_copied = false;
  DevBuf<int64_t> d_point_x, d_tail_owned, d_pt_hslot, d_g_indptr;
  DevBuf<int32_t> d_pt_hlocal;
  std::vector<double> h_pointV, h_pointr, h_Wend;   // host copies: travel by value in PcTailArgs
  // resident tail (pc_kernels.hpp, RES): one launch per evaluation, the tail runs as block 0 beside the tiles
  hipFunction_t bulk_all_res_fn = nullptr;   // multi-phase problems: pc_bulk_all with the resident tail
  bool resident = true;                      // PYCOLLO_AMD_RESIDENT=0: always two launches (bulk, then pc_tail)
  DevBuf<unsigned long long> d_erec;         // granules of the edge-node Hessian entries endpoint terms are added to
  DevBuf<int64_t> d_rec_slot;                // [n_rec] H slot of every record (-1: site absent)
  DevBuf<int32_t> d_rec_term;                // [n_rec] endpoint Hessian entry added to the record, or -1
  int32_t n_rec = 0;
  PinBuf<unsigned> h_timeout;                // host-visible: set by a tail whose granules never arrived
  uint32_t epoch = 0;                        // tag of the last resident launch's granules (never 0)
  int spin_us = 500;                         // host-pointer calls poll the stream this long before blocking (PYCOLLO_AMD_SPIN_US)
  int tail_lds_bytes = 0, lds_nred = 0;
  bool heavy_point = false;                  // the endpoint block is worth one tail workgroup per part when tiles are single waves
  int tail_blocks_env = 0;
  DevBuf<unsigned long long> d_hb_gran;      // endpoint Hessian terms handed between the tail's workgroups
  PinBuf<double> h_norms;
  std::vector<double> V_ocp, r_ocp, W_ocp;
  int n_launches = 0;
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
                    double* d_H, double* d_fobj, double* d_gradnz, int flags, double sigma) {
  auto& Q = h->Q;
  std::memset(&t, 0, sizeof(t));
  t.x = d_x;
  t.lam = d_lam;
  t.c = d_c;
  t.G = d_G;
  t.H = d_H;
  t.fobj = d_fobj;
  t.grad_nz = d_gradnz;
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
  t.lds_nred = h->lds_nred;
  t.erec = h->d_erec.p;
  t.rec_slot = h->d_rec_slot.p;
  t.rec_term = h->d_rec_term.p;
  t.n_rec = h->n_rec;
  t.n_tail_blocks = 1;
  t.hb_gran = h->d_hb_gran.p;
  t.timeout = h->h_timeout.p;
  for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
    auto& P = Q.ph[ip];
    auto& D = *h->pd[ip];
    PcTailPhase& tp = t.ph[ip];
    tp.partials = D.partials_ext ? D.partials_ext : D.partials.p;
    tp.gran = D.gran.p;
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
  a.lds_out = wpt == 4 ? D.lds_out4 : D.lds_out;
  a.wpt = wpt;
  a.hslot0 = D.hslot0.p;
  a.hslotN = D.hslotN.p;
  a.partials = D.partials_ext ? D.partials_ext : D.partials.p;
  a.gran = D.gran.p;
  a.erec = h->d_erec.p;
  a.erec0 = D.erec0;
  a.tab = D.tab.p;
  a.tile_rec = (h->mixed && h->d_trec.p) ? h->d_trec.p + h->trec_first[ip] : nullptr;
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
                double* d_fobj, double* d_gradnz, int flags, hipStream_t st, double sigma, bool bulk = true,
                bool tail = true) {
  auto& Q = h->Q;
  // One launch per evaluation when every phase runs whole on this device: the tail is block 0 of the bulk launch and
  // receives the tiles' partial sums as granules (pc_kernels.hpp, RES).  A rank of the section-sharded evaluation,
  // whose sums travel through the all-gather first, and the bulk-only / tail-only calls take two launches.
  bool res = bulk && tail && h->resident && (Q.ph.size() > 1 ? h->bulk_all_res_fn != nullptr : h->pd[0]->fn_res != nullptr);
  for (size_t ip = 0; res && ip < Q.ph.size(); ++ip) {
    auto& D = *h->pd[ip];
    res = D.tile_begin == 0 && D.tile_end == D.n_tiles && !D.partials_ext;
  }
  if (h->host_args_dirty) {
    if (h->mixed) {   // the records of the launched tile ranges, in launch order (pc_args.h::PcTileRec)
      std::vector<PcTileRec> recs;
      h->trec_first.assign(Q.ph.size(), 0);
      for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
        auto& D = *h->pd[ip];
        h->trec_first[ip] = (int)recs.size();
        for (int t = D.tile_begin; t < D.tile_end; ++t) recs.push_back(h->trec_all[h->trec_base[ip] + t]);
      }
      if (recs.empty()) recs.push_back(PcTileRec{});
      HIP_OK(hipDeviceSynchronize());   // no launch in flight may still read the old records
      h->d_trec.upload(recs);
    }
    h->host_bulk_args.resize(Q.ph.size());
    for (size_t ip = 0; ip < Q.ph.size(); ++ip)
      fill_phase_args(h, ip, h->host_bulk_args[ip].a, nullptr, nullptr, nullptr, nullptr, nullptr, 0, h->pd[ip]->wpt);
    fill_tail_args(h, h->host_tail_launch.t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1.0);
    h->host_args_dirty = false;
  }
  auto patch_tail = [&](PcTailArgs& t, int block_threads) {
    t.x = d_x; t.lam = d_lam; t.c = d_c; t.G = d_G; t.H = d_H;
    t.fobj = d_fobj; t.grad_nz = d_gradnz; t.flags = flags; t.sigma = sigma;
    t.block_threads = block_threads;
    t.epoch = h->epoch;
  };
  if (res && ++h->epoch == 0) h->epoch = 1;   // (zero is the tag of never-written granules)
  struct MultiRes {
    PcMultiArgs m;
    PcTailArgs t;
  };
  if (bulk && (res ? h->bulk_all_res_fn : h->bulk_all_fn) && Q.ph.size() > 1) {
    // several phases, one launch: the phases' workgroups run side by side instead of one kernel after another
    if (h->args_dirty) {
      std::vector<PcPhaseArgs> blocks(Q.ph.size());
      for (size_t ip = 0; ip < Q.ph.size(); ++ip)
        fill_phase_args(h, ip, blocks[ip], nullptr, nullptr, nullptr, nullptr, nullptr, 0, h->wpt_all);
      HIP_OK(hipDeviceSynchronize());   // no launch in flight may still read the old blocks
      HIP_OK(hipMemcpy(h->d_phase_args.p, blocks.data(), blocks.size() * sizeof(PcPhaseArgs), hipMemcpyHostToDevice));
      h->args_dirty = false;
    }
    MultiRes mr;
    PcMultiArgs& m = mr.m;
    std::memset(&m, 0, sizeof(m));
    m.x = d_x;
    m.lam = d_lam;
    m.c = d_c;
    m.G = d_G;
    m.H = d_H;
    m.ph = reinterpret_cast<const PcPhaseArgs*>(h->d_phase_args.p);
    m.flags = flags;
    m.n_phases = (int32_t)Q.ph.size();
    m.epoch = h->epoch;
    m.trec = h->mixed ? h->d_trec.p : nullptr;
    int nb = 0;
    for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
      m.first_block[ip] = nb;
      nb += std::max(0, h->pd[ip]->tile_end - h->pd[ip]->tile_begin);
    }
    for (size_t ip = Q.ph.size(); ip <= PC_MAX_PHASES; ++ip) m.first_block[ip] = nb;
    if (res) {
      const int bt = h->TB * h->wpt_all;
      const int ntb = h->tail_blocks_env > 0 ? h->tail_blocks_env : ((h->heavy_point && bt < 256) ? 4 : 1);
      m.tail_blocks = ntb;
      mr.t = h->host_tail_launch.t;
      patch_tail(mr.t, bt);
      mr.t.n_tail_blocks = ntb;
      size_t sz = sizeof(mr);
      void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &mr, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
      HIP_OK(hipModuleLaunchKernel(h->bulk_all_res_fn, nb + ntb, 1, 1, bt, 1, 1,
                                   std::max(h->lds_all, h->tail_lds_bytes), st, nullptr, cfg));
      return;
    }
    if (nb > 0) {
      size_t sz = sizeof(m);
      void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &m, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
      HIP_OK(hipModuleLaunchKernel(h->bulk_all_fn, nb, 1, 1, h->TB * h->wpt_all, 1, 1, h->lds_all, st, nullptr, cfg));
    }
    bulk = false;
  }
  struct BulkRes {   // what the host hands to pc_bulk_p<i>_r: (lead scalars..., PcPhaseArgs a, PcTailArgs t)
    PcBulkArgs ba;
    PcTailArgs t;
  };
  for (size_t ip = 0; bulk && ip < Q.ph.size(); ++ip) {
    auto& D = *h->pd[ip];
    if (D.tile_end <= D.tile_begin) continue;
    PcPhaseArgs& a = h->host_bulk_args[ip].a;
    a.x = d_x; a.lam = d_lam; a.c = d_c; a.G = d_G; a.H = d_H;
    a.flags = flags;
    a.wpt = D.wpt;
    a.block_threads = h->TB * D.wpt;
    a.epoch = h->epoch;
    // (lead scalars..., PcPhaseArgs): the lead is what the command processor preloads into SGPRs (pc_args.h)
    PcBulkArgs& ba = h->host_bulk_args[ip];
    const int un = a.uni_n > 0 ? a.uni_n : 0;
    ba.lead = PcLead{a.x + a.x_off, a.lam ? a.lam + a.c_off : nullptr, a.qa, a.sec_h, a.N, a.K, a.tile_begin, a.n_blocks,
                     a.flags | (a.wpt << 8) | ((a.block_threads >> 6) << 12) | (a.spt << 16),
                     un ? (a.qa_off[un] | ((a.qa_total + a.qw_off[un]) << 16)) : 0};
    if (res) {   // single phase (several phases were launched above)
      static_assert(offsetof(BulkRes, t) == sizeof(PcBulkArgs), "kernel argument layout");
      BulkRes br;
      br.ba = ba;
      br.t = h->host_tail_launch.t;
      const int ntb = h->tail_blocks_env > 0 ? h->tail_blocks_env : ((h->heavy_point && a.block_threads < 256) ? 4 : 1);
      br.ba.lead.wa |= ntb << 28;
      patch_tail(br.t, a.block_threads);
      br.t.n_tail_blocks = ntb;
      size_t sz = sizeof(br);
      void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &br, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
      HIP_OK(hipModuleLaunchKernel(D.fn_res, D.n_tiles + ntb, 1, 1, a.block_threads, 1, 1,
                                   std::max(D.lds_bytes, h->tail_lds_bytes), st, nullptr, cfg));
      return;
    }
    size_t sz = sizeof(ba);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ba, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    HIP_OK(hipModuleLaunchKernel(D.fn, D.tile_end - D.tile_begin, 1, 1, h->TB * D.wpt, 1, 1, D.lds_bytes, st, nullptr, cfg));
  }
  if (!tail) return;
  PcTailLaunch& tl = h->host_tail_launch;
  PcTailArgs& t = tl.t;
  patch_tail(t, PC_TAIL_THREADS);
  tl.lead = PcTailLead{t.x, t.ph[0].partials, t.ph[0].scal, t.ph[0].x_off, t.ph[0].n_tiles, t.ph[0].N, t.flags, t.block_threads};
  size_t sz = sizeof(tl);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &tl, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  // more than four strides of partial sums per lane in some phase: the build that overlaps their loads
  int max_tiles = 0;
  for (size_t ip = 0; ip < Q.ph.size(); ++ip) max_tiles = std::max(max_tiles, (int)t.ph[ip].n_tiles);
  hipFunction_t fn = (h->tail_big_fn && max_tiles > 4 * PC_TAIL_THREADS) ? h->tail_big_fn : h->tail_fn;
  HIP_OK(hipModuleLaunchKernel(fn, 1, 1, 1, PC_TAIL_THREADS, 1, 1, h->tail_lds_bytes, st, nullptr, cfg));
}

void upload_scaling(pc_handle* h) {
  auto& Q = h->Q;
  for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
    auto& P = Q.ph[ip];
    auto& D = *h->pd[ip];
    const int NZ = P.n_z, NQ = P.n_q, NS = P.n_w, NY = P.n_y, NP = P.n_p;
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
    for (int l = 0; l < NS; ++l) s[o++] = h->V_ocp[P.wocp(l, Q.ocp_s_off)];   // parameters: q, t or s (pc_pattern.hpp)
    for (int l = 0; l < NS; ++l) s[o++] = h->r_ocp[P.wocp(l, Q.ocp_s_off)];
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
  h->x_valid = h->fc_valid = false;
}

void require_device(pc_handle* h) {
  if (!h) throw std::runtime_error("null handle");
  if (h->device < 0)
    throw std::runtime_error("handle was created with device = -1 (structure only): no evaluation is possible "
                             "without a GPU; this library has no CPU fallback");
  if (!h->scaling_set) throw std::runtime_error("pc_set_scaling must be called before evaluating");
  HIP_OK(hipSetDevice(h->device));
}

// ---- host-pointer staging ---------------------------------------------------------------------------
// The resident tail spins (bounded) for values of the other workgroups; a tail that gave up says so here.
void check_timeout(pc_handle* h) {
  if (h->h_timeout.p && h->h_timeout.p[0]) {
    const unsigned code = h->h_timeout.p[0];
    h->h_timeout.p[0] = 0;
    throw std::runtime_error("resident tail: granules of " + std::string(code >= 100 ? "an edge-node Hessian entry" : "a phase's partial sums") +
                             " never arrived (code " + std::to_string(code) + "); the results of this evaluation are invalid");
  }
}

// Wait for the handle's stream.  A blocking wait costs an interrupt and a wake-up (~10 us on the MI355X host, a
// quarter of a 10 k-node callback); the stream is polled for `spin_us` first, which is where a callback completes.
void wait_stream(pc_handle* h) {
  if (h->spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(h->stream);
      if (e == hipSuccess) { check_timeout(h); return; }
      if (e != hipErrorNotReady) HIP_OK(e);
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > h->spin_us) break;
    }
  }
  HIP_OK(hipStreamSynchronize(h->stream));
  check_timeout(h);
}
// what the kernels are handed for x~, lambda and the outputs: the device mirror, or the pinned host block itself
const double* kx(pc_handle* h) { return (h->host_mode & 1) ? h->h_in.p : h->d_in.p; }
const double* klam(pc_handle* h) { return kx(h) + h->o_lam; }
double* kout(pc_handle* h, size_t off) { return ((h->host_mode & 2) ? h->h_out.p : h->d_out.p) + off; }

// x~ of a callback: staged (and copied up) only when the caller says the point is new
void stage_x(pc_handle* h, const double* x, int new_x) {
  if (!new_x && h->x_valid) return;
  if (!x) throw std::runtime_error("null x");
  if (x != h->h_in.p) std::memcpy(h->h_in.p, x, h->Q.num_x * sizeof(double));
  if (!(h->host_mode & 1))
    HIP_OK(hipMemcpyAsync(h->d_in.p, h->h_in.p, h->Q.num_x * sizeof(double), hipMemcpyHostToDevice, h->stream));
  h->x_valid = true;
  h->fc_valid = false;
}

void stage_lambda(pc_handle* h, const double* lambda) {
  if (!lambda) throw std::runtime_error("null lambda");
  double* dst = h->h_in.p + h->o_lam;
  if (lambda != dst) std::memcpy(dst, lambda, h->Q.num_c * sizeof(double));
  if (!(h->host_mode & 1))
    HIP_OK(hipMemcpyAsync(h->d_in.p + h->o_lam, dst, h->Q.num_c * sizeof(double), hipMemcpyHostToDevice, h->stream));
}

void copy_down(pc_handle* h, size_t begin, size_t end) {   // [begin, end) of the output block, device -> pinned host
  if (h->host_mode & 2) return;                             // the kernels wrote there themselves
  HIP_OK(hipMemcpyAsync(h->h_out.p + begin, h->d_out.p + begin, (end - begin) * sizeof(double), hipMemcpyDeviceToHost,
                        h->stream));
}

// a kernel of the handle's code object, whichever of its modules holds it
hipError_t find_fn(pc_handle* h, hipFunction_t* fn, const char* name) {
  *fn = nullptr;
  hipError_t e = hipModuleGetFunction(fn, h->module, name);
  for (size_t i = 0; e != hipSuccess && i < h->more_modules.size(); ++i) {
    (void)hipGetLastError();
    e = hipModuleGetFunction(fn, h->more_modules[i], name);
  }
  if (e != hipSuccess) *fn = nullptr;
  return e;
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
    pcp::from_desc(*d, Q);   // pc_desc.hpp: plain C++, also compiled (with sanitizers) by the CPU test harness
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
    if (auto_tb) {
      TB = Nmax >= 262144 ? 256 : (Nmax >= 65536 ? 128 : 64);
      // a one-state model has so little per node that the widest workgroup pays much earlier (hypersensitive, TB =
      // 64 / 128 / 256: 20 k nodes 5.11 / 4.94-5.21 / 5.0-5.2 us, 30 k 5.46 / 5.41 / 5.18, 50 k 6.67 / 5.81 / 5.70,
      // 65 k 8.92 / 6.76 / 5.90, 150 k 14.5 / 9.05 / 8.27, 250 k 22.1 / 12.1 / 9.97)
      bool one_state = true;
      for (int ip = 0; ip < d->n_phases; ++ip) one_state = one_state && d->phases[ip].n_y == 1 && d->phases[ip].eval_ops <= 4000;
      if (one_state && Nmax >= 25000) TB = 256;
    }
    if (TB != 64 && TB != 128 && TB != 256) throw std::runtime_error("threads_per_block must be 64, 128 or 256");
    for (auto& P : Q.ph) pcp::finalize_phase_tables(P, Q.n_s);
    const int qa_n = (int)h->qa.size(), qw_n = (int)h->qw.size();
    // bytes of LDS a workgroup of W waves needs when tiles hold at most `tc` nodes (worst phase; before the tiles exist:
    // a tile of tc nodes has at most tc - 1 defect rows per state)
    auto lds_need = [&](int tb, int tc, int W) {
      int mx = 0;
      for (auto& P : Q.ph)
        mx = std::max(mx, phase_lds_bytes(P, W > 1 ? 64 : tb, qa_n, qw_n, W * phase_lds_out(P, tc - 1, tc), P.compiled_order == 0));
      return mx;
    };
    if (auto_tb)   // largest tile whose staging fits the 64 KiB of dynamic LDS a module kernel may request
      while (TB > 64 && lds_need(TB, TB, 1) > h->lds_limit) TB /= 2;
    h->TB = TB;
    // Tile capacity in nodes (<= TB): a tile smaller than its workgroup leaves lanes idle but shortens every
    // wave's store phase and puts more waves on the chip -- what a problem with far fewer tiles than SIMDs wants.
    int TC = TB, max_nk = 2;
    for (auto& P : Q.ph)
      for (int k = 0; k < P.K; ++k) max_nk = std::max(max_nk, P.n_k[k]);
    // The two-wave build of a heavy model (codegen: pc_bulk_all_r_w2 / pc_bulk_p<i>_r_w2, <= 256 registers): two waves
    // share every 64-node tile, each with its own staging region, and a SIMD holds two of them -- if the CU's 160 KiB of
    // LDS hold four tiles.  The tile capacity is the largest that allows it (Delta III, order 5: 60 rows of 40 doubles per
    // replica; 4 x 12.5 k nodes 30.6 -> 23.2 us).  PYCOLLO_AMD_TWO_WAVE=0 turns the mode off; an explicit
    // PYCOLLO_AMD_WPT / PYCOLLO_AMD_TILE_NODES leaves the choice to the caller.
    for (auto& P : Q.ph) h->mixed = h->mixed || !P.spec_orders.empty();
    h->two_wave = d->two_wave_occupancy >= 2 && TB == 64 && !std::getenv("PYCOLLO_AMD_WPT") &&
                  !std::getenv("PYCOLLO_AMD_TILE_NODES");
    if (const char* env = std::getenv("PYCOLLO_AMD_TWO_WAVE")) h->two_wave = h->two_wave && std::atoi(env) != 0;
    {   // with few tiles (a rank's share of a sharded mesh, a coarse mesh) four waves per tile serve better: one wave
        // per SIMD either way, and the shorter one wins
      int64_t tiles64 = 0;
      for (auto& P : Q.ph) {
        int64_t N = 1;
        for (int k = 0; k < P.K; ++k) N += P.n_k[k] - 1;
        tiles64 += (N + 62) / 63;
        h->two_wave = h->two_wave && P.n_y >= 2;
      }
      // ... and with many more tiles than the chip holds at once the replicas' second evaluation of the node functions
      // buys nothing: the (split, register-capped) one-wave-per-tile kernel already runs two waves per SIMD
      int64_t max_tiles = 1 << 30;
      if (const char* env = std::getenv("PYCOLLO_AMD_TWO_WAVE_MAX_TILES")) max_tiles = std::atoll(env);
      h->two_wave = h->two_wave && tiles64 > 400 && tiles64 <= max_tiles;
    }
    constexpr int kQuarterCu = 30 * 1280;
    // A two-wave launch is one generation of waves and lasts one wave's life, which grows with the tile's rows (its
    // share of the flush): the tile capacity is the one that cuts the mesh into about 896 tiles (3.5 workgroups per CU,
    // 1 792 of the chip's 2 048 wave slots) when that is smaller than what the LDS allows.  Measured on one box, Delta III
    // 4 x 12.5 k nodes, tiles / us: order 4  796 / 25.8, 928 / 22.3, 984 / 22.4;  order 5  896 / 23.2, 964 / 26.0;  order 6
    // 836 / 32.1, 912 / 29.3, 1000 / 28.8;  order 7  836 / 31.4, 928 / 27.7;  order 8  796 / 35.7, 896 / 32.0, 1020 / 33.6
    // (profiles/r04_two_wave_tiles.txt).  PYCOLLO_AMD_TWO_WAVE_TILES overrides the target (0: the largest tile the LDS allows).
    auto fill_tc = [&](int tc_max) {
      int64_t want = 896;
      if (const char* env = std::getenv("PYCOLLO_AMD_TWO_WAVE_TILES")) want = std::atoll(env);
      if (want <= 0) return tc_max;
      int64_t rows = 0;
      for (auto& P : Q.ph)
        for (int k = 0; k < P.K; ++k) rows += P.n_k[k] - 1;
      const int per = (int)((rows + want - 1) / want);
      return std::max(std::max(max_nk, 24), std::min(tc_max, per + 1));
    };
    if (h->mixed) {
      // Mixed build: tiles are cut per order (pc_pattern.hpp::build_tiles_mixed) under row caps that keep a workgroup's
      // LDS inside the budget -- a quarter CU for the two-wave build (below), else what a workgroup may request -- and
      // the two-wave build is kept when every wave of the launch is then resident at once, as for a uniform mesh.
      if (const char* env = std::getenv("PYCOLLO_AMD_MIX_MIN_RUN_ROWS"))
        for (auto& P : Q.ph) P.min_run_rows = std::max(1, std::atoi(env));
      auto cut = [&](int W, int budget) {
        int64_t tiles = 0;
        for (auto& P : Q.ph) {
          if (!P.spec_orders.empty()) {
            pcp::phase_set_caps(P, W > 1 ? 64 : TB, W, qa_n, qw_n, budget);
            // A launch is one generation of waves and lasts as long as its slowest one: a tile of several orders runs
            // the any-order body, which takes about twice as long per row as an order-specialised one -- such tiles are
            // kept to half the rows so that they finish with the others (measured: with 63-row any-order tiles among
            // 835 order-pure ones the mixed build ran no faster than the any-order kernel alone, 51.3 / 50.6 us)
            int mix_rows = 32;
            if (const char* env = std::getenv("PYCOLLO_AMD_MIX_CAP_ROWS")) mix_rows = std::max(8, std::atoi(env));
            P.mix_cap_rows = std::min(P.mix_cap_rows, mix_rows);
          }
          pcp::build_tiles(P, TC);
          tiles += (int64_t)P.tile_k0.size() - 1;
        }
        return tiles;
      };
      if (h->two_wave) {
        // (phases of a single order inside a mixed problem: one tile capacity for them, as below)
        int tc = 64;
        auto need_uniform = [&](int t) {
          int mx = 0;
          for (auto& P : Q.ph)
            if (P.spec_orders.empty())
              mx = std::max(mx, phase_lds_bytes(P, 64, qa_n, qw_n, 2 * phase_lds_out(P, t - 1, t), P.compiled_order == 0));
          return mx;
        };
        while (tc > std::max(max_nk, 32) && need_uniform(tc) > kQuarterCu) --tc;
        int64_t max_waves = 2048;   // every wave of the launch resident at once (two per SIMD)
        if (const char* env = std::getenv("PYCOLLO_AMD_TWO_WAVE_MAX_WAVES")) max_waves = std::atoll(env);
        // the smallest tile capacity from the chip-filling one upwards whose cut keeps every wave resident
        bool fits = false, fixed = false;
        for (auto& P : Q.ph) fixed = fixed || !P.fixed_tile_k0.empty();
        if (fixed) {   // the caller's tiles (a rank-local handle): two waves per tile when its actual tiles allow it
          TC = 64;
          const int64_t tiles = cut(2, kQuarterCu);
          int need = 0;
          for (auto& P : Q.ph)
            if (!P.spec_orders.empty()) {
              bool ap = false, ag = false;
              const int out = 2 * phase_lds_out_tiles(P, &ap, &ag);
              if (ag) need = std::max(need, phase_lds_bytes(P, 64, qa_n, qw_n, out, true, true));
              if (ap) need = std::max(need, phase_lds_bytes(P, 64, qa_n, qw_n, out, false, true));
            }
          fits = need <= kQuarterCu && need_uniform(64) <= kQuarterCu && 2 * tiles <= max_waves;
        }
        for (int t = std::min(tc, fill_tc(tc)); !fixed && t <= tc && !fits; ++t) {
          TC = t;
          fits = need_uniform(t) <= kQuarterCu && 2 * cut(2, kQuarterCu) <= max_waves;
        }
        if (!fits) {
          h->two_wave = false;
          TC = TB;
        }
      }
      if (!h->two_wave) cut(1, h->lds_limit);
    } else if (h->two_wave) {
      // four workgroups per CU: a quarter of the 160 KiB less two allocation granules of 1280 B -- measured, Delta III
      // 4 x 12.5 k nodes: 37 272 B per workgroup 23.2 us, 39 832 B (nominally still a quarter) 26.8 us
      int tc = 64;
      while (tc > std::max(max_nk, 32) && lds_need(64, tc, 2) > kQuarterCu) --tc;
      // ... and every wave of the launch resident at once (2048 wave slots at two per SIMD): with smaller tiles than
      // that allows, the one-wave-per-tile kernel -- itself two waves per SIMD -- is as fast or faster (Delta III order 6:
      // 1160 tiles of 45 nodes x 2 waves 34.0 us, 872 tiles x 1 wave 32.2 us; 4 x 25 k nodes, order 5: 42.7 / 42.2 us)
      auto tiles_at = [&](int t) {
        int64_t tiles_tc = 0;
        for (auto& P : Q.ph) {
          int64_t N = 1, nmax = 2;
          for (int k = 0; k < P.K; ++k) { N += P.n_k[k] - 1; nmax = std::max<int64_t>(nmax, P.n_k[k]); }
          const int64_t per_tile = std::max<int64_t>(1, ((t - 1) / (nmax - 1)) * (nmax - 1));
          tiles_tc += (N - 1 + per_tile - 1) / per_tile;
        }
        return tiles_tc;
      };
      bool fits = false;
      if (lds_need(64, tc, 2) <= kQuarterCu)
        for (int t = std::min(tc, fill_tc(tc)); t <= tc && !fits; ++t)
          if (2 * tiles_at(t) <= 2048) { TC = t; fits = true; }
      if (!fits) h->two_wave = false;
    }
    if (const char* env = std::getenv("PYCOLLO_AMD_TILE_NODES")) {
      const int v = std::atoi(env);
      if (v >= max_nk && v <= TB) TC = v;
    }
    h->TC = TC;
    if (d->plan_only) {   // the tiling alone (a rank of the section-sharded evaluation asks for its range, sharding.py)
      pcp::build_all(Q, TC, true);
      h->device = -1;
      h->pd.resize(Q.ph.size());
      for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
        h->pd[ip].reset(new PhaseDev());
        h->pd[ip]->n_tiles = (int)Q.ph[ip].tile_k0.size() - 1;
      }
      return;
    }
    pcp::build_all(Q, TC);
    if (h->mixed) {   // one record per tile: which body runs it and where it sits in the mesh
      for (auto& P : Q.ph)
        for (int k = 0; k < P.K; ++k)
          if (h->qa_off[P.n_k[k]] < 0) throw std::runtime_error("no quadrature table for a section order in use");
      h->trec_base.assign(Q.ph.size(), 0);
      for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
        auto& P = Q.ph[ip];
        h->trec_base[ip] = h->trec_all.size();
        for (size_t t = 0; t + 1 < P.tile_k0.size(); ++t) {
          PcTileRec r{};
          const int k0 = P.tile_k0[t];
          r.phase = (int32_t)ip;
          r.tile = (int32_t)t;
          r.order = t < P.tile_order.size() ? P.tile_order[t] : 0;
          r.k0 = k0;
          r.nsec = P.tile_k0[t + 1] - k0;
          r.n0 = P.sec_s[k0];
          r.nprev = k0 > 0 ? P.n_k[k0 - 1] : 0;
          r.qa_prev = r.nprev ? h->qa_off[r.nprev] : 0;
          r.E0 = P.sec_E[k0];
          r.w_prev = r.nprev ? h->qw[h->qw_off[r.nprev] + r.nprev - 1] : 0.0;
          h->trec_all.push_back(r);
        }
      }
    }
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
    if (const char* env = std::getenv("PYCOLLO_AMD_RESIDENT")) h->resident = std::atoi(env) != 0;
    if (const char* env = std::getenv("PYCOLLO_AMD_SPIN_US")) h->spin_us = std::atoi(env);
    // the endpoint block is generated in four parts, one per wave of a 256-thread tail; a small block is not worth
    // the three extra waves every tile's workgroup then carries (they exit at their first instruction)
    h->heavy_point = Q.point_x.size() + (size_t)Q.n_b + Q.pthess_row.size() > 16;
    if (const char* env = std::getenv("PYCOLLO_AMD_TAIL_BLOCKS")) h->tail_blocks_env = std::min(4, std::max(0, std::atoi(env)));
    // the tail's LDS carve (pc::tail_lds): acc | part | sum | xb | lb | hold | hb
    for (auto& P : Q.ph) h->lds_nred = std::max(h->lds_nred, P.nred);
    h->tail_lds_bytes = 8 * (int)(Q.tail_owned.size() + 17 * (size_t)h->lds_nred + Q.point_x.size() + (size_t)Q.n_b +
                                  2 * Q.pthess_row.size() + 2);
    h->pd.resize(Q.ph.size());
    for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
      h->pd[ip].reset(new PhaseDev());
      auto& P = Q.ph[ip];
      auto& D = *h->pd[ip];
      D.n_tiles = (int)P.tile_k0.size() - 1;
      D.tile_begin = 0;
      D.tile_end = D.n_tiles;
      const int nfs = phase_nfs(P);
      D.nfs = nfs;
      const bool mesh_tables = P.compiled_order == 0;
      // uniform section order: index arithmetic replaces the section tables
      bool same = true;
      for (int k = 0; k < P.K; ++k) same = same && P.n_k[k] == P.n_k[0];
      same = same && P.fixed_tile_k0.empty();   // (a caller's tile table is read from the tables, never computed from the tile index)
      D.uni_n = same ? P.n_k[0] : 0;
      D.spt = same ? (TC - 1) / (P.n_k[0] - 1) : 0;
      if (same && P.tile_k0.size() > 1 && P.tile_k0[1] != std::min(D.spt, P.K))
        throw std::runtime_error("internal error: uniform tiling mismatch");
      D.lds_out = phase_lds_out(P, phase_max_tile_rows(P), std::min(TC, phase_max_tile_rows(P) + 1));   // the largest tile's runs
      bool any_pure = false, any_generic = false;
      D.lds_out4 = phase_lds_out(P, phase_max_tile_rows(P), std::min(TC, phase_max_tile_rows(P) + 1), false);
      if (!P.spec_orders.empty()) {   // every row with its own order
        D.lds_out = phase_lds_out_tiles(P, &any_pure, &any_generic);
        D.lds_out4 = phase_lds_out_tiles(P, nullptr, nullptr, false);
      }
      // LDS of the phase's workgroups with w staging regions: the larger of the tile bodies the phase's tiles run
      D.lds_for = [&P, &D, TB, qa_n, qw_n, mesh_tables, any_pure, any_generic](int w) {
        const int out = (w == 4 ? D.lds_out4 : D.lds_out) * w;
        if (P.spec_orders.empty()) return phase_lds_bytes(P, TB, qa_n, qw_n, out, mesh_tables);
        int b = 0;
        if (any_generic) b = std::max(b, phase_lds_bytes(P, TB, qa_n, qw_n, out, true, true));
        if (any_pure) b = std::max(b, phase_lds_bytes(P, TB, qa_n, qw_n, out, false, true));
        return b;
      };
      // Few tiles and several states: W waves share a tile and split its output runs, so that the chip's 1024
      // SIMDs each hold a wave (or two) instead of a fraction of them holding one long-running wave.  Beyond that
      // the replicas only add redundant node evaluations (measured on 64-node tiles, W = 1 / 2 / 4: shuttle
      // 381 tiles 22.7 / 19.2 / 17.6 us, 953 tiles 25.2 / 23.3 / 31.3 us, 2858 tiles 54 / 60 / 81 us).
      D.wpt = 1;
      if (TB == 64) {
        // (one state: W = 2 measured no faster, 5.58 vs 5.49 us.  A heavy model -- Delta III, ~10k operations --
        //  loses at 834 tiles, 358 / 410 / 522 us: every sharing wave re-evaluates the node functions)
        const bool heavy = P.eval_ops > 4000;
        if (P.n_y >= 2 && D.n_tiles <= (heavy ? 512 : 1024)) D.wpt = 2;
        if (P.n_y >= 3 && D.n_tiles <= 400) D.wpt = 4;
        // One state, one launch per evaluation: the last tile's store phase is what the launch waits for once the tail
        // runs beside the tiles, and four waves get a tile's runs out sooner than one (hypersensitive, W = 1 / 2 / 4:
        // 42 tiles 4.69 / 4.34 / 4.25 us, 167 tiles 4.78-4.99 / 4.59-4.87 / 4.46-4.69, 334 tiles 5.94 / 5.00 / 4.94,
        // 500 tiles 5.92 / 5.29 / 5.44; with a separate tail launch W = 2 measured no gain, 5.58 vs 5.49 us)
        if (P.n_y == 1 && !heavy && h->resident) D.wpt = D.n_tiles <= 400 ? 4 : (D.n_tiles <= 1024 ? 2 : 1);
        if (const char* env = std::getenv("PYCOLLO_AMD_WPT")) {
          const int v = std::atoi(env);
          if (v == 1 || v == 2 || v == 4) D.wpt = v;
        }
        if (h->two_wave) D.wpt = 2;
        while (D.wpt > 1 && D.lds_for(D.wpt) > h->lds_limit) D.wpt /= 2;
      }
      D.lds_bytes = D.lds_for(D.wpt);
      h->lds_max = std::max(h->lds_max, D.lds_bytes);
      if (D.lds_bytes > h->lds_limit)
        throw std::runtime_error("tile needs more dynamic LDS than a workgroup may request; use a smaller "
                                 "threads_per_block");
      if ((int)P.goff.size() > PC_MAX_GOFF || (int)P.hoff.size() > PC_MAX_HOFF)
        throw std::runtime_error("too many variables/constraints per phase for the kernel argument block");
    }
    h->n_launches = (int)Q.ph.size() + 1;   // refined after the module is loaded (resident tail: one launch)
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
          (void)P;
          mx = std::max(mx, D.lds_for(w));
        }
        return mx;
      };
      if (TB == 64) {
        if (min_ny >= 2 && total <= (heavy ? 512 : 1024)) h->wpt_all = 2;
        if (min_ny >= 3 && total <= 400) h->wpt_all = 4;
        if (const char* env = std::getenv("PYCOLLO_AMD_WPT")) {
          const int v = std::atoi(env);
          if (v == 1 || v == 2 || v == 4) h->wpt_all = v;
        }
        if (h->two_wave) h->wpt_all = 2;
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
    {
      const std::string path = d->code_object, ext = ".hsaco";
      const std::string base = path.size() > ext.size() && path.compare(path.size() - ext.size(), ext.size(), ext) == 0
                                   ? path.substr(0, path.size() - ext.size()) : path;
      for (int k = 1; k < 16; ++k) {
        const std::string part = base + ".p" + std::to_string(k) + ext;
        FILE* f = std::fopen(part.c_str(), "rb");
        if (!f) break;
        std::fclose(f);
        hipModule_t m = nullptr;
        HIP_OK(hipModuleLoad(&m, part.c_str()));
        h->more_modules.push_back(m);
      }
    }
    HIP_OK(find_fn(h.get(), &h->tail_fn, d->tail_kernel));
    if (find_fn(h.get(), &h->tail_big_fn, (std::string(d->tail_kernel) + "_big").c_str()) != hipSuccess)
      h->tail_big_fn = nullptr;
    if (Q.ph.size() > 1) {
      bool merge = true;
      if (const char* env = std::getenv("PYCOLLO_AMD_MERGE")) merge = std::atoi(env) != 0;
      if (!merge || find_fn(h.get(), &h->bulk_all_fn, "pc_bulk_all") != hipSuccess) h->bulk_all_fn = nullptr;
      if (h->bulk_all_fn) {
        h->d_phase_args.alloc(Q.ph.size() * sizeof(PcPhaseArgs));
        h->n_launches = 2;
        if (find_fn(h.get(), &h->bulk_all_res_fn, "pc_bulk_all_r") != hipSuccess) h->bulk_all_res_fn = nullptr;
        if (h->bulk_all_res_fn && h->wpt_all > 1) {   // the same launch with the replica index compiled in
          hipFunction_t fw = nullptr;
          if (find_fn(h.get(), &fw, ("pc_bulk_all_r_w" + std::to_string(h->wpt_all)).c_str()) == hipSuccess && fw)
            h->bulk_all_res_fn = fw;
          else
            (void)hipGetLastError();
        }
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
      HIP_OK(find_fn(h.get(), &D.fn, P.bulk_kernel.c_str()));
      if (Q.ph.size() == 1 && find_fn(h.get(), &D.fn_res, (P.bulk_kernel + "_r").c_str()) != hipSuccess)
        D.fn_res = nullptr;   // code object without the resident-tail variant: two launches per evaluation
      if (D.fn_res && D.wpt > 1) {   // the same kernel with the replica index compiled in (codegen: light / medium models)
        hipFunction_t fw = nullptr;
        if (find_fn(h.get(), &fw, (P.bulk_kernel + "_r_w" + std::to_string(D.wpt)).c_str()) == hipSuccess && fw)
          D.fn_res = fw;
        else
          (void)hipGetLastError();
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
      D.gran.upload(std::vector<unsigned long long>((size_t)2 * std::max(1, P.nred) * D.n_tiles, 0ull));   // tag 0: never written
    }
    {
      // Resident tail: a Hessian entry of an edge node 0 / N-1 on which an endpoint term lands reaches the tail
      // workgroup as a record (two granules); the tail adds the term and stores the entry.  Records are numbered per
      // phase, node 0 then node N-1, in the kernels' site order (z-z entries, t strips, s strips) -- the order the
      // generated kernels rank their compile-time flags in (codegen.edge_flags, S<M>::erank).
      std::map<int64_t, int32_t> term_of_slot;   // slot -> endpoint entry
      for (size_t e = 0; e < Q.pt_hslot.size(); ++e)
        if (Q.pt_hlocal[e] < 0) term_of_slot.emplace(Q.pt_hslot[e], (int32_t)e);
      std::vector<int64_t> rec_slot;
      std::vector<int32_t> rec_term;
      for (size_t ip = 0; ip < Q.ph.size(); ++ip) {
        auto& P = Q.ph[ip];
        const int NZ = P.n_z, NS = P.n_w, NHZZ = (int)P.hslot0.size(), NE = NHZZ + 2 * NZ + NS * NZ;
        h->pd[ip]->erec0 = (int32_t)rec_slot.size();
        int found[2] = {0, 0};
        for (int edge = 0; edge < 2; ++edge) {
          const int64_t node = edge ? P.N - 1 : 0;
          for (int site = 0; site < NE; ++site) {
            int64_t slot = -1;
            if (site < NHZZ) {
              slot = edge ? P.hslotN[site] : P.hslot0[site];
            } else {
              const int64_t base = P.hoff[NZ + (site - NHZZ)];   // t strips (j, z) then s strips (l, z)
              if (base >= 0) slot = base + node;
            }
            auto it = slot >= 0 ? term_of_slot.find(slot) : term_of_slot.end();
            if (it == term_of_slot.end()) continue;
            rec_slot.push_back(slot);
            rec_term.push_back(it->second);
            term_of_slot.erase(it);
            ++found[edge];
          }
        }
        if (d->phases[ip].n_edge_rec[0] != found[0] || d->phases[ip].n_edge_rec[1] != found[1])
          throw std::runtime_error("the phase descriptor's edge-record counts do not match the Hessian pattern "
                                   "(codegen.edge_flags and pc_pattern.hpp disagree)");
      }
      if (!term_of_slot.empty())
        throw std::runtime_error("internal error: an endpoint Hessian term lands on an edge-node entry no tile produces");
      h->n_rec = (int32_t)rec_slot.size();
      if (rec_slot.empty()) { rec_slot.push_back(-1); rec_term.push_back(-1); }
      h->d_rec_slot.upload(rec_slot);
      h->d_rec_term.upload(rec_term);
      h->d_erec.upload(std::vector<unsigned long long>(2 * rec_slot.size(), 0ull));
    }
    h->d_hb_gran.upload(std::vector<unsigned long long>(2 * std::max<size_t>(1, Q.pthess_row.size()), 0ull));
    h->h_timeout.alloc(16);
    std::memset(h->h_timeout.p, 0, 16 * sizeof(unsigned));
    h->d_point_x.upload(Q.point_x);
    h->d_tail_owned.upload(Q.tail_owned);
    h->d_pt_hslot.upload(Q.pt_hslot);
    h->d_pt_hlocal.upload(Q.pt_hlocal);
    h->d_g_indptr.upload(Q.g_indptr);
    const size_t nG = Q.g_row.size(), nH = Q.h_row.size();
    {
      // every part starts on a 256-byte boundary (what a separate allocation would give the kernels)
      auto up = [](size_t v) { return (v + 31) & ~(size_t)31; };
      h->o_lam = up(Q.num_x);
      h->in_total = h->o_lam + up(Q.num_c);
      h->o_f = 0;
      h->o_gn = 1;
      h->o_c = up(1 + Q.jgrad_col.size());
      h->o_G = h->o_c + up(Q.num_c);
      h->o_H = h->o_G + up(nG);
      h->out_total = h->o_H + up(nH);
    }
    h->d_in.alloc(h->in_total); h->d_out.alloc(h->out_total);
    h->h_in.alloc(h->in_total); h->h_out.alloc(h->out_total);
    h->d_norms.alloc(Q.num_c); h->h_norms.alloc(Q.num_c);
    HIP_OK(hipMemset(h->d_in.p, 0, h->in_total * sizeof(double)));
    HIP_OK(hipMemset(h->d_out.p, 0, h->out_total * sizeof(double)));
    std::memset(h->h_in.p, 0, h->in_total * sizeof(double));
    std::memset(h->h_out.p, 0, h->out_total * sizeof(double));
    HIP_OK(hipEventCreateWithFlags(&h->ev_small, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&h->ev_G, hipEventDisableTiming));
    if (const char* env = std::getenv("PYCOLLO_AMD_HOST_MODE")) h->host_mode = std::atoi(env) & 3;
    if (h->resident && (Q.ph.size() > 1 ? h->bulk_all_res_fn != nullptr : h->pd[0]->fn_res != nullptr)) h->n_launches = 1;
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
  if (h->ev_small) (void)hipEventDestroy(h->ev_small);
  if (h->ev_G) (void)hipEventDestroy(h->ev_G);
  for (hipModule_t m : h->more_modules) (void)hipModuleUnload(m);
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
    info->lds_bytes_max = (h->Q.ph.size() > 1 && h->lds_all > 0) ? h->lds_all : h->lds_max;   // the merged launch's when there is one
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
      // evaluations may be in flight on a caller's stream (pc_eval_all_device): none may still read the tables
      HIP_OK(hipSetDevice(h->device));
      HIP_OK(hipDeviceSynchronize());
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
    h->fc_valid = h->small_synced = h->G_synced = false;   // these launches write the handle's f block: a companion host-pointer call must re-evaluate
    check_timeout(h);   // a previous device-API evaluation whose tail gave up: say so before queueing more work
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    launch_all(h, d_x, d_lambda, d_g, d_jac, d_hess, h->d_out.p + h->o_f, nullptr, PC_FLAG_C | PC_FLAG_G | PC_FLAG_H, st,
               obj_factor);
  });
}

int pc_check(pc_handle* h) {
  return guarded([&] {
    require_device(h);
    check_timeout(h);
  });
}

int pc_launch_bulk_device(pc_handle* h, const double* d_x, const double* d_lambda, double* d_g, double* d_jac,
                          double* d_hess, void* stream) {
  return guarded([&] {
    require_device(h);
    h->fc_valid = h->small_synced = h->G_synced = false;   // these launches write the handle's f block: a companion host-pointer call must re-evaluate
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    launch_all(h, d_x, d_lambda, d_g, d_jac, d_hess, h->d_out.p + h->o_f, nullptr, PC_FLAG_C | PC_FLAG_G | PC_FLAG_H, st, 1.0,
               true, false);
  });
}

int pc_launch_bulk_flags_device(pc_handle* h, const double* d_x, const double* d_lambda, double* d_g, double* d_jac,
                                double* d_hess, int flags, void* stream) {
  return guarded([&] {
    require_device(h);
    h->fc_valid = h->small_synced = h->G_synced = false;   // these launches write the handle's f block: a companion host-pointer call must re-evaluate
    if (flags & ~(PC_FLAG_C | PC_FLAG_G | PC_FLAG_H)) throw std::runtime_error("flags must be a combination of 1 (g), 2 (jac_g), 4 (hess)");
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    launch_all(h, d_x, d_lambda, d_g, d_jac, d_hess, h->d_out.p + h->o_f, nullptr, flags, st, 1.0, true, false);
  });
}

int pc_launch_tail_device(pc_handle* h, const double* d_x, double obj_factor, const double* d_lambda, double* d_g,
                          double* d_jac, double* d_hess, void* stream) {
  return guarded([&] {
    require_device(h);
    h->fc_valid = h->small_synced = h->G_synced = false;   // these launches write the handle's f block: a companion host-pointer call must re-evaluate
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    launch_all(h, d_x, d_lambda, d_g, d_jac, d_hess, h->d_out.p + h->o_f, nullptr, PC_FLAG_C | PC_FLAG_G | PC_FLAG_H, st,
               obj_factor, false, true);
  });
}

// The tail launch of a sharded evaluation that also delivers the objective and its gradient (host): what a solver's
// objective / gradient callbacks need from a rank that never runs the whole evaluation (pycollo_amd/ipm_sharded.py).
// Synchronises `stream`.
int pc_launch_tail_objective_device(pc_handle* h, const double* d_x, double obj_factor, const double* d_lambda, double* d_g,
                                    double* d_jac, double* d_hess, void* stream, double* f, double* grad) {
  return guarded([&] {
    require_device(h);
    auto& Q = h->Q;
    h->fc_valid = h->small_synced = h->G_synced = false;
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    launch_all(h, d_x, d_lambda, d_g, d_jac, d_hess, h->d_out.p + h->o_f, h->d_out.p + h->o_gn,
               PC_FLAG_C | PC_FLAG_G | PC_FLAG_H, st, obj_factor, false, true);
    HIP_OK(hipMemcpyAsync(h->h_out.p, h->d_out.p, (1 + Q.jgrad_col.size()) * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (f) *f = h->h_out.p[h->o_f];
    if (grad) {
      std::memset(grad, 0, Q.num_x * sizeof(double));
      for (size_t e = 0; e < Q.jgrad_col.size(); ++e) grad[Q.point_x[Q.jgrad_col[e]]] = h->h_out.p[h->o_gn + e];
    }
  });
}

int pc_set_tile_range(pc_handle* h, int phase, int tile_begin, int tile_end) {
  return guarded([&] {
    if (!h || phase < 0 || phase >= (int)h->pd.size()) throw std::runtime_error("phase out of range");
    auto& D = *h->pd[phase];
    if (tile_begin < 0 || tile_end < tile_begin || tile_end > D.n_tiles) throw std::runtime_error("tile range out of range");
    if (h->device >= 0) {
      HIP_OK(hipSetDevice(h->device));
      HIP_OK(hipDeviceSynchronize());   // launches in flight on any stream still use the old range
    }
    D.tile_begin = tile_begin;
    D.tile_end = tile_end;
    h->fc_valid = false;
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

int pc_phase_tile_orders(const pc_handle* h, int phase, int32_t* tile_order) {
  return guarded([&] {
    if (!h || phase < 0 || phase >= (int)h->Q.ph.size() || !tile_order) throw std::runtime_error("phase out of range or null output");
    auto& P = h->Q.ph[phase];
    for (size_t t = 0; t + 1 < P.tile_k0.size(); ++t) tile_order[t] = t < P.tile_order.size() ? P.tile_order[t] : 0;
  });
}

int pc_set_partials_buffer(pc_handle* h, int phase, double* d_partials) {
  return guarded([&] {
    if (!h || phase < 0 || phase >= (int)h->pd.size()) throw std::runtime_error("phase out of range");
    if (h->device >= 0) {
      HIP_OK(hipSetDevice(h->device));
      HIP_OK(hipDeviceSynchronize());
    }
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
    if (!x || !lambda || !g || !jac || !hess) throw std::runtime_error("null argument");
    // one block up: [x~ | lambda] (contiguous in the pinned block, gap included)
    if (x != h->h_in.p) std::memcpy(h->h_in.p, x, Q.num_x * sizeof(double));
    if (lambda != h->h_in.p + h->o_lam) std::memcpy(h->h_in.p + h->o_lam, lambda, Q.num_c * sizeof(double));
    if (!(h->host_mode & 1))
      HIP_OK(hipMemcpyAsync(h->d_in.p, h->h_in.p, h->in_total * sizeof(double), hipMemcpyHostToDevice, h->stream));
    launch_all(h, kx(h), klam(h), kout(h, h->o_c), kout(h, h->o_G), kout(h, h->o_H), kout(h, h->o_f), kout(h, h->o_gn),
               PC_FLAG_C | PC_FLAG_G | PC_FLAG_H, h->stream, obj_factor);
    // one block down: [J | grad J | c~ | G~ | H~]
    copy_down(h, 0, h->o_H + Q.h_row.size());
    wait_stream(h);
    if (g != h->h_out.p + h->o_c) std::memcpy(g, h->h_out.p + h->o_c, Q.num_c * sizeof(double));
    if (jac != h->h_out.p + h->o_G) std::memcpy(jac, h->h_out.p + h->o_G, Q.g_row.size() * sizeof(double));
    if (hess != h->h_out.p + h->o_H) std::memcpy(hess, h->h_out.p + h->o_H, Q.h_row.size() * sizeof(double));
    h->x_valid = h->fc_valid = h->small_synced = h->G_synced = true;
  });
}

int pc_host_buffers(pc_handle* h, double** x, double** lambda, double** g, double** jac, double** hess) {
  return guarded([&] {
    require_device(h);
    if (x) *x = h->h_in.p;
    if (lambda) *lambda = h->h_in.p + h->o_lam;
    if (g) *g = h->h_out.p + h->o_c;
    if (jac) *jac = h->h_out.p + h->o_G;
    if (hess) *hess = h->h_out.p + h->o_H;
  });
}

int pc_set_prefetch_jac(pc_handle* h, int on) {
  return guarded([&] {
    require_device(h);
    h->prefetch_jac = on != 0;
  });
}

int pc_set_host_mode(pc_handle* h, int mode) {
  return guarded([&] {
    require_device(h);
    if (mode < 0 || mode > 3) throw std::runtime_error("host mode must be 0..3");
    HIP_OK(hipStreamSynchronize(h->stream));
    h->host_mode = mode;
    h->x_valid = h->fc_valid = false;
  });
}

// J, grad J, c~ and G~ are produced together by the first callback at a new x and kept for the companion calls
// (IPOPT evaluates f, grad f, g, jac g at the same x, passing new_x = 1 to the first of them only;
// pycollo/nlp.py:47-57).  The small results [J | grad J | c~] and the large one, G~, come down as two copies so that
// eval_f / eval_g of a line-search trial point wait for the small one only.
static void ensure_fcG(pc_handle* h, const double* x, int new_x) {
  require_device(h);
  stage_x(h, x, new_x);
  if (h->fc_valid) return;
  launch_all(h, kx(h), nullptr, kout(h, h->o_c), kout(h, h->o_G), nullptr, kout(h, h->o_f), kout(h, h->o_gn),
             PC_FLAG_C | PC_FLAG_G, h->stream, 1.0);
  copy_down(h, 0, h->o_c + h->Q.num_c);
  HIP_OK(hipEventRecord(h->ev_small, h->stream));
  h->G_copied = h->prefetch_jac || (h->host_mode & 2);
  if (h->G_copied) {
    copy_down(h, h->o_G, h->o_G + h->Q.g_row.size());
    HIP_OK(hipEventRecord(h->ev_G, h->stream));
  }
  h->fc_valid = true;
  h->small_synced = h->G_synced = false;
}
static void wait_small(pc_handle* h) {
  if (h->small_synced) return;
  if (h->spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    while (hipEventQuery(h->ev_small) == hipErrorNotReady &&
           std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() <= h->spin_us) {}
  }
  HIP_OK(hipEventSynchronize(h->ev_small));
  check_timeout(h);
  h->small_synced = true;
}
static void wait_G(pc_handle* h) {
  if (h->G_synced) return;
  if (!h->G_copied) {   // G~ stayed on the device until somebody asked for it
    copy_down(h, h->o_G, h->o_G + h->Q.g_row.size());
    HIP_OK(hipEventRecord(h->ev_G, h->stream));
    h->G_copied = true;
  }
  if (h->spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    while (hipEventQuery(h->ev_G) == hipErrorNotReady &&
           std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() <= h->spin_us) {}
  }
  HIP_OK(hipEventSynchronize(h->ev_G));
  check_timeout(h);
  h->small_synced = h->G_synced = true;
}

int pc_eval_f(pc_handle* h, const double* x, int new_x, double* f) {
  return guarded([&] {
    ensure_fcG(h, x, new_x);
    wait_small(h);
    *f = h->h_out.p[h->o_f];
  });
}

int pc_eval_grad_f(pc_handle* h, const double* x, int new_x, double* grad) {
  return guarded([&] {
    ensure_fcG(h, x, new_x);
    wait_small(h);
    auto& Q = h->Q;
    std::memset(grad, 0, Q.num_x * sizeof(double));
    for (size_t e = 0; e < Q.jgrad_col.size(); ++e) grad[Q.point_x[Q.jgrad_col[e]]] = h->h_out.p[h->o_gn + e];
  });
}

int pc_eval_g(pc_handle* h, const double* x, int new_x, double* g) {
  return guarded([&] {
    ensure_fcG(h, x, new_x);
    wait_small(h);
    if (g != h->h_out.p + h->o_c) std::memcpy(g, h->h_out.p + h->o_c, h->Q.num_c * sizeof(double));
  });
}

int pc_eval_jac_g(pc_handle* h, const double* x, int new_x, double* values) {
  return guarded([&] {
    ensure_fcG(h, x, new_x);
    wait_G(h);
    if (values != h->h_out.p + h->o_G) std::memcpy(values, h->h_out.p + h->o_G, h->Q.g_row.size() * sizeof(double));
  });
}

int pc_eval_h(pc_handle* h, const double* x, int new_x, double obj_factor, const double* lambda, int new_lambda,
              double* values) {
  (void)new_lambda;   // sigma and lambda are cheap to restage; H~ is always re-evaluated
  return guarded([&] {
    require_device(h);
    auto& Q = h->Q;
    stage_x(h, x, new_x);
    stage_lambda(h, lambda);
    launch_all(h, kx(h), klam(h), nullptr, nullptr, kout(h, h->o_H), kout(h, h->o_f), nullptr, PC_FLAG_H, h->stream,
               obj_factor);
    copy_down(h, h->o_H, h->o_H + Q.h_row.size());
    wait_stream(h);
    // the stream has drained: earlier copies are complete too -- G~'s only if it was ever issued (prefetch off and
    // kernels not writing host memory: G~ is still on the device and wait_G must fetch it)
    h->small_synced = h->fc_valid;
    h->G_synced = h->fc_valid && h->G_copied;
    if (values != h->h_out.p + h->o_H) std::memcpy(values, h->h_out.p + h->o_H, Q.h_row.size() * sizeof(double));
  });
}

int pc_eval_resident(pc_handle* h, const double* x, double obj_factor, const double* lambda, double* f, double* grad,
                     double* g) {
  return guarded([&] {
    require_device(h);
    auto& Q = h->Q;
    if (!x) throw std::runtime_error("null x");
    const int mode = h->host_mode;
    h->host_mode = 0;   // results stay in (and are read from) the device mirror
    try {
      std::memcpy(h->h_in.p, x, Q.num_x * sizeof(double));
      if (lambda) std::memcpy(h->h_in.p + h->o_lam, lambda, Q.num_c * sizeof(double));
      HIP_OK(hipMemcpyAsync(h->d_in.p, h->h_in.p, (lambda ? h->in_total : (size_t)Q.num_x) * sizeof(double),
                            hipMemcpyHostToDevice, h->stream));
      launch_all(h, kx(h), lambda ? klam(h) : nullptr, kout(h, h->o_c), kout(h, h->o_G), lambda ? kout(h, h->o_H) : nullptr,
                 kout(h, h->o_f), kout(h, h->o_gn), PC_FLAG_C | PC_FLAG_G | (lambda ? PC_FLAG_H : 0), h->stream, obj_factor);
      copy_down(h, 0, h->o_c + Q.num_c);
      wait_stream(h);
    } catch (...) {
      h->host_mode = mode;
      throw;
    }
    h->host_mode = mode;
    h->x_valid = h->fc_valid = false;   // (the callback cache describes the host copies; G~ was not copied)
    if (f) *f = h->h_out.p[h->o_f];
    if (grad) {
      std::memset(grad, 0, Q.num_x * sizeof(double));
      for (size_t e = 0; e < Q.jgrad_col.size(); ++e) grad[Q.point_x[Q.jgrad_col[e]]] = h->h_out.p[h->o_gn + e];
    }
    if (g) std::memcpy(g, h->h_out.p + h->o_c, Q.num_c * sizeof(double));
  });
}

int pc_device_results(pc_handle* h, const double** d_g, const double** d_jac, const double** d_hess) {
  return guarded([&] {
    require_device(h);
    if (d_g) *d_g = h->d_out.p + h->o_c;
    if (d_jac) *d_jac = h->d_out.p + h->o_G;
    if (d_hess) *d_hess = h->d_out.p + h->o_H;
  });
}

// ---- IPOPT's callback types (include/pycollo_amd.h) ------------------------------------------------------
static bool ipopt_sizes_ok(pc_handle* h, int n, int m, int64_t nele, int which) {
  if (!h) { set_err("null user_data"); return false; }
  const auto& Q = h->Q;
  const int64_t want = which == 1 ? (int64_t)Q.g_row.size() : (which == 2 ? (int64_t)Q.h_row.size() : -1);
  if (n != (int)Q.num_x || (m >= 0 && m != (int)Q.num_c) || (which && nele != want)) {
    set_err("IPOPT callback: n / m / number of non-zeros do not match the handle");
    return false;
  }
  return true;
}

int pc_ipopt_eval_f(int n, double* x, int new_x, double* obj_value, void* user_data) {
  pc_handle* h = static_cast<pc_handle*>(user_data);
  if (!ipopt_sizes_ok(h, n, -1, 0, 0)) return 0;
  return pc_eval_f(h, x, new_x, obj_value);
}

int pc_ipopt_eval_grad_f(int n, double* x, int new_x, double* grad_f, void* user_data) {
  pc_handle* h = static_cast<pc_handle*>(user_data);
  if (!ipopt_sizes_ok(h, n, -1, 0, 0)) return 0;
  return pc_eval_grad_f(h, x, new_x, grad_f);
}

int pc_ipopt_eval_g(int n, double* x, int new_x, int m, double* g, void* user_data) {
  pc_handle* h = static_cast<pc_handle*>(user_data);
  if (!ipopt_sizes_ok(h, n, m, 0, 0)) return 0;
  return pc_eval_g(h, x, new_x, g);
}

int pc_ipopt_eval_jac_g(int n, double* x, int new_x, int m, int nele_jac, int* iRow, int* jCol, double* values,
                        void* user_data) {
  pc_handle* h = static_cast<pc_handle*>(user_data);
  if (!ipopt_sizes_ok(h, n, m, nele_jac, 1)) return 0;
  if (!values) return pc_jac_structure(h, iRow, jCol);   // IPOPT's structure query (x may be NULL)
  return pc_eval_jac_g(h, x, new_x, values);
}

int pc_ipopt_eval_h(int n, double* x, int new_x, double obj_factor, int m, double* lambda, int new_lambda,
                    int nele_hess, int* iRow, int* jCol, double* values, void* user_data) {
  pc_handle* h = static_cast<pc_handle*>(user_data);
  if (!ipopt_sizes_ok(h, n, m, nele_hess, 2)) return 0;
  if (!values) return pc_hess_structure(h, iRow, jCol);
  return pc_eval_h(h, x, new_x, obj_factor, lambda, new_lambda, values);
}

int pc_row_norms_jac(pc_handle* h, const double* x, double* norms) {
  return guarded([&] {
    require_device(h);
    // always through the device mirror: the row reduction reads G~ back
    const int mode = h->host_mode;
    h->host_mode = 0;
    h->x_valid = h->fc_valid = false;
    try {
      ensure_fcG(h, x, 1);
    } catch (...) {
      h->host_mode = mode;
      throw;
    }
    h->host_mode = mode;
    const int64_t m = h->Q.num_c;
    const int64_t threads = m * 64;
    const int blocks = (int)((threads + 255) / 256);
    hipLaunchKernelGGL(row_norms_kernel, dim3(blocks), dim3(256), 0, h->stream, h->d_g_indptr.p, h->d_out.p + h->o_G,
                       h->d_norms.p, m);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(h->h_norms.p, h->d_norms.p, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    std::memcpy(norms, h->h_norms.p, m * sizeof(double));
    h->x_valid = h->fc_valid = (mode == 0);   // the cache describes the device mirror only
    h->small_synced = h->fc_valid;
    h->G_synced = h->fc_valid && h->G_copied;   // (see pc_eval_h)
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
    HIP_OK(find_fn(h, &fn, name.c_str()));
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
    // (always through the device mirror; the callback cache no longer describes it afterwards)
    std::memcpy(h->h_in.p, x, Q.num_x * sizeof(double));
    HIP_OK(hipMemcpyAsync(h->d_in.p, h->h_in.p, Q.num_x * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->x_valid = h->fc_valid = false;
    a.x = h->d_in.p;
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
    check_timeout(h);
  });
}

void* pc_stream(pc_handle* h) { return h ? (void*)h->stream : nullptr; }

int pc_read_symbol(pc_handle* h, const char* name, void* dst, size_t bytes) {
  return guarded([&] {
    require_device(h);
    if (!name || !dst) throw std::runtime_error("null symbol name or destination");
    hipDeviceptr_t p = nullptr;
    size_t sz = 0;
    // "symbol@k": the copy in part k of a code object that comes in parts (every part has its own device globals)
    std::string sym(name);
    hipModule_t mod = h->module;
    if (const size_t at = sym.find('@'); at != std::string::npos) {
      const int k = std::atoi(sym.c_str() + at + 1);
      sym.resize(at);
      if (k < 0 || k > (int)h->more_modules.size()) throw std::runtime_error("no such part of the code object");
      if (k > 0) mod = h->more_modules[k - 1];
    }
    HIP_OK(hipModuleGetGlobal(&p, &sz, mod, sym.c_str()));
    if (bytes > sz) throw std::runtime_error("symbol is smaller than the requested read");
    HIP_OK(hipStreamSynchronize(h->stream));
    HIP_OK(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
  });
}

}  // extern "C"

#include "pc_ipm.hpp"   // the device-resident interior-point state (same translation unit: it uses launch_all)
