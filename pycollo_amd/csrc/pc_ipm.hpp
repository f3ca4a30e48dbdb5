// Device-resident state of one interior-point solve: the iterate, its
// multipliers, the step and every right-hand side live in device memory; one C call per part of an iteration -- error
// measures, Newton step (Hessian evaluation, KKT assembly, factorisation with the inertia loop, refined solve,
// fraction-to-the-boundary rule), a line-search trial point, acceptance -- and only a handful of scalars come back.
// The algorithm is pycollo_amd/ipm.py's (the filter logic, the barrier update and the rare restoration branch stay
// there, on those scalars); the arithmetic per vector entry is the same expression as its NumPy statement.
//
// Boundary being matched: the reference crosses from Python into its NLP solver once per solve
// (pycollo/backend.py:1807-1827, ca.nlpsol "ipopt"; pycollo/nlp.py:84-115), IPOPT's algorithm then calls the
// callbacks and its linear solver (backend.py:1703-1711) without leaving native code.
//
// Included at the end of pc_engine.hip (same translation unit: it launches the evaluation through launch_all and reads
// the handle's result blocks); the KKT side goes through the exported pc_kkt_*_device calls.
#pragma once

struct pc_ipm {
  pc_handle* h = nullptr;
  pc_kkt* k = nullptr;
  int64_t n = 0, m = 0, ns = 0, nv = 0, nu = 0, ngj = 0;
  double sf = 1.0;
  DevBuf<double> v, lam, zl, zu, vl, vu, sc, rhs_c, g, c, ct, vt, sol, dzl, dzu, Sigma, gphi, dvec, dvec_true, rhs, jtl, lams, mvx,
      gradnz, ft, part, red, csoc, sol0;   // csoc / sol0: second-order correction (its constraint values, the Newton step kept aside)
  DevBuf<uint8_t> hasl, hasu, fixed;
  DevBuf<int32_t> slack_of_row;   // [m] slack index of an inequality row, -1 for an equality row
  DevBuf<int64_t> gcol;           // [ngj] x index of every structural non-zero of grad J
  DevBuf<unsigned> counter;
  PinBuf<double> h_red;
};

namespace {

constexpr int IPM_NRED = 16;      // scalars a reduction kernel may return
constexpr int IPM_BLOCKS = 128;   // workgroups of a reduction (<= 256: the last block combines the partials one per thread, fixed tree)
enum { IPM_SUM = 0, IPM_MAX = 1, IPM_MIN = 2 };

// Block reduction of K per-thread values (op per slot), then the last block to arrive combines the blocks' partials
// in block order -- a fixed order whatever the scheduling -- and writes out[K].
template <int K>
__device__ __forceinline__ void ipm_reduce(double (&val)[K], const int (&op)[K], double* part, unsigned* counter, double* out) {
  __shared__ double sh[K][256];
  __shared__ bool last;
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < K; ++q) sh[q][tid] = val[q];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const double a = sh[q][tid], b = sh[q][tid + s];
        sh[q][tid] = op[q] == IPM_SUM ? a + b : (op[q] == IPM_MAX ? (b > a || b != b ? b : a) : (b < a || b != b ? b : a));
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
#pragma unroll
    for (int q = 0; q < K; ++q) part[(size_t)blockIdx.x * K + q] = sh[q][0];
    __threadfence();
    last = atomicAdd(counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last) {
    // The last block combines the blocks' partials with the same tree, thread b holding block b's (gridDim.x <= 256): a
    // fixed shape, so the same bits every run.  (One thread walking K x gridDim.x partials in a row, as this was first
    // written, took 65-75 us of a 1.08 ms interior-point iteration at config 2 -- twice per iteration.)
    __threadfence();
#pragma unroll
    for (int q = 0; q < K; ++q)
      sh[q][tid] = tid < (int)gridDim.x ? part[(size_t)tid * K + q] : (op[q] == IPM_SUM ? 0.0 : (op[q] == IPM_MAX ? -INFINITY : INFINITY));
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const double a = sh[q][tid], b = sh[q][tid + s];
          sh[q][tid] = op[q] == IPM_SUM ? a + b : (op[q] == IPM_MAX ? (b > a || b != b ? b : a) : (b < a || b != b ? b : a));
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
#pragma unroll
      for (int q = 0; q < K; ++q) out[q] = sh[q][0];
      *counter = 0;
    }
  }
}

// c = sc (c_raw - rhs_c) - [slack of the row]   (ipm.py::_c)
__global__ void ipm_scale_c(const double* __restrict__ craw, const double* __restrict__ sc, const double* __restrict__ rhs_c,
                            const int32_t* __restrict__ slack_of_row, const double* __restrict__ v, int64_t n, double* __restrict__ c, int64_t m) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    double x = sc[i] * (craw[i] - rhs_c[i]);
    const int32_t s = slack_of_row[i];
    if (s >= 0) x -= v[n + s];
    c[i] = x;
  }
}

// g = [sf grad J ; 0]: zero fill, then the structural non-zeros
__global__ void ipm_zero(double* __restrict__ a, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] = 0.0;
}
__global__ void ipm_scatter_grad(const double* __restrict__ gradnz, const int64_t* __restrict__ col, double sf, double* __restrict__ g, int64_t ngj) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < ngj) g[col[e]] = sf * gradnz[e];
}
__global__ void ipm_mul(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}
// mvx = [0 (nv) ; lam (m)]: the vector whose product with the KKT matrix is [J^T lam ; ...]  (ipm.py::_JT)
__global__ void ipm_pad_lam(const double* __restrict__ lam, double* __restrict__ out, int64_t nv, int64_t m) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv + m; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = i < nv ? 0.0 : lam[i - nv];
}

// error measures of ipm.py::errors, as scalars from which E(mu) follows for any mu:
//   0 max |g + J^T lam - zl + zu| over the free unknowns   1 max |c|   2 theta = sum |c|
//   3 / 4 max / min of (v - vl) zl over hasl   5 / 6 max / min of (vu - v) zu over hasu
//   7 sum |lam|   8 sum zl   9 sum zu
__global__ void __launch_bounds__(256) ipm_errors_kernel(const double* __restrict__ g, const double* __restrict__ jtl, const double* __restrict__ zl,
                                                         const double* __restrict__ zu, const double* __restrict__ v, const double* __restrict__ vl,
                                                         const double* __restrict__ vu, const uint8_t* __restrict__ hasl, const uint8_t* __restrict__ hasu,
                                                         const uint8_t* __restrict__ fixed, const double* __restrict__ c, const double* __restrict__ lam,
                                                         int64_t nv, int64_t m, double* part, unsigned* counter, double* out) {
  double val[10] = {0.0, 0.0, 0.0, -INFINITY, INFINITY, -INFINITY, INFINITY, 0.0, 0.0, 0.0};
  const int op[10] = {IPM_MAX, IPM_MAX, IPM_SUM, IPM_MAX, IPM_MIN, IPM_MAX, IPM_MIN, IPM_SUM, IPM_SUM, IPM_SUM};
  auto mx = [](double a, double b) { return (b > a || b != b) ? b : a; };
  auto mn = [](double a, double b) { return (b < a || b != b) ? b : a; };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    if (!fixed[i]) val[0] = mx(val[0], fabs(g[i] + jtl[i] - zl[i] + zu[i]));
    if (hasl[i]) {
      const double s = (v[i] - vl[i]) * zl[i];
      val[3] = mx(val[3], s);
      val[4] = mn(val[4], s);
    }
    if (hasu[i]) {
      const double s = (vu[i] - v[i]) * zu[i];
      val[5] = mx(val[5], s);
      val[6] = mn(val[6], s);
    }
    val[8] += zl[i];
    val[9] += zu[i];
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const double a = fabs(c[i]);
    val[1] = mx(val[1], a);
    val[2] += a;
    val[7] += fabs(lam[i]);
  }
  ipm_reduce<10>(val, op, part, counter, out);
}

// Sigma, grad phi_mu, the right-hand side and the diagonal of the Newton system (ipm.py main loop, "Newton step")
__global__ void ipm_newton_setup(const double* __restrict__ v, const double* __restrict__ vl, const double* __restrict__ vu,
                                 const double* __restrict__ zl, const double* __restrict__ zu, const uint8_t* __restrict__ hasl,
                                 const uint8_t* __restrict__ hasu, const uint8_t* __restrict__ fixed, const double* __restrict__ g,
                                 const double* __restrict__ jtl, const double* __restrict__ c, double mu, double* __restrict__ Sigma,
                                 double* __restrict__ gphi, double* __restrict__ rhs, int64_t nv, int64_t m) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv + m; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nv) {
      const double dlv = hasl[i] ? v[i] - vl[i] : 1.0, duv = hasu[i] ? vu[i] - v[i] : 1.0;
      Sigma[i] = (hasl[i] ? zl[i] / dlv : 0.0) + (hasu[i] ? zu[i] / duv : 0.0);
      const double gp = g[i] - (hasl[i] ? mu / dlv : 0.0) + (hasu[i] ? mu / duv : 0.0);
      gphi[i] = gp;
      rhs[i] = fixed[i] ? 0.0 : -(gp + jtl[i]);
    } else {
      rhs[i] = -c[i - nv];
    }
  }
}
__global__ void ipm_diag(const double* __restrict__ Sigma, double dw, double dc, double* __restrict__ dvec, int64_t nv, int64_t m) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv + m; i += (int64_t)gridDim.x * blockDim.x)
    dvec[i] = i < nv ? Sigma[i] + dw : -dc;
}

// after the solve: the step of the bound multipliers, the fraction-to-the-boundary limits, grad phi . dv and the
// barrier sum at v:   0 a_max (primal)   1 a_zl   2 a_zu   3 dphi   4 -sum log(v - vl) - sum log(vu - v)   5 non-finite entries of the step
__global__ void __launch_bounds__(256) ipm_step_kernel(double* __restrict__ sol, const double* __restrict__ v, const double* __restrict__ vl,
                                                       const double* __restrict__ vu, const double* __restrict__ zl, const double* __restrict__ zu,
                                                       const uint8_t* __restrict__ hasl, const uint8_t* __restrict__ hasu, const uint8_t* __restrict__ fixed,
                                                       const double* __restrict__ gphi, double mu, double tau, double* __restrict__ dzl,
                                                       double* __restrict__ dzu, int64_t nv, int64_t m, double* part, unsigned* counter, double* out) {
  double val[6] = {1.0, 1.0, 1.0, 0.0, 0.0, 0.0};
  const int op[6] = {IPM_MIN, IPM_MIN, IPM_MIN, IPM_SUM, IPM_SUM, IPM_SUM};
  auto mn = [](double a, double b) { return (b < a || b != b) ? b : a; };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv + m; i += (int64_t)gridDim.x * blockDim.x) {
    if (i >= nv) {
      if (!isfinite(sol[i])) val[5] += 1.0;
      continue;
    }
    double dv = fixed[i] ? 0.0 : sol[i];
    if (!isfinite(dv)) val[5] += 1.0;
    sol[i] = dv;
    const double dl = v[i] - vl[i], du = vu[i] - v[i];
    const double dlv = hasl[i] ? dl : 1.0, duv = hasu[i] ? du : 1.0;
    const double a = hasl[i] ? mu / dlv - zl[i] - zl[i] / dlv * dv : 0.0;
    const double b = hasu[i] ? mu / duv - zu[i] + zu[i] / duv * dv : 0.0;
    dzl[i] = a;
    dzu[i] = b;
    if (hasl[i] && dv < 0) val[0] = mn(val[0], -tau * dl / dv);
    if (hasu[i] && dv > 0) val[0] = mn(val[0], tau * du / dv);
    if (hasl[i] && a < 0) val[1] = mn(val[1], -tau * zl[i] / a);
    if (hasu[i] && b < 0) val[2] = mn(val[2], -tau * zu[i] / b);
    val[3] += gphi[i] * dv;
    if (hasl[i]) val[4] -= log(dl);
    if (hasu[i]) val[4] -= log(du);
  }
  ipm_reduce<6>(val, op, part, counter, out);
}

// trial point vt = v + alpha dv and its barrier sum
__global__ void __launch_bounds__(256) ipm_trial_point(const double* __restrict__ v, const double* __restrict__ dv, double alpha, const double* __restrict__ vl,
                                                       const double* __restrict__ vu, const uint8_t* __restrict__ hasl, const uint8_t* __restrict__ hasu,
                                                       double* __restrict__ vt, int64_t nv, double* part, unsigned* counter, double* out) {
  double val[1] = {0.0};
  const int op[1] = {IPM_SUM};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const double x = v[i] + alpha * dv[i];
    vt[i] = x;
    if (hasl[i]) val[0] -= log(x - vl[i]);
    if (hasu[i]) val[0] -= log(vu[i] - x);
  }
  ipm_reduce<1>(val, op, part, counter, out);
}
// second-order correction: c_soc = alpha c_prev + c(trial) (c_prev: c at the current point the first time, c_soc after), and the
// constraint part of the right-hand side = -c_soc
__global__ void ipm_soc_rhs(const double* __restrict__ c, const double* __restrict__ ct, double* __restrict__ csoc, double alpha, int first,
                            double* __restrict__ rhs_c, int64_t m) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = alpha * (first ? c[i] : csoc[i]) + ct[i];
    csoc[i] = v;
    rhs_c[i] = -v;
  }
}
__global__ void ipm_neg(const double* __restrict__ c, double* __restrict__ out, int64_t m) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) out[i] = -c[i];
}
// theta = sum |c|, max |c|
__global__ void __launch_bounds__(256) ipm_theta(const double* __restrict__ c, int64_t m, double* part, unsigned* counter, double* out) {
  double val[2] = {0.0, 0.0};
  const int op[2] = {IPM_SUM, IPM_MAX};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const double a = fabs(c[i]);
    val[0] += a;
    val[1] = (a > val[1] || a != a) ? a : val[1];
  }
  ipm_reduce<2>(val, op, part, counter, out);
}

// acceptance: v <- vt, lam += alpha dlam, z += a_z dz, the duals kept within a factor of their central-path values
__global__ void ipm_accept_kernel(double* __restrict__ v, const double* __restrict__ vt, double* __restrict__ lam, const double* __restrict__ sol,
                                  double* __restrict__ zl, double* __restrict__ zu, const double* __restrict__ dzl, const double* __restrict__ dzu,
                                  const double* __restrict__ vl, const double* __restrict__ vu, const uint8_t* __restrict__ hasl,
                                  const uint8_t* __restrict__ hasu, double alpha, double a_z, double mu, int64_t nv, int64_t m) {
  const double ks = 1e10;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv + m; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nv) {
      const double x = vt[i];
      v[i] = x;
      double a = zl[i] + a_z * dzl[i], b = zu[i] + a_z * dzu[i];
      if (hasl[i]) {
        const double d = x - vl[i], lo = mu / (ks * d), hi = ks * mu / d;
        a = a < lo ? lo : (a > hi ? hi : a);      // np.clip
      } else a = 0.0;
      if (hasu[i]) {
        const double d = vu[i] - x, lo = mu / (ks * d), hi = ks * mu / d;
        b = b < lo ? lo : (b > hi ? hi : b);
      } else b = 0.0;
      zl[i] = a;
      zu[i] = b;
    } else {
      lam[i - nv] += alpha * sol[i];
    }
  }
}

inline unsigned ipm_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(IPM_BLOCKS, (n + 255) / 256)); }

// the scalars of the last reduction, on the host (one copy, one wait)
void ipm_fetch(pc_ipm* s, int count, double* out) {
  HIP_OK(hipMemcpyAsync(s->h_red.p, s->red.p, count * sizeof(double), hipMemcpyDeviceToHost, s->h->stream));
  wait_stream(s->h);   // (polls before it blocks; reports a resident tail that gave up)
  for (int i = 0; i < count; ++i) out[i] = s->h_red.p[i];
}

// J, grad J, c~, G~ at the point in `x` (device), into the handle's result blocks; c (scaled, slacks subtracted) and g follow
void ipm_eval_point(pc_ipm* s, const double* d_v, double* d_c_scaled, bool with_jac) {
  pc_handle* h = s->h;
  hipStream_t st = h->stream;
  h->fc_valid = h->small_synced = h->G_synced = false;
  launch_all(h, d_v, nullptr, h->d_out.p + h->o_c, h->d_out.p + h->o_G, nullptr, s->ft.p, with_jac ? s->gradnz.p : nullptr,
             with_jac ? (PC_FLAG_C | PC_FLAG_G) : PC_FLAG_C, st, 1.0);
  hipLaunchKernelGGL(ipm_scale_c, dim3(ipm_grid(s->m)), dim3(256), 0, st, h->d_out.p + h->o_c, s->sc.p, s->rhs_c.p, s->slack_of_row.p, d_v,
                     s->n, d_c_scaled, s->m);
  if (with_jac) {
    hipLaunchKernelGGL(ipm_zero, dim3(ipm_grid(s->nv)), dim3(256), 0, st, s->g.p, s->nv);
    if (s->ngj) hipLaunchKernelGGL(ipm_scatter_grad, dim3((unsigned)((s->ngj + 255) / 256)), dim3(256), 0, st, s->gradnz.p, s->gcol.p, s->sf, s->g.p, s->ngj);
  }
  HIP_OK(hipGetLastError());
}

}  // namespace

extern "C" {

int pc_ipm_create(pc_handle* h, pc_kkt* k, const pc_ipm_desc* d, pc_ipm** out) {
  if (out) *out = nullptr;
  std::unique_ptr<pc_ipm> s;
  const int ok = guarded([&] {
    require_device(h);
    if (!k || !d || !out) throw std::runtime_error("null argument");
    if (d->n != h->Q.num_x || d->m != h->Q.num_c) throw std::runtime_error("pc_ipm_create: sizes differ from the NLP's");
    s.reset(new pc_ipm());
    s->h = h;
    s->k = k;
    s->n = d->n; s->m = d->m; s->ns = d->ns; s->nv = d->n + d->ns; s->nu = s->nv + s->m;
    s->sf = d->obj_scale;
    const size_t nv = (size_t)s->nv, m = (size_t)s->m, nu = (size_t)s->nu;
    auto up = [](DevBuf<double>& b, const double* src, size_t n) { b.upload(std::vector<double>(src, src + n)); };
    up(s->vl, d->vl, nv); up(s->vu, d->vu, nv); up(s->sc, d->row_scale, m); up(s->rhs_c, d->rhs_c, m);
    s->hasl.upload(std::vector<uint8_t>(d->hasl, d->hasl + nv));
    s->hasu.upload(std::vector<uint8_t>(d->hasu, d->hasu + nv));
    s->fixed.upload(std::vector<uint8_t>(d->fixed, d->fixed + nv));
    std::vector<int32_t> sor(m, -1);
    for (int64_t i = 0; i < d->ns; ++i) {
      if (d->ineq_rows[i] < 0 || d->ineq_rows[i] >= d->m) throw std::runtime_error("pc_ipm_create: inequality row out of range");
      sor[(size_t)d->ineq_rows[i]] = (int32_t)i;
    }
    s->slack_of_row.upload(sor);
    std::vector<int64_t> gcol;
    for (size_t e = 0; e < h->Q.jgrad_col.size(); ++e) gcol.push_back(h->Q.point_x[h->Q.jgrad_col[e]]);
    s->ngj = (int64_t)gcol.size();
    if (gcol.empty()) gcol.push_back(0);
    s->gcol.upload(gcol);
    for (DevBuf<double>* b : {&s->v, &s->zl, &s->zu, &s->g, &s->vt, &s->dzl, &s->dzu, &s->Sigma, &s->gphi}) b->alloc(nv);
    for (DevBuf<double>* b : {&s->lam, &s->c, &s->ct, &s->lams, &s->csoc}) b->alloc(m);
    for (DevBuf<double>* b : {&s->sol, &s->dvec, &s->dvec_true, &s->rhs, &s->jtl, &s->mvx, &s->sol0}) b->alloc(nu);
    s->gradnz.alloc(gcol.size());
    s->ft.alloc(2);
    s->part.alloc((size_t)IPM_BLOCKS * IPM_NRED);
    s->red.alloc(IPM_NRED);
    s->counter.upload(std::vector<unsigned>(1, 0u));
    s->h_red.alloc(IPM_NRED);
    HIP_OK(hipMemset(s->g.p, 0, nv * sizeof(double)));
    // one stream for the evaluation, the vector kernels and the linear algebra: no cross-stream waits inside an iteration
    if (!pc_kkt_set_stream(k, (void*)h->stream)) throw std::runtime_error(pc_kkt_last_error());
  });
  if (!ok) return 0;
  *out = s.release();
  return 1;
}

void pc_ipm_destroy(pc_ipm* s) {
  if (!s) return;
  if (s->h && s->h->device >= 0) {
    (void)hipSetDevice(s->h->device);
    (void)hipStreamSynchronize(s->h->stream);
  }
  delete s;
}

int pc_ipm_set_state(pc_ipm* s, const double* v, const double* lam, const double* zl, const double* zu) {
  return guarded([&] {
    if (!s || !v || !lam || !zl || !zu) throw std::runtime_error("null argument");
    require_device(s->h);
    HIP_OK(hipStreamSynchronize(s->h->stream));
    HIP_OK(hipMemcpy(s->v.p, v, s->nv * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(s->lam.p, lam, s->m * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(s->zl.p, zl, s->nv * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(s->zu.p, zu, s->nv * sizeof(double), hipMemcpyHostToDevice));
  });
}

int pc_ipm_get_state(pc_ipm* s, double* v, double* lam, double* zl, double* zu, double* c, double* g) {
  return guarded([&] {
    if (!s) throw std::runtime_error("null argument");
    require_device(s->h);
    HIP_OK(hipStreamSynchronize(s->h->stream));
    if (v) HIP_OK(hipMemcpy(v, s->v.p, s->nv * sizeof(double), hipMemcpyDeviceToHost));
    if (lam) HIP_OK(hipMemcpy(lam, s->lam.p, s->m * sizeof(double), hipMemcpyDeviceToHost));
    if (zl) HIP_OK(hipMemcpy(zl, s->zl.p, s->nv * sizeof(double), hipMemcpyDeviceToHost));
    if (zu) HIP_OK(hipMemcpy(zu, s->zu.p, s->nv * sizeof(double), hipMemcpyDeviceToHost));
    if (c) HIP_OK(hipMemcpy(c, s->c.p, s->m * sizeof(double), hipMemcpyDeviceToHost));
    if (g) HIP_OK(hipMemcpy(g, s->g.p, s->nv * sizeof(double), hipMemcpyDeviceToHost));
  });
}

// f (scaled by obj_scale), theta = sum |c|, max |c| at the current point; J, grad J, c~ and G~ are evaluated there
int pc_ipm_eval_point(pc_ipm* s, double* out3) {
  return guarded([&] {
    if (!s || !out3) throw std::runtime_error("null argument");
    require_device(s->h);
    ipm_eval_point(s, s->v.p, s->c.p, true);
    hipLaunchKernelGGL(ipm_theta, dim3(ipm_grid(s->m)), dim3(256), 0, s->h->stream, s->c.p, s->m, s->part.p, s->counter.p, s->red.p);
    HIP_OK(hipMemcpyAsync(s->red.p + 2, s->ft.p, sizeof(double), hipMemcpyDeviceToDevice, s->h->stream));
    double r[3];
    ipm_fetch(s, 3, r);
    out3[0] = s->sf * r[2];
    out3[1] = r[0];
    out3[2] = r[1];
  });
}

// the ten scalars of ipm_errors_kernel at the current point (J^T lambda through the KKT product, as ipm.py::_JT)
int pc_ipm_errors(pc_ipm* s, double* out10) {
  return guarded([&] {
    if (!s || !out10) throw std::runtime_error("null argument");
    require_device(s->h);
    hipStream_t st = s->h->stream;
    hipLaunchKernelGGL(ipm_pad_lam, dim3(ipm_grid(s->nu)), dim3(256), 0, st, s->lam.p, s->mvx.p, s->nv, s->m);
    hipLaunchKernelGGL(ipm_zero, dim3(ipm_grid(s->nu)), dim3(256), 0, st, s->dvec.p, s->nu);
    if (!pc_kkt_matvec_device(s->k, 0, s->dvec.p, s->mvx.p, s->jtl.p)) throw std::runtime_error(pc_kkt_last_error());
    hipLaunchKernelGGL(ipm_errors_kernel, dim3(ipm_grid(std::max(s->nv, s->m))), dim3(256), 0, st, s->g.p, s->jtl.p, s->zl.p, s->zu.p, s->v.p,
                       s->vl.p, s->vu.p, s->hasl.p, s->hasu.p, s->fixed.p, s->c.p, s->lam.p, s->nv, s->m, s->part.p, s->counter.p, s->red.p);
    HIP_OK(hipGetLastError());
    ipm_fetch(s, 10, out10);
  });
}

// The Newton step at the current point for barrier parameter mu (pc_ipm_errors must have run at this point: it leaves
// J^T lambda): Lagrangian Hessian at (x, obj_scale, row_scale . lambda), Sigma, right-hand side, factorisation with the
// inertia-correcting regularisation loop of ipm.py::_solve_kkt (IPOPT's delta_w schedule), refined solve, dz, limits.
// out8: 0 dw   1 a_max   2 a_z   3 grad phi . dv   4 phi_mu(v) - f (= mu x barrier sum)   5 factorisations   6 back-substitutions   7 non-finite
// Returns 1 with out8[0] < 0 when the regularisation failed (the caller ends the solve as ipm.py does).
int pc_ipm_newton(pc_ipm* s, double mu, double tau, double dw_last, double* out8) {
  return guarded([&] {
    if (!s || !out8) throw std::runtime_error("null argument");
    pc_handle* h = s->h;
    require_device(h);
    hipStream_t st = h->stream;
    const unsigned gu = ipm_grid(s->nu);
    hipLaunchKernelGGL(ipm_mul, dim3(ipm_grid(s->m)), dim3(256), 0, st, s->sc.p, s->lam.p, s->lams.p, s->m);
    h->fc_valid = h->small_synced = h->G_synced = false;
    launch_all(h, s->v.p, s->lams.p, h->d_out.p + h->o_c, h->d_out.p + h->o_G, h->d_out.p + h->o_H, s->ft.p + 1, nullptr, PC_FLAG_H, st, s->sf);
    hipLaunchKernelGGL(ipm_newton_setup, dim3(gu), dim3(256), 0, st, s->v.p, s->vl.p, s->vu.p, s->zl.p, s->zu.p, s->hasl.p, s->hasu.p,
                       s->fixed.p, s->g.p, s->jtl.p, s->c.p, mu, s->Sigma.p, s->gphi.p, s->rhs.p, s->nv, s->m);
    double dw = 0.0, dc = 0.0;
    int nfac = 0, nsol = 0;
    bool done = false;
    for (int attempt = 0; attempt < 40 && !done; ++attempt) {
      const double dc_eff = std::max(dc, 1e-9);
      hipLaunchKernelGGL(ipm_diag, dim3(gu), dim3(256), 0, st, s->Sigma.p, dw, dc_eff, s->dvec.p, s->nv, s->m);
      int32_t npos = 0, nneg = 0;
      if (!pc_kkt_factor_device(s->k, 1, s->dvec.p, &npos, &nneg)) throw std::runtime_error(pc_kkt_last_error());
      ++nfac;
      if (npos == s->nv && nneg == s->m) {   // fixed unknowns are unit pivots: counted with the primal ones
        hipLaunchKernelGGL(ipm_diag, dim3(gu), dim3(256), 0, st, s->Sigma.p, dw, dc, s->dvec_true.p, s->nv, s->m);
        int32_t ns_ = 0;
        if (!pc_kkt_solve_refined_device(s->k, 1, s->dvec_true.p, s->rhs.p, 3, s->sol.p, &ns_)) throw std::runtime_error(pc_kkt_last_error());
        nsol += ns_;
        hipLaunchKernelGGL(ipm_step_kernel, dim3(gu), dim3(256), 0, st, s->sol.p, s->v.p, s->vl.p, s->vu.p, s->zl.p, s->zu.p, s->hasl.p,
                           s->hasu.p, s->fixed.p, s->gphi.p, mu, tau, s->dzl.p, s->dzu.p, s->nv, s->m, s->part.p, s->counter.p, s->red.p);
        double r[6];
        ipm_fetch(s, 6, r);
        if (r[5] == 0.0) {
          out8[0] = dw; out8[1] = r[0]; out8[2] = std::min(r[1], r[2]); out8[3] = r[3]; out8[4] = mu * r[4];
          out8[5] = nfac; out8[6] = nsol; out8[7] = 0.0;
          done = true;
          break;
        }
      }
      if (dw >= 1e20) break;
      if (dw == 0.0) dw = dw_last == 0.0 ? 1e-4 : std::max(1e-20, dw_last / 3.0);
      else dw *= dw_last == 0.0 ? 100.0 : 8.0;
    }
    if (!done) {
      out8[0] = -1.0;
      out8[5] = nfac; out8[6] = nsol; out8[7] = 1.0;
    }
  });
}

// trial point v + alpha dv: f (scaled), theta, mu x barrier sum there
int pc_ipm_trial(pc_ipm* s, double alpha, double mu, double* out3) {
  return guarded([&] {
    if (!s || !out3) throw std::runtime_error("null argument");
    require_device(s->h);
    hipStream_t st = s->h->stream;
    hipLaunchKernelGGL(ipm_trial_point, dim3(ipm_grid(s->nv)), dim3(256), 0, st, s->v.p, s->sol.p, alpha, s->vl.p, s->vu.p, s->hasl.p, s->hasu.p,
                       s->vt.p, s->nv, s->part.p, s->counter.p, s->red.p + 4);
    ipm_eval_point(s, s->vt.p, s->ct.p, false);
    hipLaunchKernelGGL(ipm_theta, dim3(ipm_grid(s->m)), dim3(256), 0, st, s->ct.p, s->m, s->part.p, s->counter.p, s->red.p);
    HIP_OK(hipMemcpyAsync(s->red.p + 2, s->ft.p, sizeof(double), hipMemcpyDeviceToDevice, st));
    double r[5];
    ipm_fetch(s, 5, r);
    out3[0] = s->sf * r[2];
    out3[1] = r[0];
    out3[2] = mu * r[4];
  });
}

// Second-order correction (ipm.py, IPOPT A-5.7 .. A-5.9) after pc_ipm_trial rejected the step of size `alpha`: the same
// factorisation solved for the constraint values c_soc = alpha c + c(trial) (first != 0), or alpha c_soc + c(trial) of the
// previous corrected trial; the corrected step replaces the Newton step (kept aside: pc_ipm_soc_restore), the bound
// multipliers' step and the limits follow it.  out8 as pc_ipm_newton's (0: unused).
int pc_ipm_soc(pc_ipm* s, double alpha, int first, double mu, double tau, double* out8) {
  return guarded([&] {
    if (!s || !out8) throw std::runtime_error("null argument");
    require_device(s->h);
    hipStream_t st = s->h->stream;
    const unsigned gu = ipm_grid(s->nu);
    if (first) HIP_OK(hipMemcpyAsync(s->sol0.p, s->sol.p, (size_t)s->nu * sizeof(double), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(ipm_soc_rhs, dim3(ipm_grid(s->m)), dim3(256), 0, st, s->c.p, s->ct.p, s->csoc.p, alpha, first, s->rhs.p + s->nv, s->m);
    int32_t ns_ = 0;
    if (!pc_kkt_solve_refined_device(s->k, 1, s->dvec_true.p, s->rhs.p, 3, s->sol.p, &ns_)) throw std::runtime_error(pc_kkt_last_error());
    hipLaunchKernelGGL(ipm_step_kernel, dim3(gu), dim3(256), 0, st, s->sol.p, s->v.p, s->vl.p, s->vu.p, s->zl.p, s->zu.p, s->hasl.p,
                       s->hasu.p, s->fixed.p, s->gphi.p, mu, tau, s->dzl.p, s->dzu.p, s->nv, s->m, s->part.p, s->counter.p, s->red.p);
    double r[6];
    ipm_fetch(s, 6, r);
    out8[0] = 0.0; out8[1] = r[0]; out8[2] = std::min(r[1], r[2]); out8[3] = r[3]; out8[4] = mu * r[4];
    out8[5] = 0.0; out8[6] = ns_; out8[7] = r[5];
  });
}

// the corrections were rejected: the Newton step, its right-hand side and its bound-multiplier steps back in place
int pc_ipm_soc_restore(pc_ipm* s, double mu, double tau) {
  return guarded([&] {
    if (!s) throw std::runtime_error("null argument");
    require_device(s->h);
    hipStream_t st = s->h->stream;
    HIP_OK(hipMemcpyAsync(s->sol.p, s->sol0.p, (size_t)s->nu * sizeof(double), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(ipm_neg, dim3(ipm_grid(s->m)), dim3(256), 0, st, s->c.p, s->rhs.p + s->nv, s->m);
    hipLaunchKernelGGL(ipm_step_kernel, dim3(ipm_grid(s->nu)), dim3(256), 0, st, s->sol.p, s->v.p, s->vl.p, s->vu.p, s->zl.p, s->zu.p, s->hasl.p,
                       s->hasu.p, s->fixed.p, s->gphi.p, mu, tau, s->dzl.p, s->dzu.p, s->nv, s->m, s->part.p, s->counter.p, s->red.p);
    double r[6];
    ipm_fetch(s, 6, r);
  });
}

// accept the last trial point: v, lambda, z updated; grad J and G~ evaluated at the new point (c~ is the trial's)
int pc_ipm_accept(pc_ipm* s, double alpha, double a_z, double mu) {
  return guarded([&] {
    if (!s) throw std::runtime_error("null argument");
    require_device(s->h);
    hipStream_t st = s->h->stream;
    hipLaunchKernelGGL(ipm_accept_kernel, dim3(ipm_grid(s->nu)), dim3(256), 0, st, s->v.p, s->vt.p, s->lam.p, s->sol.p, s->zl.p, s->zu.p,
                       s->dzl.p, s->dzu.p, s->vl.p, s->vu.p, s->hasl.p, s->hasu.p, alpha, a_z, mu, s->nv, s->m);
    ipm_eval_point(s, s->v.p, s->c.p, true);
    HIP_OK(hipGetLastError());
  });
}

}  // extern "C"
