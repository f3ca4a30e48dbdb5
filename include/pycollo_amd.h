/*
 * pycollo_amd -- C ABI of the MI355X collocation NLP-callback engine.
 *
 * Drop-in boundary for pycollo's hot path.  The callback set mirrors IPOPT's
 * IpStdCInterface.h (Eval_F_CB, Eval_Grad_F_CB, Eval_G_CB, Eval_Jac_G_CB, Eval_H_CB) so a handle can
 * be driven by IPOPT with or without Python; each entry point cites the reference interface it
 * replaces.  All functions return 1 on success and 0 on failure (IPOPT convention);
 * pc_last_error() describes the last failure on the calling thread.  NaN/Inf are written through.
 *
 * Ownership: the caller owns every pointer it passes.  The library owns all device memory, one HIP
 * stream per handle and copies of the descriptor arrays.  Calls on one handle are not re-entrant
 * (IPOPT is serial); different handles may be used from different threads.
 *
 * Index arrays are 0-based int32 (IPOPT "C_STYLE"), CSR order: row-major, ascending columns.
 * The Hessian is the lower triangle.
 */
#ifndef PYCOLLO_AMD_H
#define PYCOLLO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pc_handle pc_handle;

/* kinds of point (endpoint) variables, pycollo/backend.py:1264-1269 */
enum { PC_PT_Y0 = 0, PC_PT_YF = 1, PC_PT_Q = 2, PC_PT_T0 = 3, PC_PT_TF = 4, PC_PT_S = 5 };

/* One phase of the optimal control problem on one mesh (pycollo/mesh.py:236-356 tables,
 * pycollo/backend.py:854-1305 phase data reduced to counts and structural masks). */
typedef struct {
  int32_t n_y, n_u, n_q, n_p;      /* needed states, controls, integrals; path constraints */
  int32_t t0_free, tF_free;        /* 1 if t0 / tF is an NLP variable (bounds.py:456-480) */
  double t0_fixed, tF_fixed;       /* value used when the time is not free (backend.py:1579-1586) */
  int32_t K;                       /* mesh sections */
  const int32_t* n_k;              /* [K] nodes per section, both ends included (mesh N_K) */
  const double* h_k;               /* [K] section widths in tau (mesh h_K), sum = 2 */
  /* structural non-zeros of d[f|p|g]/d[z|w], sorted by (row, col); rows: n_y+n_p+n_q, cols: n_z+n_w.
   * w = the parameters of the node functions, i.e. everything in f, p, g that is not a node variable: the live
   * reference keeps q, t0, tF and s as global symbols there (pycollo/backend.py:1526-1539,1565-1570).  With n_w = 0
   * (w_kind = NULL) the parameters are the problem's n_s static parameters. */
  int32_t n_jac;
  const int32_t* jac_row;
  const int32_t* jac_col;
  /* structural non-zeros of the node-Lagrangian Hessian in [z|w], lower triangle, sorted by (row, col) */
  int32_t n_hess;
  const int32_t* hess_row;
  const int32_t* hess_col;
  const char* bulk_kernel;         /* kernel symbol inside the code object */
  int32_t compiled_order;          /* 0: the kernel handles any mesh; n > 0: it was compiled for sections of
                                      exactly n nodes (every n_k must equal n) */
  int32_t n_edge_rec[2];           /* Hessian entries of node 0 / node N-1 an endpoint term lands on: what the phase's
                                      kernel was generated with (codegen.edge_flags); checked against the pattern */
  int32_t eval_ops;                /* arithmetic operations in the phase's node functions and their derivatives
                                      (0 = unknown); a launch-shape hint only: heavy models share a tile between
                                      fewer waves because every sharing wave re-evaluates the node functions */
  /* parameters of the node functions in x order: [integral variables f/p/g depend on | free times they depend on |
   * all static parameters]; w_kind[l] = 0 static parameter, 1 integral variable, 2 free time; w_idx[l] = index
   * within the kind (static: global index; integral: m; time: position among the phase's free times) */
  int32_t n_w;
  const int32_t* w_kind;
  const int32_t* w_idx;
  /* mixed build (compiled_order == 0 only): the section orders the code object's tile kernels of this phase carry a
   * specialised body for, next to the any-order body.  pc_create then cuts the tiles at order changes wherever the
   * orders come in runs (what ph refinement leaves, pycollo/mesh_refinement.py:252-321) and every order-pure tile runs
   * the body of its order.  n_spec = 0: the any-order kernel alone. */
  int32_t n_spec;
  int32_t spec_orders[4];
  int32_t reserved_spec[3];
  /* tile table given by the caller instead of cut by pc_create (n_fixed_tiles = 0: pc_create cuts): first section of
   * every tile [n_fixed_tiles + 1] (0 ... K, increasing) and, for a mixed build, the order whose body runs each tile
   * [n_fixed_tiles] (NULL: the any-order body everywhere).  What a rank-local handle of the section-sharded evaluation
   * is built with (pycollo_amd/sharding.py, LocalShard): the global tiling's tiles of the rank's range behind a halo tile,
   * so that every tile computes bit for bit what it computes in the whole mesh. */
  int32_t n_fixed_tiles;
  int32_t reserved_tiles;
  const int32_t* fixed_tile_k0;
  const int32_t* fixed_tile_order;
} pc_phase_desc;

typedef struct {
  int32_t n_phases;
  const pc_phase_desc* phases;
  int32_t n_s;                     /* needed static parameters */
  /* point variables in x_point_var order (pycollo/backend.py:658-661) */
  int32_t n_point;
  const int32_t* point_phase;      /* -1 for static parameters */
  const int32_t* point_kind;       /* PC_PT_* */
  const int32_t* point_idx;        /* index within its kind */
  int32_t n_b;                     /* endpoint constraints */
  int32_t n_jgrad;                 /* structural non-zeros of dJ/dxb */
  const int32_t* jgrad_col;
  int32_t n_bjac;                  /* of db/dxb, sorted by (row, col) */
  const int32_t* bjac_row;
  const int32_t* bjac_col;
  int32_t n_pthess;                /* of d2(sigma J + lam.b)/dxb2, lower triangle, sorted */
  const int32_t* pthess_row;
  const int32_t* pthess_col;
  /* quadrature tables as data (pycollo/quadrature.py:189-261): for every order n in `orders`,
   * packed back to back: A (n-1) x n row-major in quad_A, weights n in quad_w */
  int32_t n_orders;
  const int32_t* orders;
  const double* quad_A;
  const double* quad_w;
  const char* code_object;         /* path of the gfx950 code object; may be NULL when device < 0 */
  const char* tail_kernel;
  int32_t device;                  /* HIP device ordinal; -1 = structure only, no evaluation possible */
  int32_t threads_per_block;       /* 0 = choose from the mesh size; else 64, 128 or 256 */
  int32_t two_wave_occupancy;      /* waves per SIMD the compiler reported for the code object's two-wave launch kernel
                                      (pc_bulk_all_r_w2 / pc_bulk_p<i>_r_w2 of a model with a heavy phase; codegen's
                                      resource sidecar); 0 = none.  >= 2: pc_create shares every 64-node tile between
                                      two waves and sizes the tiles so that a CU holds eight of them.  A launch-shape
                                      hint only: results do not depend on it beyond summation order of the integrals */
  int32_t plan_only;               /* 1: stop after the tiles are cut -- no layout, no patterns, no device (pc_phase_tiles /
                                      pc_phase_tile_orders / pc_get_info's tile fields work; O(sections) memory) */
} pc_problem_desc;

typedef struct {
  int32_t n, m;
  int64_t nnz_jac, nnz_hess;
  int64_t algorithmic_bytes;       /* 8*(n+m) read + 8*(m+nnz_jac+nnz_hess) written per eval_all */
  int32_t n_tiles_total, threads_per_block;   /* threads_per_block = nodes per tile (plus the shared end node) */
  int32_t lds_bytes_max, n_launches; /* kernels per eval_all */
  int32_t waves_per_tile, reserved;  /* largest replica count of a phase (1, 2 or 4): workgroup = tile x this */
} pc_info;

const char* pc_last_error(void);

/* replaces: Casadi.generate_nlp_function_callables + create_nlp_solver (pycollo/backend.py:1403-1411,
 * 1681-1693) and the legacy CompiledFunctions constructor (pycollo/compiled.py:19-30) */
int pc_create(const pc_problem_desc* desc, pc_handle** out);
void pc_destroy(pc_handle* h);
int pc_get_info(const pc_handle* h, pc_info* info);

/* replaces: Iteration.num_x / num_c (pycollo/iteration.py:231-235,287-290),
 * Casadi.evaluate_G_num_nonzero (backend.py:1763-1771), evaluate_H_num_nonzero (:1799-1805) */
int pc_sizes(const pc_handle* h, int32_t* n, int32_t* m, int64_t* nnz_jac, int64_t* nnz_hess);
/* replaces: Casadi.evaluate_G_structure (backend.py:1747-1761), IPOPTProblem.jacobianstructure
 * (pycollo/nlp.py:56-57); CSR (row-major) order */
int pc_jac_structure(const pc_handle* h, int32_t* iRow, int32_t* jCol);
/* replaces: IPOPTProblem.hessianstructure (nlp.py:62-63), evaluate_H_structure (backend.py:1790-1797) */
int pc_hess_structure(const pc_handle* h, int32_t* iRow, int32_t* jCol);

/* replaces: the V/r/W/w substitutions of create_nlp_solver (backend.py:1459-1463,1684-1689);
 * vectors are per OCP variable / constraint (scaling.py:166-167,274) */
int pc_set_scaling(pc_handle* h, const double* V_ocp, const double* r_ocp, const double* W_ocp, double w_J);

/* new_x follows IPOPT's protocol (IpStdCInterface.h): 1 on the first callback at a point, 0 on the companion
 * calls at the same x, which reuse what the first one launched (J, grad J, g and jac_g are produced together).
 * replaces: nlp_f / IPOPTProblem.objective (backend.py:1713-1715, nlp.py:47-48) */
int pc_eval_f(pc_handle* h, const double* x, int new_x, double* f);
/* replaces: nlp_grad_f / IPOPTProblem.gradient (backend.py:1717-1720, nlp.py:50-51); dense n */
int pc_eval_grad_f(pc_handle* h, const double* x, int new_x, double* grad);
/* replaces: nlp_g / IPOPTProblem.constraints (backend.py:1722-1725, nlp.py:53-54); dense m */
int pc_eval_g(pc_handle* h, const double* x, int new_x, double* g);
/* replaces: nlp_jac_g / IPOPTProblem.jacobian (backend.py:1738-1745, nlp.py:56-57) */
int pc_eval_jac_g(pc_handle* h, const double* x, int new_x, double* values);
/* replaces: nlp_hess_l / IPOPTProblem.hessian (nlp.py:59-60, pycollo/iteration.py:1057) */
int pc_eval_h(pc_handle* h, const double* x, int new_x, double obj_factor, const double* lambda,
              int new_lambda, double* values);
/* fused g + jac_g + hess at one (x, sigma, lambda): the benchmarked call (host pointers) */
int pc_eval_all(pc_handle* h, const double* x, double obj_factor, const double* lambda, double* g,
                double* jac, double* hess);
/* ---- IPOPT's C callback set, with IPOPT's own signatures ----------------------------------------------
 * IpStdCInterface.h (third-party, coin-or/Ipopt; not part of the reference tree) declares
 *   typedef double Number; typedef int Index; typedef int Bool; typedef void* UserDataPtr;
 *   Bool (*Eval_F_CB)     (Index n, Number* x, Bool new_x, Number* obj_value, UserDataPtr user_data);
 *   Bool (*Eval_Grad_F_CB)(Index n, Number* x, Bool new_x, Number* grad_f, UserDataPtr user_data);
 *   Bool (*Eval_G_CB)     (Index n, Number* x, Bool new_x, Index m, Number* g, UserDataPtr user_data);
 *   Bool (*Eval_Jac_G_CB) (Index n, Number* x, Bool new_x, Index m, Index nele_jac, Index* iRow, Index* jCol,
 *                          Number* values, UserDataPtr user_data);
 *   Bool (*Eval_H_CB)     (Index n, Number* x, Bool new_x, Number obj_factor, Index m, Number* lambda,
 *                          Bool new_lambda, Index nele_hess, Index* iRow, Index* jCol, Number* values,
 *                          UserDataPtr user_data);
 * The functions below have exactly these types and can be handed to CreateIpoptProblem(n, x_L, x_U, m, g_L, g_U,
 * nele_jac, nele_hess, 0 (C-style indices), &pc_ipopt_eval_f, &pc_ipopt_eval_g, &pc_ipopt_eval_grad_f,
 * &pc_ipopt_eval_jac_g, &pc_ipopt_eval_h) with user_data = the pc_handle*.  values == NULL asks for the structure
 * (iRow / jCol, CSR order), as IPOPT does once before the first evaluation; a size that does not match the
 * handle's returns 0.  They replace what cyipopt builds around IPOPTProblem's methods
 * (pycollo/nlp.py:84-115: ipopt.problem(n, m, problem_obj, lb, ub, cl, cu)). */
int pc_ipopt_eval_f(int n, double* x, int new_x, double* obj_value, void* user_data);
int pc_ipopt_eval_grad_f(int n, double* x, int new_x, double* grad_f, void* user_data);
int pc_ipopt_eval_g(int n, double* x, int new_x, int m, double* g, void* user_data);
int pc_ipopt_eval_jac_g(int n, double* x, int new_x, int m, int nele_jac, int* iRow, int* jCol, double* values,
                        void* user_data);
int pc_ipopt_eval_h(int n, double* x, int new_x, double obj_factor, int m, double* lambda, int new_lambda,
                    int nele_hess, int* iRow, int* jCol, double* values, void* user_data);

/* The pinned staging blocks the host-pointer calls copy through.  A caller that writes x / lambda there and passes
 * these same pointers to pc_eval_* (inputs and/or outputs) skips the corresponding host-side memcpy: the results are
 * then read in place and stay valid until the next evaluation on the handle.  (The reference hands fresh numpy
 * arrays to cyipopt, pycollo/nlp.py:47-63; this is the zero-copy form of that hand-over.) */
int pc_host_buffers(pc_handle* h, double** x, double** lambda, double** g, double** jac, double** hess);
/* How the host-pointer calls move data: bit 0 = the kernels read x / lambda straight from the pinned host block
 * (no copy up), bit 1 = the kernels write their results straight into the pinned host block (no copy down).
 * 0 = one DMA copy up, one down (default; PYCOLLO_AMD_HOST_MODE overrides at pc_create). */
int pc_set_host_mode(pc_handle* h, int mode);
/* 1 (default): the first callback at a new point also starts the copy of jac_g to the host, which IPOPT asks for
 * next; 0: jac_g stays in device memory until pc_eval_jac_g is called (a solver whose linear algebra runs on the
 * GPU, pc_kkt_*, never asks) */
int pc_set_prefetch_jac(pc_handle* h, int on);
/* same with every vector resident in device memory; asynchronous on `stream` (hipStream_t, NULL =
 * the handle's stream).  No host synchronisation is performed.
 * AT MOST ONE evaluation per handle may be in flight: a handle owns one set of hand-over buffers (per-tile partial
 * sums, edge-node records) tagged with one launch counter, so evaluations of one handle must be queued on ONE stream
 * (or ordered by events); two streams running them concurrently overwrite each other's hand-over values and the
 * earlier evaluation's tail gives up after its bounded wait.  Use one handle per concurrent stream (IPOPT is serial:
 * pycollo/nlp.py:84-115 has one problem object per solve).  A tail that gave up is reported by the next
 * pc_eval_all_device, by pc_synchronize and by pc_check. */
int pc_eval_all_device(pc_handle* h, const double* d_x, double obj_factor, const double* d_lambda,
                       double* d_g, double* d_jac, double* d_hess, void* stream);
/* After the CALLER has synchronised its own stream: 1 if every evaluation queued so far completed, 0 (and
 * pc_last_error) if a resident tail gave up waiting for values of the same launch -- the results of that evaluation
 * are invalid.  (pc_synchronize does the same for the handle's own stream.)  No reference counterpart. */
int pc_check(pc_handle* h);
/* profiling aid: launches only the per-phase bulk kernels of pc_eval_all_device (no tail kernel),
 * so that bench.py can time the dominant kernel between two HIP events */
int pc_launch_bulk_device(pc_handle* h, const double* d_x, const double* d_lambda, double* d_g, double* d_jac,
                          double* d_hess, void* stream);
/* ---- multi-GPU sharding by contiguous mesh-section ranges ---------------------
 * A phase is cut into tiles of whole sections; a rank launches the bulk kernels over its tile range only
 * (pc_set_tile_range + pc_launch_bulk_device), the ranks exchange their output segments and per-tile
 * partial sums (one all-gather), and every rank finishes with pc_launch_tail_device over the complete
 * partials.  The reference has no counterpart (single process). */
int pc_set_tile_range(pc_handle* h, int phase, int tile_begin, int tile_end);
/* n_tiles, number of partial sums per tile, and (optional) first section of every tile [n_tiles+1] */
int pc_phase_tiles(const pc_handle* h, int phase, int32_t* n_tiles, int32_t* nred, int32_t* tile_k0);
/* mixed build (pc_phase_desc::n_spec > 0): the section order whose tile body runs every tile [n_tiles], 0 = the
 * any-order body; all zero for a phase that is not mixed */
int pc_phase_tile_orders(const pc_handle* h, int phase, int32_t* tile_order);
/* redirect the per-tile partial sums of one phase, double [n_tiles][nred], into caller-owned device
 * memory so that they can travel in the same all-gather as the output segments (NULL = internal) */
int pc_set_partials_buffer(pc_handle* h, int phase, double* d_partials);
/* the tile kernels for a subset of the outputs (flags: 1 = g rows, 2 = jac_g, 4 = hess; the tiles' partial sums are
 * produced by whichever launch computes their source: integrand sums with g, parameter sums with hess).  Lets a
 * sharded evaluation start the exchange of g / jac_g while the tiles of hess still run (pycollo_amd/sharding.py). */
int pc_launch_bulk_flags_device(pc_handle* h, const double* d_x, const double* d_lambda, double* d_g, double* d_jac,
                                double* d_hess, int flags, void* stream);
int pc_launch_tail_device(pc_handle* h, const double* d_x, double obj_factor, const double* d_lambda, double* d_g,
                          double* d_jac, double* d_hess, void* stream);
/* pc_launch_tail_device that also returns the objective f and grad_f[n] (host; either may be NULL) and synchronises
 * `stream`: the eval_f / eval_grad_f of a rank of a sharded solve (pycollo/nlp.py:47-56), whose own tiles never make a
 * whole evaluation.  obj_factor scales f's part of hess only, as in pc_eval_h. */
int pc_launch_tail_objective_device(pc_handle* h, const double* d_x, double obj_factor, const double* d_lambda, double* d_g,
                                    double* d_jac, double* d_hess, void* stream, double* f, double* grad);
int pc_synchronize(pc_handle* h);

/* replaces: the sparse row norms inside IterationScaling._calculate_constraint_scaling
 * (pycollo/scaling.py:392-395) without densifying G; evaluates G at x with the current scaling */
int pc_row_norms_jac(pc_handle* h, const double* x, double* norms);

/* replaces: interpolate_to_new_mesh (pycollo/iteration.py:96-137, scipy interp1d linear + extrapolate): carries
 * the rows of vals_prev[n_vars][n_prev] from the mesh tau_prev to tau_new.  Host pointers in and out. */
int pc_interp_linear(int device, const double* tau_prev, int n_prev, const double* vals_prev, int n_vars,
                     const double* tau_new, int n_new, double* out);

/* Multi-GPU exchange helper (no counterpart in the reference, which is single-process):
 * copy n_chunks contiguous runs between two device buffers on `stream`; d_chunks = [n_chunks][3] int64
 * (source offset, destination offset, length <= pc_run_chunk()), all in doubles.  Used to pack one rank's CSR
 * runs before the all-gather and to unpack the other ranks' runs after it. */
int pc_copy_runs(const double* d_src, double* d_dst, const int64_t* d_chunks, int64_t n_chunks, void* stream);
int pc_run_chunk(void);

/* replaces: PattersonRaoMeshRefinement.mesh_error / phase_mesh_error (pycollo/mesh_refinement.py:63-240) with the
 * section polynomial fits of pycollo/solution/solution_abc.py:60-107 folded into per-order tables:
 * for every order n in `orders`, packed back to back, tabB / tabE are (n-1) x n (integral / value of the Lagrange
 * basis on the n section nodes at the n-1 interior nodes of the order-(n+1) rule) and tabA is the n x (n+1)
 * integration matrix of order n+1.  Writes the section maxima of the relative error, max_rel[K], and of the
 * absolute error per state, max_abs[K][n_y] (may be NULL).  x is the scaled solution (host pointer). */
int pc_mesh_error(pc_handle* h, int phase, const double* x, int n_orders, const int32_t* orders, const double* tabB,
                  const double* tabE, const double* tabA, double* max_rel, double* max_abs);

/* ---- KKT solve on the GPU ---------------------------------------------------
 * Replaces the sparse symmetric-indefinite linear solver IPOPT calls once per iteration -- MUMPS, which the reference
 * selects by name only (``linear_solver``, pycollo/backend.py:1703-1711, pycollo/settings.py:49-59) -- for the
 * interior-point system  [[W + Sigma + dw I, J^T], [J, -dc I]]  of the collocation NLP.  W = H~ and J = G~ are read
 * where the evaluation left them in device memory (pc_eval_resident + pc_device_results): they never travel to the
 * host.  The matrix is eliminated block-wise along the mesh (section interiors in parallel, then the chain of section
 * boundary nodes, then the dense border of global unknowns) with 1 x 1 pivots in a fixed order; the signs of the pivots
 * are the inertia.  The index tables (pc_kkt_desc) are built by pycollo_amd/kkt.py, which documents the layout.
 * Unknowns, in every vector below: the nv primal ones (x~, then the slacks of the inequality rows), then the m
 * multipliers. */
typedef struct pc_kkt pc_kkt;
typedef struct {
  int64_t nu, nv, n_leaf, n_chain, n_phase, nb, total_vals, border_off, n_dst, n_src, n_mv;
  const int64_t *perm, *leaf_ptr, *chain_ptr, *chain_phase_ptr, *leaf_left;
  const int64_t *leafA_off, *leafS_off, *chainD_off, *chainS_off;
  const int64_t *dst, *run_ptr;       /* assembly: destination positions and their source runs */
  const int32_t *src_kind, *src_idx;  /* 0: G~ value, 1: H~ value, 2: the constant 1 */
  const double* src_coef;
  const int64_t* diag_pos;
  const uint8_t* fixed;
  const int64_t* mv_ptr;              /* the whole symmetric matrix as CSR over the unknowns (products) */
  const int32_t *mv_col, *mv_kind, *mv_idx;
  const double* mv_coef;
  const uint8_t* chain_export;        /* [n_chain] or NULL: 1 = the chain node is not eliminated and its assembled panel is
                                         handed out (pc_kkt_export_*): the nodes a rank shares with its neighbours in a
                                         factorisation cut across ranks; first / last node of a chain segment only */
} pc_kkt_desc;
/* Host-only helper of the table build (no device): value-buffer position of K[u[e], v[e]] for n pairs of natural
 * unknowns under the elimination plan (class 0 leaf / 1 chain / 2 border, block and local index per unknown; block
 * geometry per leaf and per chain node); -1 for a pair the plan keeps apart.  pycollo_amd/kkt.py::build_tables. */
typedef struct {
  int64_t nu, nb, border_off;
  const int8_t* cls;
  const int64_t *blk, *local;
  const int64_t *leafA_off, *m_l, *w_l, *leaf_left;        /* per leaf */
  const int64_t *chainD_off, *nzb, *nzb_next, *wc;          /* per chain node */
  const uint8_t* last_of_phase;
} pc_kkt_plan;
int pc_kkt_plan_positions(const pc_kkt_plan* plan, int64_t n, const int64_t* u, const int64_t* v, int64_t* out);
/* Host-only: the tables of the chain's cyclic reduction as pc_kkt_create builds them (csrc/pc_kkt_cr.hpp) -- per chain node
 * its separators (-1: none), the node whose Schur block holds its coupling to each (-1: neighbours in the chain), its level,
 * and in CSR form the eliminated nodes it is a separator of (node << 1 | 1 if it is that node's left separator; pull_e holds
 * pull_cap entries, 2 n_chain suffice).  chain_export as in pc_kkt_desc (may be NULL).  For the CPU tests, which hold the
 * tables against a symbolic elimination of the chain graph. */
int pc_kkt_cr_plan(int64_t n_chain, int64_t n_phase, const int64_t* chain_phase_ptr, const int64_t* chain_ptr, int64_t nb,
                   const uint8_t* chain_export, int64_t* cr_a, int64_t* cr_b, int64_t* mid_a, int64_t* mid_b, int32_t* level,
                   int64_t* pull_ptr, int32_t* pull_e, int64_t pull_cap);
/* Host-only: the entry tables of pc_kkt_desc (dst, run_ptr, src_*, mv_*) from the CSR structures of H~ (hr, hc; lower
 * triangle), G~ (jr, jc; rows scaled by row_scale[m]), the slack columns of the ns inequality rows ineq_rows and the fixed
 * flags [nu], under the plan.  n = number of NLP variables, nv = n + ns.  counts[3] receives (n_src, n_dst, n_mv).
 * Either call with dst == NULL for the counts alone and again with exact buffers, or once with buffers for the most
 * there can be: nH + nG + ns entries for dst / src_* (run_ptr one more), twice that for mv_col / mv_kind / mv_idx /
 * mv_coef, nu + 1 for mv_ptr. */
int pc_kkt_plan_entries(const pc_kkt_plan* plan, int64_t n, int64_t nv, int64_t nH, const int64_t* hr, const int64_t* hc,
                        int64_t nG, const int64_t* jr, const int64_t* jc, const double* row_scale, int64_t ns,
                        const int64_t* ineq_rows, const uint8_t* fixed, int64_t* counts, int64_t* dst, int64_t* run_ptr,
                        int32_t* src_kind, int32_t* src_idx, double* src_coef, int64_t* mv_ptr, int32_t* mv_col,
                        int32_t* mv_kind, int32_t* mv_idx, double* mv_coef);
const char* pc_kkt_last_error(void);
int pc_kkt_create(const pc_kkt_desc* desc, const double* d_jac, const double* d_hess, int device, pc_kkt** out);
void pc_kkt_destroy(pc_kkt* k);
/* assemble from the current device G~ / H~ (use_hess = 0: W = 0, e.g. least-squares multipliers) and dvec[nu]
 * (host: Sigma + dw on primal unknowns, -dc on multipliers), factorise; returns the pivot signs */
int pc_kkt_factor(pc_kkt* k, int use_hess, const double* dvec, int32_t* n_pos, int32_t* n_neg);
int pc_kkt_solve(pc_kkt* k, const double* rhs, double* x);                       /* host vectors [nu] */
int pc_kkt_matvec(pc_kkt* k, int use_hess, const double* dvec, const double* x, double* y);   /* y = K x */
/* x = K^-1 rhs with the current factors, iteratively refined ON THE DEVICE against the system with dvec_true on its
 * diagonal (the factors may carry a slightly different diagonal): solve, residual, up to max_steps corrections, a
 * correction kept only while it halves the residual 2-norm and stays finite.  One call per linear step of the
 * interior-point method -- what IPOPT's linear-solver interface does inside IpPDFullSpaceSolver (the reference only
 * names the solver, pycollo/backend.py:1703-1711).  n_solves (may be NULL): back-substitutions performed. */
int pc_kkt_solve_refined(pc_kkt* k, int use_hess, const double* dvec_true, const double* rhs, int max_steps, double* x,
                         int32_t* n_solves);
/* The same three with every vector in device memory, queued on the handle's stream (no host copy of a vector; the
 * factorisation still returns its two pivot counts).  pc_kkt_set_stream moves the handle onto the caller's stream --
 * the evaluation's -- so that one stream orders evaluation, assembly and solve. */
int pc_kkt_set_stream(pc_kkt* k, void* stream);
int pc_kkt_factor_device(pc_kkt* k, int use_hess, const double* d_dvec, int32_t* n_pos, int32_t* n_neg);
int pc_kkt_matvec_device(pc_kkt* k, int use_hess, const double* d_dvec, const double* d_x, double* d_y);
int pc_kkt_solve_refined_device(pc_kkt* k, int use_hess, const double* d_dvec_true, const double* d_rhs, int max_steps,
                                double* d_x, int32_t* n_solves);
/* A rank's part of a factorisation cut across ranks (the mesh is sharded by contiguous section
 * ranges, pycollo/mesh.py:297-335 is the partition those follow; pycollo_amd/kkt_sharded.py builds the tables).  The
 * reference has no counterpart -- IPOPT hands the whole matrix to one MUMPS process (pycollo/backend.py:1703-1711).  The
 * handle holds the rank's leaves, chain segments and its local border (the NLP's border + the nodes it shares with its
 * neighbours); G~ / H~ are read where the rank's own tile kernels wrote them.
 *   pc_kkt_factor_partial      assemble, eliminate leaves and chain; border_out[nb * nb] (host) = the border block with
 *                              every Schur complement added, NOT factorised, lower triangle valid -- the term this rank adds
 *                              to the reduced system; pivot counts of leaves and chain
 *   pc_kkt_border_load_factor  factorise the border block given in border[nb * nb] (lower triangle read) as it stands:
 *                              the reduced system, in a handle that has a border only
 *   pc_kkt_forward_partial     forward elimination of rhs[nu]; border_rhs_out[nb] = the border's right-hand side minus what
 *                              leaves and chain owe it (block order) -- the term this rank adds to the reduced right-hand side
 *   pc_kkt_backward_partial    back-substitution from border_x[nb] (block order); x[nu] = the local solution */
int pc_kkt_factor_partial(pc_kkt* k, int use_hess, const double* dvec, double* border_out, int32_t* n_pos, int32_t* n_neg);
int pc_kkt_border_load_factor(pc_kkt* k, const double* border, int32_t* n_pos, int32_t* n_neg);
int pc_kkt_forward_partial(pc_kkt* k, const double* rhs, double* border_rhs_out);
int pc_kkt_backward_partial(pc_kkt* k, const double* border_x, double* x);
/* Exported chain nodes (pc_kkt_desc::chain_export: the nodes a rank shares with its neighbours stay chain nodes of both
 * and are eliminated by neither): after pc_kkt_factor_partial their assembled panels, concatenated in ascending node
 * order, each [nz][nz + nr + nb] row-major = [D | K(node, its segment's exported last node) | F] (nr = 0 for a last node
 * or when the segment's last node is not exported); after pc_kkt_forward_partial their right-hand sides minus what the
 * eliminated blocks owe them; before pc_kkt_backward_partial their solution from the reduced system.  Host arrays. */
int pc_kkt_export_panels(pc_kkt* k, double* out);
int pc_kkt_export_rhs(pc_kkt* k, double* out);
int pc_kkt_import_solution(pc_kkt* k, const double* in);

/* ---- a device-resident interior-point iteration --------------------------
 * The reference enters its NLP solver once per solve (pycollo/backend.py:1807-1827: ca.nlpsol "ipopt"; legacy
 * pycollo/nlp.py:84-115) and IPOPT iterates in native code, calling the callbacks and its linear solver
 * (backend.py:1703-1711).  Here the iterate v = [x ; slacks], lambda, the bound multipliers, the step and every
 * right-hand side stay in device memory; one call per part of an iteration, a few scalars back each.  The algorithm
 * -- and its scalar logic: filter, barrier update, termination -- is pycollo_amd/ipm.py's. */
typedef struct pc_ipm pc_ipm;
typedef struct {
  int64_t n, m, ns;                 /* NLP variables, constraints, slacks (inequality rows) */
  const int64_t* ineq_rows;         /* [ns] */
  const double *vl, *vu;            /* [n + ns] bounds on v */
  const uint8_t *hasl, *hasu, *fixed;   /* [n + ns] finite lower / upper bound, fixed unknown */
  const double *row_scale, *rhs_c;  /* [m] IPOPT-style gradient-based row scaling; equality right-hand sides */
  double obj_scale;                 /* the objective's scaling factor */
} pc_ipm_desc;
int pc_ipm_create(pc_handle* h, pc_kkt* k, const pc_ipm_desc* desc, pc_ipm** out);
void pc_ipm_destroy(pc_ipm* s);
int pc_ipm_set_state(pc_ipm* s, const double* v, const double* lambda, const double* zl, const double* zu);
int pc_ipm_get_state(pc_ipm* s, double* v, double* lambda, double* zl, double* zu, double* c, double* g);   /* NULL: skipped */
/* evaluate J, grad J, g, jac_g at the current v; out3 = scaled objective, sum |c|, max |c| */
int pc_ipm_eval_point(pc_ipm* s, double* out3);
/* out10 = max |grad L| over free unknowns, max |c|, sum |c|, max / min of (v - vl) zl, max / min of (vu - v) zu,
 * sum |lambda|, sum zl, sum zu -- E_mu of Waechter & Biegler eq. (5) follows for any mu */
int pc_ipm_errors(pc_ipm* s, double* out10);
/* the Newton step for barrier parameter mu (after pc_ipm_errors at this point): Hessian, KKT assembly, factorisation
 * with the inertia-correcting regularisation loop (delta_w schedule from dw_last), refined solve, bound-multiplier
 * steps, fraction-to-the-boundary limits.  out8 = dw (< 0: regularisation failed), alpha_max, alpha_z, grad phi . dv,
 * mu x barrier sum at v, factorisations, back-substitutions, non-finite flag */
int pc_ipm_newton(pc_ipm* s, double mu, double tau, double dw_last, double* out8);
/* trial point v + alpha dv: out3 = scaled objective, sum |c|, mu x barrier sum there */
int pc_ipm_trial(pc_ipm* s, double alpha, double mu, double* out3);
/* Second-order correction after pc_ipm_trial rejected the step of size alpha (IPOPT's A-5.7 .. A-5.9, which the reference's
 * solver applies inside ipopt.Problem.solve, backend.py:1711): the factorisation of the last pc_ipm_newton solved again for the
 * constraint values alpha c + c(trial) (first != 0) or alpha c_soc + c(trial) of the previous corrected trial; the corrected step
 * replaces the Newton step for pc_ipm_trial / pc_ipm_accept.  out8 as pc_ipm_newton's.  pc_ipm_soc_restore puts the Newton step
 * back when the corrections were rejected. */
int pc_ipm_soc(pc_ipm* s, double alpha, int first, double mu, double tau, double* out8);
int pc_ipm_soc_restore(pc_ipm* s, double mu, double tau);
/* accept the last trial point: v, lambda (+= alpha), z (+= alpha_z, kept near the central path); grad J, jac_g at the new v */
int pc_ipm_accept(pc_ipm* s, double alpha, double alpha_z, double mu);

/* Evaluate at (x, obj_factor, lambda) and leave g, jac_g and the Lagrangian Hessian in device memory; only J,
 * grad J (dense n, may be NULL) and g (may be NULL) come back.  lambda == NULL: g and jac_g only. */
int pc_eval_resident(pc_handle* h, const double* x, double obj_factor, const double* lambda, double* f, double* grad,
                     double* g);
/* device pointers of the results of the last evaluation that went through the handle's own buffers */
int pc_device_results(pc_handle* h, const double** d_g, const double** d_jac, const double** d_hess);

/* timing of the last n pc_eval_all_device launches is measured by the caller with HIP events on the
 * stream it passed; this returns the stream the handle owns (hipStream_t) */
void* pc_stream(pc_handle* h);
/* diagnostic: copy `bytes` of a __device__ variable of the problem's code object to the host (the clock stamps of a
 * -DPC_STAMPS build, tools/stamps.py); synchronises the handle's stream first.  No reference counterpart. */
int pc_read_symbol(pc_handle* h, const char* name, void* dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* PYCOLLO_AMD_H */
