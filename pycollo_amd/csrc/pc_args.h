// Shared host/device argument blocks for the collocation kernels.
// Plain C structs: the host library fills them, generated kernels receive them by value.
#ifndef PC_ARGS_H
#define PC_ARGS_H

#include <stdint.h>

#define PC_MAX_ORDER 20
#define PC_FLAG_C 1
#define PC_FLAG_G 2
#define PC_FLAG_H 4

// One record per workgroup of a launch over a phase whose sections differ in order and whose code object carries
// order-specialised tile bodies next to the any-order one ("mixed" build, codegen): which body runs the tile and where
// the tile sits in the mesh, so that an order-pure tile needs no section tables.  Records are in launch order (the
// launched tile ranges of the phases one after the other): workgroup b of the launch reads record b.
struct PcTileRec {
  int32_t phase;     // phase of the tile
  int32_t tile;      // tile index inside its phase
  int32_t order;     // n > 0: every section of the tile has n nodes and the code object has a body for n; 0: any-order body
  int32_t k0, nsec;  // first section, number of sections
  int32_t n0;        // first node
  int32_t nprev;     // nodes of section k0 - 1 (the tile's first node closes it); 0 at k0 == 0
  int32_t qa_prev;   // offset of A_nprev inside the packed A tables
  int64_t E0;        // sec_E[k0]
  double w_prev;     // last quadrature weight of order nprev
  int32_t reserved[4];
};

#define PC_MAX_GOFF 32
#define PC_MAX_HOFF 96
#define PC_MAX_SCAL 96

// Per-phase arguments of the bulk kernel (passed by value in the kernarg segment: everything a wave
// needs before its first global load -- offsets, geometry -- arrives with the dispatch packet).
struct PcPhaseArgs {
  // NLP vectors (device)
  const double* x;      // [num_x] scaled variables
  const double* lam;    // [num_c] multipliers (may be null unless PC_FLAG_H)
  double* c;            // [num_c]
  double* G;            // [nnz_G]
  double* H;            // [nnz_H]
  // mesh (device); tile_* / sec_s / sec_E are only read when the section orders differ (uni_n == 0)
  const int32_t* tile_k0;  // [n_tiles+1] first section of every tile
  const int32_t* tile_n0;  // [n_tiles+1] first node of every tile
  const int32_t* sec_s;    // [K+1] first node of every section; sec_s[K] = N-1
  const double* sec_h;     // [K] section widths in tau
  const int64_t* sec_E;    // [K+1] prefix sum of (n_k-1)*n_k
  const double* qa;        // packed A tables of the orders in use
  const double* qw;        // packed weight tables; the host places them right behind the A tables (qw == qa + qa_total)
  const int64_t* hslot0;  // [NHZZ] slots of the node block at node 0
  const int64_t* hslotN;  // [NHZZ] slots of the node block at node N-1
  double* partials;       // [n_tiles][NRED] per-tile partial sums (two-launch build)
  // resident-tail build (pc_kernels.hpp, RES): values handed to the tail workgroup of the same launch as granules
  unsigned long long* gran;      // [n_tiles][NRED][2] the per-tile partial sums
  unsigned long long* erec;      // [n_rec][2] edge-node Hessian entries an endpoint term is added to (all phases)
  const double* tab;      // device copy of scal | goff | hoff (packed, used entries only): staged into LDS by
                          // the kernels of models whose tables do not fit the scalar register file
  const PcTileRec* tile_rec;  // mixed build: the records of this launch's tiles of this phase (record of workgroup 0 first)
  int64_t x_off, s_off;   // first x index of the phase / of the static parameters
  int64_t c_off, c_path_off, c_int_off;
  double t_fixed[2];
  int32_t N, K, n_tiles, flags;
  int32_t qa_total, qw_total;
  int32_t tile_begin;     // first tile of this launch (section-range sharding across GPUs)
  int32_t uni_n;          // > 0: every section has uni_n nodes (index arithmetic replaces the section tables)
  int32_t spt;            // sections per tile when uniform
  int32_t lds_out;        // doubles of the output staging buffer
  uint32_t epoch;         // tag of this launch's granules (resident-tail build; pc_bulk_all takes it from PcMultiArgs)
  // Resident-tail build: every Hessian entry of the edge nodes 0 / N-1 goes to the tail workgroup as a record of two
  // granules instead of being stored (the tail adds the endpoint term that lands on it, if any, and stores it).  Record
  // of a site = erec0 + (node N-1 ? NEDGE : 0) + site; sites: the NHZZ z-z entries, then the t strips (j, z), then the
  // s strips (l, z) -- NEDGE = NHZZ + 2 NZ + NS NZ of them.
  int32_t erec0, reserved0;
  int32_t wpt;            // waves (replicas) per tile: 1, 2 or 4; > 1 only with 64-node tiles
  int32_t block_threads;  // threads per workgroup of this launch (= blockDim.x, passed for the same reason as n_blocks)
  int32_t n_blocks;       // workgroups of this launch (tile_end - tile_begin): the kernel must not read gridDim,
                          // which lives in the dispatch packet in host memory
  int32_t qa_off[PC_MAX_ORDER + 1];
  int32_t qw_off[PC_MAX_ORDER + 1];
  // packed scaling doubles: Vz[NZ] rz[NZ] Vq[NQ] rq[NQ] Vt[2] rt[2] Vs[NS] rs[NS] Wd[NY] Wp[NP] Wi[NQ]
  double scal[PC_MAX_SCAL];
  // CSR value offsets
  int64_t goff[PC_MAX_GOFF];  // [NY] defect block bases | [NP] path bases | [NQ] integral bases
  int64_t hoff[PC_MAX_HOFF];  // [NZ] hz_base | [2*NZ] ht_base | [NS*NZ] hs_base  (-1 where absent)
};

// LDS carve-up of the bulk kernel, shared by the host (size query, launch) and the device.  All offsets in doubles.
//   tab_doubles  entries of the packed scal | goff | hoff table of the phase's model (NSCAL + NFN + 3 NZ + NS NZ):
//                staged by the kernels of models whose tables do not fit the scalar register file
//   mesh_tables  the kernel handles any mesh (compiled order 0) and stages the tile's section tables and the
//                per-order table offsets; an order-specialised kernel does index arithmetic instead
//   mixed        order-specialised body inside a mixed build: the section before the tile may have another order
// LDS per tile is what bounds the waves a CU holds for the multi-state models: nothing is reserved that the
// kernel at hand does not use.
struct LdsPlan {
  int qa, qw, off, tab, h, E, s, kr, cp, f, yu, fs, lam, red, out, total;
};
#ifdef __HIPCC__
__host__ __device__
#endif
inline LdsPlan lds_plan(int TB, int qa_total, int qw_total, int NY, int NFS, int NRED, int lds_out, int tab_doubles,
                        bool mesh_tables, bool mixed = false) {
  LdsPlan p;
  int o = 0;
  p.qa = o; o += qa_total;
  p.qw = o; o += qw_total;
  p.off = o; o += mesh_tables ? PC_MAX_ORDER + 1 : 0;   // int32 x 2 x (PC_MAX_ORDER+1): table offsets by order
  p.tab = o; o += tab_doubles;
  p.h = o; o += TB + 2;
  p.E = o; o += mesh_tables ? TB + 2 : 0;                 // int64 entries
  p.s = o; o += mesh_tables ? (TB + 4) / 2 + 1 : 0;       // int32 entries, (TB+3) of them
  p.kr = o; o += mesh_tables ? (TB + 1) / 2 + 1 : 0;      // int32 entries
  p.cp = o; o += (mixed && !mesh_tables) ? PC_MAX_ORDER : 0;   // order-pure tile of a mixed mesh: the previous section's last A column
  p.fs = o; o += NFS * TB;
  p.red = o; o += (NRED > 0 ? NRED : 1) * 16;
  // f, y and the staged multipliers are dead once the defect values are formed; the output staging buffer
  // (CSR runs are written to HBM fully coalesced) is laid over them
  p.f = o;
  p.yu = p.f + NY * TB;
  p.lam = p.yu + NY * TB;
  p.out = o;
  {
    const int node_arrays = 2 * NY * TB + NY * (TB + PC_MAX_ORDER);
    o += node_arrays > lds_out ? node_arrays : lds_out;
  }
  p.total = o;
  return p;
}

// Row groups of the order-specialised tile bodies (pc::bulk, "row groups"): in how many passes over the section rows
// a state's defect-Jacobian block of a tile of n-node sections is staged and flushed, given the longest row of any
// state (D n + C doubles).  Two staging regions per workgroup (the two-wave build) must fit PC_STAGE_BUDGET doubles --
// what leaves the workgroup inside a quarter of a CU's LDS next to its tables.  One pass while a tile of at least 85 %
// of the full tile's rows fits (the host then shortens the tile a little: Delta III at order 5 runs 56-node tiles);
// else the fewest passes with which the full tile fits.  The kernels and the host's LDS sizing (pc_desc.hpp) both
// call this.
#define PC_STAGE_BUDGET 4480   // doubles: 2 regions x 17.5 KiB
#ifdef __HIPCC__
__host__ __device__
#endif
constexpr int pc_row_passes(int n, int max_row_len) {
  if (n < 3 || max_row_len <= 0) return 1;
  const int nq = 63 / (n - 1);   // sections of a full tile
  if (20 * (PC_STAGE_BUDGET / 2 / max_row_len) >= 17 * nq * (n - 1)) return 1;
  int np = 2;
  while (np < n - 1 && 2 * nq * ((n - 1 + np - 1) / np) * max_row_len > PC_STAGE_BUDGET) ++np;
  return np;
}
// rows a pass holds
#ifdef __HIPCC__
__host__ __device__
#endif
constexpr int pc_row_group(int n, int max_row_len) {
  const int np = pc_row_passes(n, max_row_len);
  return (n - 1 + np - 1) / np;
}

// The first 14 dwords of pc_bulk_p<i>'s argument block, passed as leading scalar kernel parameters: the command
// processor preloads them into SGPRs with the dispatch (kernarg preload), so a wave can address its node loads
// without waiting for a scalar load of the block first.  Copies of the PcPhaseArgs fields of the same name.
struct PcLead {
  const double* xz;     // x + x_off: the phase's first node value
  const double* lamd;   // lam + c_off: the phase's first defect multiplier (null without PC_FLAG_H)
  const double* qa;     // packed A tables, the packed weight tables right behind them (qw == qa + qa_total)
  const double* sec_h;  // section widths
  int32_t N, K, tile_begin, n_blocks;
  int32_t wa;           // flags | wpt << 8 | (block_threads / 64) << 12 | spt << 16 | tail blocks (resident build) << 28
  int32_t wb;           // uniform order n: qa_off[n] | (qa_total + qw_off[n]) << 16 (both relative to qa); else 0
};
// what the host hands to pc_bulk_p<i>: (lead scalars..., PcPhaseArgs a)
struct PcBulkArgs {
  PcLead lead;
  PcPhaseArgs a;
};

#define PC_MAX_PHASES 8

// One launch for the bulk kernels of every phase of a multi-phase problem (`pc_bulk_all`): workgroup b belongs to
// the phase p with first_block[p] <= b < first_block[p + 1] and runs that phase's tile b - first_block[p] +
// tile_begin.  The per-phase argument blocks live in device memory (they only change with the scaling or the
// tile ranges); what changes from call to call travels in this small kernarg block.
struct PcMultiArgs {
  const double* x;
  const double* lam;
  double* c;
  double* G;
  double* H;
  const PcPhaseArgs* ph;                  // [n_phases], device memory
  int32_t flags, n_phases;
  int32_t first_block[PC_MAX_PHASES + 1];
  uint32_t epoch;                         // tag of this launch's granules (resident-tail build)
  int32_t tail_blocks;                    // resident-tail build: leading workgroups that run the tail (1, or one per part
                                          //   of a heavy endpoint block); the tiles follow
  const PcTileRec* trec;                  // mixed build: one record per tile workgroup of the launch, else null
};
#define PC_MAX_POINT 96           // endpoint (point) variables: y(t0), y(tF), q, t of every phase, s
#define PC_MAX_ENDPOINT_ROWS 32   // endpoint constraint rows
#define PC_TAIL_THREADS 256        // workgroup of the tail kernel: 4 waves, one per part of the endpoint block (8 waves
                                   // measured: +0.9 us on every evaluation, no gain on the heaviest endpoint block)
#define PC_TAIL_OWNED_MAX 1024   // Hessian entries the tail accumulates in LDS

struct PcTailPhase {
  const double* partials;  // [n_tiles][NRED]
  const unsigned long long* gran;   // resident-tail build: the same sums as granules, [n_tiles][NRED][2]
  const double* scal;
  int64_t x_off, s_off, c_int_off;
  int64_t gq_base[8];      // CSR offset of the q column of every integral row (then t, s follow)
  const int64_t* hsum_slot;  // [2*NS] (t_j, s_l) slots then [NS*(NS+1)/2] (s_l, s_l') slots; -1 absent
  const int32_t* hsum_local; // same shape: index of the slot in tail_owned (the tail accumulates in LDS)
  double t_fixed[2];
  int32_t n_tiles, N;
};

struct PcTailArgs {
  const double* x;
  const double* lam;
  double* c;
  double* G;
  double* H;
  double* fobj;                 // [1] objective value (scaled by w_J)
  double* grad_nz;              // [NGJ] structural non-zeros of grad J (scaled by w_J), or null
  double sigma, wJ;             // objective factor and objective scaling
  const int64_t* point_x;       // [n_point] x index of every point variable
  const double* point_V;        // [n_point]
  const double* point_r;        // [n_point]
  const double* W_end;          // [n_b]
  const int64_t* tail_owned;    // [n_tail_owned] H slots written by the tail only (zeroed first)
  const int64_t* pt_hslot;      // [n_pt_hess] H slot of every endpoint Hessian entry
  const int32_t* pt_hlocal;     // [n_pt_hess] its index in tail_owned, or -1: the slot belongs to an edge node of
                                //   the bulk kernels and the endpoint term is added to the value they wrote
  int64_t c_end_off;            // first endpoint row of c
  int64_t g_end_base;           // CSR offset of the first endpoint row of G
  int32_t n_tail_owned, flags;
  int32_t block_threads;        // threads of the workgroup that runs the tail: 64, 128 or 256 (blockDim.x would be a
                                //   late scalar load)
  int32_t lds_nred;             // largest NRED of any phase: sizes the tail's LDS carve (pc::tail_lds)
  // resident-tail build
  const unsigned long long* erec;   // [n_rec][2] granules of the edge-node Hessian entries (see PcPhaseArgs)
  const int64_t* rec_slot;          // [n_rec] H slot of every record, -1: the site does not exist in this model
  const int32_t* rec_term;          // [n_rec] endpoint Hessian entry whose term is added to the record, or -1
  int32_t n_rec;
  int32_t n_tail_blocks;            // workgroups sharing the tail's endpoint block (see pc_kernels.hpp, Tail)
  unsigned long long* hb_gran;      // [n_pt_hess][2] endpoint Hessian terms handed from the helper tail blocks to block 0
  unsigned* timeout;                // host-visible word the tail sets when a granule never arrives (bounded spin)
  uint32_t epoch, reserved;
  PcTailPhase ph[PC_MAX_PHASES];
  // the endpoint block's small tables by value: their loads join the argument fetch instead of forming a
  // second dependent round trip through device memory (point_x / point_V / point_r / W_end hold the same data)
  int64_t pt_x[PC_MAX_POINT];
  double pt_V[PC_MAX_POINT];
  double pt_r[PC_MAX_POINT];
  double pt_W[PC_MAX_ENDPOINT_ROWS];
};

// The first 12 dwords of pc_tail's argument block, as leading scalar kernel parameters (preloaded into SGPRs like
// PcLead): what the partial-sum loads and lane 0's loads of the FIRST phase are addressed with.  Copies of
// PcTailArgs::x / flags / block_threads and of ph[0].{partials, scal, x_off, n_tiles, N}.
struct PcTailLead {
  const double* x;
  const double* partials0;
  const double* scal0;
  int64_t x_off0;
  int32_t n_tiles0, N0, flags, block_threads;
};
struct PcTailLaunch {   // what the host hands to pc_tail: (lead scalars..., PcTailArgs a)
  PcTailLead lead;
  PcTailArgs t;
};

// Arguments of the ph mesh-error kernel (pycollo/mesh_refinement.py:63-240).
struct PcRefineArgs {
  const double* x;          // [num_x] scaled solution
  const int32_t* tile_k0;   // [n_tiles+1] first section of every tile (sum of n_k+1 lanes <= blockDim)
  const int32_t* lane0;     // [K] first lane of every section inside its tile
  const int32_t* sec_s;     // [K+1] first solution node of every section
  const double* sec_h;      // [K]
  const double* tabB;       // per order n: (n-1) x n integration of the solution-node Lagrange basis up to ph node j
  const double* tabE;       // per order n: (n-1) x n evaluation of that basis at the interior ph nodes
  const double* tabA;       // per order n: n x (n+1) integration matrix of order n+1 (quadrature A(n+1))
  double* max_rel;          // [K] section maximum of the relative error
  double* max_abs;          // [K][NY] section maximum of the absolute error per state
  int64_t x_off, s_off;
  double t_fixed[2];
  int32_t N, K, tab_total_BE, tab_total_A;
  int32_t offBE[PC_MAX_ORDER + 1];
  int32_t offA[PC_MAX_ORDER + 1];
  double scal[PC_MAX_SCAL];
};

#endif  // PC_ARGS_H
