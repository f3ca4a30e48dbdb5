"""HIP source generation + hipcc driver for the per-problem collocation code object.

The hand-written kernels live in ``csrc/pc_kernels.hpp``; this module only emits what is specific to
one problem: compile-time sizes, the structural non-zero lists and one straight-line, CSE'd
``eval`` per phase (the role ``numbafy`` / ``numbafy_hessian`` play in the reference --
pycollo/numbafy.py:90-103,231-245: precomputed constants -> tiered intermediates -> stacked outputs),
and the endpoint function block.  The translation unit is compiled for gfx950 into a code object
(``.hsaco``) that the C-ABI library loads with ``hipModuleLoad``; objects are cached by model digest.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

import sympy as sym
from sympy.printing.c import C99CodePrinter

from .model import Model, PhaseModel, PointModel

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cache")
ARCH = "gfx950"


class _Printer(C99CodePrinter):
    """C printer that keeps fp64 integer powers off the transcendental path."""

    def _print_Pow(self, expr):
        base, exp = expr.base, expr.exp
        b = self.parenthesize(base, 200)  # always parenthesise non-atomic bases
        if exp.is_Integer:
            n = int(exp)
            if n == 0:
                return "1.0"
            a = abs(n)
            if a <= 4:
                body = "*".join([b] * a)
                body = f"({body})" if a > 1 else b
            else:
                body = f"pc_powi<{a}>({self._print(base)})"
            return body if n > 0 else f"(1.0/{body})"
        if exp == sym.Rational(1, 2):
            return f"sqrt({self._print(base)})"
        if exp == sym.Rational(-1, 2):
            return f"(1.0/sqrt({self._print(base)}))"
        if exp == sym.Rational(3, 2):
            return f"({b}*sqrt({self._print(base)}))"
        if exp == sym.Rational(-3, 2):
            return f"(1.0/({b}*sqrt({self._print(base)})))"
        if exp.is_Rational and exp.q == 2 and abs(exp.p) <= 15:
            # other half-integer powers (r^-5/2, r^7/2 of gravity gradients): integer power times one square root --
            # libm's pow() is a log and an exp, some two hundred fp64 instructions per call
            k = (abs(int(exp.p)) - 1) // 2
            ip = "*".join([b] * k) if k <= 4 else f"pc_powi<{k}>({self._print(base)})"
            body = f"(({ip})*sqrt({self._print(base)}))"
            return body if exp.p > 0 else f"(1.0/{body})"
        return f"pow({self._print(base)}, {self._print(exp)})"

    def parenthesize(self, item, level, strict=False):
        s = self._print(item)
        if item.is_Atom and not (item.is_Number and item < 0):
            return s
        return f"({s})"

    def _print_sec(self, expr):
        return f"(1.0/cos({self._print(expr.args[0])}))"

    def _print_csc(self, expr):
        return f"(1.0/sin({self._print(expr.args[0])}))"

    def _print_cot(self, expr):
        return f"(1.0/tan({self._print(expr.args[0])}))"

    def _print_Float(self, expr):
        return repr(float(expr))

    def _print_Integer(self, expr):
        return f"{int(expr)}.0"

    def _print_Rational(self, expr):
        return f"({int(expr.p)}.0/{int(expr.q)}.0)"


_PRINTER = _Printer({"allow_unknown_functions": False})


def _c(expr) -> str:
    return _PRINTER.doprint(sym.sympify(expr))


def _emit_block(inputs: dict, outputs: list[tuple[str, sym.Expr]], tmp: str) -> list[str]:
    """CSE a list of (lvalue, expr) assignments; ``inputs`` maps symbols to C rvalues."""
    if not outputs:
        return []
    exprs = [sym.sympify(e) for _, e in outputs]
    repl, reduced = sym.cse(exprs, symbols=sym.numbered_symbols(tmp), order="none")
    lines = []
    sub = dict(inputs)
    for s, e in repl:
        lines.append(f"    const double {s} = {_c(e.xreplace(sub))};")
    for (lhs, _), e in zip(outputs, reduced):
        lines.append(f"    {lhs} = {_c(e.xreplace(sub))};")
    return lines


STATIC_W_MAX_OPS = 4000    # models up to this many operations get per-replica kernels (pc_engine's "heavy" threshold)


def is_heavy(pm: PhaseModel) -> bool:
    """Models past this size run in the two-wave build (pc::bulk SPLIT rule, pc_create's tile choice)."""
    return pm.eval_ops > STATIC_W_MAX_OPS


def _static_w_list(pm: PhaseModel, single_phase: bool = False) -> list[int]:
    """Waves-per-tile counts that get a kernel with the replica index compiled in.  Light and medium models: 2 and 4
    (6-8 % over the run-time replica index).  Heavy models: 2 -- the *two-wave build*: each replica's copy holds
    its own items only and evaluates the node functions in two passes (M::HEAVY), which is what brings a Delta III
    tile body from 308 to <= 256 registers, i.e. two resident waves per SIMD instead of one -- and, for a single-phase
    model, 4 as well: on a small mesh (space station, 96 tiles) a replica's copy of the node functions keeps only what
    its own items need, 23.8 -> 17.9 us (a multi-phase kernel would carry phases x 4 heavy bodies: not built)."""
    if os.environ.get("PYCOLLO_AMD_STATIC_W", "1") == "0":
        return []
    if is_heavy(pm):
        return [2, 4] if single_phase and _heavy_w4_enabled() else [2]
    return [2, 4]


def _heavy_w4_enabled() -> bool:
    """PYCOLLO_AMD_HEAVY_W4=0 leaves the four-wave per-replica kernel of heavy single-phase models out (compile time)."""
    return os.environ.get("PYCOLLO_AMD_HEAVY_W4", "1") != "0"


def _constexpr_table(name: str, values: list[int]) -> str:
    if not values:
        return f"  static constexpr int {name}(int) {{ return 0; }}"
    body = ", ".join(str(int(v)) for v in values)
    return (f"  static constexpr int {name}(int e) {{ constexpr int t[{len(values)}] = {{{body}}}; return t[e]; }}")


def edge_flags(model: Model, pm: PhaseModel) -> list[int]:
    """Which Hessian entries of the phase's edge nodes (node 0, then node N-1) an endpoint term lands on -- a
    property of the model alone.  Sites per edge, in the kernels' order: the z-z entries of the node block (rows of
    ``pm.hess`` with both indices < n_z), the t strips (j, z), the s strips (l, z).  In the resident-tail build these
    entries travel from the edge tile to the tail workgroup as records; all others are stored by the tile."""
    nz, ns = pm.n_z, pm.n_s
    zz = [(r, c) for r, c, _ in pm.hess if r < nz and c < nz]
    hm = pm.hess_mask()
    tz = pm.t_strip_mask()
    n_t = int(pm.t_free[0]) + int(pm.t_free[1])
    ne = len(zz) + 2 * nz + ns * nz
    flags = [0] * (2 * ne)
    pv = model.point.vars
    for r, c, _ in model.point.hess:
        a, b = pv[r], pv[c]
        for u, v in ((a, b), (b, a)):
            if u.phase != pm.index or u.kind not in ("y0", "yF"):
                continue
            edge = 0 if u.kind == "y0" else 1
            site = None
            if v.phase == pm.index and v.kind == u.kind:                      # node block
                key = (max(u.idx, v.idx), min(u.idx, v.idx))
                if key in zz:
                    site = zz.index(key)
            elif v.phase == pm.index and v.kind in ("t0", "tF") and n_t > 0 and tz[u.idx]:
                j = 0 if v.kind == "t0" else (1 if pm.t_free[0] else 0)
                site = len(zz) + j * nz + u.idx
            elif v.kind == "s" or (v.phase == pm.index and v.kind == "q"):  # strips of the non-time parameters
                l = pm.param_index(0 if v.kind == "s" else 1, v.idx)
                if l >= 0 and hm[nz + l, u.idx]:
                    site = len(zz) + 2 * nz + l * nz + u.idx
            if site is not None:
                flags[edge * ne + site] = 1
    return flags


def _phase_struct(pm: PhaseModel, model: Model | None = None) -> str:
    name = f"Phase{pm.index}"
    nfn, nv = pm.n_fn, pm.n_v
    v_in = {s: sym.Symbol(f"v[{i}]") for i, s in enumerate(pm.z + pm.s)}
    mult = pm.mf + pm.mp + pm.mg
    m_in = {s: sym.Symbol(f"mult[{i}]") for i, s in enumerate(mult)}
    inputs = {**v_in, **m_in}
    outs = [(f"F[{i}]", e) for i, e in enumerate(pm.f + pm.p + pm.g)]
    outs += [(f"Jv[{i}]", e) for i, (_, _, e) in enumerate(pm.jac)]
    outs += [(f"Hv[{i}]", e) for i, (_, _, e) in enumerate(pm.hess)]
    body = [f"    constexpr double {k} = {float(val)!r};" for k, val in pm.consts]
    body += _emit_block(inputs, outs, "w")
    lines = [f"struct {name} {{",
             f"  static constexpr int NY = {pm.n_y}, NU = {pm.n_u}, NQ = {pm.n_q}, NP = {pm.n_p}, NS = {pm.n_s};",
             f"  static constexpr bool T0_FREE = {'true' if pm.t_free[0] else 'false'}, "
             f"TF_FREE = {'true' if pm.t_free[1] else 'false'};",
             f"  static constexpr int NJ = {len(pm.jac)}, NH = {len(pm.hess)};",
             f"  static constexpr bool HEAVY = {'true' if is_heavy(pm) else 'false'};   // two-wave build: node functions in two passes",
             _constexpr_table("jr", [r for r, _, _ in pm.jac]),
             _constexpr_table("jc", [c for _, c, _ in pm.jac]),
             _constexpr_table("hr", [r for r, _, _ in pm.hess]),
             _constexpr_table("hc", [c for _, c, _ in pm.hess]),
             "  // what each parameter v[NZ + l] is: kind 0 static parameter, 1 integral variable, 2 free time; index within its kind",
             _constexpr_table("wk", pm.w_kind or [0] * pm.n_s),
             _constexpr_table("wi", pm.w_idx or list(range(pm.n_s))),
             "  // edge-node Hessian entry sites (node 0, then node N-1) whose value goes to the tail as a record",
             _constexpr_table("efl", edge_flags(model, pm) if model is not None else []),
             "  __device__ static __forceinline__ void eval(const double* __restrict__ v, const double* __restrict__ mult,",
             "      double* __restrict__ F, double* __restrict__ Jv, double* __restrict__ Hv) {",
             "    (void)v; (void)mult; (void)F; (void)Jv; (void)Hv;"]
    lines += body
    lines += ["  }"]
    # the same outputs in two passes (values and first partials / second partials): heavy models run them one after
    # the other so that the first and second partials are never live together (pc::bulk, SPLIT)
    cdecl = [f"    constexpr double {k} = {float(val)!r};" for k, val in pm.consts]
    fj = [(f"F[{i}]", e) for i, e in enumerate(pm.f + pm.p + pm.g)] + [(f"Jv[{i}]", e) for i, (_, _, e) in enumerate(pm.jac)]
    lines += ["  __device__ static __forceinline__ void eval_fj(const double* __restrict__ v, double* __restrict__ F,",
              "      double* __restrict__ Jv) {", "    (void)v; (void)F; (void)Jv;"]
    lines += cdecl + _emit_block(v_in, fj, "w") + ["  }"]
    lines += ["  __device__ static __forceinline__ void eval_h(const double* __restrict__ v, const double* __restrict__ mult,",
              "      double* __restrict__ Hv) {", "    (void)v; (void)mult; (void)Hv;"]
    lines += cdecl + _emit_block(inputs, [(f"Hv[{i}]", e) for i, (_, _, e) in enumerate(pm.hess)], "w") + ["  }"]
    # state equations only (ph mesh-error estimate, mesh_refinement.py:199-201)
    fbody = [f"    constexpr double {k} = {float(val)!r};" for k, val in pm.consts]
    fbody += _emit_block(v_in, [(f"F[{i}]", e) for i, e in enumerate(pm.f)], "w")
    lines += ["  __device__ static __forceinline__ void eval_f(const double* __restrict__ v, double* __restrict__ F) {",
              "    (void)v; (void)F;"] + fbody + ["  }", "};", ""]
    assert nfn >= 0 and nv >= 0
    return "\n".join(lines)


POINT_PARTS = 4   # the endpoint block is cut into this many parts, one per wave of the tail workgroup (PC_TAIL_THREADS / 64)


def _point_struct(pt: PointModel) -> str:
    xb_in = {pv.symbol: sym.Symbol(f"xb[{i}]") for i, pv in enumerate(pt.vars)}
    inputs = dict(xb_in)
    inputs[pt.sigma] = sym.Symbol("sw")
    inputs.update({s: sym.Symbol(f"lb[{i}]") for i, s in enumerate(pt.lam)})
    outs = [("Jval", pt.J)]
    outs += [(f"gJ[{i}]", e) for i, (_, e) in enumerate(pt.J_grad)]
    outs += [(f"b[{i}]", e) for i, e in enumerate(pt.b)]
    outs += [(f"jb[{i}]", e) for i, (_, _, e) in enumerate(pt.b_jac)]
    outs += [(f"hb[{i}]", e) for i, (_, _, e) in enumerate(pt.hess)]
    # deal the outputs to POINT_PARTS parts, heaviest first onto the lightest part (operation counts before CSE);
    # the objective value and gradient stay together in part 0's list so that part 0 always exists
    cost = [max(1, int(sym.count_ops(sym.sympify(e)))) for _, e in outs]
    part = [0] * len(outs)
    load = [0] * POINT_PARTS
    n_obj = 1 + len(pt.J_grad)
    load[0] = sum(cost[:n_obj])
    for i in sorted(range(n_obj, len(outs)), key=lambda i: -cost[i]):
        g = min(range(POINT_PARTS), key=lambda g: load[g])
        part[i] = g
        load[g] += cost[i]
    consts = [f"    constexpr double {k} = {float(val)!r};" for k, val in pt.consts]
    o_b, o_jb, o_hb = n_obj, n_obj + len(pt.b), n_obj + len(pt.b) + len(pt.b_jac)
    lines = ["struct Point {",
             f"  static constexpr int NPV = {len(pt.vars)}, NB = {len(pt.b)}, NGJ = {len(pt.J_grad)}, "
             f"NBJ = {len(pt.b_jac)}, NPH = {len(pt.hess)}, NPARTS = {POINT_PARTS};",
             _constexpr_table("gc", [c for c, _ in pt.J_grad]),
             _constexpr_table("br", [r for r, _, _ in pt.b_jac]),
             _constexpr_table("bc", [c for _, c, _ in pt.b_jac]),
             _constexpr_table("phr", [r for r, _, _ in pt.hess]),
             _constexpr_table("phc", [c for _, c, _ in pt.hess]),
             "  // which part evaluates an endpoint row / Jacobian entry / Hessian entry (the objective is part 0's)",
             _constexpr_table("part_b", part[o_b:o_jb]),
             _constexpr_table("part_jb", part[o_jb:o_hb]),
             _constexpr_table("part_hb", part[o_hb:])]
    sig = ("(const double* __restrict__ xb, double sw, const double* __restrict__ lb,\n"
           "      double& Jval, double* __restrict__ gJ, double* __restrict__ b, double* __restrict__ jb, double* __restrict__ hb) {")
    void = "    (void)xb; (void)sw; (void)lb; (void)Jval; (void)gJ; (void)b; (void)jb; (void)hb;"
    lines += ["  __device__ static __forceinline__ void eval" + sig, void] + consts + _emit_block(inputs, outs, "w") + ["  }"]
    for g in range(POINT_PARTS):
        mine = [o for o, pg in zip(outs, part) if pg == g]
        lines += [f"  __device__ static __forceinline__ void eval_part{g}" + sig, void] + consts + _emit_block(inputs, mine, "w") + ["  }"]
    lines += ["  template <int G> __device__ static __forceinline__ void eval_part" + sig]
    for g in range(POINT_PARTS):
        lines.append(f"    if constexpr (G == {g}) eval_part{g}(xb, sw, lb, Jval, gJ, b, jb, hb);")
    lines += ["  }", "};", ""]
    return "\n".join(lines)


def _norm_mixed(model: Model, orders, mixed):
    """Per phase the sorted tuple of section orders a mixed build specialises (empty: the phase is not mixed)."""
    if mixed is None:
        return tuple(() for _ in model.phases)
    mixed = tuple(tuple(sorted({int(n) for n in m})) for m in mixed)
    if len(mixed) != len(model.phases):
        raise ValueError("one tuple of specialised orders (or an empty one) per phase is required")
    for o, m in zip(orders, mixed):
        if m and int(o) != 0:
            raise ValueError("a phase is either compiled for one order or mixed (orders[p] == 0)")
        if any(n < 2 or n > 20 for n in m):
            raise ValueError("section orders lie in [2, 20]")
    return mixed


def generate_source(model: Model, orders=None, heavy_cap: bool | None = None, mixed=None) -> str:
    """``mixed[p]`` non-empty: phase p's mesh has sections of several orders and its tile kernels carry one body per
    listed order next to the any-order body (pc::bulk_mix picks per tile from the tile's record)."""
    orders = tuple(orders) if orders is not None else tuple(0 for _ in model.phases)
    mixed = _norm_mixed(model, orders, mixed)
    any_mixed = any(mixed)

    def seq(pm):
        return "std::integer_sequence<int, " + ", ".join(str(n) for n in mixed[pm.index]) + ">{}"
    if heavy_cap is None:
        heavy_cap = _heavy_cap_enabled()
    heavy_any = any(is_heavy(pm) for pm in model.phases)
    parts = [f"// generated by pycollo_amd.codegen for model '{model.name}' digest {model.digest} -- do not edit",
             '#include "pc_kernels.hpp"',
             "",
             "template <int N> __device__ __forceinline__ double pc_powi(double x) {",
             "  double r = 1.0;",
             "#pragma unroll",
             "  for (int i = 0; i < N; ++i) r *= x;",
             "  return r;",
             "}",
             "",
             "namespace gen {"]
    for pm in model.phases:
        parts.append(_phase_struct(pm, model))
    parts.append(_point_struct(model.point))
    parts.append("}  // namespace gen\n")
    parts.append("namespace gen {")
    parts.append("struct Tail {   // finishes one evaluation: cross-tile sums of every phase and the endpoint block")
    parts.append("  // RES: block 0 of a resident-tail bulk launch (values of other workgroups arrive as granules)")
    parts.append("  template <bool RES = false, bool BIG = false>")
    parts.append("  __device__ static __forceinline__ void run(const PcTailArgs& a, const PcTailLead* ld = nullptr, int arg_off = 0,")
    parts.append("                                             int tb = 0, int ntb = 1) {")
    parts.append("    extern __shared__ double pc_tail_smem[];")
    parts.append("    const pc::TailLds L = pc::tail_lds<Point>(a, pc_tail_smem);")
    first = model.phases[0].index
    parts.append("    if constexpr (RES) {")
    parts.append("      pc::KernargWarm warm;")
    parts.append("      warm.issue<(int)sizeof(PcTailArgs)>(arg_off);   // every line of the tail's argument block, at once")
    parts.append("      pc::tail_point_load<Point, true>(a, L);")
    parts.append("      pc::tail_begin(a, L);")
    parts.append("      warm.settle();")
    parts.append("      pc::tail_point_eval<Point>(a, L, tb, ntb);   // the endpoint rows are out before the first tile's sums arrive")
    parts.append("      if (tb > 0) { pc::tail_point_publish<Point>(a, L, tb, ntb); return; }   // helper block: its parts only")
    for pm in model.phases:
        parts.append(f"      pc::tail_phase<Phase{pm.index}, true>(a, {pm.index}, L);")
    parts.append("      pc::tail_point_collect<Point>(a, L, ntb);")
    parts.append("      pc::tail_point_apply<Point, true>(a, L);")
    parts.append("      pc::tail_end(a, L);")
    parts.append("    } else {")
    parts.append(f"      pc::TailPhaseRegs<Phase{first}> r0;   // the first phase's loads go out before anything else")
    parts.append(f"      pc::tail_phase_issue<Phase{first}, false, BIG>(a, {first}, r0, ld);")
    parts.append("      __builtin_amdgcn_sched_barrier(0);")
    parts.append("      pc::tail_point_load<Point, false>(a, L);")
    parts.append("      pc::tail_begin(a, L);")
    parts.append("      pc::tail_point_eval<Point>(a, L);")
    parts.append(f"      pc::tail_phase_finish<Phase{first}>(a, {first}, r0, L);")
    for pm in model.phases[1:]:
        parts.append(f"      pc::tail_phase<Phase{pm.index}, false, BIG>(a, {pm.index}, L);")
    parts.append("      pc::tail_point_apply<Point, false>(a, L);")
    parts.append("      pc::tail_end(a, L);")
    parts.append("    }")
    parts.append("  }")
    parts.append("};")
    parts.append("}  // namespace gen\n")
    occ = _occupancy_attr()
    lead_sig = ('const double* xz, const double* lamd, const double* qa, const double* sec_h, int N, int K, '
                'int tile_begin, int n_blocks, int wa, int wb')
    # A heavy model's code object is compiled in parts, side by side (build_code_object: -DPC_PART=k, one module each):
    # part 0 holds the per-phase kernels, the mesh-error kernels and the tails, every merged / resident / per-replica
    # launch kernel is a part of its own.  PC_PART < 0: everything in one object.
    parts.append("#ifndef PC_PART\n#define PC_PART -1\n#endif")
    parts.append("#if PC_PART <= 0")
    for pm in model.phases:
        # leading scalars = struct PcLead, member by member: the command processor preloads them into SGPRs
        hv = _heavy_attr(is_heavy(pm), heavy_cap)
        parts.append(f'extern "C" __global__ void __launch_bounds__(256) {occ}{hv}pc_bulk_p{pm.index}({lead_sig}, PcPhaseArgs a) {{')
        if mixed[pm.index] and len(model.phases) == 1:
            parts.append('  const PcPhaseArgs& ka = pc::kernarg_phase_args();   // (a itself is not named: see there)')
            parts.append('  const int blk = pc::xcd_major((int)blockIdx.x, n_blocks);')
            parts.append(f'  pc::bulk_mix<gen::Phase{pm.index}, false, 0, 0>({seq(pm)}, ka.tile_rec + blk, ka, false, 0, blk);')
        else:   # (the per-phase kernels of a multi-phase mixed build keep the any-order body: only the merged launch is specialised)
            parts.append('  const PcLead ld{xz, lamd, qa, sec_h, N, K, tile_begin, n_blocks, wa, wb};')
            parts.append(f'  pc::bulk<gen::Phase{pm.index}, {int(orders[pm.index])}>(a, false, 0, -1, &ld);')
        parts.append('}')
    parts.append("#endif")
    if len(model.phases) == 1:
        pm = model.phases[0]
        parts.append("// resident-tail build: block 0 runs the tail beside the tiles, one launch per evaluation")
        hv = _heavy_attr(is_heavy(pm), heavy_cap)
        parts.append(f'#if PC_PART < 0 || PC_PART == {1}')
        parts.append(f'extern "C" __global__ void __launch_bounds__(256) {occ}{hv}pc_bulk_p{pm.index}_r({lead_sig}, PcPhaseArgs a, PcTailArgs t) {{')
        parts.append('  const int ntb = (wa >> 28) & 7;   // leading workgroups that run the tail')
        parts.append('  if ((int)blockIdx.x < ntb) { gen::Tail::run<true>(t, nullptr, (int)sizeof(PcBulkArgs), (int)blockIdx.x, ntb); return; }')
        if mixed[pm.index]:
            parts.append('  const PcPhaseArgs& ka = pc::kernarg_phase_args();')
            parts.append('  const int blk = pc::xcd_major((int)blockIdx.x - ntb, n_blocks);')
            parts.append(f'  pc::bulk_mix<gen::Phase{pm.index}, true, 0, 0>({seq(pm)}, ka.tile_rec + blk, ka, false, 0, blk);')
        else:
            parts.append('  const PcLead ld{xz, lamd, qa, sec_h, N, K, tile_begin, n_blocks, wa, wb};')
            parts.append(f'  pc::bulk<gen::Phase{pm.index}, {int(orders[pm.index])}, true>(a, false, 0, pc::xcd_major((int)blockIdx.x - ntb, n_blocks), &ld);')
        parts.append('}')
        parts.append('#endif')
        # The same launch with the replica index as a template argument, one kernel per waves-per-tile count: a
        # replica's copy of the tile body then holds only the items dealt to it (everything else is dead code in that
        # copy) -- 6-8 % on every workload that shares tiles (config 2, W = 4: 4.45-4.7 -> 4.14-4.32 us).  pc_create
        # picks pc_bulk_p0_r_w<W> when the code object has it (which W exist: _static_w_list).
        static_ws = _static_w_list(pm, single_phase=True)
        for wn in static_ws:
            parts.append(f'#if PC_PART < 0 || PC_PART == {2 + static_ws.index(wn)}')
            parts.append(f'extern "C" __global__ void __launch_bounds__(256) {occ}{hv}pc_bulk_p{pm.index}_r_w{wn}({lead_sig}, PcPhaseArgs a, PcTailArgs t) {{')
            parts.append('  const int ntb = (wa >> 28) & 7;')
            parts.append('  if ((int)blockIdx.x < ntb) { gen::Tail::run<true>(t, nullptr, (int)sizeof(PcBulkArgs), (int)blockIdx.x, ntb); return; }')
            if mixed[pm.index]:
                parts.append('  const PcPhaseArgs& ka = pc::kernarg_phase_args();')
            else:
                parts.append('  const PcLead ld{xz, lamd, qa, sec_h, N, K, tile_begin, n_blocks, wa, wb};')
            parts.append(f'  if (((wa >> 8) & 0xf) != {wn}) return;   // (built for exactly that many waves per tile; the host checks too)')
            parts.append('  const int blk = pc::xcd_major((int)blockIdx.x - ntb, n_blocks);')
            parts.append('  switch (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6)) {')
            for wv in range(wn):
                if mixed[pm.index]:
                    parts.append(f'    case {wv}: pc::bulk_mix<gen::Phase{pm.index}, true, {wn}, {wv}>({seq(pm)}, ka.tile_rec + blk, ka, false, 0, blk); return;')
                else:
                    parts.append(f'    case {wv}: pc::bulk<gen::Phase{pm.index}, {int(orders[pm.index])}, true, {wn}, {wv}>(a, false, 0, blk, &ld); return;')
            parts.append('    default: return;')
            parts.append('  }')
            parts.append('}')
            parts.append('#endif')
    else:
        np_ = len(model.phases)

        def all_body(res: bool, wn: int = 0):
            """wn = 0: replica index at run time (any waves-per-tile count); wn > 0: one instantiation of the tile body
            per replica, for launches with exactly wn waves per tile."""
            out = []
            blk = "(int)blockIdx.x - tail_blocks" if res else "(int)blockIdx.x"
            out.append(f"  const int b = pc::xcd_major({blk}, fb{np_});")
            rs = 'true' if res else 'false'
            for i, pm in enumerate(model.phases):
                cond = f"if (b < fb{i + 1}) " if i + 1 < np_ else ""
                args = f"(ph[{i}], true, fb{i}, b, nullptr, x, lam, c, G, H, flags, epoch)"
                margs = f"({seq(pm)}, trec + b, ph[{i}], true, fb{i}, b, nullptr, x, lam, c, G, H, flags, epoch)" if mixed[pm.index] else ""

                def call(wn_, wv_):
                    if mixed[pm.index]:
                        return f"pc::bulk_mix<gen::Phase{pm.index}, {rs}, {wn_}, {wv_}>{margs}"
                    return f"pc::bulk<gen::Phase{pm.index}, {int(orders[pm.index])}, {rs}, {wn_}, {wv_}>{args}"
                if wn > 0:
                    out.append(f"  {cond}{{ switch (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6)) {{")
                    for wv in range(wn):
                        out.append(f"    case {wv}: {call(wn, wv)}; return;")
                    out.append("    default: return; } }")
                else:
                    out.append(f"  {cond}{{ {call(0, 0)}; return; }}")
            return out
        parts.append("// every phase in one launch: a workgroup finds its phase from the cumulative block counts")
        # The members of PcMultiArgs travel as separate scalar parameters (same order, same offsets: the host still
        # hands over the struct) and reach pc::bulk as values: as ONE by-value struct whose address is passed on, a
        # large kernel keeps it in scratch memory (seen: 112 B per lane, a scratch set-up on every wave of the launch).
        multi_sig = ("const double* x, const double* lam, double* c, double* G, double* H, const PcPhaseArgs* ph, int flags, "
                     "int n_phases, " + ", ".join(f"int fb{i}" for i in range(9)) + ", unsigned epoch, int tail_blocks, const PcTileRec* trec")
        parts.append("static_assert(PC_MAX_PHASES + 1 == 9, \"pc_bulk_all spells PcMultiArgs::first_block out\");")
        hva = _heavy_attr(heavy_any, heavy_cap)
        parts.append(f'#if PC_PART < 0 || PC_PART == {1}')
        parts.append('extern "C" __global__ void __launch_bounds__(256) ' + occ + hva + 'pc_bulk_all(' + multi_sig + ') {')
        parts += all_body(False)
        parts.append("}")
        parts.append('#endif')
        parts.append("// the same with the resident tail as block 0")
        parts.append(f'#if PC_PART < 0 || PC_PART == {2}')
        parts.append('extern "C" __global__ void __launch_bounds__(256) ' + occ + hva + 'pc_bulk_all_r(' + multi_sig + ', PcTailArgs t) {')
        parts.append('  if ((int)blockIdx.x < tail_blocks) { gen::Tail::run<true>(t, nullptr, (int)((sizeof(PcMultiArgs) + 7) & ~7), (int)blockIdx.x, tail_blocks); return; }')
        parts += all_body(True)
        parts.append("}")
        parts.append('#endif')
        # replica index compiled in (see pc_bulk_p<i>_r_w<W> above): W = 2 always -- for heavy models this is the two-wave
        # build --, W = 4 when every phase is light or medium
        multi_ws = sorted(set.intersection(*[set(_static_w_list(pm)) for pm in model.phases]))
        for wn in multi_ws:
            parts.append(f'#if PC_PART < 0 || PC_PART == {3 + multi_ws.index(wn)}')
            parts.append(f'extern "C" __global__ void __launch_bounds__(256) {occ}{hva}pc_bulk_all_r_w{wn}(' + multi_sig + ', PcTailArgs t) {')
            parts.append('  if ((int)blockIdx.x < tail_blocks) { gen::Tail::run<true>(t, nullptr, (int)((sizeof(PcMultiArgs) + 7) & ~7), (int)blockIdx.x, tail_blocks); return; }')
            parts += all_body(True, wn)
            parts.append("}")
            parts.append('#endif')
    parts.append("")
    parts.append("#if PC_PART <= 0")
    for pm in model.phases:
        parts.append(f'extern "C" __global__ void __launch_bounds__(256) pc_mesh_err_p{pm.index}(PcRefineArgs a) '
                     f'{{ pc::mesh_error<gen::Phase{pm.index}>(a); }}')
    parts.append('extern "C" __global__ void __launch_bounds__(PC_TAIL_THREADS) pc_tail(const double* x, const double* partials0, '
                 'const double* scal0, long long x_off0, int n_tiles0, int N0, int flags, int block_threads, PcTailArgs a) {')
    parts.append('  const PcTailLead ld{x, partials0, scal0, x_off0, n_tiles0, N0, flags, block_threads};')
    parts.append('  gen::Tail::run<false, false>(a, &ld);')
    parts.append('}')
    parts.append('// the same for many tiles: several strides of partial sums in flight per lane')
    parts.append('extern "C" __global__ void __launch_bounds__(PC_TAIL_THREADS) pc_tail_big(const double* x, const double* partials0, '
                 'const double* scal0, long long x_off0, int n_tiles0, int N0, int flags, int block_threads, PcTailArgs a) {')
    parts.append('  const PcTailLead ld{x, partials0, scal0, x_off0, n_tiles0, N0, flags, block_threads};')
    parts.append('  gen::Tail::run<false, true>(a, &ld);')
    parts.append('}')
    parts.append("#endif")
    parts.append("")
    return "\n".join(parts)


def n_parts(model: Model) -> int:
    """Parts a model's code object is compiled in (generate_source: PC_PART): 1 unless a phase is heavy."""
    if not any(is_heavy(pm) for pm in model.phases) or os.environ.get("PYCOLLO_AMD_SPLIT_BUILD", "1") == "0":
        return 1
    if len(model.phases) == 1:
        return 2 + len(_static_w_list(model.phases[0], single_phase=True))
    return 3 + len(sorted(set.intersection(*[set(_static_w_list(pm)) for pm in model.phases])))


def part_paths(code_object: str) -> list[str]:
    """The modules of a code object: the file itself and its siblings <base>.p<k>.hsaco (pc_create loads them all)."""
    out, k = [code_object], 1
    while os.path.exists(f"{code_object[:-6]}.p{k}.hsaco"):
        out.append(f"{code_object[:-6]}.p{k}.hsaco")
        k += 1
    return out


def _waves_per_eu() -> int:
    """Experiment knob: PYCOLLO_AMD_WAVES_PER_EU=3|4 asks the compiler to fit the bulk kernels into the VGPR budget
    of that many waves per SIMD (168 / 128 registers), spilling if it must.  0 (default) = no constraint."""
    try:
        return int(os.environ.get("PYCOLLO_AMD_WAVES_PER_EU", "0"))
    except ValueError:
        return 0


def _extra_defines() -> list[str]:
    """Experiment knob: PYCOLLO_AMD_DEFINES="PC_FLUSH_DEPTH=8 PC_PIN_BUDGET=0" adds -D flags to the code-object build."""
    return [d for d in os.environ.get("PYCOLLO_AMD_DEFINES", "").split() if d]


HEAVY_SCRATCH_LIMIT = int(os.environ.get("PYCOLLO_AMD_HEAVY_SCRATCH_LIMIT", "384"))   # bytes per lane a capped heavy kernel may spill (env: experiments)


def _heavy_cap_enabled() -> bool:
    """PYCOLLO_AMD_HEAVY_CAP=0: never cap the registers of heavy models' kernels (A/B knob)."""
    return os.environ.get("PYCOLLO_AMD_HEAVY_CAP", "1") != "0" and _waves_per_eu() == 0


def _heavy_attr(heavy: bool, cap: bool) -> str:
    """Heavy models' tile kernels are compiled for two waves per SIMD (``amdgpu_waves_per_eu(2)``: at most 256
    registers; what does not fit is spilled -- Delta III, order 4: 266 -> 256 registers + 52 B of scratch per lane,
    29.5 -> 25.9 us at 4 x 12.5 k nodes).  ``build_code_object`` drops the cap again when a kernel spills more than
    HEAVY_SCRATCH_LIMIT bytes (space station: 472 registers uncapped)."""
    return "__attribute__((amdgpu_waves_per_eu(2))) " if (heavy and cap) else ""


def _occupancy_attr() -> str:
    w = _waves_per_eu()
    return f"__attribute__((amdgpu_waves_per_eu({w},{w}))) " if w > 0 else ""


def _fp_contract() -> str:
    """Experiment knob PYCOLLO_AMD_FP_CONTRACT: 'off' (default, see build_code_object) or 'fast'."""
    v = os.environ.get("PYCOLLO_AMD_FP_CONTRACT", "off").strip().lower()
    return v if v in ("off", "fast", "on") else "off"


def _preload_count() -> int:
    """Leading kernel arguments the command processor preloads into SGPRs (pc_bulk_p<i>: the ten of PcLead = 14
    dwords, all the user SGPRs there are next to the kernarg pointer).  PYCOLLO_AMD_PRELOAD=0 turns it off."""
    try:
        return max(0, int(os.environ.get("PYCOLLO_AMD_PRELOAD", "10")))
    except ValueError:
        return 10


def _kernels_stamp() -> str:
    h = hashlib.sha256()
    for fn in ("pc_kernels.hpp", "pc_args.h"):
        with open(os.path.join(CSRC, fn), "rb") as f:
            h.update(f.read())
    with open(os.path.abspath(__file__), "rb") as f:   # the generator itself: printer rules, kernel entry points
        h.update(f.read())
    h.update(b"flags:-O3 -ffp-contract=off")
    return h.hexdigest()[:12]


def hipcc_path() -> str | None:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _orders_tag(model: Model, orders, mixed=None) -> str:
    orders = tuple(int(o) for o in orders) if orders is not None else tuple(0 for _ in model.phases)
    if len(orders) != len(model.phases):
        raise ValueError("one section order (or 0) per phase is required")
    tag = "n" + "_".join(str(o) for o in orders)
    mixed = _norm_mixed(model, orders, mixed)
    if any(mixed):
        tag += "-m" + "_".join(".".join(str(n) for n in m) if m else "x" for m in mixed)
    return tag


def code_object_path(model: Model, orders=None, mixed=None) -> str:
    occ = f"_w{_waves_per_eu()}" if _waves_per_eu() > 0 else ""
    if _fp_contract() != "off":
        occ += "_fc" + _fp_contract()
    if _preload_count() != 10:
        occ += f"_pl{_preload_count()}"
    if not _heavy_cap_enabled() and _waves_per_eu() == 0:
        occ += "_nocap"
    if not _heavy_w4_enabled():
        occ += "_nohw4"
    if HEAVY_SCRATCH_LIMIT != 384:          # (decides whether a heavy model's kernels keep their register cap)
        occ += f"_sl{HEAVY_SCRATCH_LIMIT}"
    if _extra_defines():
        occ += "_d" + hashlib.sha256(" ".join(_extra_defines()).encode()).hexdigest()[:8]
    return os.path.join(CACHE, f"model_{model.digest}_{_kernels_stamp()}_{_orders_tag(model, orders, mixed)}{occ}.hsaco")


def build_code_object(model: Model, orders=None, force: bool = False, verbose: bool = False, mixed=None) -> str:
    """Return the path of the gfx950 code object for ``model``, compiling it if it is not cached.

    ``orders[p] = n > 0`` specialises phase p's kernel for meshes whose sections all have n nodes;
    0 keeps it generic (any mesh)."""
    os.makedirs(CACHE, exist_ok=True)
    out = code_object_path(model, orders, mixed)
    if os.path.exists(out) and not force:
        return out
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError(f"code object {out} is not built and hipcc is not available to build it")
    # Under rocprofv3 every child inherits the profiler's preload, which initialises the GPU; hipcc then execs clang --
    # an exec from a process that holds the GPU, which must not happen on this pool.  Build beforehand instead.
    preload = os.environ.get("LD_PRELOAD", "")
    if "rocprof" in preload or any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ):
        raise RuntimeError(f"code object {out} is not built and this process runs under a profiler preload: build it "
                           f"first in a plain process (python -c 'import __graft_entry__ as g; g.build()' or one "
                           f"unprofiled run of the same command)")
    src = out[:-6] + ".hip"
    import json
    tmp_tag = f".tmp{os.getpid()}"          # (per process: two processes may build the same object side by side)
    tmp_out = out + tmp_tag
    # Heavy models: first with their tile kernels capped at two waves per SIMD (_heavy_attr); when a capped kernel
    # spills more than HEAVY_SCRATCH_LIMIT bytes per lane the model does not fit (space station) and the object is
    # built again without the cap.
    attempts = [True, False] if (any(is_heavy(pm) for pm in model.phases) and _heavy_cap_enabled()) else [False]
    nparts = n_parts(model)
    part_out = [out] + [f"{out[:-6]}.p{k}.hsaco" for k in range(1, nparts)]
    for cap in attempts:
        # (the source of an attempt is per process: another process may be compiling the other attempt of the same object)
        with open(src + tmp_tag, "w") as f:
            f.write(generate_source(model, orders, heavy_cap=cap, mixed=mixed))
        # -ffp-contract=off: no fused multiply-add, so V*x~ + r and every model expression round exactly like
        # the reference's CasADi / NumPy arithmetic (tests/unit/test_iteration.py:302 asserts J == 100 exactly)
        base = [hipcc, f"--offload-arch={ARCH}", "--genco", "-O3", "-std=c++17", f"-ffp-contract={_fp_contract()}",
                "-mllvm", f"-amdgpu-kernarg-preload-count={_preload_count()}", "-Rpass-analysis=kernel-resource-usage",
                f"-I{CSRC}", "-x", "hip"] + [f"-D{d}" for d in _extra_defines()]
        # a heavy model's object is compiled in parts (generate_source: PC_PART), side by side: one module per launch kernel
        procs = [subprocess.Popen(base + [f"-DPC_PART={k if nparts > 1 else -1}", "-o", part_out[k] + tmp_tag, src + tmp_tag],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(nparts)]
        resources = {}
        for pr in procs:
            _, err = pr.communicate()
            if pr.returncode != 0:
                for q in procs:
                    if q.poll() is None:
                        q.kill()
                raise RuntimeError(f"hipcc failed for {src}:\n{err[-4000:]}")
            if verbose:
                print(err)
            resources.update(_parse_resources(err))
        worst = max((k.get("scratch", 0) for name, k in resources.items() if name.startswith("pc_bulk")), default=0)
        if not cap or worst <= HEAVY_SCRATCH_LIMIT:
            break
    os.replace(src + tmp_tag, src)               # (kept for inspection: tools/isa_lines.py)
    # what the compiler made of every kernel travels with the object: pc_create's launch shape depends on it (a
    # two-wave build that did not fit 256 registers must not be launched as one)
    resources["_build"] = {"heavy_cap": bool(cap), "parts": nparts, "scratch_limit": HEAVY_SCRATCH_LIMIT}
    with open(resources_path(out) + tmp_tag, "w") as f:
        json.dump(resources, f, indent=1, sort_keys=True)
    os.replace(resources_path(out) + tmp_tag, resources_path(out))
    for k in range(nparts - 1, -1, -1):          # the object itself last: its presence says the build is complete
        os.replace(part_out[k] + tmp_tag, part_out[k])
    return out


def resources_path(code_object: str) -> str:
    return code_object[:-6] + ".json" if code_object.endswith(".hsaco") else code_object + ".json"


def _parse_resources(remarks: str) -> dict:
    """{kernel: {"vgprs", "agprs", "sgprs", "scratch", "occupancy"}} from -Rpass-analysis=kernel-resource-usage."""
    import re
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("agprs", r" AGPRs: (\d+)"), ("sgprs", r"TotalSGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return out


def code_object_resources(code_object: str) -> dict:
    """The sidecar written by :func:`build_code_object` ({} when absent: an object built by an older generator)."""
    import json
    try:
        with open(resources_path(code_object)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def two_wave_occupancy(model: Model, code_object: str) -> int:
    """Waves per SIMD (by registers) of the code object's two-wave launch kernel -- ``pc_bulk_all_r_w2`` /
    ``pc_bulk_p<i>_r_w2`` of a model with a heavy phase --, 0 when the object has none or the model is not heavy."""
    if not any(is_heavy(pm) for pm in model.phases):
        return 0
    res = code_object_resources(code_object)
    name = "pc_bulk_all_r_w2" if len(model.phases) > 1 else f"pc_bulk_p{model.phases[0].index}_r_w2"
    k = res.get(name)
    if not k or k.get("scratch", HEAVY_SCRATCH_LIMIT + 1) > HEAVY_SCRATCH_LIMIT:
        return 0
    return int(k.get("occupancy", 0))


def kernel_resources(model: Model, orders=None, mixed=None) -> dict:
    """Registers and scratch memory of every kernel of the model's code object, as hipcc reports them
    (``-Rpass-analysis=kernel-resource-usage``): {kernel: {"vgprs", "sgprs", "scratch", "occupancy"}}.  A bulk kernel
    with scratch memory is a bug of the templates (an array subscripted by a run-time value), not a tuning matter:
    tests assert it stays at zero."""
    import re
    import tempfile
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc is not available")
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "m.hip")
        with open(src, "w") as f:
            f.write(generate_source(model, orders, mixed=mixed))
        cmd = [hipcc, f"--offload-arch={ARCH}", "--genco", "-O3", "-std=c++17", f"-ffp-contract={_fp_contract()}",
               "-mllvm", f"-amdgpu-kernarg-preload-count={_preload_count()}", "-Rpass-analysis=kernel-resource-usage",
               f"-I{CSRC}", "-o", os.path.join(tmp, "m.hsaco"), src] + [f"-D{d}" for d in _extra_defines()]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{res.stderr[-4000:]}")
    return _parse_resources(res.stderr)
