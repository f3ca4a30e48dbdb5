"""ctypes binding of the C-ABI engine + the Python-side mirror of pycollo's callback surfaces.

Two reference surfaces are mirrored:

* the legacy cyipopt object protocol ``IPOPTProblem`` (pycollo/nlp.py:36-76): ``objective``,
  ``gradient``, ``constraints``, ``jacobian``, ``jacobianstructure``, ``hessian``,
  ``hessianstructure``, ``intermediate`` -> :class:`PycolloGpuProblem`
* the live CasADi probes ``evaluate_J/g/c/G/G_nonzeros/G_structure/G_num_nonzero`` and the
  (unimplemented in the reference, implemented here) ``evaluate_H*`` (pycollo/backend.py:1713-1805)
  -> the same methods on :class:`NlpEngine`

There is no CPU fallback: without the HIP library, the code object or a GPU every evaluation raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.sparse as sparse

from . import codegen
from .layout import NlpLayout
from .mesh import PhaseMesh, build_phase_mesh
from .model import Model, compile_model
from .problem import ProblemSpec
from .quadrature import QuadratureTables

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpycollo_amd.so")

_KIND = {"y0": 0, "yF": 1, "q": 2, "t0": 3, "tF": 4, "s": 5}

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


class _PhaseDesc(C.Structure):
    _fields_ = [("n_y", C.c_int32), ("n_u", C.c_int32), ("n_q", C.c_int32), ("n_p", C.c_int32),
                ("t0_free", C.c_int32), ("tF_free", C.c_int32),
                ("t0_fixed", C.c_double), ("tF_fixed", C.c_double),
                ("K", C.c_int32), ("n_k", _i32p), ("h_k", _f64p),
                ("n_jac", C.c_int32), ("jac_row", _i32p), ("jac_col", _i32p),
                ("n_hess", C.c_int32), ("hess_row", _i32p), ("hess_col", _i32p),
                ("bulk_kernel", C.c_char_p), ("compiled_order", C.c_int32), ("n_edge_rec", C.c_int32 * 2),
                ("eval_ops", C.c_int32), ("n_w", C.c_int32), ("w_kind", _i32p), ("w_idx", _i32p),
                ("n_spec", C.c_int32), ("spec_orders", C.c_int32 * 4), ("reserved_spec", C.c_int32 * 3),
                ("n_fixed_tiles", C.c_int32), ("reserved_tiles", C.c_int32), ("fixed_tile_k0", _i32p), ("fixed_tile_order", _i32p)]


class _ProblemDesc(C.Structure):
    _fields_ = [("n_phases", C.c_int32), ("phases", C.POINTER(_PhaseDesc)), ("n_s", C.c_int32),
                ("n_point", C.c_int32), ("point_phase", _i32p), ("point_kind", _i32p), ("point_idx", _i32p),
                ("n_b", C.c_int32),
                ("n_jgrad", C.c_int32), ("jgrad_col", _i32p),
                ("n_bjac", C.c_int32), ("bjac_row", _i32p), ("bjac_col", _i32p),
                ("n_pthess", C.c_int32), ("pthess_row", _i32p), ("pthess_col", _i32p),
                ("n_orders", C.c_int32), ("orders", _i32p), ("quad_A", _f64p), ("quad_w", _f64p),
                ("code_object", C.c_char_p), ("tail_kernel", C.c_char_p),
                ("device", C.c_int32), ("threads_per_block", C.c_int32), ("two_wave_occupancy", C.c_int32),
                ("plan_only", C.c_int32)]


class _Info(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("nnz_jac", C.c_int64), ("nnz_hess", C.c_int64),
                ("algorithmic_bytes", C.c_int64), ("n_tiles_total", C.c_int32), ("threads_per_block", C.c_int32),
                ("lds_bytes_max", C.c_int32), ("n_launches", C.c_int32), ("waves_per_tile", C.c_int32),
                ("reserved", C.c_int32)]


_lib = None


def load_library() -> C.CDLL:
    """Load libpycollo_amd.so (built in-tree by ``__graft_entry__.build``); fail loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(hipcc required).  pycollo_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.
    # Importing torch first makes the dynamic loader resolve this library's libamdhip64.so.7 to the copy
    # torch already mapped; the other order leaves two HSA runtimes fighting over the device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.pc_last_error.restype = C.c_char_p
    lib.pc_create.argtypes = [C.POINTER(_ProblemDesc), C.POINTER(vp)]
    lib.pc_destroy.argtypes = [vp]
    lib.pc_destroy.restype = None
    lib.pc_get_info.argtypes = [vp, C.POINTER(_Info)]
    lib.pc_sizes.argtypes = [vp, _i32p, _i32p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.pc_jac_structure.argtypes = [vp, vp, vp]
    lib.pc_hess_structure.argtypes = [vp, vp, vp]
    lib.pc_set_scaling.argtypes = [vp, vp, vp, vp, C.c_double]
    lib.pc_eval_f.argtypes = [vp, vp, C.c_int, vp]
    lib.pc_eval_grad_f.argtypes = [vp, vp, C.c_int, vp]
    lib.pc_eval_g.argtypes = [vp, vp, C.c_int, vp]
    lib.pc_eval_jac_g.argtypes = [vp, vp, C.c_int, vp]
    lib.pc_eval_h.argtypes = [vp, vp, C.c_int, C.c_double, vp, C.c_int, vp]
    lib.pc_eval_all.argtypes = [vp, vp, C.c_double, vp, vp, vp, vp]
    lib.pc_eval_resident.argtypes = [vp, vp, C.c_double, vp, vp, vp, vp]
    lib.pc_host_buffers.argtypes = [vp] + [C.POINTER(vp)] * 5
    lib.pc_set_host_mode.argtypes = [vp, C.c_int]
    lib.pc_set_prefetch_jac.argtypes = [vp, C.c_int]
    lib.pc_eval_all_device.argtypes = [vp, vp, C.c_double, vp, vp, vp, vp, vp]
    lib.pc_launch_bulk_device.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.pc_launch_tail_device.argtypes = [vp, vp, C.c_double, vp, vp, vp, vp, vp]
    lib.pc_launch_bulk_flags_device.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp]
    lib.pc_set_tile_range.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.pc_phase_tiles.argtypes = [vp, C.c_int, _i32p, _i32p, vp]
    lib.pc_phase_tile_orders.argtypes = [vp, C.c_int, vp]
    lib.pc_set_partials_buffer.argtypes = [vp, C.c_int, vp]
    lib.pc_synchronize.argtypes = [vp]
    lib.pc_check.argtypes = [vp]
    lib.pc_read_symbol.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.pc_row_norms_jac.argtypes = [vp, vp, vp]
    lib.pc_interp_linear.argtypes = [C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp]
    lib.pc_copy_runs.argtypes = [vp, vp, vp, C.c_int64, vp]
    lib.pc_run_chunk.argtypes = []
    lib.pc_mesh_error.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.pc_stream.argtypes = [vp]
    lib.pc_stream.restype = vp
    _lib = lib
    return lib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(-1))


MIX_MIN_RUN_ROWS = 24      # pc_pattern.hpp Phase::min_run_rows: a run of equal sections gets tiles of its own from here
MIX_MIN_NODES = int(os.environ.get("PYCOLLO_AMD_MIX_MIN_NODES", "20000"))   # smaller NLPs keep the any-order kernel


def choose_spec_orders(n_k, max_orders: int = 4, min_share: float = 0.05) -> tuple:
    """Section orders worth a tile body of their own on a mesh of mixed orders: those whose sections come in runs
    (>= MIX_MIN_RUN_ROWS defect rows of equal sections in a row, pc_pattern.hpp::build_tiles_mixed) covering at least
    ``min_share`` of the phase's rows; at most ``max_orders`` of them, by coverage.  ph refinement leaves such meshes:
    subdivided and merged stretches are runs of the minimum order (pycollo/mesh_refinement.py:252-321)."""
    n_k = np.asarray(n_k, dtype=np.int64)
    if n_k.size == 0 or np.all(n_k == n_k[0]):
        return ()
    cut = np.flatnonzero(np.diff(n_k)) + 1
    starts = np.concatenate([[0], cut])
    lengths = np.diff(np.concatenate([starts, [n_k.size]]))
    order = n_k[starts]
    rows = lengths * (order - 1)
    total = float(np.sum(n_k - 1))
    cover = {}
    for o, r in zip(order, rows):
        if r >= MIX_MIN_RUN_ROWS:
            cover[int(o)] = cover.get(int(o), 0) + int(r)
    picks = sorted((o for o, r in cover.items() if r >= min_share * total), key=lambda o: -cover[o])[:max_orders]
    return tuple(sorted(picks))


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a.size else C.cast(None, typ)


def interp_linear(tau_prev, vals_prev, tau_new, device: int = 0) -> np.ndarray:
    """Rows of ``vals_prev`` carried from ``tau_prev`` to ``tau_new`` on the GPU (pycollo/iteration.py:96-137)."""
    lib = load_library()
    tp = np.ascontiguousarray(tau_prev, dtype=np.float64)
    vp_ = np.ascontiguousarray(np.atleast_2d(vals_prev), dtype=np.float64)
    tn = np.ascontiguousarray(tau_new, dtype=np.float64)
    if vp_.shape[1] != tp.shape[0]:
        raise ValueError("vals_prev must have one column per previous abscissa")
    out = np.empty((vp_.shape[0], tn.shape[0]))
    if vp_.shape[0] == 0:
        return out
    if not lib.pc_interp_linear(int(device), tp.ctypes.data, tp.shape[0], vp_.ctypes.data, vp_.shape[0], tn.ctypes.data,
                                tn.shape[0], out.ctypes.data):
        raise RuntimeError(lib.pc_last_error().decode())
    return out


class NlpEngine:
    """One transcribed NLP (model x meshes x scaling) bound to one GPU (or structure-only)."""

    def __init__(self, problem: ProblemSpec | Model, meshes: list[PhaseMesh] | None = None, *, device: int | None = 0,
                 threads_per_block: int = 0, quad: QuadratureTables | None = None, build: bool = True,
                 specialise: bool = True, mixed="auto", orders=None, fixed_tiles=None, plan_only: bool = False,
                 plan_device: int = -1):
        """``orders`` / ``mixed``: the build to use instead of the one chosen from the meshes (a rank-local handle runs the
        global handle's build on its part of the mesh).  ``fixed_tiles``: per phase ``(tile_k0, tile_order or None)`` --
        the caller's tile table (``pc_phase_desc.fixed_tile_k0``).  ``plan_only``: cut the tiles and stop
        (``phase_tiles`` / ``phase_tile_orders`` / ``info`` work; no patterns, no device memory); ``plan_device``: the
        device whose LDS limit the plan assumes (-1: the 64 KiB default of a structure-only handle)."""
        self.model = problem if isinstance(problem, Model) else compile_model(problem)
        self.quad = quad or QuadratureTables(self.model.quadrature_method)
        if meshes is None:
            if isinstance(problem, Model):
                raise ValueError("meshes are required when a compiled Model is passed")
            meshes = [build_phase_mesh(self.quad, *ph.mesh.resolved()) for ph in problem.phases]
        self.meshes = meshes
        self.layout = NlpLayout(self.model, meshes)
        self.device = -1 if device is None else int(device)
        self._lib = load_library()
        self._h = C.c_void_p()
        self._keep = []
        code_object = None
        # kernels are specialised for the section order of every phase whose mesh has a single order
        self.orders = tuple(int(m.n[0]) if np.all(m.n == m.n[0]) else 0 for m in meshes)
        if not specialise:
            self.orders = tuple(0 for _ in meshes)
        if orders is not None:
            self.orders = tuple(int(o) for o in orders)
        self._fixed_tiles = fixed_tiles
        self._plan_only = bool(plan_only)
        wants_device = self.device >= 0 or (plan_only and plan_device >= 0)
        # ... and a phase whose sections differ in order gets the mixed build when its orders come in runs (large NLPs
        # only: a code object per set of orders is not worth compiling for a mesh of a few hundred nodes).
        # mixed: "auto" | None | one tuple of orders per phase
        if mixed == "auto":
            big = sum(int(np.sum(m.n - 1)) + 1 for m in meshes) >= MIX_MIN_NODES
            mixed = (tuple(choose_spec_orders(m.n) if o == 0 else () for m, o in zip(meshes, self.orders))
                     if (specialise and big and wants_device) else None)
        self.mixed = (tuple(tuple(int(n) for n in mm) for mm in mixed) if mixed is not None and any(mixed)
                      else tuple(() for _ in meshes))
        if self.device >= 0:
            code_object = (codegen.build_code_object(self.model, self.orders, mixed=self.mixed) if build
                           else codegen.code_object_path(self.model, self.orders, self.mixed))
            if not os.path.exists(code_object):
                raise RuntimeError(f"code object {code_object} is missing")
        if plan_only and plan_device >= 0:   # the plan of a device handle: its code object's register figures decide the launch shape
            code_object = codegen.code_object_path(self.model, self.orders, self.mixed)
            if not os.path.exists(code_object):
                code_object = codegen.build_code_object(self.model, self.orders, mixed=self.mixed) if build else None
        self.code_object = code_object
        desc = self._make_desc(code_object, threads_per_block)
        if plan_only:
            desc.plan_only = 1
            desc.device = int(plan_device)
            desc.code_object = None
        if not self._lib.pc_create(C.byref(desc), C.byref(self._h)):
            raise RuntimeError("pc_create failed: " + self._lib.pc_last_error().decode())
        info = _Info()
        self._check(self._lib.pc_get_info(self._h, C.byref(info)))
        self.info = {k: getattr(info, k) for k, _ in _Info._fields_}
        if plan_only:
            return
        self.num_x, self.num_c = info.n, info.m
        self.nnz_jac, self.nnz_hess = int(info.nnz_jac), int(info.nnz_hess)
        if (self.num_x, self.num_c) != (self.layout.num_x, self.layout.num_c):
            raise RuntimeError("layout mismatch between host library and Python bookkeeping")
        self._jac_struct = None
        self._hess_struct = None
        # default scaling: base variable scaling from bounds, base constraint scaling, w_J = 1
        V, r = self.layout.base_variable_scaling()
        self.set_scaling(V, r, self.layout.base_constraint_scaling(V), 1.0)

    # ---- descriptor ----------------------------------------------------------------------------
    def _make_desc(self, code_object, tpb) -> _ProblemDesc:
        m = self.model
        keep = self._keep
        phases = (_PhaseDesc * len(m.phases))()
        for i, (pm, mesh) in enumerate(zip(m.phases, self.meshes)):
            n_k = _i32(mesh.n)
            h_k = np.ascontiguousarray(mesh.h, dtype=np.float64)
            jr, jc = _i32([r for r, _, _ in pm.jac]), _i32([c for _, c, _ in pm.jac])
            hr, hc = _i32([r for r, _, _ in pm.hess]), _i32([c for _, c, _ in pm.hess])
            keep += [n_k, h_k, jr, jc, hr, hc]
            d = phases[i]
            d.n_y, d.n_u, d.n_q, d.n_p = pm.n_y, pm.n_u, pm.n_q, pm.n_p
            d.t0_free, d.tF_free = int(pm.t_free[0]), int(pm.t_free[1])
            d.t0_fixed, d.tF_fixed = float(pm.t_fixed[0]), float(pm.t_fixed[1])
            d.K = mesh.K
            d.n_k, d.h_k = _ptr(n_k, _i32p), _ptr(h_k, _f64p)
            d.n_jac, d.jac_row, d.jac_col = len(jr), _ptr(jr, _i32p), _ptr(jc, _i32p)
            d.n_hess, d.hess_row, d.hess_col = len(hr), _ptr(hr, _i32p), _ptr(hc, _i32p)
            d.bulk_kernel = f"pc_bulk_p{pm.index}".encode()
            d.compiled_order = self.orders[i] if (self.device >= 0 or self._plan_only) else 0
            if self._fixed_tiles is not None and self._fixed_tiles[i] is not None:
                tk, to = self._fixed_tiles[i]
                tk = _i32(tk)
                keep.append(tk)
                d.n_fixed_tiles, d.fixed_tile_k0 = len(tk) - 1, _ptr(tk, _i32p)
                if to is not None:
                    to = _i32(to)
                    keep.append(to)
                    d.fixed_tile_order = _ptr(to, _i32p)
            d.n_spec = len(self.mixed[i])
            for j, n in enumerate(self.mixed[i]):
                d.spec_orders[j] = n
            d.eval_ops = pm.eval_ops
            if pm.w_kind:
                wk, wi = _i32(pm.w_kind), _i32(pm.w_idx)
                keep += [wk, wi]
                d.n_w, d.w_kind, d.w_idx = len(wk), _ptr(wk, _i32p), _ptr(wi, _i32p)
            fl = codegen.edge_flags(m, pm)
            d.n_edge_rec[0], d.n_edge_rec[1] = sum(fl[:len(fl) // 2]), sum(fl[len(fl) // 2:])
        pt = m.point
        pp = _i32([v.phase for v in pt.vars])
        pk = _i32([_KIND[v.kind] for v in pt.vars])
        pi = _i32([v.idx for v in pt.vars])
        jg = _i32([c for c, _ in pt.J_grad])
        br, bc = _i32([r for r, _, _ in pt.b_jac]), _i32([c for _, c, _ in pt.b_jac])
        phr, phc = _i32([r for r, _, _ in pt.hess]), _i32([c for _, c, _ in pt.hess])
        orders = sorted({int(n) for mesh in self.meshes for n in np.unique(mesh.n)})
        od = _i32(orders)
        qa = np.ascontiguousarray(np.concatenate([self.quad.A(n).ravel() for n in orders]), dtype=np.float64)
        qw = np.ascontiguousarray(np.concatenate([self.quad.weights(n).ravel() for n in orders]), dtype=np.float64)
        keep += [phases, pp, pk, pi, jg, br, bc, phr, phc, od, qa, qw]
        desc = _ProblemDesc()
        desc.n_phases, desc.phases, desc.n_s = len(m.phases), phases, m.n_s
        desc.n_point, desc.point_phase, desc.point_kind, desc.point_idx = len(pt.vars), _ptr(pp, _i32p), _ptr(pk, _i32p), _ptr(pi, _i32p)
        desc.n_b = len(pt.b)
        desc.n_jgrad, desc.jgrad_col = len(jg), _ptr(jg, _i32p)
        desc.n_bjac, desc.bjac_row, desc.bjac_col = len(br), _ptr(br, _i32p), _ptr(bc, _i32p)
        desc.n_pthess, desc.pthess_row, desc.pthess_col = len(phr), _ptr(phr, _i32p), _ptr(phc, _i32p)
        desc.n_orders, desc.orders, desc.quad_A, desc.quad_w = len(orders), _ptr(od, _i32p), _ptr(qa, _f64p), _ptr(qw, _f64p)
        desc.code_object = code_object.encode() if code_object else None
        desc.tail_kernel = b"pc_tail"
        desc.device = self.device
        desc.threads_per_block = int(tpb)
        desc.two_wave_occupancy = codegen.two_wave_occupancy(m, code_object) if code_object else 0
        return desc

    def _check(self, ok):
        if not ok:
            raise RuntimeError(self._lib.pc_last_error().decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scaling -------------------------------------------------------------------------------
    def set_scaling(self, V_ocp, r_ocp, W_ocp, w_J: float = 1.0):
        """V, r per OCP variable; W per OCP constraint; objective weight (backend.py:1459-1463,1684-1689)."""
        self.V_ocp = np.ascontiguousarray(V_ocp, dtype=np.float64)
        self.r_ocp = np.ascontiguousarray(r_ocp, dtype=np.float64)
        self.W_ocp = np.ascontiguousarray(W_ocp, dtype=np.float64)
        self.w_J = float(w_J)
        if self.V_ocp.shape != (self.layout.num_ocp_x,) or self.r_ocp.shape != (self.layout.num_ocp_x,):
            raise ValueError(f"V_ocp / r_ocp must have {self.layout.num_ocp_x} entries")
        if self.W_ocp.shape != (self.layout.num_ocp_c,):
            raise ValueError(f"W_ocp must have {self.layout.num_ocp_c} entries")
        self._cached_x = None
        self._check(self._lib.pc_set_scaling(self._h, self.V_ocp.ctypes.data, self.r_ocp.ctypes.data,
                                             self.W_ocp.ctypes.data, self.w_J))

    # ---- structure -----------------------------------------------------------------------------
    def evaluate_G_structure(self):
        """(row_indices, col_indices), CSR row-major (backend.py:1747-1761 returns CCS order)."""
        if self._jac_struct is None:
            r = np.empty(self.nnz_jac, dtype=np.int32)
            c = np.empty(self.nnz_jac, dtype=np.int32)
            self._check(self._lib.pc_jac_structure(self._h, r.ctypes.data, c.ctypes.data))
            self._jac_struct = (r, c)
        return self._jac_struct

    def evaluate_H_structure(self):
        if self._hess_struct is None:
            r = np.empty(self.nnz_hess, dtype=np.int32)
            c = np.empty(self.nnz_hess, dtype=np.int32)
            self._check(self._lib.pc_hess_structure(self._h, r.ctypes.data, c.ctypes.data))
            self._hess_struct = (r, c)
        return self._hess_struct

    def evaluate_G_num_nonzero(self):
        return self.nnz_jac

    def evaluate_H_num_nonzero(self):
        return self.nnz_hess

    def csr_to_ccs_permutation(self):
        """Permutation p with ``values_ccs = values_csr[p]`` (CasADi column-major order, backend.py:1754-1761)."""
        r, c = self.evaluate_G_structure()
        return np.lexsort((r, c))

    # ---- evaluation (host pointers) ------------------------------------------------------------
    def _x(self, x, new_x=None):
        """``new_x``: True / False for the IPOPT-protocol callbacks (the point the library's cache describes is
        tracked in ``_cached_x``), None for every other entry point, after which no cached point is assumed."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        if x.shape[0] != self.num_x:
            raise ValueError(f"x must have {self.num_x} entries")
        if new_x is None:
            self._cached_x = None
        elif new_x:
            self._cached_x = x.copy()
        return x

    def cache_holds(self, x) -> bool:
        """True when the library's J / grad J / c~ / G~ cache was filled at exactly ``x`` by the IPOPT-protocol
        callbacks and nothing else has touched the handle since: the test a caller without IPOPT's ``new_x`` flag
        (cyipopt) needs before it may pass ``new_x = 0``."""
        c = getattr(self, "_cached_x", None)
        return c is not None and np.array_equal(np.asarray(x, dtype=np.float64).reshape(-1), c)

    def evaluate_J(self, x, new_x=True):
        x = self._x(x, bool(new_x))
        out = C.c_double()
        self._check(self._lib.pc_eval_f(self._h, x.ctypes.data, int(new_x), C.addressof(out)))
        return out.value

    def evaluate_g(self, x, new_x=True):
        x = self._x(x, bool(new_x))
        g = np.empty(self.num_x)
        self._check(self._lib.pc_eval_grad_f(self._h, x.ctypes.data, int(new_x), g.ctypes.data))
        return g

    def evaluate_c(self, x, new_x=True):
        x = self._x(x, bool(new_x))
        c = np.empty(self.num_c)
        self._check(self._lib.pc_eval_g(self._h, x.ctypes.data, int(new_x), c.ctypes.data))
        return c

    def evaluate_G_nonzeros(self, x, new_x=True):
        x = self._x(x, bool(new_x))
        v = np.empty(self.nnz_jac)
        self._check(self._lib.pc_eval_jac_g(self._h, x.ctypes.data, int(new_x), v.ctypes.data))
        return v

    def evaluate_G(self, x):
        r, c = self.evaluate_G_structure()
        return sparse.coo_matrix((self.evaluate_G_nonzeros(x), (r, c)), shape=(self.num_c, self.num_x))

    def evaluate_H_nonzeros(self, x, obj_factor, lagrange, new_x=True):
        x = self._x(x, bool(new_x))
        lam = np.ascontiguousarray(lagrange, dtype=np.float64).reshape(-1)
        if lam.shape[0] != self.num_c:
            raise ValueError(f"lagrange must have {self.num_c} entries")
        v = np.empty(self.nnz_hess)
        self._check(self._lib.pc_eval_h(self._h, x.ctypes.data, int(new_x), float(obj_factor), lam.ctypes.data, 1,
                                        v.ctypes.data))
        return v

    def evaluate_H(self, x, obj_factor, lagrange):
        r, c = self.evaluate_H_structure()
        return sparse.coo_matrix((self.evaluate_H_nonzeros(x, obj_factor, lagrange), (r, c)),
                                 shape=(self.num_x, self.num_x))

    def evaluate_all(self, x, obj_factor, lagrange):
        """Fused c, G values, H values at one point (one launch sequence, one round trip)."""
        x = self._x(x)
        lam = np.ascontiguousarray(lagrange, dtype=np.float64).reshape(-1)
        c, g, h = np.empty(self.num_c), np.empty(self.nnz_jac), np.empty(self.nnz_hess)
        self._check(self._lib.pc_eval_all(self._h, x.ctypes.data, float(obj_factor), lam.ctypes.data, c.ctypes.data,
                                          g.ctypes.data, h.ctypes.data))
        return c, g, h

    # ---- zero-copy host hand-over ---------------------------------------------------------------
    def host_buffers(self):
        """NumPy views ``(x, lam, c, G, H)`` of the library's pinned staging blocks (``pc_host_buffers``).  Write x~ /
        lambda into the first two, call :meth:`evaluate_all_inplace`, read the results from the last three: no host
        memcpy on either side.  The views stay valid for the life of the engine; their contents until the next
        evaluation."""
        if getattr(self, "_host_views", None) is None:
            ptrs = [C.c_void_p() for _ in range(5)]
            self._check(self._lib.pc_host_buffers(self._h, *[C.byref(p) for p in ptrs]))
            sizes = (self.num_x, self.num_c, self.num_c, self.nnz_jac, self.nnz_hess)
            self._host_views = tuple(
                np.ctypeslib.as_array(C.cast(p, _f64p), shape=(max(n, 1),))[:n] for p, n in zip(ptrs, sizes))
        return self._host_views

    def evaluate_all_inplace(self, obj_factor=1.0):
        """Fused c, G, H at the (x~, lambda) already written into :meth:`host_buffers`; results are read there."""
        x, lam, c, g, h = self.host_buffers()
        self._cached_x = None
        self._check(self._lib.pc_eval_all(self._h, x.ctypes.data, float(obj_factor), lam.ctypes.data, c.ctypes.data,
                                          g.ctypes.data, h.ctypes.data))
        return c, g, h

    def set_host_mode(self, mode: int):
        """0: one DMA copy up / down; 1: kernels read x~, lambda from pinned host memory; 2: kernels write c~, G~, H~
        to pinned host memory; 3: both (``pc_set_host_mode``)."""
        self._cached_x = None
        self._check(self._lib.pc_set_host_mode(self._h, int(mode)))

    def evaluate_resident(self, x, obj_factor=1.0, lagrange=None, want_grad=True):
        """Evaluate and LEAVE c~, G~ (and H~ when ``lagrange`` is given) in device memory for the GPU KKT solver
        (``pycollo_amd.kkt.GpuKkt``); returns the small results ``(J, grad J or None, c~)``."""
        x = self._x(x)
        lam = None if lagrange is None else np.ascontiguousarray(lagrange, dtype=np.float64).reshape(-1)
        f = C.c_double()
        grad = np.empty(self.num_x) if want_grad else None
        c = np.empty(self.num_c)
        self._check(self._lib.pc_eval_resident(self._h, x.ctypes.data, float(obj_factor), None if lam is None else lam.ctypes.data,
                                               C.addressof(f), None if grad is None else grad.ctypes.data, c.ctypes.data))
        return f.value, grad, c

    def set_prefetch_jac(self, on: bool):
        """Whether a new point's G~ is copied to the host at once (``pc_set_prefetch_jac``)."""
        self._check(self._lib.pc_set_prefetch_jac(self._h, int(bool(on))))

    def G_row_norms(self, x):
        x = self._x(x)
        out = np.empty(self.num_c)
        self._check(self._lib.pc_row_norms_jac(self._h, x.ctypes.data, out.ctypes.data))
        return out

    # ---- evaluation (device pointers; torch tensors or raw addresses) ---------------------------
    def evaluate_all_device(self, d_x, obj_factor, d_lam, d_c, d_G, d_H, stream=None):
        def addr(t):
            return t.data_ptr() if hasattr(t, "data_ptr") else int(t)
        self._cached_x = None   # (the launch overwrites the handle's f block)
        self._check(self._lib.pc_eval_all_device(self._h, addr(d_x), float(obj_factor), addr(d_lam), addr(d_c),
                                                 addr(d_G), addr(d_H), stream))

    def bind_device(self, d_x, d_lam, d_c, d_G, d_H, stream=None):
        """``f(obj_factor)`` = :meth:`evaluate_all_device` on fixed device buffers, with the addresses resolved once.
        At 10 k nodes an evaluation is ~8 us; resolving five ``data_ptr()`` per call is a visible part of that."""
        def addr(t):
            return t.data_ptr() if hasattr(t, "data_ptr") else int(t)
        fn, h, err = self._lib.pc_eval_all_device, self._h, self._lib.pc_last_error
        px, pl, pc, pG, pH = addr(d_x), addr(d_lam), addr(d_c), addr(d_G), addr(d_H)
        keep = (d_x, d_lam, d_c, d_G, d_H)       # the buffers must outlive the callable

        def call(obj_factor=1.0, _keep=keep):
            self._cached_x = None
            if not fn(h, px, obj_factor, pl, pc, pG, pH, stream):
                raise RuntimeError("pc_eval_all_device failed: " + err().decode())
        return call

    def launch_bulk_only(self, d_x, d_lam, d_c, d_G, d_H, stream=None):
        """Profiling aid: only the bulk kernels of :meth:`evaluate_all_device`."""
        def addr(t):
            return t.data_ptr() if hasattr(t, "data_ptr") else int(t)
        self._cached_x = None   # (the launch overwrites the handle's f block)
        self._check(self._lib.pc_launch_bulk_device(self._h, addr(d_x), addr(d_lam), addr(d_c), addr(d_G), addr(d_H),
                                                    stream))

    def launch_bulk_flags(self, d_x, d_lam, d_c, d_G, d_H, flags: int, stream=None):
        """The tile kernels for a subset of the outputs: ``flags`` = 1 (c~) | 2 (G~) | 4 (H~)."""
        def addr(t):
            return t.data_ptr() if hasattr(t, "data_ptr") else int(t)
        self._cached_x = None   # (the launch overwrites the handle's f block)
        self._check(self._lib.pc_launch_bulk_flags_device(self._h, addr(d_x), addr(d_lam), addr(d_c), addr(d_G), addr(d_H),
                                                          int(flags), stream))

    def launch_tail_only(self, d_x, obj_factor, d_lam, d_c, d_G, d_H, stream=None):
        def addr(t):
            return t.data_ptr() if hasattr(t, "data_ptr") else int(t)
        self._cached_x = None   # (the launch overwrites the handle's f block)
        self._check(self._lib.pc_launch_tail_device(self._h, addr(d_x), float(obj_factor), addr(d_lam), addr(d_c),
                                                    addr(d_G), addr(d_H), stream))


    def launch_tail_objective(self, d_x, obj_factor, d_lam, d_c, d_G, d_H, stream=None, want_grad=True):
        """``launch_tail_only`` that also returns (J, grad J or None) and synchronises the stream
        (``pc_launch_tail_objective_device``): the objective callbacks of a rank of a sharded solve."""
        def addr(t):
            return t.data_ptr() if hasattr(t, "data_ptr") else int(t)
        self._cached_x = None
        f = C.c_double()
        grad = np.empty(self.num_x) if want_grad else None
        self._lib.pc_launch_tail_objective_device.argtypes = [C.c_void_p, C.c_void_p, C.c_double] + [C.c_void_p] * 5 + [C.c_void_p, C.c_void_p]
        self._check(self._lib.pc_launch_tail_objective_device(self._h, addr(d_x), float(obj_factor), addr(d_lam), addr(d_c), addr(d_G),
                                                              addr(d_H), stream, C.addressof(f), None if grad is None else grad.ctypes.data))
        return f.value, grad

    def set_tile_range(self, phase: int, begin: int, end: int):
        self._check(self._lib.pc_set_tile_range(self._h, phase, begin, end))

    def phase_tiles(self, phase: int):
        """(tile_k0 [n_tiles+1], nred): first section of every tile and partial sums per tile."""
        n, r = C.c_int32(), C.c_int32()
        self._check(self._lib.pc_phase_tiles(self._h, phase, C.byref(n), C.byref(r), None))
        k0 = np.empty(n.value + 1, dtype=np.int32)
        self._check(self._lib.pc_phase_tiles(self._h, phase, C.byref(n), C.byref(r), k0.ctypes.data))
        return k0, r.value

    def phase_tile_orders(self, phase: int) -> np.ndarray:
        """Mixed build: the section order whose tile body runs every tile of the phase (0 = the any-order body)."""
        k0, _ = self.phase_tiles(phase)
        out = np.zeros(len(k0) - 1, dtype=np.int32)
        self._check(self._lib.pc_phase_tile_orders(self._h, phase, out.ctypes.data))
        return out

    def set_partials_buffer(self, phase: int, d_partials):
        ptr = d_partials.data_ptr() if hasattr(d_partials, "data_ptr") else d_partials
        self._check(self._lib.pc_set_partials_buffer(self._h, phase, ptr))

    def mesh_error(self, phase: int, x, orders, tabB, tabE, tabA):
        """Section maxima of the ph relative / absolute mesh error of one phase (mesh_refinement.py:198-233)."""
        x = self._x(x)
        K, n_y = self.meshes[phase].K, self.model.phases[phase].n_y
        od = _i32(orders)
        B, E, A = (np.ascontiguousarray(t, dtype=np.float64) for t in (tabB, tabE, tabA))
        rel, ab = np.empty(K), np.empty((K, max(n_y, 1)))
        self._check(self._lib.pc_mesh_error(self._h, phase, x.ctypes.data, len(od), od.ctypes.data, B.ctypes.data,
                                            E.ctypes.data, A.ctypes.data, rel.ctypes.data, ab.ctypes.data))
        return rel, ab[:, :n_y]

    def synchronize(self):
        self._check(self._lib.pc_synchronize(self._h))

    def read_symbol(self, name: str, dtype, count: int):
        """Diagnostic: ``count`` items of a ``__device__`` array of the code object (``pc_read_symbol``)."""
        out = np.empty(count, dtype=dtype)
        self._check(self._lib.pc_read_symbol(self._h, name.encode(), out.ctypes.data, out.nbytes))
        return out

    def check(self):
        """After the caller synchronised its OWN stream (device API): raises if an evaluation's resident tail gave up
        waiting for values of its launch (``pc_check``) -- e.g. two evaluations of this handle in flight at once."""
        self._check(self._lib.pc_check(self._h))

    @property
    def stream(self):
        return self._lib.pc_stream(self._h)


class PycolloGpuProblem:
    """cyipopt ``problem_obj`` with the method names of ``IPOPTProblem`` (pycollo/nlp.py:36-76).

    ``ipopt.problem(n=p.n, m=p.m, problem_obj=p, lb=..., ub=..., cl=..., cu=...)`` works wherever
    cyipopt exists; x, lagrange and obj_factor are in IPOPT's (scaled) space like the reference's."""

    def __init__(self, engine: NlpEngine):
        self.engine = engine
        self.n, self.m = engine.num_x, engine.num_c
        self.obj_func_eval_counter = 0  # nlp.py:45

    def _new_x(self, x) -> bool:
        """cyipopt does not forward IPOPT's ``new_x`` flag: it is recovered by comparing with the point the ENGINE's
        cache was filled at (n doubles; far cheaper than an evaluation), so that objective / gradient / constraints /
        jacobian at one point share one launch (``pc_eval_*`` with ``new_x = 0``).  The engine owns that record
        (``NlpEngine.cache_holds``): any other call on the same engine in between (evaluate_all, set_scaling, a
        resident evaluation, a mesh-error pass) clears it, and the next callback re-evaluates."""
        return not self.engine.cache_holds(x)

    def objective(self, x):
        self.obj_func_eval_counter += 1
        return self.engine.evaluate_J(x, self._new_x(x))

    def gradient(self, x):
        return self.engine.evaluate_g(x, self._new_x(x))

    def constraints(self, x):
        return self.engine.evaluate_c(x, self._new_x(x))

    def jacobian(self, x):
        return self.engine.evaluate_G_nonzeros(x, self._new_x(x))

    def jacobianstructure(self):
        return self.engine.evaluate_G_structure()

    def hessian(self, x, lagrange, obj_factor):
        return self.engine.evaluate_H_nonzeros(x, obj_factor, lagrange, self._new_x(x))

    def hessianstructure(self):
        return self.engine.evaluate_H_structure()

    def intermediate(self, alg_mod, iter_count, obj_value, inf_pr, inf_du, mu, d_norm, regularization_size,
                     alpha_du, alpha_pr, ls_trials):
        pass
