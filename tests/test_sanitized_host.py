"""The host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only).

``tests/c/pattern_sanitize.cpp`` compiles the headers ``pc_create`` runs on a structure-only handle -- ``pc_desc.hpp``
(descriptor -> problem, LDS sizing) and ``pc_pattern.hpp`` (tiles, layout, CSR patterns, producer-slot tables) -- with
``g++ -fsanitize=address,undefined -fno-sanitize-recover=all``.  It is fed the very descriptor the engine hands to the
library (serialised from the ctypes structure) and must (1) exit cleanly with no sanitizer report and (2) print index
arrays and tile tables identical to the library's.  No GPU sanitizer, no XNACK: those are not available on the pool.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from pycollo_amd import problems

SRC = os.path.join(ROOT, "tests", "c", "pattern_sanitize.cpp")
EXE = os.path.join(ROOT, "tests", "_build", "pattern_sanitize")


@pytest.fixture(scope="module")
def harness():
    deps = [SRC] + [os.path.join(ROOT, "pycollo_amd", "csrc", f) for f in ("pc_desc.hpp", "pc_pattern.hpp", "pc_args.h")]
    deps.append(os.path.join(ROOT, "include", "pycollo_amd.h"))
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
               "-fno-omit-frame-pointer", "-o", EXE + f".tmp{os.getpid()}", SRC]   # (per process: pytest-xdist workers build side by side)
        res = subprocess.run(cmd, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-3000:]
        os.replace(EXE + f".tmp{os.getpid()}", EXE)
    return EXE


def _arr(ptr, n):
    return [] if not ptr or n <= 0 else [ptr[i] for i in range(n)]


def _serialise(eng, desc, tile_nodes, path):
    """The descriptor the library received, as the harness's text format."""
    tok = []
    qa_total = sum((n - 1) * n for n in _arr(desc.orders, desc.n_orders))
    qw_total = sum(_arr(desc.orders, desc.n_orders))
    tok += [desc.n_phases, desc.n_s, desc.n_point, desc.n_b, desc.n_jgrad, desc.n_bjac, desc.n_pthess, tile_nodes, qa_total, qw_total]
    for ip in range(desc.n_phases):
        s = desc.phases[ip]
        tok += [s.n_y, s.n_u, s.n_q, s.n_p, s.t0_free, s.tF_free, s.K, s.n_jac, s.n_hess, s.n_w, s.compiled_order]
        tok += [s.n_spec] + [s.spec_orders[j] for j in range(4)] + [s.n_fixed_tiles]
        tok += _arr(s.n_k, s.K)
        tok += [repr(float(v)) for v in _arr(s.h_k, s.K)]
        tok += _arr(s.jac_row, s.n_jac) + _arr(s.jac_col, s.n_jac) + _arr(s.hess_row, s.n_hess) + _arr(s.hess_col, s.n_hess)
        tok += _arr(s.w_kind, s.n_w) + _arr(s.w_idx, s.n_w)
        if s.n_fixed_tiles > 0:
            tok += _arr(s.fixed_tile_k0, s.n_fixed_tiles + 1)
            tok += _arr(s.fixed_tile_order, s.n_fixed_tiles) if s.fixed_tile_order else [0] * s.n_fixed_tiles
    tok += _arr(desc.point_phase, desc.n_point) + _arr(desc.point_kind, desc.n_point) + _arr(desc.point_idx, desc.n_point)
    tok += _arr(desc.jgrad_col, desc.n_jgrad) + _arr(desc.bjac_row, desc.n_bjac) + _arr(desc.bjac_col, desc.n_bjac)
    tok += _arr(desc.pthess_row, desc.n_pthess) + _arr(desc.pthess_col, desc.n_pthess)
    with open(path, "w") as f:
        f.write(" ".join(str(t) for t in tok) + "\n")


def _parse(path):
    out, tiles, lds, orders, mixed = {}, [], [], [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if p[0] == "sizes":
                out["sizes"] = tuple(int(v) for v in p[1:])
            elif p[0] == "ok":
                out["ok"] = True
            else:
                vals = np.array([int(v) for v in p[2:]], dtype=np.int64)
                assert len(vals) == int(p[1])
                if p[0] == "tile_k0":
                    tiles.append(vals)
                elif p[0] == "lds":
                    lds.append(vals)
                elif p[0] == "tile_order":
                    orders.append(vals)
                elif p[0] == "mixed":
                    mixed.append(vals)
                elif p[0] not in ("goff", "hoff", "hslot0", "hslotN"):
                    out[p[0]] = vals
    out["tile_k0"], out["lds"], out["tile_order"], out["mixed"] = tiles, lds, orders, mixed
    return out


def _ragged(prob, K, seed=3):
    rr = np.random.default_rng(seed)
    for ph in prob.phases:
        ph.mesh.mesh_section_sizes = rr.uniform(0.5, 1.5, K)
        ph.mesh.number_mesh_section_nodes = rr.integers(2, 9, K)
    return prob


CASES = [("brachistochrone", {}, False), ("hypersensitive", dict(K=2000, order=6), False), ("cart_pole", dict(K=50, order=4), False),
         ("shuttle", dict(K=300, order=5), True), ("double_pendulum", {}, False), ("two_phase_transfer", {}, False),
         ("delta_iii", dict(K=40, order=4), True), ("delta_iii", dict(K=7, order=5), False),
         ("time_coupled_transfer", dict(K=40, order=6), False), ("space_station", dict(K=12, order=4), False),
         ("sliding_mass", dict(num_phases=4, K=5, order=4), False), ("hypersensitive", dict(K=1, order=2), False)]


@pytest.mark.parametrize("name,kw,ragged", CASES)
def test_pattern_builder_under_asan_ubsan(built, harness, tmp_path, name, kw, ragged):
    from pycollo_amd.engine import NlpEngine
    prob = problems.REGISTRY[name](**kw)
    if ragged:
        prob = _ragged(prob, kw["K"])
    eng = NlpEngine(prob, device=None)
    desc = eng._make_desc(None, 0)
    tile_nodes = eng.info["threads_per_block"]
    fin, fout = str(tmp_path / "in.txt"), str(tmp_path / "out.txt")
    _serialise(eng, desc, tile_nodes, fin)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    res = subprocess.run([harness, fin, fout], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-4000:]
    assert "Sanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-4000:]
    got = _parse(fout)
    assert got.get("ok")
    assert got["sizes"] == (eng.num_x, eng.num_c, eng.nnz_jac, eng.nnz_hess)
    (gr, gc), (hr, hc) = eng.evaluate_G_structure(), eng.evaluate_H_structure()
    for key, ref in (("g_row", gr), ("g_col", gc), ("h_row", hr), ("h_col", hc)):
        np.testing.assert_array_equal(got[key], ref)
    for ip in range(len(prob.phases)):   # the tiling the sharding plan is built on (tests/test_sharding.py)
        k0, _ = eng.phase_tiles(ip)
        np.testing.assert_array_equal(got["tile_k0"][ip], k0)
        rows, lds_out, b1, b2, b4 = got["lds"][ip]
        assert 0 < rows < tile_nodes and lds_out > 0 and b1 <= b2 <= b4
    assert max(int(v[2]) for v in got["lds"]) <= eng.info["lds_bytes_max"] or eng.info["waves_per_tile"] > 1
    eng.close()


def test_harness_reports_a_planted_overflow(harness, tmp_path):
    """The sanitizers are live in this build: a descriptor whose Jacobian list points outside the function / variable
    ranges must be rejected by the builder's own checks or caught by the sanitizer -- never run through silently."""
    from pycollo_amd.engine import NlpEngine
    prob = problems.cart_pole(K=5, order=4)
    eng = NlpEngine(prob, device=None)
    desc = eng._make_desc(None, 0)
    fin, fout = str(tmp_path / "in.txt"), str(tmp_path / "out.txt")
    _serialise(eng, desc, 64, fin)
    toks = open(fin).read().split()
    # header (10) + phase header (11 + 6) + K n_k + K h_k, then the first jac_row: make it a row far outside n_fn
    K = desc.phases[0].K
    toks[10 + 17 + 2 * K] = "1000"
    open(fin, "w").write(" ".join(toks) + "\n")
    res = subprocess.run([harness, fin, fout], capture_output=True, text=True, timeout=600)
    assert res.returncode != 0
    eng.close()


def _run_mesh(seed):
    rr = np.random.default_rng(seed)
    n = np.concatenate([np.full(30, 4), rr.integers(4, 9, 12), np.full(3, 4), np.full(20, 6), [7, 5], np.full(10, 5),
                        np.full(40, 4), [8], np.full(9, 6), np.full(100, 8)]).astype(np.int64)
    return rr.uniform(0.5, 1.5, n.size), n


@pytest.mark.parametrize("name,spec", [("shuttle", (4, 6)), ("delta_iii", (4, 6, 8)), ("two_phase_transfer", (5,)),
                                       ("time_coupled_transfer", (4, 6, 7, 8))])
def test_mixed_tile_cutter_under_asan_ubsan(built, harness, tmp_path, name, spec):
    """Round 4's host index arithmetic -- runs of equal sections cut into order-pure tiles under row caps
    (pc_pattern.hpp::build_tiles_mixed, pc_desc.hpp::phase_set_caps), the exact staging size of the tiles as cut, the
    row-group rule (pc_args.h::pc_row_passes) -- under the sanitizers, tile table and orders identical to the library's."""
    from pycollo_amd.engine import NlpEngine
    prob = problems.REGISTRY[name]()
    for i, ph in enumerate(prob.phases):
        sizes, nodes = _run_mesh(3 + i)
        ph.mesh.number_mesh_sections, ph.mesh.mesh_section_sizes, ph.mesh.number_mesh_section_nodes = nodes.size, sizes, nodes
    eng = NlpEngine(prob, device=None, mixed=tuple(spec for _ in prob.phases))
    desc = eng._make_desc(None, 0)
    fin, fout = str(tmp_path / "in.txt"), str(tmp_path / "out.txt")
    _serialise(eng, desc, eng.info["threads_per_block"], fin)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    res = subprocess.run([harness, fin, fout], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-4000:]
    assert "Sanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-4000:]
    got = _parse(fout)
    assert got.get("ok") and got["sizes"] == (eng.num_x, eng.num_c, eng.nnz_jac, eng.nnz_hess)
    for ip in range(len(prob.phases)):
        np.testing.assert_array_equal(got["tile_k0"][ip], eng.phase_tiles(ip)[0])
        np.testing.assert_array_equal(got["tile_order"][ip], eng.phase_tile_orders(ip))
        grouped, whole, any_pure, any_generic = got["mixed"][ip][:4]
        assert 0 < grouped <= whole and any_pure == 1 and any_generic == 1
        assert all(1 <= int(v) <= 19 for v in got["mixed"][ip][4:])
    eng.close()


def test_fixed_tile_table_under_asan_ubsan(built, harness, tmp_path):
    """A rank-local handle's descriptor (sharding.LocalShard: the global tiles of its range between two halo tiles)."""
    from pycollo_amd.sharding import LocalShard
    prob = problems.two_phase_transfer(K=40, order=4)
    ls = LocalShard(prob, 1, 3, device=None)
    eng = ls.engine
    desc = eng._make_desc(None, eng.info["threads_per_block"])
    fin, fout = str(tmp_path / "in.txt"), str(tmp_path / "out.txt")
    _serialise(eng, desc, eng.info["threads_per_block"], fin)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    res = subprocess.run([harness, fin, fout], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-4000:]
    assert "Sanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-4000:]
    got = _parse(fout)
    for ip in range(len(prob.phases)):
        np.testing.assert_array_equal(got["tile_k0"][ip], eng.phase_tiles(ip)[0])
        np.testing.assert_array_equal(got["tile_order"][ip], eng.phase_tile_orders(ip))
    ls.close()
