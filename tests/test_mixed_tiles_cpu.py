"""CPU tests of the mixed build's host side: which section orders get a tile body (engine.choose_spec_orders), how the
tiles are cut (pc_pattern.hpp::build_tiles_mixed through a structure-only handle), the synthetic ph-refined mesh, and
the generated source.  No evaluation happens here (no GPU)."""
import numpy as np
import pytest

from pycollo_amd import codegen, problems
from pycollo_amd.engine import MIX_MIN_RUN_ROWS, NlpEngine, choose_spec_orders
from pycollo_amd.model import compile_model
from pycollo_amd.refinement import synthetic_refined_mesh


def _problem(nodes, sizes=None, name="hypersensitive"):
    prob = problems.REGISTRY[name]()
    for ph in prob.phases:
        ph.mesh.number_mesh_sections = len(nodes)
        ph.mesh.mesh_section_sizes = np.ones(len(nodes)) if sizes is None else sizes
        ph.mesh.number_mesh_section_nodes = np.asarray(nodes, dtype=np.int64)
    return prob


def test_choose_spec_orders():
    assert choose_spec_orders(np.full(100, 5)) == ()                    # a single order is not mixed at all
    rr = np.random.default_rng(0)
    assert choose_spec_orders(rr.integers(4, 9, 3000)) == ()            # orders assigned at random: no runs
    n = np.concatenate([np.full(200, 4), rr.integers(5, 9, 30), np.full(50, 7), np.full(3, 9), np.full(40, 4)])
    assert choose_spec_orders(n) == (4, 7)
    # at most four, by coverage
    n = np.concatenate([np.full(100, o) for o in (4, 5, 6, 7, 8)] + [np.full(300, 9)])
    picks = choose_spec_orders(n)
    assert len(picks) == 4 and 9 in picks


@pytest.mark.parametrize("spec", [(4, 6), (4,), (5, 6, 8)])
def test_tiles_cover_the_mesh_and_respect_orders(built, spec):
    rr = np.random.default_rng(3)
    n = np.concatenate([np.full(30, 4), rr.integers(4, 9, 12), np.full(3, 4), np.full(20, 6), [7, 5], np.full(10, 5),
                        np.full(40, 4), [8], np.full(9, 6), np.full(100, 8)])
    eng = NlpEngine(_problem(n), device=None, mixed=(spec,))
    k0, _ = eng.phase_tiles(0)
    od = eng.phase_tile_orders(0)
    assert k0[0] == 0 and k0[-1] == n.size and np.all(np.diff(k0) > 0) and od.size == k0.size - 1
    seen = set()
    for i, o in enumerate(od):
        sec = n[k0[i]:k0[i + 1]]
        assert np.sum(sec - 1) <= 63                                    # a tile fits one wave, shared end node included
        if o:
            assert o in spec and np.all(sec == o)
            seen.add(int(o))
    # every run of an order with a body that is long enough got tiles of its own
    cut = np.flatnonzero(np.diff(n)) + 1
    starts = np.concatenate([[0], cut])
    ends = np.concatenate([cut, [n.size]])
    for s, e in zip(starts, ends):
        if n[s] in spec and (e - s) * (n[s] - 1) >= MIX_MIN_RUN_ROWS:
            assert s in k0 and e in k0
            assert all(od[i] == n[s] for i in range(len(od)) if s <= k0[i] < e)
    assert seen == {o for o in spec if o in (4, 5, 6, 8)}
    # the patterns do not depend on the tiling
    ref = NlpEngine(_problem(n), device=None, mixed=None)
    for a, b in zip(eng.evaluate_G_structure() + eng.evaluate_H_structure(), ref.evaluate_G_structure() + ref.evaluate_H_structure()):
        np.testing.assert_array_equal(a, b)
    assert np.all(ref.phase_tile_orders(0) == 0)


def test_synthetic_refined_mesh_has_runs():
    sizes, nodes = synthetic_refined_mesh(12500, seed=7)
    s2, n2 = synthetic_refined_mesh(12500, seed=7)
    np.testing.assert_array_equal(nodes, n2)
    np.testing.assert_array_equal(sizes, s2)
    N = int(np.sum(nodes - 1)) + 1
    assert abs(N - 12500) < 0.1 * 12500 and abs(sizes.sum() - 1.0) < 1e-12
    assert nodes.min() >= 4 and nodes.max() <= 10 and np.unique(nodes).size >= 3
    spec = choose_spec_orders(nodes)
    assert len(spec) >= 2
    eng = NlpEngine(_problem(nodes, sizes), device=None, mixed=(spec,))
    od = eng.phase_tile_orders(0)
    k0, _ = eng.phase_tiles(0)
    rows = np.array([np.sum(nodes[k0[i]:k0[i + 1]] - 1) for i in range(od.size)])
    assert rows[od > 0].sum() > 0.75 * rows.sum()                       # most of the mesh runs order-specialised bodies


def test_generated_source_of_a_mixed_build():
    m = compile_model(problems.REGISTRY["two_phase_transfer"]())
    src = codegen.generate_source(m, (0, 0), mixed=((4, 6), ()))
    assert "pc::bulk_mix<gen::Phase0" in src and "std::integer_sequence<int, 4, 6>" in src
    assert "pc::bulk_mix<gen::Phase1" not in src
    assert codegen.code_object_path(m, (0, 0), ((4, 6), ())).endswith("n0_0-m4.6_x.hsaco")
    assert codegen.code_object_path(m, (0, 0), None) == codegen.code_object_path(m, (0, 0), ((), ()))
    with pytest.raises(ValueError):
        codegen.generate_source(m, (4, 0), mixed=((4, 6), ()))          # a phase is either single-order or mixed
