"""Model compiler + code generator (host logic, no GPU)."""
import numpy as np
import pytest
import sympy as sym

from pycollo_amd import codegen, problems
from pycollo_amd.model import compile_model
from pycollo_amd.problem import ProblemSpec


def test_counts_match_reference_fixtures():
    m = compile_model(problems.brachistochrone())
    p = m.phases[0]
    assert (p.n_y, p.n_u, p.n_q, p.n_p, p.n_t, m.n_s) == (3, 1, 0, 0, 1, 0)     # t0 eliminated (bounds.py:456-480)
    assert p.t_free == (False, True) and p.t_fixed[0] == 0.0
    m = compile_model(problems.double_pendulum())
    p = m.phases[0]
    assert (p.n_y, p.n_u, p.n_q, p.n_t, m.n_s) == (4, 2, 1, 1, 2)
    m = compile_model(problems.hypersensitive())
    assert m.phases[0].n_t == 0                                                # both times fixed


def test_phase_aux_overrides_problem_aux():
    # conftest.py:118-124,127: phase-level g = -9.81 wins over problem-level g = 0
    m = compile_model(problems.double_pendulum())
    vals = [v for _, v in m.phases[0].consts]
    assert -9.81 in vals and 0.0 not in vals


def test_numeric_constants_are_not_folded():
    """exp(-(r - R_E)/h_0) must not become exp(R_E/h_0) * exp(-r/h_0) (= inf * 0 in fp64)."""
    m = compile_model(problems.delta_iii())
    src = codegen.generate_source(m)
    assert "inf" not in src and "nan" not in src
    assert "constexpr double K0" in src


def test_structural_masks_hypersensitive():
    p = compile_model(problems.hypersensitive()).phases[0]
    assert [(r, c) for r, c, _ in p.jac] == [(0, 0), (0, 1), (1, 0), (1, 1)]
    assert [(r, c) for r, c, _ in p.hess] == [(0, 0), (1, 1)]


def test_errors():
    y, u, k = sym.symbols("y u k")
    prob = ProblemSpec("bad")
    ph = prob.new_phase("A")
    ph.state_variables = [y]
    ph.control_variables = [u]
    ph.state_equations = [k * y + u]          # k is undefined
    ph.bounds.initial_time = 0
    ph.bounds.final_time = 1
    ph.bounds.state_variables = [[0, 1]]
    ph.bounds.control_variables = [[0, 1]]
    prob.objective_function = ph.final_state_variables[0]
    with pytest.raises(ValueError, match="neither variables nor auxiliary data"):
        compile_model(prob)
    prob.auxiliary_data = {k: k + 1}
    with pytest.raises(ValueError, match="cyclic"):
        compile_model(prob)
    prob.auxiliary_data = {k: 2.0}
    prob.endpoint_constraints = [ph.final_state_variables[0]]
    prob.bounds.endpoint_constraints = [[0, 1]]
    with pytest.raises(ValueError, match="bare point variable"):       # backend.py:764-770
        compile_model(prob)


def test_digest_tracks_model():
    a = compile_model(problems.hypersensitive()).digest
    b = compile_model(problems.hypersensitive(K=99, order=7)).digest     # mesh does not enter the kernels
    c = compile_model(problems.hypersensitive(test_fixture_bounds=True)).digest
    assert a == b == c                                                    # bounds only affect scaling data
    assert a != compile_model(problems.cart_pole()).digest


def test_generated_source_is_straight_line_fp64():
    src = codegen.generate_source(compile_model(problems.shuttle()))
    assert "pow(" not in src            # integer powers expanded to multiplications
    assert "float " not in src
    # bulk, bulk with the resident tail (replica index at run time, and compiled in for 2 and 4 waves per tile),
    # mesh error, tail, tail for many tiles
    assert src.count("__global__") == 7
    assert "pc_bulk_p0_r_w2(" in src and "pc_bulk_p0_r_w4(" in src


@pytest.mark.parametrize("name,kw,orders,mixed", [
    ("hypersensitive", dict(K=10, order=6), (6,), None), ("cart_pole", dict(K=10, order=4), (4,), None),
    ("two_phase_transfer", {}, (0, 0), None), ("time_coupled_transfer", {}, (3, 4), None), ("double_pendulum", {}, (4,), None),
    ("delta_iii", dict(K=10, order=5), (5, 5, 5, 5), None),
    # mixed builds (a body per listed order next to the any-order one in every launch kernel: up to twelve bodies)
    ("hypersensitive", {}, (0,), ((4, 6),)), ("shuttle", {}, (0,), ((4, 6),)), ("two_phase_transfer", {}, (0, 0), ((4, 6), (4, 6))),
    ("delta_iii", {}, (0, 0, 0, 0), ((4, 6),) * 4)])
def test_kernels_use_no_scratch_memory(name, kw, orders, mixed):
    """A run-time subscript into a register array sends the array -- and with it the dispatch -- to scratch memory
    (it cost the resident-tail kernel 2.5 us of a 4.6 us evaluation before it was found; hundreds of bytes per lane):
    every bulk / tail kernel of the headline models must compile to zero scratch bytes.  One deliberate exception: a
    HEAVY model's tile kernels are compiled for two waves per SIMD (codegen._heavy_attr) and may spill the few
    registers past 256 -- at most codegen.HEAVY_SCRATCH_LIMIT bytes, recorded in the object's resource sidecar."""
    from pycollo_amd import codegen, problems
    from pycollo_amd.model import compile_model
    model = compile_model(problems.REGISTRY[name](**kw))
    res = codegen.code_object_resources(codegen.build_code_object(model, orders, mixed=mixed))   # what the compiler reported at build time
    assert any(k.endswith("_r") for k in res), sorted(res)
    heavy = any(codegen.is_heavy(pm) for pm in model.phases)
    capped = res.get("_build", {}).get("heavy_cap", False)
    assert capped == (heavy and codegen._heavy_cap_enabled()) or not heavy
    for kern, r in res.items():
        if kern.startswith("pc_tail") or (kern.startswith("pc_bulk") and not (heavy and capped)):
            assert r["scratch"] == 0, (kern, r)
        elif kern.startswith("pc_bulk"):
            assert r["scratch"] <= codegen.HEAVY_SCRATCH_LIMIT and r["occupancy"] >= 2, (kern, r)
    if name == "delta_iii" and mixed is None:   # the two-wave launch kernel of config 5's uniform mesh fits without spilling at all
        assert res["pc_bulk_all_r_w2"]["scratch"] == 0 and res["pc_bulk_all_r_w2"]["vgprs"] <= 256


def test_mfma_build_of_the_defect_contraction_uses_the_matrix_cores(monkeypatch, tmp_path):
    """-DPC_MFMA_DEFECT: the code object then carries v_mfma_f64_16x16x4_f64 in its tile kernels and no
    scratch; the default build carries no matrix instruction at all (DESIGN.md section 8 says why it is not the default)."""
    import os
    import subprocess
    from pycollo_amd import codegen, problems
    from pycollo_amd.model import compile_model
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("llvm-objdump not found")
    model = compile_model(problems.hypersensitive(K=10, order=6))

    def mfma_count(path):
        co = str(tmp_path / (os.path.basename(path) + ".co"))
        subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={path}", f"--output={co}"], check=True, capture_output=True)
        text = subprocess.run([f"{llvm}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
        return text.count("v_mfma_f64_16x16x4")
    assert mfma_count(codegen.build_code_object(model, (6,))) == 0
    monkeypatch.setenv("PYCOLLO_AMD_DEFINES", "PC_MFMA_DEFECT")
    flagged = codegen.build_code_object(model, (6,))
    assert mfma_count(flagged) > 0
    res = codegen.code_object_resources(flagged)
    assert all(r["scratch"] == 0 for k, r in res.items() if k.startswith("pc_bulk")), res
