"""The Delta III example's data (examples/delta_iii_launch_vehicle/delta_iii_launch_vehicle.py:198-459, restated in
problems.delta_iii): why the NLP is infeasible as published, and the guess that replaces the example's.  No GPU."""
import numpy as np
import sympy as sym

from pycollo_amd import problems


def _burn(ph):
    xi = float(ph.auxiliary_data[sym.Symbol("xi")])
    return xi * (float(ph.bounds.final_time) - float(ph.bounds.initial_time))


def test_published_mass_pins_contradict_the_burn():
    prob = problems.delta_iii()
    gaps = []
    for ph in prob.phases:
        m = sym.Symbol("m")
        pinned = float(ph.bounds.initial_state_constraints[m]) - float(ph.bounds.final_state_constraints[m])
        gaps.append(pinned - _burn(ph))
    # phases A-C: the tabulated propellant masses against thrust / (g0 Isp), about a gram (of either sign)
    assert all(1e-4 < abs(g) < 1e-2 for g in gaps[:3])
    # phase D: pinned 19 300 kg apart, burns 16 820 kg -- the second stage's 2 480 kg of structure
    assert abs(gaps[3] - 2480.0) < 1.0


def test_burnout_mass_variant_is_consistent():
    prob = problems.delta_iii(burnout_mass=True)
    m = sym.Symbol("m")
    for ph in prob.phases:
        lo, hi = ph.bounds.final_state_constraints[m]
        end = float(ph.bounds.initial_state_constraints[m]) - _burn(ph)
        assert lo < end < hi
        assert ph.bounds.state_variables[m][0] < end


def test_flown_guess_satisfies_dynamics_and_path_constraints():
    prob = problems.delta_iii_flown_guess(problems.delta_iii(burnout_mass=True), points=40)
    R_E = 6378145.0
    prev = None
    for ph in prob.phases:
        t, y, u = np.asarray(ph.guess.time), np.asarray(ph.guess.state_variables), np.asarray(ph.guess.control_variables)
        assert y.shape == (7, 40) and u.shape == (3, 40) and np.all(np.diff(t) > 0)
        np.testing.assert_allclose(np.linalg.norm(u, axis=0), 1.0, rtol=1e-12)           # path constraint |u| = 1
        r = np.linalg.norm(y[:3], axis=0)
        assert np.all(r >= R_E * (1 - 1e-12))                                            # above ground
        assert np.all(np.diff(r) > 0)                                                    # it flies
        np.testing.assert_allclose(y[6, 0] - y[6, -1], _burn(ph), rtol=1e-9)             # the mass follows the burn
        if prev is not None:
            np.testing.assert_allclose(y[:6, 0], prev[:6], rtol=1e-12)                   # linkages
        prev = y[:, -1]
        lo = np.array([b[0] for b in ph.bounds.state_variables.values()])
        hi = np.array([b[1] for b in ph.bounds.state_variables.values()])
        assert np.all(y >= lo[:, None]) and np.all(y <= hi[:, None])
    assert 2.5e6 < np.linalg.norm(prev[:3]) - R_E < 2.8e6                                # final altitude of the vertical flight
