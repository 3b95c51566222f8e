"""CPU tests of what the solver adapters share (pockit_amd/optimizer/_common.py): guess -> start vector, solver vector
-> ``Variable`` objects with the dictated boundary slots restored.  Behaviour pinned against the reference's helpers
(/root/reference/pockit/optimizer/_common.py:9-63): accepted guess shapes, error texts, FIXED / FUNC slots."""
import numpy as np
import pytest

import models
import pockit_amd.lobatto as lobatto
import pockit_amd.radau as radau
from pockit_amd.model import FIXED, FREE, FUNC
from pockit_amd.optimizer._common import dictated_slots, postprocess, preprocess
from pockit_amd.variable import Variable


def test_start_vector_is_phase_blocks_then_static_values():
    system, phases, guess = models.two_stage_rocket(radau, 6, 3)
    start, bare, opts = preprocess(system, guess)
    assert not bare and opts == {}
    assert start.shape == (system.L,)
    for k, v in enumerate(guess[: system.n_p]):
        assert np.array_equal(start[system.l_p[k]: system.r_p[k]], v.data)
    assert np.array_equal(start[system.l_s: system.r_s], np.array(list(guess[-1]), dtype=np.float64))
    assert np.array_equal(start, models.pack_guess(system, guess))
    given = {"tol": 1e-9}
    assert preprocess(system, guess, given)[2] is given


def test_one_variable_in_one_variable_out():
    system, _, guess = models.brachistochrone(radau, 4, 3)
    (guess,) = guess
    assert isinstance(guess, Variable)
    start, bare, _ = preprocess(system, guess)
    assert bare and np.array_equal(start, guess.data)
    back = postprocess(system, start, bare)
    assert isinstance(back, Variable)


def test_wrong_guess_lengths_and_half_configured_systems_raise_the_reference_messages():
    system, _, guess = models.two_stage_rocket(radau, 4, 2)
    with pytest.raises(ValueError, match=r"number of phases \+ 1 \(for static variables\)"):
        preprocess(system, guess[:-1])
    plain, _, g1 = models.brachistochrone(lobatto, 3, 3)
    with pytest.raises(ValueError, match="len\\(guess\\) must be equal to the number of phases$"):
        preprocess(plain, g1 + g1)
    empty = radau.System(0)
    with pytest.raises(ValueError, match="not fully configured"):
        preprocess(empty, [])


@pytest.mark.parametrize("ns", [radau, lobatto])
def test_dictated_slots_come_back_with_their_dictated_values(ns):
    system, _, guess = models.two_stage_rocket(ns, 5, 3)
    plan = system.plan
    rng = np.random.default_rng(3)
    x = rng.standard_normal(plan.n)
    keep = x.copy()
    out = postprocess(system, x, False)
    assert np.array_equal(x, keep)                       # the solver's vector is not written
    static = out[-1]
    assert np.array_equal(static, keep[plan.l_s: plan.r_s])
    fixed_at, fixed_to, func = dictated_slots(system)
    touched = set(fixed_at.tolist()) | {at for at, _, _ in func}
    n_fixed = n_func = 0
    for k, phase in enumerate(system.p):
        data, lay, base = out[k].data, phase.layout, int(plan.l_p[k])
        slots = [(int(lay.l_v[i]), phase.info_bc_0[i]) for i in range(phase.n_x)]
        slots += [(int(lay.r_v[i]) - 1, phase.info_bc_f[i]) for i in range(phase.n_x)]
        slots += [(lay.L - 2, phase.info_t_0), (lay.L - 1, phase.info_t_f)]
        for local, info in slots:
            if info.t == FREE:
                assert data[local] == keep[base + local]
            elif info.t == FIXED:
                assert data[local] == info.v
                n_fixed += 1
            else:
                assert info.t == FUNC
                assert data[local] == phase._value_boundary_condition(info, 0.0, static)
                n_func += 1
        rest = [i for i in range(lay.L) if base + i not in touched]
        assert np.array_equal(data[rest], keep[base + np.array(rest)])
    assert n_fixed == len(fixed_at) and n_func == len(func) and n_func > 0
