"""Segment-grouped code generation (pockit_amd/codegen.py) on the CPU: the partition itself, and --
hipcc cross-compiles gfx950 without a GPU -- that the three example models of the reference whose per-node derivative set
exceeded one workgroup's LDS / the register file in round 3 now generate code objects that fit and spill no vector register."""
import json
import os

import pytest

import model_io
from pockit_amd import hipbuild
from pockit_amd.codegen import ModelSource, split_chunks, split_groups

HERE = os.path.dirname(os.path.abspath(__file__))
LDS_LIMIT = 160 * 1024


@pytest.mark.parametrize("cap", [2, 3, 8, 32])
def test_groups_partition_every_segment_exactly_once(cap):
    for n_i in (0, 1, cap, cap + 1, 3 * cap + 1, 250):
        for n_n in (0, 1, cap // 2 + 1, 2 * cap, 33):
            groups = split_groups(n_i, n_n, cap)
            assert sorted(i for i0, ni, _, _ in groups for i in range(i0, i0 + ni)) == list(range(n_i))
            assert sorted(i for _, _, n0, nn in groups for i in range(n0, n0 + nn)) == list(range(n_n))
            if len(groups) == 1:            # the single-pass code of rounds 1-3: small sets only
                assert n_i <= cap and n_i + n_n <= cap + cap // 2
            else:                           # runs of one kind, none larger than the cap, balanced to within one
                assert all((ni == 0) != (nn == 0) and max(ni, nn) <= cap for _, ni, _, nn in groups)
                sizes = [ni for _, ni, _, nn in groups if ni]
                assert not sizes or max(sizes) - min(sizes) <= 1
    for n in (0, 1, 47, 48, 49, 200):
        chunks = split_chunks(n, 32)
        assert sorted(i for lo, cnt in chunks for i in range(lo, lo + cnt)) == list(range(n))
        assert len(chunks) == 1 or max(c for _, c in chunks) <= 32


@pytest.mark.parametrize("name", ["orbit_transfer", "rocket_powered_descent", "drone_stabilization"])
def test_the_three_large_example_models_fit_and_do_not_spill(name):
    from pockit_amd.evaluator import compile_plan

    with open(os.path.join(HERE, "golden", "examples", name + ".model.json")) as fh:
        system = model_io.load_system(json.load(fh))
    src, code = compile_plan(system.plan)
    assert src.grouped and len(code) > 0
    single = ModelSource.__new__(ModelSource)        # what round 3 asked of LDS: every segment of the Hessian staged at once
    n_h = max(pp.nx + sum(1 for sg in system.plan.hess.segs[k] if sg.kind == "I") for k, pp in enumerate(system.plan.phase_plans))
    assert 64 * n_h * 4 * 8 > LDS_LIMIT, "the model no longer needs groups?"
    tab = 4 * 8 * (2 * src.tab_cap + 2 * 64 + src.tab_cap // 2)
    for rows in (src.lds_g, src.lds_j, src.lds_h, src.lds_x, src.lds_jc):
        assert rows * 4 * 8 + tab <= LDS_LIMIT
    usage = hipbuild.resource_usage(src.source, fastmath=system._fastmath)
    assert usage and "pk_cycle" in usage and "pk_cyclec" in usage
    assert not src.spilling_kernels, src.spilling_kernels
    for kernel, u in usage.items():
        assert u.get("vgpr_spill", 0) == 0 and u.get("scratch", 0) == 0, (kernel, u)
    del single


def test_group_size_follows_the_launch_not_only_the_model(monkeypatch):
    """evaluator.compile_plan's rule for launches that underfill the chip: the humanoid on a small or
    medium mesh is generated with groups of 16 and runs its Hessian passes as workgroups of their own; on BASELINE's C5
    mesh (474 workgroups) it keeps the single pass; a model with few segments is the same code either way.  (The rule and
    the generator only -- no hipcc.)"""
    from pockit_amd import benchmarks, radau
    from pockit_amd.evaluator import _intervals_per_wave, _launch_underfills_the_chip

    for key in ("POCKIT_AMD_GROUP_CAP", "POCKIT_AMD_PASS_PARALLEL", "POCKIT_AMD_IPW"):
        monkeypatch.delenv(key, raising=False)
    small, _, _ = benchmarks.humanoid_wbc(radau, mesh=25, num_point=8)
    full, _, _ = benchmarks.humanoid_wbc(radau, mesh=5000, num_point=8)
    quad, _, _ = benchmarks.planar_quadrotor(radau, mesh=100, num_point=6)
    assert _launch_underfills_the_chip(small.plan) and _launch_underfills_the_chip(quad.plan)
    assert not _launch_underfills_the_chip(full.plan)
    assert _intervals_per_wave(full.plan, want_workgroups=True) == 474
    half = ModelSource(small.plan, group_cap=ModelSource.GROUP_CAP // 2)
    assert half.grouped and half.cycle_subs == 1 + half.j_ngmax + half.h_ngmax and half.h_ngmax >= 2
    assert "GROUPED = true" in half.source
    whole = ModelSource(small.plan)
    assert not whole.grouped and whole.cycle_subs == 0 and "GROUPED = false" in whole.source
    # a model with at most 16 segments per role: the same source text, hence the same code object, whatever the rule says
    assert ModelSource(quad.plan, group_cap=ModelSource.GROUP_CAP // 2).source == ModelSource(quad.plan).source
    # the tiling model counts the pass-parallel workgroups: fuller waves for a model with many passes
    assert _intervals_per_wave(small.plan, subs=half.cycle_subs) >= _intervals_per_wave(small.plan)
