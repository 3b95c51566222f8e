"""The line ``bench.py`` prints last is what the driver parses: strict JSON (no NaN / Infinity tokens), shorter than 4 KB,
with fixed keys -- assembled here from a canned result record, no GPU needed (tools/benchlib/line.py)."""
import argparse
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools.benchlib import line as L  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "device_resident", "end_to_end", "parity")


def canned(nan=False):
    bad = float("nan") if nan else 0.5
    res = dict(name="planar_quadrotor", intervals=2000, nodes=12000, n=96008, m=96000, nnz_J=755925, nnz_H=359972, steps=20, batches=500,
               median_batch_ms=0.094, ms_per_step=0.0047, event_group=10, batch_launch={"form": "20 kernel launches per batch"},
               batch_ms_p10=0.09, batch_ms_p90=0.1, batch_ms_min=0.09, batch_ms_max=float("inf") if nan else 0.2, region_wall_s=0.05,
               wall_ms_per_step=0.005, untimed_launches=1500, setup_s=0.2, compile_s_in_setup=0.0, dominant="pk_cycle", exchange="single GPU",
               tiles=336, ipw=6, bytes={"cycle": 15071568, "cycle_x_once": 11999312}, no_exchange_ms_per_step=None,
               exchange_forms_ms_per_step=None, dispatch_isolated_us=4.4, dispatch_in_flight_us=bad, dispatch_samples=[200, 40],
               kernel_us={"pk_cycle": 4.5}, ranks=None, finite=True, side={"mesh_error_estimation": {"pk_err_us": bad}}, end_to_end=None)
    e2e = {"headline": {"ms_per_step": 0.265, "cycles_per_s": 3772.0}, "pcie_wire_frac": 0.74, "pcie_frac": bad,
           "fresh_arrays_compact_layouts": {"cycles_per_s": 4378.0}, "one_call_cycle": {"cycles_per_s": 4100.0},
           "what": "x" * 3000, "pcie": {"floor_us": 230.0}}
    cb = {"value": 175.2, "unit": "cycles/s", "cores": 1, "kind": "port", "sample": "2100 full cycles " + "y" * 500}
    parity = {"max_rel_err": 3e-15, "tol": 1e-11, "ok": True}
    return res, e2e, cb, parity


@pytest.mark.parametrize("nan", [False, True])
def test_the_short_line_is_strict_json_below_four_kilobytes_with_the_fixed_keys(nan):
    args = argparse.Namespace(steps=20, warmup=5, workload="planar_quadrotor", gpus=1)
    res, e2e, cb, parity = canned(nan)
    line = L.short_line(args, res, e2e, 1, 2000, res["ms_per_step"], ROOT, cpu_baseline=cb, parity=parity, detail_file="bench_detail.json")
    text = L.dumps_line(line)
    assert len(text) < 4096 and "\n" not in text
    for token in ("NaN", "Infinity"):
        assert token not in text
    back = json.loads(text)                                   # round trip
    assert json.loads(json.dumps(back)) == back
    for key in REQUIRED:
        assert key in back, key
    assert back["value"] == pytest.approx(1e3 / res["ms_per_step"], rel=1e-6) and back["unit"] == "cycles/s"
    assert back["value_is"] == "device_resident" and back["device_resident"]["value"] == back["value"]
    assert back["end_to_end"]["value"] == pytest.approx(1e3 / 0.265, rel=1e-5) and back["end_to_end"]["pcie_wire_frac"] == 0.74
    roof = back["roofline"]
    for key in ("kernel", "bound", "algorithmic_bytes_per_launch", "avg_launch_us", "achieved", "peak", "frac", "traffic", "profiled"):
        assert key in roof, key
    # frac is reproducible from the line itself: algorithmic bytes / launch duration / peak
    assert roof["frac"] == pytest.approx(roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e9 / roof["peak"], rel=1e-3)
    # beside it the fraction on the bytes that actually move (x read once), the fraction on the committed
    # kernel trace -- reproducible from the line: bytes / profiled.avg_ns / peak -- and a note of at most 80 characters
    assert roof["frac_x_once"] == pytest.approx(11999312 / (roof["avg_launch_us"] * 1e-6) / 1e9 / roof["peak"], rel=1e-3)
    assert roof["frac_x_once"] < roof["frac"] and isinstance(roof["note"], str) and 0 < len(roof["note"]) <= 80
    if roof["profiled"]:
        assert roof["frac_profiled"] == pytest.approx(roof["algorithmic_bytes_per_launch"] / roof["profiled"]["avg_ns"] / roof["peak"], rel=1e-3)
        assert set(roof["profiled"]) == {"file", "avg_ns", "calls", "same_lease_ms_per_step"}
    assert set(back["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"} and len(back["cpu_baseline"]["sample"]) <= 200
    assert back["parity"]["ok"] is True and back["config"]["workload"].startswith("planar_quadrotor LGR 2000")


def test_the_detail_record_is_strict_json_too_and_keeps_the_long_tables():
    args = argparse.Namespace(steps=20, warmup=5, workload="planar_quadrotor", gpus=1)
    res, e2e, cb, parity = canned(True)
    d = L.detail_record(args, res, e2e, 1, 2000, res["ms_per_step"], 0.005, ROOT)
    text = json.dumps(L.sanitize(d), allow_nan=False)
    back = json.loads(text)
    assert back["end_to_end"]["what"] == "x" * 3000 and back["mesh_error_estimation"]["pk_err_us"] is None
    assert back["roofline"]["kernel"] == "pk_cycle" and back["device_resident"]["timing"]["batches"] == 500
