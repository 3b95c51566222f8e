"""Assembly of what ``bench.py`` prints: ONE short JSON line (strict JSON, < 4 KB, fixed keys -- what the driver parses)
and the detail record written beside it (``bench_detail.json``: timing statistics, per-callback tables, side kernels,
the other workloads, prose notes).  Pure Python: importable without torch or a GPU (tests/test_bench_line.py).

Which figure is ``value``: the task text of this tier defines it as the whole-job throughput "with inputs already resident
in HBM when the timed region starts" and rules that the PCIe-inclusive rate "is never ``value``"; so ``value`` /
``ms_per_step`` are the DEVICE-RESIDENT cycle (x, lambda and all outputs stay in HBM, one pk_cycle launch per cycle) and the
host-landed (solver-visible) cycle stands beside it under ``end_to_end``.  Both are in every line, under these names."""
import json
import math
import os

from .workloads import HBM_ACHIEVABLE_GBPS, HBM_PEAK_GBPS, LAUNCH_FLOOR_US, POINTS

LINE_LIMIT = 4096
METRIC = "NLP-callback cycles/sec (f + grad f + g + J + H)"
SHARDING_NOTE = {
    "sums": "mesh intervals over {n} GPUs (shares balanced by output volume), one pk_cycle launch per rank on its tiles; every "
            "rank's slices of grad/g/J/H stay in its own HBM at the reference positions; the sums over all nodes (integrals -> "
            "f, gradient entries of t0/tf/static parameters) are exchanged through peer-mapped mailboxes by the launch's finalize "
            "workgroup (or by a one-workgroup launch pk_xchg behind it) -- no collective in the data path",
    "direct": "mesh intervals over {n} GPUs, one pk_cycle launch per rank; the other ranks' kernels store their slices "
              "straight into rank 0's buffer through hipIpc peer mappings (xGMI), pk_xchg flags completion",
    "gather": "mesh intervals over {n} GPUs, one pk_cycle launch per rank, run-copy pack + RCCL gather to rank 0 + run-copy "
              "unpack of the owned runs of grad/g/J/H (+ the partial sums)",
    "allgather": "mesh intervals over {n} GPUs, one pk_cycle launch per rank, run-copy pack + RCCL all-gather + run-copy "
                 "unpack of the owned runs of grad/g/J/H, tiny all-reduce of the partial sums",
    "single GPU": "single GPU",
}


def _num(v, digits=6):
    """A float for the short line: finite or None, rounded to ``digits`` significant digits."""
    if v is None:
        return None
    try:
        v = float(v)
    except (TypeError, ValueError):
        return None
    if not math.isfinite(v):
        return None
    if v == 0.0:
        return 0.0
    return round(v, digits - 1 - int(math.floor(math.log10(abs(v)))))


def sanitize(obj):
    """NaN / Infinity -> None, NumPy scalars -> Python numbers: what ``json.dumps(allow_nan=False)`` accepts."""
    if isinstance(obj, dict):
        return {str(k): sanitize(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [sanitize(v) for v in obj]
    if isinstance(obj, float):
        return obj if math.isfinite(obj) else None
    if isinstance(obj, (str, int, bool)) or obj is None:
        return obj
    if hasattr(obj, "item"):
        return sanitize(obj.item())
    return repr(obj)


def workload_text(workload, intervals, res):
    return (f"{workload} LGR {intervals} intervals x {POINTS.get(workload.replace('_lgl', ''), 0)} points ({res['nodes']} nodes; "
            f"n={res['n']}, m={res['m']}, nnz_J={res['nnz_J']}, nnz_H={res['nnz_H']})")


def roofline_of(res, n_gpus, ms, workload, intervals, root):
    """The dominant kernel against the HBM roof: algorithmic bytes of one launch over its average launch
    duration measured by HIP events on the launch stream over the timed region."""
    dom = res["dominant"]
    dom_bytes = res["bytes"][dom[3:]] / n_gpus
    x_once = res["bytes"]["cycle_x_once"] / n_gpus if dom == "pk_cycle" else None
    # one launch per cycle: events over the region / launches (launch + gap to the next: an upper bound of the kernel's own
    # duration); a cycle of several launches, or N > 1 (the region holds the exchange too): per-dispatch events
    dom_us = ms * 1e3 if (dom == "pk_cycle" and n_gpus == 1) else res["dispatch_isolated_us"]
    achieved = dom_bytes / (dom_us * 1e-6) / 1e9 if dom_us else None
    traffic, profiled = None, None
    tpath = os.path.join(root, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get(f"{workload}_{intervals}", {})
            traffic = rec.get(dom)
            profiled = rec.get(dom + "_profiled")     # {"avg_ns", "calls", "file"} of the committed kernel trace
        except Exception:  # noqa: BLE001
            traffic = None
    out_bytes = 8 * (1 + res["n"] + res["m"] + res["nnz_J"] + res["nnz_H"]) / n_gpus
    regime = "hbm" if out_bytes > 256 * 2**20 else ("latency" if (dom_us and dom_us < 2 * LAUNCH_FLOOR_US) else "mall")
    iso = res.get("dispatch_isolated_us")
    return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBPS if achieved else None), "traffic": traffic,
            "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": dom_us, "regime": regime,
            "frac_of_achievable": (achieved / HBM_ACHIEVABLE_GBPS if achieved else None), "achievable_peak": HBM_ACHIEVABLE_GBPS,
            "frac_x_once": (x_once / (dom_us * 1e-6) / 1e9 / HBM_PEAK_GBPS if (x_once and dom_us) else None),
            "algorithmic_bytes_per_launch_x_counted_once": x_once,
            "frac_profiled": ((dom_bytes / (profiled["avg_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBPS)
                              if (profiled and profiled.get("avg_ns")) else None),
            "profiled": profiled, "dispatch_isolated_us": iso,
            "frac_dispatch_isolated": (dom_bytes / (iso * 1e-6) / 1e9 / HBM_PEAK_GBPS if iso else None),
            "dispatch_in_flight_us": res.get("dispatch_in_flight_us"),
            "dispatch_samples_isolated_in_flight": res.get("dispatch_samples")}


def roofline_note(roof):
    """<= 80 characters: what the committed kernel trace says beside the events of THIS run.  ``frac`` always rests on the HIP
    events of this run (the contract); the committed pair -- trace and event-timed line of ONE lease, tools/profile_round.sh --
    shows how far rocprofv3's begin-at-dispatch duration sits from the event-timed step on the same box."""
    prof, us = roof.get("profiled"), roof.get("avg_launch_us")
    if not (prof and prof.get("avg_ns") and us):
        return "frac: HIP events of this run; no committed trace for this workload"
    same = prof.get("same_lease_ms_per_step")
    if same:
        rel = prof["avg_ns"] * 1e-6 / same - 1.0
        return f"frac: events; committed pair: trace {prof['avg_ns'] * 1e-3:.2f} us vs events {same * 1e3:.2f} us ({rel * 100:+.1f} %)"[:80]
    rel = prof["avg_ns"] * 1e-3 / us - 1.0
    return f"frac: events of this run; trace of another lease {rel * 100:+.0f} %"[:80]


def multi_gpu_of(res, e2e, n_gpus, strong):
    """Fixed keys of an N > 1 line beside ``value`` (which stays the device-resident weak-scaled cycle, as the task text rules):
    ``gather_rccl`` -- the form north_star words (pack, RCCL gather / all-gather over xGMI, unpack into the reference's triplet
    order), ``strong`` -- BASELINE's workloads at their own size split over the N GPUs, plain cycles/s, ``host_landed`` -- every
    GPU landing its slices in one host array.  A form that did not run is null (or its error text)."""
    mg = res.get("multi_gpu") or {}
    forms = res.get("exchange_forms_ms_per_step") or {}
    is_rccl = str(mg.get("backend", "")).startswith("nccl")

    def form(key):
        ms = forms.get(key)
        if isinstance(ms, (int, float)):
            return {"value": _num(n_gpus * 1e3 / ms), "ms_per_step": _num(ms)}
        return {"value": None, "ms_per_step": None, "error": (str(ms)[:100] if ms else "not measured")}

    g = form("gather")
    g.update(unit="12k-node-equivalent cycles/s", allgather=form("allgather"), backend=mg.get("backend"),
             what="pack + collective + unpack into reference triplet order on the device, in the timed loop")
    st = {}
    for key, rec in (strong or {}).items():
        name = ("C3_12k" if key.startswith("planar_quadrotor") else "C4_2x1000x4" if key.startswith("two_stage_rocket") else
                "C5_5000x8" if key.startswith("humanoid_wbc") else key)
        if isinstance(rec, dict) and "device_resident" in rec:
            dr = rec["device_resident"]
            gat = (dr.get("exchange_forms_ms_per_step") or {}).get("gather")
            st[name] = {"value": _num(dr.get("cycles_per_s")), "ms_per_step": _num(dr.get("ms_per_step")),
                        "gather_rccl_value": (_num(1e3 / gat) if isinstance(gat, (int, float)) and gat else None),
                        "host_landed_value": _num(rec.get("end_to_end_cycles_per_s"))}
        else:
            st[name] = {"value": None, "error": str((rec or {}).get("error", "not measured"))[:100]}
    ee = end_to_end_of(e2e, n_gpus) or {}
    return {"ranks_seen_by_rccl": (mg.get("ranks") if is_rccl else None), "ranks": mg.get("ranks"), "backend": mg.get("backend"),
            "device_resident_form": mg.get("device_resident_form"),
            "fallback": (str(mg.get("device_resident_form_fallback"))[:120] if mg.get("device_resident_form_fallback") else None),
            "gather_rccl": g, "strong": (st or None), "strong_unit": "cycles/s (same workload as N = 1, intervals split over N)",
            "host_landed": {"value": ee.get("value"), "ms_per_step": ee.get("ms_per_step")}}


def end_to_end_of(e2e, n_gpus):
    """The host-landed (solver-visible) cycle in the short line: the five callbacks on a new x with host arrays in and
    out; N > 1: every GPU landing its slices in one pinned host array."""
    if not isinstance(e2e, dict):
        return None
    if "error" in e2e:
        return {"value": None, "ms_per_step": None, "error": str(e2e["error"])[:160]}
    if n_gpus == 1:
        head = e2e.get("headline") or {}
        ms = head.get("ms_per_step")
        out = {"value": _num(1e3 / ms) if ms else None, "ms_per_step": _num(ms), "pcie_wire_frac": _num(e2e.get("pcie_wire_frac"), 3),
               "pcie_frac": _num(e2e.get("pcie_frac"), 3)}
        for key, name in (("fresh_arrays_compact_layouts", "compact_layouts_value"), ("fresh_arrays_compact_hessian", "compact_hessian_value"),
                          ("one_call_cycle", "one_call_value")):
            if isinstance(e2e.get(key), dict):
                out[name] = _num(e2e[key].get("cycles_per_s"))
        return out
    ms = e2e.get("ms_per_cycle")
    return {"value": _num(n_gpus * 1e3 / ms) if ms else None, "ms_per_step": _num(ms), "ranks": e2e.get("ranks"),
            "form": "every GPU lands its slices in one pinned host array over its own PCIe link"}


def detail_record(args, res, e2e, n_gpus, intervals, ms, wall_ms, root):
    """Everything the run measured about the headline workload (the long record of rounds 1-3, now a file)."""
    roof = roofline_of(res, n_gpus, ms, args.workload, intervals, root)
    rec = {
        "metric": METRIC, "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "config": {"workload": workload_text(args.workload, intervals, res),
                   "sharding": SHARDING_NOTE.get(res["exchange"], res["exchange"]).format(n=n_gpus),
                   "tiles": res["tiles"], "intervals_per_wave": res["ipw"],
                   "inputs": "example guess*(1+1e-3 U(-1,1)) seed 0; lambda N(0,1) seed 1; sigma 1"},
        "device_resident": {
            "value": n_gpus * 1e3 / ms, "ms_per_step": ms,
            "unit": "cycles/s" if n_gpus == 1 else "12k-node-equivalent cycles/s",
            "what": "x and lambda resident in HBM, all outputs left in HBM: ONE pk_cycle launch per cycle and GPU",
            "timing": {"region": f"{res['batches']} back-to-back batches of exactly {args.steps} cycles, a HIP event on the launch "
                                 f"stream after every {res['event_group']} batches, barrier + synchronize around the region; "
                                 f"ms_per_step = median over the timed units / batches / steps (max over ranks)",
                       "batches": res["batches"], "batches_per_timing_event": res["event_group"], "batch_launch": res["batch_launch"],
                       "median_batch_ms": res["median_batch_ms"],
                       "batch_ms_min_p10_p90_max": [res["batch_ms_min"], res["batch_ms_p10"], res["batch_ms_p90"], res["batch_ms_max"]],
                       "region_wall_s": res["region_wall_s"], "wall_ms_per_step_whole_region": wall_ms,
                       "untimed_launches_before_the_region": res["untimed_launches"]}},
        "roofline": roof,
        "kernels_only_without_exchange": (None if res.get("no_exchange_ms_per_step") is None else {
            "value": n_gpus * 1e3 / res["no_exchange_ms_per_step"], "unit": "12k-node-equivalent cycles/s"}),
        "exchange_forms": (None if res.get("exchange_forms_ms_per_step") is None else {
            "ms_per_step": res["exchange_forms_ms_per_step"],
            "equivalent_cycles_per_s": {k: (n_gpus * 1e3 / v if isinstance(v, float) else v)
                                        for k, v in res["exchange_forms_ms_per_step"].items()}}),
        "ranks": res.get("ranks"), "multi_gpu": res.get("multi_gpu"), "exchange_check": res.get("exchange_check"),
        "kernel_us": res["kernel_us"], "cycle_algorithmic_bytes": res["bytes"]["cycle"],
        "setup_s": res["setup_s"], "compile_s_in_setup": res["compile_s_in_setup"], "outputs_finite": res["finite"],
        "end_to_end": e2e,
    }
    rec.update(res.get("side") or {})
    return rec


def short_line(args, res, e2e, n_gpus, intervals, ms, root, cpu_baseline=None, parity=None, detail_file=None, strong=None):
    """The ONE line the driver parses.  Keys are fixed; everything else lives in the detail file."""
    roof = roofline_of(res, n_gpus, ms, args.workload, intervals, root)
    prof = roof.get("profiled")
    value = n_gpus * 1e3 / ms
    line = {
        "metric": METRIC, "value": _num(value, 7), "unit": "cycles/s" if n_gpus == 1 else "12k-node-equivalent cycles/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": _num(ms, 7),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "value_is": "device_resident",
        "config": {"workload": workload_text(args.workload, intervals, res),
                   "sharding": "single GPU" if n_gpus == 1 else f"mesh intervals over {n_gpus} GPUs, form '{res['exchange']}'",
                   "inputs": "guess*(1+1e-3 U) seed 0; lambda N(0,1) seed 1; sigma 1"},
        "roofline": {"kernel": roof["kernel"], "bound": "hbm", "algorithmic_bytes_per_launch": int(roof["algorithmic_bytes_per_launch"]),
                     "avg_launch_us": _num(roof["avg_launch_us"]), "achieved": _num(roof["achieved"]), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": _num(roof["frac"], 4), "traffic": roof["traffic"], "regime": roof["regime"],
                     "dispatch_isolated_us": _num(roof["dispatch_isolated_us"]), "frac_profiled": _num(roof["frac_profiled"], 4),
                     "frac_x_once": _num(roof["frac_x_once"], 4),
                     "profiled": ({"file": prof.get("file"), "avg_ns": prof.get("avg_ns"), "calls": prof.get("calls"),
                                   "same_lease_ms_per_step": prof.get("same_lease_ms_per_step")} if prof else None),
                     "note": roofline_note(roof)},
        "cpu_baseline": None,
        "device_resident": {"value": _num(value, 7), "ms_per_step": _num(ms, 7),
                            "compact_layouts_value": _num(((res.get("side") or {}).get("compact_cycle_mode") or {}).get("cycles_per_s"))},
        "end_to_end": end_to_end_of(e2e, n_gpus),
        "parity": parity,
        "outputs_finite": bool(res.get("finite")),
        "detail_file": detail_file,
    }
    if cpu_baseline is not None:
        cb = cpu_baseline
        line["cpu_baseline"] = {"value": _num(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": str(cb.get("sample", ""))[:200]}
        line["speedup_vs_cpu_baseline"] = {"device_resident": _num(value / cb["value"], 4)}
        ee = line["end_to_end"]
        if ee and ee.get("value"):
            line["speedup_vs_cpu_baseline"]["end_to_end"] = _num(ee["value"] / cb["value"], 4)
    if n_gpus > 1:
        line["multi_gpu"] = multi_gpu_of(res, e2e, n_gpus, strong)
    return line


def dumps_line(line):
    """Strict JSON on one line, shorter than LINE_LIMIT: optional keys are dropped (never the contract's) if it is not."""
    line = sanitize(line)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    for key in ("speedup_vs_cpu_baseline", "parity", "detail_file"):
        if len(text) < LINE_LIMIT:
            break
        line.pop(key, None)
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:          # (cannot happen with the fixed keys above; a guard, not a path)
        for obj in (line.get("config"), line.get("cpu_baseline"), line.get("end_to_end")):
            if isinstance(obj, dict):
                for k, v in list(obj.items()):
                    if isinstance(v, str) and len(v) > 60:
                        obj[k] = v[:60]
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(text) < LINE_LIMIT and "\n" not in text
    return text
