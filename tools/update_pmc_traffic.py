#!/usr/bin/env python3
"""profiles/pmc_traffic.json from ONE lease's evidence (tools/profile_round.sh <tag>): the plain bench line, the rocprofv3
kernel stats and -- when collected -- the PMC passes of the same box and minute.

    update_pmc_traffic.py <tag> <dir with the lease's files> [output json]      (default output: profiles/pmc_traffic.json)

Writes, for the headline workload (planar_quadrotor 2000 intervals):
  pk_cycle_profiled = {avg_ns, calls, min_ns, file, same_lease_ms_per_step, same_lease_line}   -- AverageNs of pk_cycle in the
      kernel trace, and the event-timed ms_per_step of the plain run that preceded it on the same box
  pk_cycle          = FETCH_SIZE + WRITE_SIZE bytes per launch (KB x 1024; no gfx950 doubling on 8-byte-per-lane loads --
      calibrated in round 1, see the file's _note), when the PMC passes ran
and re-assembles the plain run's line with them (<tag>_bench_n1_with_profile.json): what the line reads once the updated
pmc_traffic.json is committed."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, src = sys.argv[1], sys.argv[2]
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "pmc_traffic.json")
    base = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    rec = json.load(open(base)) if os.path.exists(base) else {}
    wl = rec.setdefault("planar_quadrotor_2000", {})
    line = None
    try:
        text = [ln for ln in open(os.path.join(src, f"{tag}_bench_n1.json")) if ln.startswith("{")][-1]
        line = json.loads(text)
    except (OSError, IndexError, ValueError):
        pass
    stats = os.path.join(src, f"{tag}_kernel_stats.csv")
    if os.path.exists(stats):
        for r in csv.DictReader(open(stats)):
            if r["Name"] == "pk_cycle":
                wl["pk_cycle_profiled"] = {
                    "avg_ns": float(r["AverageNs"]), "calls": int(r["Calls"]), "min_ns": float(r["MinNs"]),
                    "file": f"profiles/{tag}_kernel_stats.csv",
                    "same_lease_ms_per_step": (line or {}).get("ms_per_step"), "same_lease_line": f"profiles/{tag}_bench_n1.json",
                    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-extra "
                               f"--no-cpu-baseline --no-end-to-end (tools/profile_round.sh {tag}; the plain line of the same "
                               "lease ran right before it)"}
    pmc = os.path.join(src, f"{tag}_pmc_summary.json")
    if os.path.exists(pmc):
        p = json.load(open(pmc)).get("pk_cycle", {})
        if "FETCH_SIZE_KB_per_launch" in p and "WRITE_SIZE_KB_per_launch" in p:
            fetch, write = p["FETCH_SIZE_KB_per_launch"] * 1024, p["WRITE_SIZE_KB_per_launch"] * 1024
            wl["pk_cycle"] = int(round(fetch + write))
            wl.setdefault("detail", {})[f"pk_cycle_{tag}"] = {
                "FETCH_SIZE_bytes": int(round(fetch)), "WRITE_SIZE_bytes": int(round(write)), "algorithmic_bytes": 15071568,
                "algorithmic_bytes_x_counted_once": 11999312, "source": f"profiles/{tag}_pmc_*.csv (same lease as {tag}_kernel_stats.csv)"}
    with open(dst, "w") as fh:
        json.dump(rec, fh, indent=1)
    prof = wl.get("pk_cycle_profiled")
    if line and prof:
        roof = line["roofline"]
        B = roof["algorithmic_bytes_per_launch"]
        roof["frac_profiled"] = round(B / (prof["avg_ns"] * 1e-9) / 8e12, 4)
        roof["profiled"] = {"file": prof["file"], "avg_ns": prof["avg_ns"], "calls": prof["calls"],
                            "same_lease_ms_per_step": prof["same_lease_ms_per_step"]}
        roof["traffic"] = wl.get("pk_cycle", roof.get("traffic"))
        rel = prof["avg_ns"] * 1e-3 / roof["avg_launch_us"] - 1.0
        roof["note"] = (f"events and same-lease trace agree ({rel * 100:+.1f} %)" if abs(rel) <= 0.03 else
                        f"frac: events of this run; same-lease trace (begin at dispatch) {rel * 100:+.0f} %")[:80]
        with open(os.path.join(src, f"{tag}_bench_n1_with_profile.json"), "w") as fh:
            fh.write(json.dumps(line, separators=(",", ":")) + "\n")
        print(f"[{tag}] ms_per_step {line['ms_per_step']} (events) | trace avg {prof['avg_ns']:.0f} ns over {prof['calls']} | "
              f"frac {roof['frac']} frac_profiled {roof['frac_profiled']} frac_x_once {roof.get('frac_x_once')} | {roof['note']}")


if __name__ == "__main__":
    main()
