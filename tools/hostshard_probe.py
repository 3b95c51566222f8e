"""GPU box: where the host-landed sharded cycle spends its time with ONE rank (the same five callbacks as the single-GPU
host shim, through pockit_amd.hostshard): stage by stage, next to the single-GPU shim.
usage: python3 tools/hostshard_probe.py [intervals] [reps]"""
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    intervals = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    import torch

    import pockit_amd.radau as radau
    from pockit_amd.hostshard import CMD_HESS, CMD_X, HostShardedEvaluator

    torch.cuda.set_device(0)
    system, _, guess = bench.build_workload("planar_quadrotor", intervals, radau)
    xs, lam, sigma = bench.solver_inputs(system, guess)
    hs = HostShardedEvaluator(system.plan, 0, 1, None, device=0)
    lib, h = hs.lib, hs.h
    print(f"n = {hs.plan.n}, helper threads of rank 0: {hs.helper_threads}")
    pc = time.perf_counter
    for k in range(20):
        bench.five_callbacks(hs, xs[k & 1], lam, sigma)
    rows = []
    for k in range(reps):
        x = xs[k & 1]
        t = [pc()]
        same = hs._is_prepared(x)
        t.append(pc())                                   # 0 compare
        lib.pk_copy_bits(hs.h_x.ctypes.data, x.ctypes.data, hs.plan.n)
        t.append(pc())                                   # 1 x into the segment
        seq = hs._post(CMD_X)
        hs._do_x_part(seq)
        t.append(pc())                                   # 2 enqueue the x-part
        hs._wait_marks("early", seq)
        t.append(pc())                                   # 3 wait: sums, grad f, g
        small = hs.h_part[0].copy()
        I = small[: hs.n_I]
        s = x[hs.plan.l_s: hs.plan.r_s]
        args = [float(v) for v in I[: len(hs.plan.I_syms)]] + [float(v) for v in s]
        hs.h_f[0] = hs._F_o(*args) if hs._F_o is not None else hs._F_const
        hs.out["grad"][hs.shared] = small[hs.n_I:]
        hs._x_seq = seq
        t.append(pc())                                   # 4 host sums, f
        hs._is_prepared(x)
        t.append(pc())                                   # 5 a same-x compare (gradient / constraints / jacobian each pay one)
        hs._wait_marks("x", seq)
        t.append(pc())                                   # 6 wait: J
        lib.pk_copy_bits(hs.h_lam.ctypes.data, lam.ctypes.data, hs.plan.m)
        hs.h_sigma[0] = sigma
        t.append(pc())                                   # 7 lambda into the segment
        seq = hs._post(CMD_HESS)
        hs._do_hess(seq)
        t.append(pc())                                   # 8 enqueue H
        hs._wait_marks("h", seq)
        t.append(pc())                                   # 9 wait H
        rows.append([t[i + 1] - t[i] for i in range(len(t) - 1)])
        assert not same
    names = ["compare x (differs)", "x -> segment", "enqueue x-part", "wait early mark", "host sums + f", "compare x (same)",
             "wait J mark", "lambda -> segment", "enqueue H + run copy", "wait H mark"]
    tot = 0.0
    for i, nm in enumerate(names):
        v = statistics.median(r[i] for r in rows) * 1e6
        tot += v
        print(f"{nm:28s} {v:8.1f} us")
    print(f"{'sum':28s} {tot:8.1f} us")
    batches = bench.timed_cycles(lambda k: bench.five_callbacks(hs, xs[k & 1], lam, sigma), 20, 5)
    print(f"five callbacks through hostshard (1 rank): {statistics.median(batches) / 20 * 1e6:.1f} us per cycle")
    n_tab = [("tab_early", hs.tab_early[1]), ("tab_j", hs.tab_j[1]), ("tab_j_changing", hs.tab_j_changing[1]), ("tab_h", hs.tab_h[1]),
             ("tab_xin", hs.tab_xin[1])]
    print("run-table chunks:", n_tab)
    hs.close()
    system2, _, guess2 = bench.build_workload("planar_quadrotor", intervals, radau)
    batches = bench.timed_cycles(lambda k: bench.five_callbacks(system2, xs[k & 1], lam, sigma), 20, 5)
    print(f"five callbacks through the single-GPU shim:  {statistics.median(batches) / 20 * 1e6:.1f} us per cycle")


if __name__ == "__main__":
    main()
