#!/usr/bin/env python3
"""Developer helper (GPU box): what the PCIe link of this box delivers for the transfer sizes of the host cycle --
device -> pinned host and pinned host -> device copies (one DMA each, synchronized), so that the end-to-end figures of
bench.py can be read against the link instead of against a data-sheet number."""
import time

import torch

dev = torch.device("cuda", 0)
print(f"{'MB':>9s} {'D2H us':>9s} {'D2H GB/s':>9s} {'H2D us':>9s} {'H2D GB/s':>9s}")
for nbytes in (8, 768064, 2879776, 6047400, 12_000_000, 47_359_088, 57_278_608, 120_637_896):
    n = max(nbytes // 8, 1)
    d = torch.zeros(n, dtype=torch.float64, device=dev)
    h = torch.zeros(n, dtype=torch.float64).pin_memory()
    out = []
    for src, dst in ((d, h), (h, d)):
        for _ in range(5):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        reps = 50 if nbytes < 20_000_000 else 15
        t0 = time.perf_counter()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out += [dt * 1e6, nbytes / dt / 1e9]
    print(f"{nbytes / 1e6:9.3f} {out[0]:9.1f} {out[1]:9.1f} {out[2]:9.1f} {out[3]:9.1f}", flush=True)

# the same device -> host transfers split over two streams (two copies in flight: do two DMA engines share the link better
# than one uses it?)
print(f"\n{'MB':>9s} {'1 stream us':>12s} {'2 streams us':>13s} {'4 streams us':>13s}")
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
for nbytes in (1_536_064, 2_879_776, 6_047_400, 57_278_608):
    n = nbytes // 8
    d = torch.zeros(n, dtype=torch.float64, device=dev)
    h = torch.zeros(n, dtype=torch.float64).pin_memory()
    row = []
    for parts in (1, 2, 4):
        cuts = [n * k // parts for k in range(parts + 1)]

        def go():
            for k in range(parts):
                with torch.cuda.stream(streams[k]):
                    h[cuts[k]: cuts[k + 1]].copy_(d[cuts[k]: cuts[k + 1]], non_blocking=True)
            torch.cuda.synchronize()

        for _ in range(5):
            go()
        t0 = time.perf_counter()
        for _ in range(40):
            go()
        row.append((time.perf_counter() - t0) / 40 * 1e6)
    print(f"{nbytes / 1e6:9.3f} {row[0]:12.1f} {row[1]:13.1f} {row[2]:13.1f}   ({nbytes / row[0] / 1e3:.1f} / {nbytes / row[1] / 1e3:.1f} / {nbytes / row[2] / 1e3:.1f} GB/s)", flush=True)
