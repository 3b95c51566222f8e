"""What the helper threads of pk_host_threads buy on THIS host: passes of pk_same_bits / pk_copy_bits over arrays of several
sizes with 0 ... 7 helpers (median of 200 back-to-back passes, and of passes 3 ms apart: helpers cold).  No GPU needed.
usage: python3 tools/host_pool_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pockit_amd import runtime  # noqa: E402

lib = runtime.load_library()
print(f"cpus: os.cpu_count() = {os.cpu_count()}, affinity = {len(os.sched_getaffinity(0))}")


def med(fn, reps, gap=0.0):
    fn()
    ts = []
    for _ in range(reps):
        if gap:
            time.sleep(gap)
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return np.median(ts) * 1e6


for n in (96_000, 240_000, 768_000, 1_920_000):
    a = np.random.rand(n)
    b = a.copy()
    c = np.empty(n)
    print(f"n = {n} doubles ({8 * n / 1e6:.2f} MB)")
    for k in (0, 1, 2, 3, 5, 7):
        lib.pk_host_threads(k)
        time.sleep(0.01)
        t1 = med(lambda: lib.pk_same_bits(a.ctypes.data, b.ctypes.data, n), 200)
        t2 = med(lambda: lib.pk_copy_bits(c.ctypes.data, a.ctypes.data, n), 200)
        t3 = med(lambda: lib.pk_same_bits(a.ctypes.data, b.ctypes.data, n), 30, gap=3e-3)
        print(f"   {k} helpers: compare {t1:7.1f} us   copy {t2:7.1f} us   compare after 3 ms of silence {t3:7.1f} us   hot now: "
              f"{lib.pk_host_threads_hot()}", flush=True)
lib.pk_host_threads(0)
