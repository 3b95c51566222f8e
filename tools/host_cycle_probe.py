#!/usr/bin/env python3
"""Developer helper (GPU box): the host-landed NLP-callback cycle (x and lambda in host NumPy arrays, f / grad f / g / J / H
back in host arrays, a new x every cycle) -- where its time goes and what each switch of the host shim is worth.

  1. the C ABI alone: pk_callback_x x 4 + pk_callback_hess per cycle, time of every call (medians);
  2. the five callbacks of ``System`` (what cyipopt calls): caller-owned arrays / zero-copy views / compact Hessian;
  3. A/B of the shim's switches (pk_set_host_option, System.writable_results) on 2.;
  4. the link: one DMA + wait for every transfer of the cycle (the PCIe floor the cycle is read against).

CASE=quadrotor|humanoid|brachistochrone selects the model (default quadrotor 2000 x 6)."""
import ctypes as C
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402

from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

CASES = {"lqr": (models.lqr, 10, 10), "brach20": (models.brachistochrone, 20, 8), "quad100": (models.planar_quadrotor, 100, 6),
         "quad500": (models.planar_quadrotor, 500, 6), "quadrotor": (models.planar_quadrotor, 2000, 6), "humanoid": (models.humanoid_wbc, 5000, 8),
         "brachistochrone": (models.brachistochrone, 1250, 8), "rocket": (models.two_stage_rocket, 1000, 4)}
case = os.environ.get("CASE", "quadrotor")
builder, mesh, K = CASES[case]
med = lambda v: statistics.median(v) * 1e6  # noqa: E731


def build():
    system, _, guess = builder(radau, mesh, K)
    x, lam, sigma = models.bench_inputs(system, guess)
    return system, [x * (1 + 1e-9 * k) for k in range(8)], lam, sigma


def python_cycle(system, xs, lam, sigma, n=100, warm=15):
    rows = []
    for k in range(warm + n):
        xk = xs[k % 8]
        t = [time.perf_counter()]
        system.objective(xk); t.append(time.perf_counter())
        system.gradient(xk); t.append(time.perf_counter())
        system.constraints(xk); t.append(time.perf_counter())
        system.jacobian(xk); t.append(time.perf_counter())
        system.hessian(xk, lam, sigma); t.append(time.perf_counter())
        if k >= warm:
            rows.append([t[i + 1] - t[i] for i in range(5)] + [t[5] - t[0]])
    m = [med([r[i] for r in rows]) for i in range(6)]
    return m


system, xs, lam, sigma = build()
ev = system.evaluator
lib, h, plan = ev.ctx.lib, ev.ctx.handle, system.plan
print(f"{case} {mesh}x{K}: n={plan.n} m={plan.m} nnz_J={plan.nnz_J} nnz_H={plan.nnz_H}; constant runs of J left out of the "
      f"copy: {ev.jac_constant_runs} ({sum(b - a for a, b in ev.jac_constant_runs)} values)", flush=True)

# ---- 1. the C ABI alone (blocks of our own, filled once)
blocks = []
for _ in range(3):
    p = C.c_void_p()
    assert lib.pk_host_alloc(8 * (plan.nnz_J + plan.n + plan.m), C.byref(p)) == 0
    ev.ctx.check(lib.pk_fill_jac_constants(h, p))
    hp = C.c_void_p()
    assert lib.pk_host_alloc(8 * plan.nnz_H, C.byref(hp)) == 0
    blocks.append((p, hp))
f, fresh = C.c_double(), C.c_int()
rows = []
for k in range(220):
    xk = xs[k % 8]
    blk, hb = blocks[k % 3]
    t = [time.perf_counter()]
    for what in range(4):
        rc = lib.pk_callback_x(h, what, xk.ctypes.data, blk, C.addressof(f), C.addressof(fresh))
        assert rc == 0, ev.ctx.lib.pk_last_error(h)
        t.append(time.perf_counter())
    rc = lib.pk_callback_hess(h, xk.ctypes.data, lam.ctypes.data, float(sigma), blk, hb, 0, C.addressof(fresh))
    assert rc == 0
    t.append(time.perf_counter())
    if k >= 20:
        rows.append([t[i + 1] - t[i] for i in range(5)] + [t[5] - t[0]])
m = [med([r[i] for r in rows]) for i in range(6)]
print(f"1. C ABI: objective {m[0]:.1f} | gradient {m[1]:.1f} | constraints {m[2]:.1f} | jacobian {m[3]:.1f} | hessian {m[4]:.1f} | "
      f"cycle {m[5]:.1f} us = {1e6 / m[5]:.0f} cycles/s", flush=True)
ev._invalidate_x()

# ---- 2. / 3. the callbacks of System under every switch
def report(label, modes=("fresh arrays", "fresh arrays, compact Hessian")):
    for mode in modes:
        ev.zero_copy = mode.startswith("zero")
        system.set_hessian_layout("compact" if "compact" in mode else "reference")
        if "compact" in mode and not ev.src.compact:
            continue
        m = python_cycle(system, xs, lam, sigma)
        print(f"   {label:34s} {mode:30s} f/grad/g/J/H {m[0]:6.1f} {m[1]:6.1f} {m[2]:6.1f} {m[3]:6.1f} {m[4]:6.1f} | cycle {m[5]:7.1f} us "
              f"= {1e6 / m[5]:6.0f} cycles/s", flush=True)
    ev.zero_copy = False
    system.set_hessian_layout("reference")


print("2. System callbacks (defaults)")
report("defaults", ("fresh arrays", "zero-copy views", "fresh arrays, compact Hessian"))
print("3. switches")
DEFAULTS = {"small_x_kb": 128, "small_direct": 1, "xpart_single": 1, "mark_wait": 1, "hess_direct": 1, "spin_wait": 1, "lambda_direct": 1, "chunk_upload": 1, "kernel_upload": 1, "kernel_download": 8, "split_copy": 1,
            "speculative_hess": 1}
if os.environ.get("PROBE_ONLY"):          # PROBE_ONLY=hess_direct,split_copy: A/B of these switches only
    DEFAULTS = {k: v for k, v in DEFAULTS.items() if k in os.environ["PROBE_ONLY"].split(",")}
for name, dflt in DEFAULTS.items():
    for other in ((0, 4, 64, 1024) if name == "kernel_download" else (0, 512, 1024, 8192) if name == "small_x_kb" else (1 - dflt,)):
        ev.ctx.check(lib.pk_set_host_option(h, name.encode(), other))
        report(f"{name} = {other}")
    ev.ctx.check(lib.pk_set_host_option(h, name.encode(), dflt))
for combo in (() if os.environ.get("PROBE_ONLY") else ({"lambda_direct": 0, "kernel_upload": 0}, {"kernel_download": 0, "split_copy": 0})):
    for k, v in combo.items():
        ev.ctx.check(lib.pk_set_host_option(h, k.encode(), v))
    report(" ".join(f"{k}={v}" for k, v in combo.items())[:34])
    for k in combo:
        ev.ctx.check(lib.pk_set_host_option(h, k.encode(), DEFAULTS[k]))
ev.set_host_mode(False, False)
report("prefetch = 0")
ev.set_host_mode(True, False)
system._invalidate()
system, xs, lam, sigma = build()
system.writable_results = True             # (writable arrays: the constant entries of J are filled in again per iterate)
ev = system.evaluator
lib, h = ev.ctx.lib, ev.ctx.handle
report("writable results")
system._invalidate()

# ---- 4. the link
import torch  # noqa: E402

dev = torch.device("cuda", 0)
n_const = sum(b - a for a, b in plan.jac_constant_runs() if b - a >= 80_000 or (a == 0 and b >= 1024))
sizes = {"x": 8 * plan.n, "lambda": 8 * plan.m, "J (changing part) | grad f | g": 8 * (plan.nnz_J - n_const + plan.n + plan.m),
         "J | grad f | g": 8 * (plan.nnz_J + plan.n + plan.m), "H": 8 * plan.nnz_H}
print("4. one DMA + wait per transfer of the cycle")
tot = 0.0
for name, nbytes in sizes.items():
    d = torch.zeros(nbytes // 8, dtype=torch.float64, device=dev)
    hbuf = torch.zeros(nbytes // 8, dtype=torch.float64).pin_memory()
    up = name in ("x", "lambda")
    src, dst = (hbuf, d) if up else (d, hbuf)
    for _ in range(5):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    us = med(ts)
    print(f"   {name:32s} {nbytes / 1e6:8.3f} MB {us:8.1f} us {nbytes / us / 1e3:6.1f} GB/s")
    if name != "J | grad f | g":
        tot += us
print(f"   floor of the cycle as four ideal DMAs (x, [J' | grad f | g], lambda, H): {tot:.1f} us = {1e6 / tot:.0f} cycles/s")
