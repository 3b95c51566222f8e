#!/usr/bin/env python3
"""Developer helper (GPU box): where the host-buffer cycle of the C ABI spends its time -- pk_prepare_x (enqueue only), the
waits of pk_fetch for f / grad f | g / J, pk_stage_lambda, pk_eval_hess_prepared -- medians over 200 cycles of the
12k-node quadrotor, called through ctypes without the Python callbacks of System."""
import sys, time, statistics, ctypes as C
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pockit_amd import benchmarks as models
import pockit_amd.radau as radau
system, _, guess = models.planar_quadrotor(radau, 2000, 6)
x, lam, sigma = models.bench_inputs(system, guess)
ev = system.evaluator
lib, h = ev.ctx.lib, ev.ctx.handle
xs = [x * (1 + 1e-9 * k) for k in range(8)]
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
for k in range(20):
    system.objective(xs[k % 8]); system.gradient(xs[k % 8]); system.constraints(xs[k % 8]); system.jacobian(xs[k % 8]); system.hessian(xs[k % 8], lam, sigma)
tp, tf, tg, tj, th1, th2 = [], [], [], [], [], []
for k in range(200):
    xk = xs[k % 8]
    t0 = time.perf_counter(); ev.ctx.check(lib.pk_prepare_x(h, dp(xk))); t1 = time.perf_counter()
    ev.ctx.check(lib.pk_fetch(h, 0, None)); t2 = time.perf_counter()
    ev.ctx.check(lib.pk_fetch(h, 1, None)); t3 = time.perf_counter()
    ev.ctx.check(lib.pk_fetch(h, 3, None)); t4 = time.perf_counter()
    ev.ctx.check(lib.pk_stage_lambda(h, dp(lam))); t5 = time.perf_counter()
    ev.ctx.check(lib.pk_eval_hess_prepared(h, None, C.c_double(sigma), None)); t6 = time.perf_counter()
    tp.append(t1 - t0); tf.append(t2 - t1); tg.append(t3 - t2); tj.append(t4 - t3); th1.append(t5 - t4); th2.append(t6 - t5)
med = lambda v: statistics.median(v) * 1e6
print(f"prepare_x (enqueue only) {med(tp):.1f} us | wait f {med(tf):.1f} | wait grad {med(tg):.1f} | wait J {med(tj):.1f} | stage lambda {med(th1):.1f} | hess (enqueue + wait) {med(th2):.1f} | total {med(tp)+med(tf)+med(tg)+med(tj)+med(th1)+med(th2):.1f}")
