"""Test infrastructure: seeded RANDOM optimal-control models built through the modeling API of a given namespace (the
reference's ``pockit.radau`` / ``pockit.lobatto`` in the build container, ``pockit_amd.*`` on the GPU box, ``oracle.*``), the
same calls in the same order for all of them.  One seed decides everything the transcription depends on: the number of
phases, states, controls, static parameters, integrals and path constraints; which boundary values and times are FREE (None),
FIXED (a number) or FUNC (an expression of the static parameters); bare-symbol constraints (variable / time / static bounds);
objectives linear or nonlinear in the integrals; system constraints; ragged hp meshes.  tests/golden/make_random.py stores
the reference's callback vectors for a list of seeds; tests/test_gpu_random_models.py compares the HIP kernels with them."""
import numpy as np
import sympy as sp


def _term(rng, syms):
    """One smooth term of 1-3 of ``syms``; safe for values in about [-2, 2]."""
    a, b, c = (syms[int(rng.integers(len(syms)))] for _ in range(3))
    k = int(rng.integers(12))
    coef = float(np.round(rng.uniform(0.3, 1.5), 3)) * (1 if rng.random() < 0.7 else -1)
    body = [a, a * b, sp.sin(a), sp.cos(a + b), a**2, a * b * c, sp.exp(-a * a), 1 / (2 + b * b), sp.sqrt(1 + a * a),
            sp.sin(a) * sp.cos(b), a**3 / 3, sp.tanh(a - b)][k]
    return coef * body


def _expr(rng, syms, n_terms=None):
    n = int(rng.integers(1, 4)) if n_terms is None else n_terms
    return sum((_term(rng, syms) for _ in range(n)), sp.Integer(0))


def random_model(ns, seed, scheme="radau", mesh_scale=1):
    """-> (system, phases).  ``scheme`` only bounds the points per interval from below (LGL needs two); ``mesh_scale`` > 1
    multiplies the number of mesh intervals (the soak run: several wave tiles per phase; the fixtures use 1)."""
    rng = np.random.default_rng(seed)
    n_s = int(rng.integers(0, 3))
    system = ns.System(n_s)
    S = list(system.s)
    n_p = int(rng.choice([1, 1, 2]))
    phases, integrals = [], []
    for _ in range(n_p):
        n_x, n_u = int(rng.integers(1, 5)), int(rng.integers(1, 3))
        p = system.new_phase(n_x, n_u)
        X, U, t = list(p.x), list(p.u), p.t
        pool = X + U + ([t] if rng.random() < 0.5 else []) + S
        p.set_dynamics([_expr(rng, pool) for _ in range(n_x)])
        n_I = int(rng.integers(0, 3))
        if n_I:
            p.set_integral([_expr(rng, pool) for _ in range(n_I)])
            integrals += list(p.I)
        # path constraints: expressions and bare symbols (bare symbols become bounds of variables / time / static parameters)
        exprs, lo, hi = [], [], []
        for _ in range(int(rng.integers(0, 3))):
            exprs.append(_expr(rng, X + U, 2))
            lo.append(-float(np.round(rng.uniform(1, 5), 2)))
            hi.append(float(np.round(rng.uniform(1, 5), 2)))
        for sym in ([U[0]] if rng.random() < 0.5 else []) + ([X[-1]] if rng.random() < 0.3 else []) + \
                   ([t] if rng.random() < 0.2 else []) + ([S[0]] if (S and rng.random() < 0.3) else []):
            exprs.append(sym)
            lo.append(-3.0)
            hi.append(4.0)
        if exprs:
            p.set_phase_constraint(exprs, lo, hi)

        def boundary(kind_p=(0.4, 0.4, 0.2)):
            r = rng.random()
            if r < kind_p[0] or (not S and r >= kind_p[0] + kind_p[1]):
                return None
            if r < kind_p[0] + kind_p[1]:
                return float(np.round(rng.uniform(0.5, 1.5), 3))
            return _expr(rng, S, 1) + float(np.round(rng.uniform(0.5, 1.0), 3))
        x0 = [boundary() for _ in range(n_x)]
        xf = [boundary() for _ in range(n_x)]
        t0 = [0.0, None, (S[0] * 0.1 if S else 0.0)][int(rng.integers(3))]
        tf = [None, float(np.round(rng.uniform(1.5, 3.0), 2)), ((S[-1] ** 2 + 2.0) if S else None)][int(rng.integers(3))]
        p.set_boundary_condition(x0, xf, t0, tf)
        n_int = int(rng.integers(1, 6)) * int(mesh_scale)
        cuts = np.sort(rng.uniform(0.1, 0.9, size=n_int - 1)) if n_int > 1 else np.zeros(0)
        mesh = np.concatenate([[0.0], np.round(cuts, 3), [1.0]])
        if np.any(np.diff(mesh) <= 1e-3):
            mesh = np.linspace(0.0, 1.0, n_int + 1)
        k_min = 2 if scheme == "lobatto" else 1
        num_point = [int(rng.integers(k_min, 8)) for _ in range(n_int)]
        p.set_discretization(mesh, num_point)
        phases.append(p)
    system.set_phase(phases)
    terms = integrals + S
    if not terms:
        objective = sp.Integer(0)
    else:
        objective = sum(float(np.round(rng.uniform(0.5, 2.0), 2)) * v for v in terms)
        if integrals and rng.random() < 0.35:            # nonlinear in the integrals: outer-product Hessian blocks
            objective += integrals[0] ** 2 + (integrals[0] * integrals[-1] if len(integrals) > 1 else 0)
        if S and rng.random() < 0.5:
            objective += sp.sin(S[0]) * (integrals[0] if integrals else 1)
    system.set_objective(objective)
    if terms and rng.random() < 0.6:
        cons = [_expr(rng, terms, 2) for _ in range(int(rng.integers(1, 3)))]
        system.set_system_constraint(cons, [-5.0] * len(cons), [5.0] * len(cons))
    return system, phases


def random_inputs(system, seed):
    """x in [0.6, 1.4] everywhere (FIXED / FUNC slots are dictated by the transcription anyway), lambda ~ N(0, 1), sigma 0.7."""
    rng = np.random.default_rng(10_000 + seed)
    n, m = int(system.L), len(system.c_lb)
    return rng.uniform(0.6, 1.4, size=n), rng.standard_normal(m), 0.7


SEEDS = {"radau": list(range(1, 17)), "lobatto": list(range(101, 117))}
