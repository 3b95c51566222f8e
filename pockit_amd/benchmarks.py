"""Benchmark / parity models, written against the *shared* modeling API only (used by bench.py,
``__graft_entry__`` and, through ``tests/models.py``, by the tests and the golden-vector generator).

Every builder takes a namespace ``ns`` exposing ``System``, ``linear_guess`` and
``constant_guess`` (``pockit.radau`` / ``pockit.lobatto`` of the reference in the golden
generator, ``pockit_amd.radau`` / ``pockit_amd.lobatto`` for the product, ``oracle.radau`` /
``oracle.lobatto`` for the CPU restatement), so that the same model text drives all three.

The model constants restate the reference's example inputs (they are benchmark data, not code):
  * LQR ......................... README.md:95-118, examples/linear_quadratic_regulator.py:47-62
  * brachistochrone ............. examples/brachistochrone.py:38-75
  * planar_quadrotor ............ examples/planar_quadrotor.py:40-186
  * two_stage_rocket ............ examples/multiphase_two_stage_rocket.py:36-190
  * humanoid_wbc ................ examples/humanoid_whole_body_control.py:42-297
  * derivative_model ............ tests/test_radau/test_derivative_radau.py:11-41
  * worked_model ................ SURVEY.md Appendix A.4
Meshes are parameters (BASELINE.json re-meshes the example models on LGR).
"""
from __future__ import annotations

import numpy as np
import sympy as sp


# --------------------------------------------------------------------------- helpers
def pack_guess(system, guess):
    """x0 = [phase data ... | static]  (reference: pockit/optimizer/_common.py:30-34)."""
    x0 = np.zeros(int(system.L))
    for i in range(system.n_p):
        x0[system.l_p[i]: system.r_p[i]] = guess[i].data
    if system.n_s > 0:
        x0[system.l_s: system.r_s] = np.asarray(list(guess[-1]), dtype=np.float64)
    return x0


def bench_inputs(system, guess):
    """Seeded evaluation point of SURVEY.md section 8(d): x0*(1+1e-3 U), lambda ~ N(0,1), sigma=1."""
    x = pack_guess(system, guess)
    x = x * (1.0 + 1.0e-3 * np.random.default_rng(0).uniform(-1.0, 1.0, x.shape))
    lam = np.random.default_rng(1).standard_normal(len(system.c_lb))
    return x, lam, 1.0


# --------------------------------------------------------------------------- LQR
def lqr(ns, mesh=10, num_point=10):
    a, b, s_w, q, r = -1.0, 1.0, 1.0, 1.0, 0.1
    system = ns.System(["x_f"])
    (x_f,) = system.s
    phase = system.new_phase(["x"], ["u"])
    (x,) = phase.x
    (u,) = phase.u
    phase.set_dynamics([a * x + b * u])
    phase.set_integral([q * x**2 + r * u**2])
    phase.set_boundary_condition([1], [x_f], 0, 1)
    phase.set_discretization(mesh, num_point)
    system.set_phase([phase])
    system.set_objective(phase.I[0] + s_w * x_f**2 / 2)
    guess = [ns.constant_guess(phase, 0.3), [0.2]]
    return system, [phase], guess


# --------------------------------------------------------------------------- brachistochrone
def brachistochrone(ns, mesh=10, num_point=8):
    gravity, target_x, target_y = 9.81, 2.0, 2.0
    system = ns.System(0)
    phase = system.new_phase(["x", "y", "speed"], ["path_angle"])
    _, _, speed = phase.x
    (path_angle,) = phase.u
    phase.set_dynamics(
        [speed * sp.sin(path_angle), speed * sp.cos(path_angle), gravity * sp.cos(path_angle)]
    )
    phase.set_integral([1.0])
    phase.set_phase_constraint([speed, path_angle], [0.0, 0.0], [np.inf, np.pi / 2.0])
    phase.set_boundary_condition([0.0, 0.0, 0.0], [target_x, target_y, None], 0.0, None)
    phase.set_discretization(mesh, num_point)
    system.set_phase([phase])
    system.set_objective(phase.I[0])

    guess = ns.linear_guess(phase, 0.0)
    guess.t_f = 1.0
    tau = guess.t_x / guess.t_f
    guess.x[0] = target_x * tau
    guess.x[1] = target_y * tau
    guess.x[2] = np.sqrt(2.0 * gravity * guess.x[1])
    guess.u[0] = np.arctan2(target_x, target_y)
    return system, [phase], [guess]


# --------------------------------------------------------------------------- planar quadrotor
_Q = dict(
    MASS=1.20, PITCH_INERTIA=0.025, GRAVITY=9.81, HORIZON=5.0,
    START=(0.0, 0.0), TARGET=(5.0, 0.0), OBST=(2.5, 0.80), OBST_R=0.80, CLEAR=0.12,
    MAX_TORQUE=0.25, MAX_PITCH=float(np.deg2rad(65.0)), MAX_PITCH_RATE=3.0, GUARD=0.03,
)


def _quadrotor_profiles(time):
    c = _Q
    f = np.asarray(time) / c["HORIZON"]
    prog = 10.0 * f**3 - 15.0 * f**4 + 6.0 * f**5
    prog_r = (30.0 * f**2 - 60.0 * f**3 + 30.0 * f**4) / c["HORIZON"]
    prog_a = (60.0 * f - 180.0 * f**2 + 120.0 * f**3) / c["HORIZON"] ** 2
    arch = 1.82
    span = c["TARGET"][0] - c["START"][0]
    x = c["START"][0] + span * prog
    vx = span * prog_r
    ax = span * prog_a
    z = arch * np.sin(np.pi * f) ** 2
    vz = arch * np.pi * np.sin(2.0 * np.pi * f) / c["HORIZON"]
    az = 2.0 * arch * np.pi**2 * np.cos(2.0 * np.pi * f) / c["HORIZON"] ** 2
    sf = az + c["GRAVITY"]
    pitch = -np.arctan2(ax, sf)
    thrust = c["MASS"] * np.hypot(ax, sf)
    return x, z, vx, vz, pitch, thrust


def planar_quadrotor(ns, mesh=14, num_point=6, fastmath=True):
    c = _Q
    max_thrust = 2.2 * c["MASS"] * c["GRAVITY"]
    safe_radius = c["OBST_R"] + c["CLEAR"]
    system = ns.System(0, fastmath=fastmath)
    phase = system.new_phase(
        ["x", "z", "velocity_x", "velocity_z", "pitch", "pitch_rate"], ["thrust", "torque"]
    )
    x, z, vx, vz, pitch, pitch_rate = phase.x
    thrust, torque = phase.u
    phase.set_dynamics(
        [
            vx,
            vz,
            -thrust * sp.sin(pitch) / c["MASS"],
            thrust * sp.cos(pitch) / c["MASS"] - c["GRAVITY"],
            pitch_rate,
            torque / c["PITCH_INERTIA"],
        ]
    )
    hover = c["MASS"] * c["GRAVITY"]
    phase.set_integral(
        [
            0.025 * ((thrust - hover) / hover) ** 2
            + 0.012 * (torque / c["MAX_TORQUE"]) ** 2
            + 0.002 * pitch_rate**2
            + 0.004 * z**2
        ]
    )
    dist2 = (x - c["OBST"][0]) ** 2 + (z - c["OBST"][1]) ** 2
    nt = phase.t / c["HORIZON"]
    ground_guard = 16.0 * c["GUARD"] * nt**2 * (1.0 - nt) ** 2
    enforced_radius = safe_radius + 0.004
    phase.set_phase_constraint(
        [x, z, pitch, pitch_rate, thrust, torque, dist2, z - ground_guard],
        [-0.20, 0.0, -c["MAX_PITCH"], -c["MAX_PITCH_RATE"], 0.0, -c["MAX_TORQUE"],
         enforced_radius**2, 0.0],
        [5.20, 3.0, c["MAX_PITCH"], c["MAX_PITCH_RATE"], max_thrust, c["MAX_TORQUE"],
         np.inf, np.inf],
    )
    phase.set_boundary_condition(
        [c["START"][0], c["START"][1], 0.0, 0.0, 0.0, 0.0],
        [c["TARGET"][0], c["TARGET"][1], 0.0, 0.0, 0.0, 0.0],
        0.0,
        c["HORIZON"],
    )
    phase.set_discretization(mesh, num_point)
    system.set_phase([phase])
    system.set_objective(phase.I[0])

    guess = ns.linear_guess(phase, 0.0)
    gx, gz, gvx, gvz, gpitch, _ = _quadrotor_profiles(guess.t_x)
    guess.x[0] = gx
    guess.x[1] = gz
    guess.x[2] = gvx
    guess.x[3] = gvz
    guess.x[4] = gpitch
    guess.x[5] = np.gradient(gpitch, guess.t_x, edge_order=2)
    _, _, _, _, pitch_u, thrust_u = _quadrotor_profiles(guess.t_u)
    rate_u = np.gradient(pitch_u, guess.t_u, edge_order=2)
    torque_u = c["PITCH_INERTIA"] * np.gradient(rate_u, guess.t_u, edge_order=2)
    guess.u[0] = np.clip(thrust_u, 0.0, max_thrust)
    guess.u[1] = np.clip(torque_u, -c["MAX_TORQUE"], c["MAX_TORQUE"])
    return system, [phase], [guess]


# --------------------------------------------------------------------------- two-stage rocket
def two_stage_rocket(ns, mesh=96, num_point=1):
    gravity, m0 = 1.0, 1.0
    prop1, drop, prop2 = 0.06, 0.20, 0.12
    burnout1 = m0 - prop1
    m2_0 = burnout1 - drop
    m2_dry = m2_0 - prop2
    target_alt = 2.0

    system = ns.System(
        ["h_separation", "v_separation", "m_before_drop", "m_after_drop", "t_separation",
         "m_final", "t_final"]
    )
    h_s, v_s, m_before, m_after, t_s, m_f, t_f = system.s

    def stage(phase, thrust, mass_flow, mass_bounds, boundaries, times):
        altitude, velocity, mass = phase.x
        (throttle,) = phase.u
        phase.set_dynamics([velocity, thrust * throttle / mass - gravity, -mass_flow * throttle])
        phase.set_integral([throttle**2])
        phase.set_phase_constraint(
            [throttle, altitude, velocity, mass],
            [0.0, 0.0, 0.0, mass_bounds[0]],
            [1.0, target_alt, 3.0, mass_bounds[1]],
            [True, False, False, False],
        )
        phase.set_boundary_condition(*boundaries, *times)
        phase.set_discretization(mesh, num_point)

    p1 = system.new_phase(["altitude_1", "velocity_1", "mass_1"], ["throttle_1"])
    stage(p1, 2.40, 0.080, (burnout1, m0), ([0.0, 0.0, m0], [h_s, v_s, m_before]), (0.0, t_s))
    p2 = system.new_phase(["altitude_2", "velocity_2", "mass_2"], ["throttle_2"])
    stage(p2, 1.60, 0.045, (m2_dry, m2_0), ([h_s, v_s, m_after], [target_alt, 0.0, m_f]), (t_s, t_f))

    system.set_phase([p1, p2])
    system.set_objective(t_f + 0.04 * (p1.I[0] + p2.I[0]))
    system.set_system_constraint(
        [h_s, v_s, m_before, m_after, m_after - m_before, t_s, m_f, t_f - t_s, t_f],
        [0.10, 0.05, burnout1, m2_dry, -drop, 0.20, m2_dry, 0.40, 1.00],
        [1.80, 2.50, burnout1, m2_0, -drop, 3.00, m2_0, 5.00, 7.00],
    )

    sep = np.array([0.40, 0.80, burnout1])
    mass_after, final_mass, sep_t, fin_t = m2_0, 0.68, 1.10, 3.60
    g1 = ns.linear_guess(p1, 0.0)
    g1.t_f = sep_t
    tau1 = g1.t_x / sep_t
    g1.x[0] = sep[0] * tau1**2
    g1.x[1] = sep[1] * tau1
    g1.x[2] = m0 + (sep[2] - m0) * tau1
    g1.u[0] = 0.75
    g2 = ns.linear_guess(p2, 0.0)
    g2.t_0 = sep_t
    g2.t_f = fin_t
    tau2 = (g2.t_x - sep_t) / (fin_t - sep_t)
    g2.x[0] = sep[0] + (target_alt - sep[0]) * tau2
    g2.x[1] = sep[1] * (1 - tau2) + 0.55 * np.sin(np.pi * tau2)
    g2.x[2] = mass_after + (final_mass - mass_after) * tau2
    tau2u = (g2.t_u - sep_t) / (fin_t - sep_t)
    g2.u[0] = np.where(tau2u < 0.6, 0.9, 0.0)
    static = [sep[0], sep[1], sep[2], mass_after, sep_t, final_mass, fin_t]
    return system, [p1, p2], [g1, g2, static]


def three_stage_rocket(ns, mesh=96, num_point=1):
    """Stand-in for BASELINE.json's literal configs[3] ("multiphase_two_stage_rocket, 3 phases x 1000 intervals"): the
    reference's example has TWO phases (examples/multiphase_two_stage_rocket.py:90-113); this is the same vertical-ascent
    model with a THIRD burn stage and a second mass drop, linked through static parameters in the same way (FUNC boundary
    values and times of twelve static parameters).  Not a reference program -- a synthetic workload for the phase fan-out."""
    gravity, m0 = 1.0, 1.0
    prop = (0.06, 0.08, 0.06)
    drop = (0.20, 0.12)
    thrust, flow = (2.40, 1.60, 1.10), (0.080, 0.045, 0.030)
    target_alt = 2.6
    m_start = [m0, m0 - prop[0] - drop[0], m0 - prop[0] - drop[0] - prop[1] - drop[1]]
    m_dry = [m_start[k] - prop[k] for k in range(3)]

    system = ns.System(
        ["h_sep_1", "v_sep_1", "m_before_1", "m_after_1", "t_sep_1",
         "h_sep_2", "v_sep_2", "m_before_2", "m_after_2", "t_sep_2", "m_final", "t_final"]
    )
    h1, v1, mb1, ma1, t1, h2, v2, mb2, ma2, t2, m_f, t_f = system.s

    def stage(phase, k, boundaries, times):
        altitude, velocity, mass = phase.x
        (throttle,) = phase.u
        phase.set_dynamics([velocity, thrust[k] * throttle / mass - gravity, -flow[k] * throttle])
        phase.set_integral([throttle**2])
        phase.set_phase_constraint(
            [throttle, altitude, velocity, mass],
            [0.0, 0.0, 0.0, m_dry[k]],
            [1.0, target_alt, 3.0, m_start[k]],
            [True, False, False, False],
        )
        phase.set_boundary_condition(*boundaries, *times)
        phase.set_discretization(mesh, num_point)

    p1 = system.new_phase(["altitude_1", "velocity_1", "mass_1"], ["throttle_1"])
    stage(p1, 0, ([0.0, 0.0, m0], [h1, v1, mb1]), (0.0, t1))
    p2 = system.new_phase(["altitude_2", "velocity_2", "mass_2"], ["throttle_2"])
    stage(p2, 1, ([h1, v1, ma1], [h2, v2, mb2]), (t1, t2))
    p3 = system.new_phase(["altitude_3", "velocity_3", "mass_3"], ["throttle_3"])
    stage(p3, 2, ([h2, v2, ma2], [target_alt, 0.0, m_f]), (t2, t_f))

    system.set_phase([p1, p2, p3])
    system.set_objective(t_f + 0.04 * (p1.I[0] + p2.I[0] + p3.I[0]))
    system.set_system_constraint(
        [h1, v1, mb1, ma1 - mb1, t1, h2 - h1, v2, mb2, ma2 - mb2, t2 - t1, m_f, t_f - t2, t_f],
        [0.10, 0.05, m_dry[0], -drop[0], 0.20, 0.05, 0.05, m_dry[1], -drop[1], 0.20, m_dry[2], 0.30, 1.00],
        [1.80, 2.50, m_dry[0], -drop[0], 3.00, 1.80, 2.50, m_start[1], -drop[1], 4.00, m_start[2], 5.00, 9.00],
    )

    knots_t = [0.0, 1.10, 2.40, 4.20]
    knots_h = [0.0, 0.40, 1.30, target_alt]
    knots_v = [0.0, 0.80, 0.90, 0.0]
    guesses = []
    for k, p in enumerate((p1, p2, p3)):
        g = ns.linear_guess(p, 0.0)
        g.t_0, g.t_f = knots_t[k], knots_t[k + 1]
        span = knots_t[k + 1] - knots_t[k]
        tau = (g.t_x - knots_t[k]) / span
        g.x[0] = knots_h[k] + (knots_h[k + 1] - knots_h[k]) * tau
        g.x[1] = knots_v[k] * (1 - tau) + knots_v[k + 1] * tau + 0.3 * np.sin(np.pi * tau)
        g.x[2] = m_start[k] + (m_dry[k] + 0.4 * prop[k] - m_start[k]) * tau
        g.u[0] = np.where((g.t_u - knots_t[k]) / span < 0.6, 0.85, 0.1)
        guesses.append(g)
    static = [knots_h[1], knots_v[1], m_dry[0] + 0.4 * prop[0], m_start[1], knots_t[1],
              knots_h[2], knots_v[2], m_dry[1] + 0.4 * prop[1], m_start[2], knots_t[2], m_dry[2] + 0.4 * prop[2], knots_t[3]]
    return system, [p1, p2, p3], guesses + [static]


# --------------------------------------------------------------------------- humanoid WBC
_H = dict(
    TORSO=0.60, UPPER=0.38, FORE=0.30, HORIZON=2.5, KP=36.0, KD=12.0,
    W_LEFT_POS=120.0, W_LEFT_VEL=3.0, W_TORSO=20.0, W_QD=0.03, W_NULL=0.01,
    MAX_QD=3.0, MAX_NULL=10.0,
    Q_LO=(-0.55, -1.8, 0.35, -2.2, -2.2), Q_HI=(0.55, 1.2, 1.8, 2.2, 2.2),
    Q0=(0.25, -0.45, 0.95, 0.60, -1.10), Q_LEFT_TARGET=(0.0, -0.70, 1.10, -0.45, 1.00),
    RIGHT_DISP=(0.04, 0.08),
)


def _humanoid_hands_numeric(q):
    c = _H
    _, rs, re, ls, le = np.asarray(q, dtype=float)
    sh = np.array([0.0, c["TORSO"]])
    r_elbow = sh + c["UPPER"] * np.array([np.cos(rs), np.sin(rs)])
    r_hand = r_elbow + c["FORE"] * np.array([np.cos(rs + re), np.sin(rs + re)])
    l_elbow = sh + c["UPPER"] * np.array([-np.cos(ls), np.sin(ls)])
    l_hand = l_elbow + c["FORE"] * np.array([-np.cos(ls + le), np.sin(ls + le)])
    return r_hand, l_hand


def humanoid_wbc(ns, mesh=10, num_point=4):
    c = _H
    right0 = _humanoid_hands_numeric(c["Q0"])[0]
    left_target = _humanoid_hands_numeric(c["Q_LEFT_TARGET"])[1]

    system = ns.System(0)
    phase = system.new_phase(
        ["torso_angle", "right_shoulder_angle", "right_elbow_angle", "left_shoulder_angle",
         "left_elbow_angle", "torso_rate", "right_shoulder_rate", "right_elbow_rate",
         "left_shoulder_rate", "left_elbow_rate"],
        ["null_torso_acceleration", "null_right_shoulder_acceleration",
         "null_right_elbow_acceleration", "null_left_shoulder_acceleration",
         "null_left_elbow_acceleration"],
    )
    q = sp.Matrix(phase.x[:5])
    qd = sp.Matrix(phase.x[5:])
    z = sp.Matrix(phase.u)

    _, rs, re, ls, le = q
    sh = sp.Matrix([0.0, c["TORSO"]])
    right_hand = sh + sp.Matrix(
        [c["UPPER"] * sp.cos(rs) + c["FORE"] * sp.cos(rs + re),
         c["UPPER"] * sp.sin(rs) + c["FORE"] * sp.sin(rs + re)]
    )
    left_hand = sh + sp.Matrix(
        [-c["UPPER"] * sp.cos(ls) - c["FORE"] * sp.cos(ls + le),
         c["UPPER"] * sp.sin(ls) + c["FORE"] * sp.sin(ls + le)]
    )
    Jr = right_hand.jacobian(q)
    Jl = left_hand.jacobian(q)
    Jr_dot = sp.zeros(2, 5)
    for k in range(5):
        Jr_dot += Jr.diff(q[k]) * qd[k]
    arm = Jr[:, 1:3]
    det = arm[0, 0] * arm[1, 1] - arm[0, 1] * arm[1, 0]
    arm_inv = sp.Matrix([[arm[1, 1], -arm[0, 1]], [-arm[1, 0], arm[0, 0]]]) / det
    pinv = sp.zeros(5, 2)
    pinv[1:3, :] = arm_inv
    null_proj = sp.diag(1.0, 0.0, 0.0, 1.0, 1.0)

    tau = phase.t / c["HORIZON"]
    prog = 10.0 * tau**3 - 15.0 * tau**4 + 6.0 * tau**5
    prog_r = (30.0 * tau**2 - 60.0 * tau**3 + 30.0 * tau**4) / c["HORIZON"]
    prog_a = (60.0 * tau - 180.0 * tau**2 + 120.0 * tau**3) / c["HORIZON"] ** 2
    disp = sp.Matrix(c["RIGHT_DISP"])
    des_p = sp.Matrix(right0) + disp * prog
    des_v = disp * prog_r
    des_a = disp * prog_a
    right_vel = Jr * qd
    a_ref = des_a + c["KP"] * (des_p - right_hand) + c["KD"] * (des_v - right_vel)
    primary = pinv * (a_ref - Jr_dot * qd)
    qdd = primary + null_proj * z
    phase.set_dynamics([*qd, *qdd])

    left_err = left_hand - sp.Matrix(left_target)
    left_vel = Jl * qd
    cost = (
        c["W_LEFT_POS"] * left_err.dot(left_err)
        + c["W_LEFT_VEL"] * left_vel.dot(left_vel)
        + c["W_TORSO"] * q[0] ** 2
        + c["W_QD"] * qd.dot(qd)
        + c["W_NULL"] * z.dot(z)
    )
    phase.set_integral([cost])
    phase.set_phase_constraint(
        [*q, *qd, *z],
        [*c["Q_LO"], *([-c["MAX_QD"]] * 5), *([-c["MAX_NULL"]] * 5)],
        [*c["Q_HI"], *([c["MAX_QD"]] * 5), *([c["MAX_NULL"]] * 5)],
    )
    phase.set_boundary_condition([*c["Q0"], *np.zeros(5)], [None] * 10, 0.0, c["HORIZON"])
    phase.set_discretization(mesh, num_point)
    system.set_phase([phase])
    system.set_objective(phase.I[0])

    guess = ns.linear_guess(phase, 0.0)
    final = np.array([0.03, -0.36, 0.92, -0.45, 1.00])
    d = final - np.array(c["Q0"])
    tx = guess.t_x / c["HORIZON"]
    px = 10.0 * tx**3 - 15.0 * tx**4 + 6.0 * tx**5
    rx = (30.0 * tx**2 - 60.0 * tx**3 + 30.0 * tx**4) / c["HORIZON"]
    for j in range(5):
        guess.x[j] = c["Q0"][j] + d[j] * px
        guess.x[5 + j] = d[j] * rx
    tu = guess.t_u / c["HORIZON"]
    au = (60.0 * tu - 180.0 * tu**2 + 120.0 * tu**3) / c["HORIZON"] ** 2
    for j in range(5):
        guess.u[j] = d[j] * au
    return system, [phase], [guess]


def humanoid_team(ns, mesh=10, num_point=4, copies=4, coupling=6.0):
    """Stand-in for BASELINE.json's literal configs[4] ("humanoid_whole_body_control, ~40-state"): the reference's example
    has 10 states + 5 controls (examples/humanoid_whole_body_control.py:160-180).  This is ``copies`` of that whole-body
    model in ONE phase -- 10 x copies states, 5 x copies controls -- whose torsos are coupled by springs (torso k is pulled
    towards torso k + 1), so the derivative set does not fall apart into independent blocks.  Not a reference program: a
    synthetic wide model for the width scaling of the kernels (its derivative set is evaluated in groups)."""
    c = _H
    right0 = _humanoid_hands_numeric(c["Q0"])[0]
    left_target = _humanoid_hands_numeric(c["Q_LEFT_TARGET"])[1]
    joints = ("torso", "right_shoulder", "right_elbow", "left_shoulder", "left_elbow")
    system = ns.System(0)
    phase = system.new_phase(
        [f"{j}_angle_{k}" for k in range(copies) for j in joints] + [f"{j}_rate_{k}" for k in range(copies) for j in joints],
        [f"null_{j}_acceleration_{k}" for k in range(copies) for j in joints])
    nq = 5 * copies
    Q = [sp.Matrix(phase.x[5 * k:5 * k + 5]) for k in range(copies)]
    QD = [sp.Matrix(phase.x[nq + 5 * k:nq + 5 * k + 5]) for k in range(copies)]
    Z = [sp.Matrix(phase.u[5 * k:5 * k + 5]) for k in range(copies)]
    tau = phase.t / c["HORIZON"]
    prog = 10.0 * tau**3 - 15.0 * tau**4 + 6.0 * tau**5
    prog_r = (30.0 * tau**2 - 60.0 * tau**3 + 30.0 * tau**4) / c["HORIZON"]
    prog_a = (60.0 * tau - 180.0 * tau**2 + 120.0 * tau**3) / c["HORIZON"] ** 2
    disp = sp.Matrix(c["RIGHT_DISP"])
    accel, cost = [], 0
    for k in range(copies):
        q, qd, z = Q[k], QD[k], Z[k]
        _, rs, re, ls, le = q
        sh = sp.Matrix([0.0, c["TORSO"]])
        right_hand = sh + sp.Matrix([c["UPPER"] * sp.cos(rs) + c["FORE"] * sp.cos(rs + re),
                                     c["UPPER"] * sp.sin(rs) + c["FORE"] * sp.sin(rs + re)])
        left_hand = sh + sp.Matrix([-c["UPPER"] * sp.cos(ls) - c["FORE"] * sp.cos(ls + le),
                                    c["UPPER"] * sp.sin(ls) + c["FORE"] * sp.sin(ls + le)])
        Jr = right_hand.jacobian(q)
        Jl = left_hand.jacobian(q)
        Jr_dot = sp.zeros(2, 5)
        for i in range(5):
            Jr_dot += Jr.diff(q[i]) * qd[i]
        arm = Jr[:, 1:3]
        det = arm[0, 0] * arm[1, 1] - arm[0, 1] * arm[1, 0]
        arm_inv = sp.Matrix([[arm[1, 1], -arm[0, 1]], [-arm[1, 0], arm[0, 0]]]) / det
        pinv = sp.zeros(5, 2)
        pinv[1:3, :] = arm_inv
        null_proj = sp.diag(1.0, 0.0, 0.0, 1.0, 1.0)
        scale = 1.0 + 0.1 * k                                    # (the copies follow slightly different reference motions)
        des_p = sp.Matrix(right0) + disp * prog * scale
        des_v = disp * prog_r * scale
        des_a = disp * prog_a * scale
        a_ref = des_a + c["KP"] * (des_p - right_hand) + c["KD"] * (des_v - Jr * qd)
        qdd = pinv * (a_ref - Jr_dot * qd) + null_proj * z
        qdd[0] += -coupling * (q[0] - Q[(k + 1) % copies][0])     # torso spring to the next copy
        accel.append(qdd)
        left_err = left_hand - sp.Matrix(left_target)
        left_vel = Jl * qd
        cost += (c["W_LEFT_POS"] * left_err.dot(left_err) + c["W_LEFT_VEL"] * left_vel.dot(left_vel)
                 + c["W_TORSO"] * q[0] ** 2 + c["W_QD"] * qd.dot(qd) + c["W_NULL"] * z.dot(z))
    phase.set_dynamics([v for qd in QD for v in qd] + [a for qdd in accel for a in qdd])
    phase.set_integral([cost])
    phase.set_phase_constraint(
        [*phase.x, *phase.u],
        [*(list(c["Q_LO"]) * copies), *([-c["MAX_QD"]] * nq), *([-c["MAX_NULL"]] * nq)],
        [*(list(c["Q_HI"]) * copies), *([c["MAX_QD"]] * nq), *([c["MAX_NULL"]] * nq)],
    )
    phase.set_boundary_condition([*(list(c["Q0"]) * copies), *np.zeros(nq)], [None] * (2 * nq), 0.0, c["HORIZON"])
    phase.set_discretization(mesh, num_point)
    system.set_phase([phase])
    system.set_objective(phase.I[0])

    guess = ns.linear_guess(phase, 0.0)
    final = np.array([0.03, -0.36, 0.92, -0.45, 1.00])
    d = final - np.array(c["Q0"])
    tx = guess.t_x / c["HORIZON"]
    px = 10.0 * tx**3 - 15.0 * tx**4 + 6.0 * tx**5
    rx = (30.0 * tx**2 - 60.0 * tx**3 + 30.0 * tx**4) / c["HORIZON"]
    tu = guess.t_u / c["HORIZON"]
    au = (60.0 * tu - 180.0 * tu**2 + 120.0 * tu**3) / c["HORIZON"] ** 2
    for k in range(copies):
        for j in range(5):
            guess.x[5 * k + j] = c["Q0"][j] + d[j] * px
            guess.x[nq + 5 * k + j] = d[j] * rx
            guess.u[5 * k + j] = d[j] * au
    return system, [phase], [guess]


def state_chain(ns, states=52, mesh=40, num_point=4, window=0):
    """Synthetic WIDE model for the width scaling of the kernels (the reference loops per state with no limit,
    phasebase.py:1083-1124, 1234-1285): ``states`` states x_0 ... x_(n-1) and one control u in one phase,
    x_0' = -x_0 + u,  x_i' = -x_i + x_(i-1) x_((i+1) mod n), integral u^2 + x_0^2, fixed start, free end, fixed times.
    ``window`` = W > 0: every dynamics function also carries 0.01 sin of the mean of the W states from its own on (cyclic) --
    W more Jacobian entries and W (W + 1) / 2 Hessian pairs per state, every pair shared by up to W dynamics functions (the
    compact Hessian then sums several contracted multipliers per entry).  Not a reference program."""
    n = int(states)
    system = ns.System(0)
    phase = system.new_phase([f"x_{i}" for i in range(n)], ["u"])
    x, (u,) = list(phase.x), phase.u
    dyn = [-x[0] + u] + [-x[i] + x[i - 1] * x[(i + 1) % n] for i in range(1, n)]
    if window:
        dyn = [d + 0.01 * sp.sin(sum(x[(i + j) % n] for j in range(int(window))) / int(window)) for i, d in enumerate(dyn)]
    phase.set_dynamics(dyn)
    phase.set_integral([u**2 + x[0] ** 2])
    phase.set_phase_constraint([u], [-3.0], [3.0])
    phase.set_boundary_condition([0.5 + 0.4 * np.cos(0.7 * i) for i in range(n)], [None] * n, 0.0, 2.0)
    phase.set_discretization(mesh, num_point)
    system.set_phase([phase])
    system.set_objective(phase.I[0])
    guess = ns.linear_guess(phase, 0.0)
    for i in range(n):
        guess.x[i] = (0.5 + 0.4 * np.cos(0.7 * i)) * np.exp(-0.4 * guess.t_x) + 0.05 * np.sin(3.0 * guess.t_x + i)
    guess.u[0] = 0.3 * np.cos(2.0 * guess.t_u)
    return system, [phase], [guess]


def wide_mix(ns, shapes=((60, 3, 10, 4),), statics=6, mesh=40, num_point=4, free_time=True):
    """Synthetic models that are wide in EVERY direction the modeling API has (the reference loops over states, controls,
    path constraints, integrals and static parameters with no limit: phasebase.py:1083-1124, 1234-1285, systembase.py:
    148-187): one phase per entry of ``shapes`` = (states, controls, path constraints, integrals); ``statics`` static
    parameters enter every dynamics function, the first ones link the phases (terminal time of phase k = initial time of
    phase k + 1 = a static parameter, FUNC boundaries; the first state of phase k + 1 starts where a static parameter says,
    which a system constraint ties to nothing else) and, with ``free_time``, the last terminal time.  Not a reference
    program."""
    n_ph = len(shapes)
    n_s = max(int(statics), n_ph + 1)
    system = ns.System(n_s)
    s = list(system.s)
    phases, guesses, obj = [], [], 0
    for k, (nx, nu, nc, ni) in enumerate(shapes):
        phase = system.new_phase([f"x{k}_{i}" for i in range(nx)], [f"u{k}_{i}" for i in range(nu)])
        x, u = list(phase.x), list(phase.u)
        dyn = [-x[i] + x[i - 1] * u[i % nu] + 0.1 * s[(i + k) % n_s] * x[(i + 1) % nx] for i in range(nx)]
        phase.set_dynamics(dyn)
        phase.set_integral([x[j % nx] * u[j % nu] + u[j % nu] ** 2 + 0.05 * sp.cos(x[(3 * j + 1) % nx]) for j in range(ni)])
        if nc:
            phase.set_phase_constraint([x[j % nx] ** 2 + u[(j + 1) % nu] ** 2 - 0.2 * x[(j + 2) % nx] for j in range(nc)],
                                       [-1.0] * nc, [9.0] * nc)
        t0 = 0.0 if k == 0 else s[k - 1] + 1.0 * k
        tf = (s[k] + 1.0 * (k + 1)) if (k + 1 < n_ph or free_time) else 1.0 * (k + 1)
        start = [0.3 + 0.02 * i for i in range(nx)]
        if k:
            start[0] = s[n_ph]
        phase.set_boundary_condition(start, [None] * nx, t0, tf)
        phase.set_discretization(mesh, num_point)
        phases.append(phase)
        obj = obj + sum(phase.I[j] for j in range(ni))
        g = ns.linear_guess(phase, 0.0)
        g.t_0, g.t_f = 1.0 * k + (0.1 if k else 0.0), 1.0 * (k + 1) + 0.1
        for i in range(nx):
            g.x[i] = (0.3 + 0.02 * i) * np.exp(-0.3 * (g.t_x - g.t_0)) + 0.04 * np.sin(2.0 * g.t_x + i)
        for i in range(nu):
            g.u[i] = 0.3 * np.cos(1.5 * g.t_u + i)
        guesses.append(g)
    system.set_phase(phases)
    system.set_objective(obj + sum(0.5 * si**2 for si in s) + (s[n_ph - 1] if free_time else 0))
    system.set_system_constraint([s[i] - s[i + 1] for i in range(n_ph)] + [s[n_ph] * s[0]], [-5.0] * (n_ph + 1), [5.0] * (n_ph + 1))
    static = [0.1] * n_s
    static[n_ph] = 0.3
    return system, phases, guesses + [static]


# --------------------------------------------------------------------------- semantic pins
def derivative_model(ns, mesh=(0, 0.2, 1), num_point=(3, 4)):
    """Feature-dense model of the reference's FD derivative tests: 2 static params, FUNC state
    boundary, FUNC t_f, free t_0, 2 integrals, 2 path constraints (one a bare symbol -> bound),
    nonlinear objective in the integrals, 2 system constraints."""
    s = ns.System(2)
    p = s.new_phase(1, 1)
    p.set_dynamics([p.x[0] * sp.cos(s.s[0]) / p.u[0] + p.t**2])
    p.set_boundary_condition([0], [sp.cos(s.s[0] * 0.1)], None, 3 * sp.sin(s.s[1]))
    p.set_integral(
        [
            sp.cos(p.x[0]) * p.u[0] + 2 * p.x[0] * sp.cos(s.s[0]) + 3 * sp.cos(p.x[0]) * p.t
            + 4 * p.u[0] * sp.cos(s.s[0]) + 5 * sp.cos(p.u[0]) * p.t + 6 * s.s[1] * sp.cos(p.t),
            6 * sp.cos(p.x[0]) * p.u[0] + 5 * p.x[0] * sp.cos(s.s[0]) + 4 * sp.cos(p.x[0]) * p.t
            + 3 * p.u[0] * sp.cos(s.s[0]) + 2 * sp.cos(p.u[0]) * p.t + s.s[1] * sp.cos(p.t),
        ]
    )
    p.set_phase_constraint([p.t - p.x[0] * p.u[0] * s.s[0] * s.s[1], p.x[0]], [0, 0], [0, 1])
    p.set_discretization(list(mesh), list(num_point))
    s.set_phase([p])
    s.set_objective((p.I[0] + p.I[1] + s.s[0]) ** 2)
    s.set_system_constraint([(s.s[0] + 1) ** 2, s.s[1] / 2 * p.I[0]], [0, 0], [0, 0])
    guess = [ns.constant_guess(p, 1.3), [0.7, 0.4]]
    return s, [p], guess


def worked_model(ns, mesh=2, num_point=3):
    """states a,b; control u; mixed FIXED/FREE boundaries; free t_f."""
    s = ns.System(0)
    p = s.new_phase(["a", "b"], ["u"])
    a, b = p.x
    (u,) = p.u
    p.set_dynamics([b * u, sp.sin(a) + u**2])
    p.set_integral([a**2 + u**2])
    p.set_phase_constraint([a * b], [-1], [1])
    p.set_boundary_condition([1.0, None], [None, 0.0], 0.0, None)
    p.set_discretization(mesh, num_point)
    s.set_phase([p])
    s.set_objective(p.I[0])
    guess = [ns.constant_guess(p, 0.8)]
    return s, [p], guess


def func_times_model(ns, mesh=(0, 0.3, 0.55, 1), num_point=(2, 3, 2)):
    """Two linked phases whose boundary states and times are nonlinear functions of the static
    parameters (exercises FUNC Hessian blocks tiled over T_f/T_b and (t,s) couplings), with
    system constraints mixing integrals of both phases."""
    s = ns.System(["sa", "sb", "sc"])
    sa, sb, sc = s.s
    p1 = s.new_phase(["y", "v"], ["w"])
    y, v = p1.x
    (w,) = p1.u
    p1.set_dynamics([v * sp.cos(sa) + p1.t * w, -y * w + sb**2])
    p1.set_integral([w**2 + y * sc, sp.sin(v) * p1.t])
    p1.set_phase_constraint([y * w - p1.t, w], [-2, -3], [2, 3])
    p1.set_boundary_condition([sa**2, 0.5], [None, sp.sin(sb) * sc], sc * 0.1, sa * sb + 2)
    p1.set_discretization(list(mesh), list(num_point))
    p2 = s.new_phase(1, 1)
    p2.set_dynamics([p2.x[0] * p2.u[0] + sc])
    p2.set_integral([p2.x[0] ** 2 * p2.u[0] ** 2])
    p2.set_boundary_condition([sp.sin(sb) * sc], [None], sa * sb + 2, None)
    p2.set_phase_constraint([p2.t], [0], [9])
    p2.set_discretization(2, 3)
    s.set_phase([p1, p2])
    s.set_objective(p1.I[0] * p2.I[0] + sp.exp(p1.I[1]) + sa * sc)
    s.set_system_constraint([p1.I[0] + p2.I[0] ** 2 - sa, sb * sc, sc], [0, -1, 0.1], [5, 1, 3])
    guess = [ns.constant_guess(p1, 0.6), ns.constant_guess(p2, 0.9), [0.3, 0.5, 0.8]]
    return s, [p1, p2], guess


def elementary_functions_model(ns, mesh=(0, 0.3, 1.0), num_point=(4, 5)):
    """Every elementary function the reference's NumPy / Numba printing handles smoothly (fastfunc.py:41-43: lambdarepr with
    the math. prefix stripped): tan, asin, acos, atan, atan2, the hyperbolic functions and their inverses, exp, log, sqrt,
    rational and negative powers -- in dynamics, integrands, path and system constraints, with static parameters, a FUNC
    boundary value and a free final time."""
    system = ns.System(["p", "q"])
    p_, q_ = system.s
    phase = system.new_phase(["a", "b", "c"], ["u", "v"])
    a, b, c = phase.x
    u, v = phase.u
    t = phase.t
    phase.set_dynamics([
        sp.tan(a / 4) + sp.asin(u / 3) * sp.exp(-b) + sp.sinh(v / 2) * p_,
        sp.acos(v / 3) * sp.log(2 + a**2) + sp.atan(b * u) + sp.cosh(c / 3),
        sp.tanh(a * b) + sp.sqrt(1 + c**2) * sp.atan2(u + 2, v + 3) + (2 + sp.cos(t))**sp.Rational(5, 2) + q_ * (1 + u**2)**(-2),
    ])
    phase.set_integral([sp.sqrt((a - 0.37)**2 + 0.01) * u**2 + sp.exp(sp.sin(b)) * v**2, sp.asinh(a * b) * c + (1 + v**2)**sp.Rational(1, 3)])
    phase.set_phase_constraint([sp.log(1 + u**2) + sp.acosh(2 + c**2) + sp.atanh(a / 3), u, v], [-1.0, -1.0, -1.0], [3.0, 1.0, 1.0])
    phase.set_boundary_condition([0.1, 0.2, None], [None, 0.5 * p_, None], 0.0, None)
    phase.set_discretization(mesh, num_point)
    system.set_phase([phase])
    system.set_objective(phase.I[0] + phase.I[1] * p_ + q_**2)
    system.set_system_constraint([phase.I[1] + p_ * q_], [-10.0], [10.0])
    guess = ns.linear_guess(phase, 0.3)
    guess.t_f = 1.5
    guess.x[0] = 0.1 + 0.3 * guess.t_x
    guess.x[1] = 0.2 + 0.2 * np.sin(guess.t_x)
    guess.x[2] = 0.4 * np.cos(guess.t_x)
    guess.u[0] = 0.3 * np.sin(2 * guess.t_u)
    guess.u[1] = -0.2 + 0.1 * guess.t_u
    return system, [phase], [guess, np.array([0.7, -0.4])]


def bang_bang_model(ns, mesh=6, num_point=5, second=False):
    """Bang-bang test problem in the style of the reference's check tests (tests/test_radau/test_check_radau.py:
    9-16): one state driven by the controls, a bang-bang path constraint u0 + s0 in [0, 2] and, with ``second``,
    a bang-bang bound -1 <= u1 <= 1 given as a bare symbol (it becomes a variable bound, not a path row)."""
    s = ns.System(1)
    p = s.new_phase(1, 2 if second else 1)
    p.set_dynamics([p.u[0] + (0.5 * p.u[1] if second else 0)])
    p.set_boundary_condition([0.0], [None], 0.0, 1.0)
    cons, lo, hi = [p.u[0] + p.s[0]], [0.0], [2.0]
    if second:
        cons, lo, hi = cons + [p.u[1]], lo + [-1.0], hi + [1.0]
    p.set_phase_constraint(cons, lo, hi, True)
    p.set_discretization(mesh, num_point)
    s.set_phase([p])
    s.set_objective(s.s[0] ** 2)
    return s, [p], [ns.constant_guess(p, 0.0), [0.0]]


def bang_bang_controls(t, profile):
    """Control histories u0(t) (and u1(t)) on node times t in [0, 1]: smooth-but-steep switching functions, so that
    the bang-bang check fails on the intervals that contain a switch."""
    t = np.asarray(t, dtype=np.float64)
    step = lambda c, w: 0.5 * (1.0 + np.tanh((t - c) / w))  # noqa: E731
    if profile == "one_switch":
        return [2.0 * step(0.37, 0.004)]
    if profile == "two_switches":
        return [2.0 * (step(0.22, 0.003) - step(0.71, 0.006))]
    if profile == "near_mesh_point":
        return [2.0 * (1.0 - step(0.5004, 0.002))]
    if profile == "ramp":
        return [2.0 * np.clip((t - 0.3) / 0.25, 0.0, 1.0)]
    if profile == "two_controls":
        return [2.0 * step(0.41, 0.005), 2.0 * step(0.63, 0.004) - 1.0]
    raise ValueError(profile)


# bang-bang refinement fixtures: (model kwargs, control profile)
BANG_BANG_CASES = {
    "bb_one_switch_6x5": (dict(mesh=6, num_point=5), "one_switch"),
    "bb_one_switch_hp": (dict(mesh=[0, 0.1, 0.3, 0.45, 0.8, 1.0], num_point=[3, 6, 4, 7, 5]), "one_switch"),
    "bb_two_switches_10x4": (dict(mesh=10, num_point=4), "two_switches"),
    "bb_near_mesh_point_8x6": (dict(mesh=8, num_point=6), "near_mesh_point"),
    "bb_ramp_5x6": (dict(mesh=5, num_point=6), "ramp"),
    "bb_two_controls_9x5": (dict(mesh=9, num_point=5, second=True), "two_controls"),
    "bb_two_controls_3x8": (dict(mesh=3, num_point=8, second=True), "two_controls"),
}


SMALL_CASES = {
    # name: (builder, scheme, kwargs)
    "derivative_lgr": (derivative_model, "radau", {}),
    "derivative_lgl": (derivative_model, "lobatto", {}),
    "worked_lgr": (worked_model, "radau", {}),
    "worked_lgl": (worked_model, "lobatto", {}),
    "functimes_lgr": (func_times_model, "radau", {}),
    "functimes_lgl": (func_times_model, "lobatto", dict(mesh=(0, 0.3, 0.55, 1), num_point=(2, 3, 4))),
    "lqr_lgl_10x10": (lqr, "lobatto", dict(mesh=10, num_point=10)),
    "lqr_lgr_4x3": (lqr, "radau", dict(mesh=4, num_point=3)),
    "brach_lgr_3x4": (brachistochrone, "radau", dict(mesh=3, num_point=4)),
    "brach_lgr_1x5": (brachistochrone, "radau", dict(mesh=1, num_point=5)),
    "brach_lgr_ragged": (brachistochrone, "radau", dict(mesh=[0, 0.2, 0.5, 1.0], num_point=[3, 1, 6])),
    "brach_lgl_3x4": (brachistochrone, "lobatto", dict(mesh=3, num_point=4)),
    "brach_lgl_1x3": (brachistochrone, "lobatto", dict(mesh=1, num_point=3)),
    "quad_lgr_3x4": (planar_quadrotor, "radau", dict(mesh=3, num_point=4)),
    "quad_lgl_4x5": (planar_quadrotor, "lobatto", dict(mesh=4, num_point=5)),
    "rocket_lgr_5x1": (two_stage_rocket, "radau", dict(mesh=5, num_point=1)),
    "rocket_lgr_3x4": (two_stage_rocket, "radau", dict(mesh=3, num_point=4)),
    "rocket_lgl_3x3": (two_stage_rocket, "lobatto", dict(mesh=3, num_point=3)),
    "humanoid_lgr_2x3": (humanoid_wbc, "radau", dict(mesh=2, num_point=3)),
    "humanoid_lgl_2x4": (humanoid_wbc, "lobatto", dict(mesh=2, num_point=4)),
}

# callback vectors of the reference beyond 12 points per interval (tests/golden/small_hi): its np.roots-based tables lose
# digits there, the comparison's tolerance follows from the measured table error (tests/test_product_host.py)
HIGH_ORDER_CASES = {
    "brach_lgr_2x13": (brachistochrone, "radau", dict(mesh=2, num_point=13)),
    "brach_lgr_3x16": (brachistochrone, "radau", dict(mesh=3, num_point=16)),
    "quad_lgl_2x16": (planar_quadrotor, "lobatto", dict(mesh=2, num_point=16)),
    "rocket_lgr_hp20": (two_stage_rocket, "radau", dict(mesh=[0, 0.4, 1.0], num_point=[20, 14])),
    "brach_lgl_2x20": (brachistochrone, "lobatto", dict(mesh=2, num_point=20)),
}

# mesh error estimation / continuous refinement fixtures (K >= 2 everywhere: the reference's refinement
# formula divides by log(K), phasebase.py:1579-1587)
ERROR_CASES = {k: SMALL_CASES[k] for k in (
    "brach_lgr_3x4", "brach_lgl_3x4", "quad_lgr_3x4", "quad_lgl_4x5", "rocket_lgr_3x4", "rocket_lgl_3x3",
    "humanoid_lgr_2x3", "humanoid_lgl_2x4", "worked_lgr", "lqr_lgl_10x10")}
ERROR_CASES["brach_lgr_hp"] = (brachistochrone, "radau", dict(mesh=[0, 0.2, 0.5, 1.0], num_point=[3, 2, 6]))
ERROR_CASES["rocket_lgl_hp"] = (two_stage_rocket, "lobatto", dict(mesh=[0, 0.3, 0.4, 1.0], num_point=[4, 2, 5]))

# BASELINE.json configs (model re-meshed on LGR) + the exact-10k supplemental.
FULL_CASES = {
    "C2_brach_lgr_200x8": (brachistochrone, "radau", dict(mesh=200, num_point=8)),
    "S_brach_lgr_1250x8": (brachistochrone, "radau", dict(mesh=1250, num_point=8)),
    "C3_quad_lgr_2000x6": (planar_quadrotor, "radau", dict(mesh=2000, num_point=6)),
    "C4_rocket_lgr_2x1000x4": (two_stage_rocket, "radau", dict(mesh=1000, num_point=4)),
    "C5_humanoid_lgr_5000x8": (humanoid_wbc, "radau", dict(mesh=5000, num_point=8)),
}


def phase_relay(ns, phases=10, mesh=3, num_point=3):
    """Synthetic workload for the phase fan-out (the reference puts no limit on the number of phases, systembase.py:148-187):
    ``phases`` short phases of one state and one control handed over through static parameters -- phase k starts at the
    value s_(k-1) its predecessor ends with (FUNC boundary values), fixed times [k, k + 1], damping and an integrand that
    differ per phase; objective: the sum of the integrals plus a pull on the hand-over values.  Not a reference program."""
    system = ns.System(max(phases - 1, 0))
    S = list(system.s)
    out, guesses = [], []
    for k in range(phases):
        p = system.new_phase(1, 1)
        (x,), (u,) = p.x, p.u
        p.set_dynamics([u - (0.1 + 0.05 * k) * x + 0.02 * sp.sin(p.t)])
        p.set_integral([u**2 + (0.3 + 0.01 * k) * (x - 1.0) ** 2])
        p.set_phase_constraint([u], [-4.0], [4.0])
        x0 = 1.5 if k == 0 else S[k - 1]
        xf = S[k] if k < phases - 1 else 1.0
        p.set_boundary_condition([x0], [xf], float(k), float(k + 1))
        p.set_discretization(mesh, num_point)
        out.append(p)
    system.set_phase(out)
    system.set_objective(sum(p.I[0] for p in out) + sum(0.05 * (s - 1.2) ** 2 for s in S))
    for k, p in enumerate(out):
        g = ns.linear_guess(p, 0.0)
        g.x[0] = 1.5 - 0.5 * (k + (g.t_x - k)) / phases
        g.u[0] = 0.1
        guesses.append(g)
    if S:
        guesses.append([1.5 - 0.5 * (k + 1) / phases for k in range(len(S))])
    return system, out, guesses


# (reference-generated fixtures of the synthetic workloads as well: tests/golden/small)
SMALL_CASES["relay_lgr_10"] = (phase_relay, "radau", dict(phases=10, mesh=3, num_point=3))
SMALL_CASES["relay_lgl_12"] = (phase_relay, "lobatto", dict(phases=12, mesh=[0, 0.3, 1.0], num_point=[3, 4]))
SMALL_CASES["rocket3_lgr_3x4"] = (three_stage_rocket, "radau", dict(mesh=3, num_point=4))
SMALL_CASES["team_lgl_2x3"] = (humanoid_team, "lobatto", dict(mesh=2, num_point=3))
# (cases the CPU suite's NumPy plan interpreter needs minutes for; the oracle test and the GPU tests run them)
SLOW_ON_CPU = {"team_lgl_2x3"}
