"""Effector front-end on the GPU: user-written effectors traced, compiled into the fused step kernel and checked
against (a) the C oracle where a built-in equivalent exists, (b) the reference's ball golden CSV, (c) the numpy
stepper evaluating the same DAG for effectors that have no built-in twin (Apollo's)."""
import numpy as np
import pytest

import elodin_amd as el
from elodin_amd import _lib as L
from elodin_amd import dsl, workloads
from oracle import oracle as orc
from tests import dsl_numpy, golden_util as gu, np_sixdof, parity

pytestmark = pytest.mark.gpu
np_ = dsl.np


@dsl.effector
def gravity(force, inertia):                                     # examples/ball/sim.py:57-59
    return force + dsl.SpatialForce(linear=np_.array([0.0, 0.0, -9.81]) * inertia.mass())


@dsl.effector(body_torque=3)
def rcs(force, pos, body_torque):                                # apollo-lander/sim.py:396-398
    return force + dsl.SpatialForce(torque=pos.angular() @ body_torque)


@dsl.effector(wind=3)
def apply_drag(wind, vel, force):                                # examples/ball/sim.py:96-116, verbatim structure
    fluid_movement_vector = wind - vel.linear()
    fluid_velocity = np_.linalg.norm(fluid_movement_vector)
    drag_force = 0.5 * (0.5 * 1.225 * fluid_velocity ** 2 * (2 * 3.1415 * 0.2 ** 2))
    return dsl.SpatialForce(linear=force.force() + drag_force * (fluid_movement_vector / fluid_velocity))


@pytest.mark.parametrize("integrator", [L.RK4, L.SEMI_IMPLICIT])
@pytest.mark.parametrize("k", [1, 16])
def test_user_written_config2_effectors_match_c_oracle(integrator, k):
    n = 5000
    w = workloads.independent_bodies(n)
    hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                     integrator=integrator, effectors=gravity | rcs, columns={"body_torque": w["body_torque"]},
                     ticks_per_launch=k)
    ref = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                          integrator=integrator, ops=[(orc.EFF_UNIFORM_GRAVITY, (0, 0, -9.81), None),
                                                      (orc.EFF_BODY_TORQUE, (), w["body_torque"])])
    hip.run(200)
    ref.step(200, threads=8)
    errs = parity.state_errors(hip, ref)
    assert max(errs.values()) < parity.F64_RTOL, errs


def test_ball_example_with_user_written_effectors_matches_reference_golden():
    g = gu.load("ball")
    w = el.World()
    w.spawn([el.Body(world_pos=el.SpatialTransform(linear=[0.0, 0.0, 6.0])), el.C("wind", g["ball.wind"][1])], name="ball")
    exec = w.build(el.six_dof(sys=gravity | apply_drag), simulation_rate=120.0)
    for r in range(1, 101):
        exec.run(1)
        assert parity.pos_rel_err(exec.column_array("world_pos"), g["ball.world_pos"][r][None]) < parity.F64_RTOL
        assert parity.field_rel_err(exec.column_array("world_vel")[:, 3:], g["ball.world_vel"][r][None, 3:]) < parity.F64_RTOL
        assert parity.field_rel_err(exec.column_array("force")[:, 3:], g["ball.force"][r][None, 3:]) < parity.F64_RTOL


def test_apollo_effectors_written_by_the_user_vs_numpy_stepper():
    """lunar_gravity | apply_main_thrust | apply_rcs_torque exactly as in apollo-lander/sim.py:380-398."""
    LUNAR_G, R_MOON = 1.622, 1_737_400.0

    @dsl.effector
    def lunar_gravity(force, inertia, vel):
        v_h_sq = np_.sum(vel.linear()[:2] ** 2)
        g_eff = np_.maximum(LUNAR_G - v_h_sq / R_MOON, 0.0)
        return force + dsl.SpatialForce(linear=np_.array([0.0, 0.0, -1.0]) * g_eff * inertia.mass())

    @dsl.effector(thrust=1)
    def apply_main_thrust(thrust, force, pos):
        return force + dsl.SpatialForce(linear=pos.angular() @ np_.array([0.0, 0.0, thrust[0]]))

    @dsl.effector(rcs_torque=3)
    def apply_rcs_torque(rcs_torque, force, pos):
        return force + dsl.SpatialForce(torque=pos.angular() @ rcs_torque)

    pipe = lunar_gravity | apply_main_thrust | apply_rcs_torque
    n = 3000
    rng = np.random.default_rng(3)
    w = workloads.independent_bodies(n)
    vel = w["world_vel"].copy()
    vel[:, 3] = rng.uniform(0, 2500.0, n)          # horizontal speeds on both sides of the relief clamp (1679 m/s)
    thrust = rng.uniform(4670.0, 45040.0, (n, 1))
    torque = rng.uniform(-3560.0, 3560.0, (n, 3))
    cols = {"thrust": thrust, "rcs_torque": torque}
    hip = el.HipExec(w["world_pos"], vel, w["inertia"], simulation_time_step=workloads.DT_120HZ,
                     integrator=L.SEMI_IMPLICIT, effectors=pipe, columns=cols)
    tp = pipe.trace()
    assert tp.reads_velocity and tp.columns == [("thrust", 1), ("rcs_torque", 3)]
    pos_r, vel_r, acc_r = w["world_pos"].copy(), vel.copy(), np.zeros((n, 6))
    eff = lambda xs, vs: dsl_numpy.evaluate(tp, xs, vs, w["inertia"], cols)
    for _ in range(50):
        pos_r, vel_r, acc_r, F_r = np_sixdof.tick(pos_r, vel_r, acc_r, w["inertia"], eff, workloads.DT_120HZ, integrator=1)
    hip.run(50)
    assert parity.pos_rel_err(hip.world_pos, pos_r) < parity.F64_RTOL
    for got, ref in ((hip.world_vel, vel_r), (hip.world_accel, acc_r), (hip.force, F_r)):
        assert max(parity.field_rel_err(got[:, :3], ref[:, :3]), parity.field_rel_err(got[:, 3:], ref[:, 3:])) < parity.F64_RTOL
    assert (np.maximum(LUNAR_G - vel[:, 3] ** 2 / R_MOON, 0.0) == 0.0).any()     # the clamp branch is exercised


def test_select_clip_and_transcendentals_f32_and_f64():
    @dsl.effector(cmd=2)
    def odd(force, pos, vel, cmd, inertia):
        s = np_.where(np_.logical_and(pos.linear()[2] > 0.0, cmd[0] < 0.5), np_.sin(cmd[1]), np_.cos(cmd[1]))
        lim = np_.clip(vel.linear(), -5.0, 5.0)
        e = np_.exp(-np_.abs(vel.angular()[0])) + np_.arctan2(cmd[0], 1.0 + np_.hypot(cmd[0], cmd[1]))
        return force + dsl.SpatialForce(torque=np_.array([s, e, np_.sqrt(inertia.inertia_diag()[1])]), linear=lim * inertia.mass())
    n = 1000
    w = workloads.independent_bodies(n)
    cmd = np.random.default_rng(0).uniform(-1, 1, (n, 2))
    tp = dsl.pipe(odd).trace()
    # third leg: f32 with hardware transcendentals / polynomial atan2 (codegen fast_math), same f32 tolerance
    for dtype, tol, fast in ((np.float64, parity.F64_RTOL, False), (np.float32, 2e-4, False), (np.float32, 2e-4, True)):
        hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=dtype, simulation_time_step=workloads.DT_120HZ,
                         effectors=odd, columns={"cmd": cmd}, fast_math=fast)
        pos_r, vel_r, acc_r = w["world_pos"].copy(), w["world_vel"].copy(), np.zeros((n, 6))
        eff = lambda xs, vs: dsl_numpy.evaluate(tp, xs, vs, w["inertia"], {"cmd": cmd})
        for _ in range(10):
            pos_r, vel_r, acc_r, F_r = np_sixdof.tick(pos_r, vel_r, acc_r, w["inertia"], eff, workloads.DT_120HZ)
        hip.run(10)
        assert parity.pos_rel_err(hip.world_pos, pos_r) < tol
        assert max(parity.field_rel_err(hip.force[:, :3], F_r[:, :3]), parity.field_rel_err(hip.force[:, 3:], F_r[:, 3:])) < tol


def test_missing_component_column_is_reported():
    w = workloads.independent_bodies(8)
    with pytest.raises(KeyError, match="body_torque"):
        el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], effectors=gravity | rcs)


# ---- whole programs: systems piped around six_dof, generated -----------------------------------------------------------

def test_apollo_sim_written_as_user_code_matches_the_handwritten_model_and_numpy():
    """examples/apollo-lander's pipe (sim.py:517-526) written against the dsl, compiled into the fused kernel, vs
    (a) the numpy stepper evaluating the same program and (b) the hand-written apollo_rollout_kernel with guidance
    switched off (commands constant) — two independent implementations of the same systems."""
    from pathlib import Path
    from elodin_amd import monte_carlo as mc
    from elodin_amd.models import apollo
    from tests import apollo_dsl as A
    ref = apollo.load_reference()
    P = mc.materialize(mc.load_spec(Path(__file__).parent / "golden" / "plans" / "apollo_512.toml")).table()
    cols = apollo.initial_columns(P, ref)
    comps = A.components_from(cols)
    n, ticks = P.shape[0], 600
    w = el.World()
    # build through the C ABI directly: bodies + reference-named component columns
    prog = dsl.Program(A.NON_EFFECTORS, A.EFFECTORS, A.POST)
    hip = el.HipExec(cols["world_pos"], cols["world_vel"], cols["inertia"], simulation_time_step=0.008333333,
                     integrator=L.SEMI_IMPLICIT, effectors=prog, columns=comps, ticks_per_launch=50)
    tp = prog.trace()
    assert tp.writes_inertia and len(tp.columns) == 11
    hip.run(ticks)
    # (a) numpy evaluation of the same program
    pos, vel, acc, inertia = cols["world_pos"].copy(), cols["world_vel"].copy(), np.zeros((n, 6)), cols["inertia"].copy()
    cn = {k: v.copy() for k, v in comps.items()}
    for t in range(1, ticks + 1):
        dsl_numpy.program_tick(tp, pos, vel, acc, inertia, cn, t, 0.008333333, 1)
    assert parity.pos_rel_err(hip.world_pos, pos) < parity.F64_RTOL
    # the setpoint equals the initial attitude, so body rates stay at rounding-noise level (exactly 0 in numpy,
    # ~1e-18 with FMA contraction): absolute floor on the angular half
    assert np.allclose(hip.world_vel[:, :3], vel[:, :3], rtol=1e-9, atol=1e-12)
    assert parity.field_rel_err(hip.world_vel[:, 3:], vel[:, 3:]) < parity.F64_RTOL
    assert parity.field_rel_err(hip.inertia, inertia) < parity.F64_RTOL
    for name in ("throttle", "thrust", "propellant", "rcs_propellant", "rcs_torque", "pitch", "landed"):
        got, want = hip._aux[name], cn[name]
        assert np.allclose(got, want, rtol=1e-8, atol=1e-9), name
    # (b) the hand-written model kernel, guidance never firing
    hw = apollo.ApolloExec(P, ref=ref, ticks_per_launch=50)
    t = apollo.Tables()
    tabs = [np.ascontiguousarray(ref[k]) for k in ("time_s", "altitude_m", "descent_rate_mps", "pitch_deg", "horizontal_speed_mps", "downrange_m")]
    t.time_s, t.altitude_m, t.descent_rate_mps, t.pitch_deg, t.horizontal_speed_mps, t.downrange_m = [a.ctypes.data for a in tabs]
    t.n, t.guidance_period_ticks, t.max_ticks = len(tabs[0]), 10 ** 9, 10 ** 9
    import ctypes as C
    fn = hw._lib.sixdof_set_model_apollo
    fn.argtypes, fn.restype = [C.c_void_p, C.POINTER(apollo.Tables)], C.c_int
    assert fn(hw._h, C.byref(t)) == L.OK
    hw.run(ticks)
    assert parity.pos_rel_err(hip.world_pos, hw.world_pos) < parity.F64_RTOL
    st = hw.model["apollo_state"]
    assert np.max(np.abs(hip._aux["propellant"][:, 0] - st[:, 6]) / st[:, 6]) < 1e-9
    assert np.max(np.abs(hip._aux["pitch"][:, 0] - st[:, 15]) / np.maximum(st[:, 15], 1e-3)) < 1e-8


def test_system_pipe_through_the_world_api_with_cadenced_system():
    """`pre | six_dof(effectors) | post` through World.build, with a system that only runs every 5th tick."""
    @dsl.system(every=5)
    def bump(counter, tick):
        return {"counter": counter + 1.0, "last": tick}

    @dsl.system
    def drain(fuel):
        return {"fuel": np_.maximum(fuel - 0.5, 0.0)}

    @dsl.effector
    def push(force, fuel, inertia):
        return force + dsl.SpatialForce(linear=np_.array([1.0, 0.0, 0.0]) * np_.where(fuel[0] > 0.0, 2.0, 0.0) * inertia.mass())

    w = el.World()
    for k in range(3):
        w.spawn([el.Body(inertia=el.SpatialInertia(1.0 + k)), el.C("fuel", [2.0 + k]), el.C("counter", [0.0]), el.C("last", [0.0])])
    exec = w.build(bump | drain | el.six_dof(1.0, push, el.Integrator.SemiImplicit), simulation_rate=1.0)
    exec.run(12)
    assert exec.component("counter")[:, 0].tolist() == [2.0, 2.0, 2.0]          # ticks 5 and 10
    assert exec.component("last")[:, 0].tolist() == [10.0, 10.0, 10.0]
    assert exec.component("fuel")[:, 0].tolist() == [0.0, 0.0, 0.0]
    # fuel lasts 4 / 6 / 8 half-unit drains -> thrust on for 3 / 5 / 7 ticks (drain runs before six_dof)
    assert exec.column_array("world_vel")[:, 3].tolist() == [6.0, 10.0, 14.0]


def test_table_interpolation_atmosphere_drag():
    """jnp.interp over constant tables, as the rocket example's `mach` system uses it (examples/rocket/main.py:356-375):
    density and temperature from the standard-atmosphere table, quadratic drag against the local flow."""
    H = [0.0, 11_000.0, 20_000.0, 32_000.0, 47_000.0, 51_000.0, 71_000.0, 84_852.0]
    TEMP = [15.0, -56.5, -56.5, -44.5, -2.5, -2.5, -58.5, -86.2]
    RHO = [1.225, 0.3639, 0.0880, 0.0132, 0.0014, 0.0009, 0.0001, 0.0]

    @dsl.effector
    def aero_drag(force, pos, vel):
        altitude = pos.linear()[2]
        temperature = np_.interp(altitude, H, TEMP) + 273.15
        density = np_.interp(altitude, H, RHO)
        speed_of_sound = np_.sqrt(1.4 * 287.05 * temperature)
        v = vel.linear()
        speed = np_.linalg.norm(v)
        mach = speed / speed_of_sound
        cd = 0.3 + 0.2 * np_.tanh(4.0 * (mach - 1.0))
        q = np_.clip(0.5 * density * speed ** 2, 1e-6, 1e12)
        return force + dsl.SpatialForce(linear=v * (-(cd * q * 0.01) / np_.maximum(speed, 1e-6)))

    n = 4000
    w = workloads.independent_bodies(n)
    rng = np.random.default_rng(9)
    pos = w["world_pos"].copy()
    pos[:, 6] = rng.uniform(-2_000.0, 95_000.0, n)          # below, inside and above the table
    pos[:8, 6] = H                                           # exactly on the breakpoints
    vel = w["world_vel"].copy()
    vel[:, 3:] *= rng.uniform(1.0, 60.0, (n, 1))
    pipe = gravity | aero_drag
    tp = pipe.trace()
    hip = el.HipExec(pos, vel, w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=pipe)
    pos_r, vel_r, acc_r = pos.copy(), vel.copy(), np.zeros((n, 6))
    eff = lambda xs, vs: dsl_numpy.evaluate(tp, xs, vs, w["inertia"], {})
    for _ in range(20):
        pos_r, vel_r, acc_r, F_r = np_sixdof.tick(pos_r, vel_r, acc_r, w["inertia"], eff, workloads.DT_120HZ)
    hip.run(20)
    assert parity.pos_rel_err(hip.world_pos, pos_r) < parity.F64_RTOL
    assert parity.field_rel_err(hip.world_vel[:, 3:], vel_r[:, 3:]) < parity.F64_RTOL
    assert parity.field_rel_err(hip.force[:, 3:], F_r[:, 3:]) < parity.F64_RTOL


def test_whole_apollo_campaign_as_user_code_equals_the_handwritten_model():
    """Sim systems + 24 Hz guidance law + command hold + campaign scoring, all written against the dsl and compiled
    into ONE kernel, flown to the surface on the example's own 30-rollout plan: same touchdown ticks and verdicts as
    the hand-written apollo_rollout_kernel (itself checked against the C restatement), continuous results to 1e-6."""
    from pathlib import Path
    from elodin_amd import monte_carlo as mc
    from elodin_amd.models import apollo
    from tests import apollo_dsl as A
    ref = apollo.load_reference()
    P = mc.materialize(mc.load_spec(Path(__file__).parent / "golden" / "plans" / "apollo.toml")).table()
    n, n_ticks = P.shape[0], apollo.max_ticks(ref)
    cols = apollo.initial_columns(P, ref)
    comps = A.components_from(cols)
    comps.update(guid=cols["apollo_guidance"].copy(), score=np.zeros((n, 4)), result=np.zeros((n, 8)),
                 result2=np.zeros((n, 4)), cfg2=np.stack([P[:, 15], P[:, 16], P[:, 3], np.zeros(n)], axis=1))
    guidance, hold, score = A.closed_loop_systems(ref, n_ticks)
    # tick order of the hand-written model: sim systems, six_dof, contact, telemetry, then post_step (score, guidance, hold)
    prog = dsl.Program(A.NON_EFFECTORS, A.EFFECTORS, A.POST + [score, guidance, hold])
    hip = el.HipExec(cols["world_pos"], cols["world_vel"], cols["inertia"], simulation_time_step=0.008333333,
                     integrator=L.SEMI_IMPLICIT, effectors=prog, columns=comps, ticks_per_launch=500)
    hw = apollo.ApolloExec(P, ref=ref, ticks_per_launch=500)
    t0 = hip.invoke_batch(n_ticks)
    hip.download()
    hw.run(n_ticks)
    res = np.concatenate([hip._aux["result"], hip._aux["result2"]], axis=1)
    assert np.array_equal(res[:, 8], hw.result[:, 8]) and np.all(res[:, 8] == 1.0)      # all landed
    assert np.array_equal(res[:, 10], hw.result[:, 10])                                   # on the same tick
    assert np.array_equal(res[:, 9], hw.result[:, 9])                                     # same soft-landing verdicts
    rel = np.abs(res[:, :8] - hw.result[:, :8]) / np.maximum(np.abs(hw.result[:, :8]), 1e-3)
    print("generated closed loop vs hand-written: worst result rel err", rel.max(),
          "device ms", t0.kernel_device_ms, "launches", t0.launches)
    assert rel.max() < 1e-6


# ---- user-written edge_fold functions (GraphQuery.edge_fold with an arbitrary fn, graph.rs:177-282) ----------------

G_NEWTON = 6.6743e-11


@dsl.edge_fold
def gravity_fn(force, a_pos, a_inertia, b_pos, b_inertia):       # examples/three-body/main.py:61-70, verbatim structure
    r = a_pos.linear() - b_pos.linear()
    m = a_inertia.mass()
    M = b_inertia.mass()
    norm = np_.linalg.norm(r)
    f = G_NEWTON * M * m * r / (norm * norm * norm)
    return dsl.SpatialForce(linear=force.force() - f)


def _three_body_state():
    g = gu.load("three_body")
    pos = np.stack([g[f"{e}.world_pos"][0] for e in "abc"])
    vel = np.stack([g[f"{e}.world_vel"][0] for e in "abc"])
    inertia = np.stack([g[f"{e}.inertia"][0] for e in "abc"])
    edge_names = ["a_>_b", "b_>_a", "a_>_c", "b_>_c", "c_>_a", "c_>_b"]
    frm = np.array([g[f"{e}.gravity_edge"][0, 0] for e in edge_names], dtype=np.uint64)
    to = np.array([g[f"{e}.gravity_edge"][0, 1] for e in edge_names], dtype=np.uint64)
    return g, pos, vel, inertia, frm, to


@pytest.mark.parametrize("small", [True, False])
def test_three_body_gravity_written_by_the_user_matches_reference_golden(small, monkeypatch):
    """The example's gravity_fn as user code -> generated PAIR functor -> the same pair kernels; checked against the
    reference's golden CSV (scripts/ci/baseline/three-body-csv) and against the built-in Newton op."""
    if not small:
        monkeypatch.setenv("SIXDOF_PAIR_SMALL", "0")
    g, pos, vel, inertia, frm, to = _three_body_state()
    kw = dict(entity_ids=[1, 2, 3], simulation_time_step=float(g["globals.simulation_time_step"][0, 0]), edges=(frm, to))
    user = el.HipExec(pos, vel, inertia, effectors=[gravity_fn], **kw)
    builtin = el.HipExec(pos, vel, inertia, effectors=[el.Effector(L.EFF_EDGE_GRAVITY_NEWTON, (G_NEWTON,))], **kw)
    worst = 0.0
    for r in range(1, 101):
        t = user.run(1)
        assert t.launches == (1 if small else 2)      # small: the one-launch kernel; else pack + the fused fold-and-integrate launch (no hub sources)
        for i, e in enumerate("abc"):
            worst = max(worst, parity.pos_rel_err(user.world_pos[i:i + 1], g[f"{e}.world_pos"][r][None]))
            for comp in ("world_vel", "world_accel", "force"):
                worst = max(worst, parity.field_rel_err(getattr(user, comp)[i:i + 1, 3:], g[f"{e}.{comp}"][r][None, 3:]))
    builtin.run(100)
    vs_builtin = max(parity.field_rel_err(getattr(user, f), getattr(builtin, f)) for f in parity.FIELDS)
    print("user-written three-body fold: vs golden", worst, "vs built-in op", vs_builtin)
    assert worst < parity.F64_RTOL and vs_builtin < 1e-12


K_SPRING, L0 = 40.0, 1.5


@dsl.edge_fold(edge_component="spring")
def spring(acc, a_pos, a_inertia, b_pos, b_inertia):
    """A fold that accumulates force AND torque and uses both masses: damped-length spring along the edge."""
    r = b_pos.linear() - a_pos.linear()
    d = np_.linalg.norm(r)
    mu = a_inertia.mass() * b_inertia.mass() / (a_inertia.mass() + b_inertia.mass())
    f = (K_SPRING * mu * np_.tanh(d - L0) / d) * r
    arm = np_.array([0.0, 0.0, 0.25])
    return acc + dsl.SpatialForce(torque=np_.cross(arm, f), linear=f)


@pytest.mark.parametrize("integrator", [L.RK4, L.SEMI_IMPLICIT])
@pytest.mark.parametrize("n", [200, 3000])
def test_user_fold_with_torque_after_builtin_ops_vs_numpy_fold(integrator, n):
    """Sparse random graph (some rows are no edge's source and keep the per-entity pipe's Force), user fold returning
    a full wrench, both launch shapes (single-workgroup n <= 256, three-kernel otherwise) vs the numpy stepper
    evaluating the same DAG sequentially in spawn order."""
    w = workloads.independent_bodies(n, seed=11)
    rng = np.random.default_rng(5)
    m = 3 * n
    src = rng.integers(0, n - n // 8, size=m)          # the last n/8 rows are never sources
    dst = (src + 1 + rng.integers(0, n - 1, size=m)) % n
    ids = np.arange(1, n + 1, dtype=np.uint64)
    g_vec = (0.0, 0.0, -9.81)
    hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=integrator, simulation_time_step=1 / 240.0,
                     effectors=[el.Effector(L.EFF_UNIFORM_GRAVITY, g_vec), spring], edges=(ids[src], ids[dst]))
    hs, hd = hip.edge_rows()
    tf = spring.trace()

    def effectors(xs, vs):
        prior = np.zeros((n, 6))
        prior[:, 3:] = np.array(g_vec) * w["inertia"][:, 6:7]
        return dsl_numpy.fold_force(tf, xs, w["inertia"], src, dst, prior)

    pos, vel, acc = w["world_pos"].copy(), w["world_vel"].copy(), np.zeros((n, 6))
    for _ in range(6):
        pos, vel, acc, F = np_sixdof.tick(pos, vel, acc, w["inertia"], effectors, 1 / 240.0, integrator=integrator)
    hip.run(6)
    assert np.array_equal(np.sort(hs.astype(np.int64) * n + hd), np.sort(src * n + dst))
    errs = {"world_pos": parity.pos_rel_err(hip.world_pos, pos), "world_vel": parity.field_rel_err(hip.world_vel, vel),
            "world_accel": parity.field_rel_err(hip.world_accel, acc), "force": parity.field_rel_err(hip.force, F)}
    print("user fold with torque", n, integrator, errs)
    assert max(errs.values()) < parity.F64_RTOL, errs
    never = np.setdiff1d(np.arange(n), src)
    assert len(never) >= n // 8 and np.all(hip.force[never, :3] == 0.0)
    assert np.allclose(hip.force[never, 5], -9.81 * w["inertia"][never, 6], rtol=1e-15)
    assert np.any(hip.force[np.unique(src), :3] != 0.0)


def test_user_fold_through_the_world_api():
    """three-body/main.py end to end on this framework: GravityEdge spawns + six_dof(sys=<user fold>)."""
    g, pos, vel, inertia, frm, to = _three_body_state()
    w = el.World()
    bodies = [w.spawn(el.Body(world_pos=el.SpatialTransform(linear=pos[i, 4:]), world_vel=el.SpatialMotion(linear=vel[i, 3:]),
                              inertia=el.SpatialInertia(inertia[i, 6])), name="abc"[i]) for i in range(3)]
    for a, b in ((0, 1), (1, 0), (0, 2), (1, 2), (2, 0), (2, 1)):
        w.spawn(el.GravityEdge(bodies[a], bodies[b]), name=f"{'abc'[a]} -> {'abc'[b]}")
    ex = w.build(el.six_dof(sys=gravity_fn), simulation_rate=120.0)
    ex.run(100)
    got = ex.column_array("world_pos")
    worst = max(parity.pos_rel_err(got[i:i + 1], g[f"{e}.world_pos"][100][None]) for i, e in enumerate("abc"))
    assert worst < parity.F64_RTOL, worst


def test_edge_fold_needs_edges_and_must_close_the_pipe():
    g, pos, vel, inertia, frm, to = _three_body_state()
    with pytest.raises(ValueError):
        el.HipExec(pos, vel, inertia, effectors=[gravity_fn])
    with pytest.raises(ValueError):
        el.HipExec(pos, vel, inertia, effectors=[gravity_fn, el.Effector(L.EFF_UNIFORM_GRAVITY, (0, 0, -1))], edges=(frm, to))


def test_fast_math_functions_over_their_domains():
    """codegen fast_math (f32): every replaced function against numpy in f64 over a grid of arguments, through a program
    that just evaluates them into a component column.  Bound: |err| <= 2e-6 + 4e-6 |value| for every function
    (sin / cos / tan / atan2 / atan / exp / pow / log / division / sqrt / hypot) on the ranges the campaigns use."""
    @dsl.system
    def table(a, b):
        x, y = a[0], a[1]
        return {"b": np_.array([np_.sin(x), np_.cos(x), np_.tan(x * 0.4), np_.arctan2(y, x), np_.arctan2(x, y), np_.exp(-np_.abs(x)),
                                np_.power(np_.abs(x) + 0.5, y), x / (np_.abs(y) + 0.25), np_.sqrt(np_.abs(x)), np_.hypot(x, y),
                                np_.arctan(x * 3.0), np_.log(np_.abs(y) + 0.1)])}
    rng = np.random.default_rng(4)
    n = 4096
    a = np.stack([rng.uniform(-3.1, 3.1, n), rng.uniform(-3.1, 3.1, n)], axis=1)
    a[:8] = [[0, 1], [1, 0], [0, -1], [-1, 0], [1, 1], [-1, -1], [0.0, 0.0], [-2.5, 1e-3]]
    w = workloads.independent_bodies(n)
    x, y = a[:, 0], a[:, 1]
    want = np.stack([np.sin(x), np.cos(x), np.tan(x * 0.4), np.arctan2(y, x), np.arctan2(x, y), np.exp(-np.abs(x)),
                     np.power(np.abs(x) + 0.5, y), x / (np.abs(y) + 0.25), np.sqrt(np.abs(x)), np.hypot(x, y), np.arctan(x * 3.0),
                     np.log(np.abs(y) + 0.1)], axis=1)
    prog = dsl.Program([table], dsl.Pipe([]), [])
    hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=np.float32, integrator=L.INTEGRATOR_NONE,
                     effectors=prog, columns={"a": a, "b": np.zeros((n, 12))}, fast_math=True)
    hip.run(1)
    got = np.asarray(hip._aux["b"], dtype=np.float64)
    a32 = a.astype(np.float32).astype(np.float64)       # the kernel sees the f32-rounded arguments
    x, y = a32[:, 0], a32[:, 1]
    want = np.stack([np.sin(x), np.cos(x), np.tan(x * 0.4), np.arctan2(y, x), np.arctan2(x, y), np.exp(-np.abs(x)),
                     np.power(np.abs(x) + 0.5, y), x / (np.abs(y) + 0.25), np.sqrt(np.abs(x)), np.hypot(x, y), np.arctan(x * 3.0),
                     np.log(np.abs(y) + 0.1)], axis=1)
    err = np.abs(got - want)
    excess = err - (2e-6 + 4e-6 * np.abs(want))
    assert excess.max() <= 0.0, (np.unravel_index(np.argmax(excess), excess.shape), excess.max())


def test_fast_math_normal_samples_are_the_reference_draws_to_single_precision():
    """fast_math programs run erfinv in single precision (codegen._FAST_ERFINV: Giles' polynomial on a double-formed 1 - u^2):
    12,288 normal samples of a float32 fast-math program against the same draws in double (numpy walk of the trace: threefry
    bits -> uniform -> scipy erfinv), same bound as the other replaced functions; the plain float32 program draws the
    double samples, rounded."""
    @dsl.system
    def draw(s, z):
        key = dsl.random.fold_in(dsl.random.key(20170814), s)
        return {"z": dsl.random.normal(key, shape=(3,)) * 2.0}
    n = 4096
    seeds = np.arange(n, dtype=np.float64)[:, None] * 7.0 + 3.0
    w = workloads.independent_bodies(n)
    tp = dsl.Program([draw], dsl.Pipe([]), []).trace({"s": 1, "z": 3})
    comps = {"s": seeds.copy(), "z": np.zeros((n, 3))}
    dsl_numpy.program_tick_systems_only(tp, w["world_pos"].copy(), w["world_vel"].copy(), np.zeros((n, 6)), w["inertia"].copy(), comps, 1)
    want = comps["z"]
    assert 3.5 < np.abs(want).max() / 2.0 < 5.4 and abs(want.std() / 2.0 - 1.0) < 0.02
    for fast, bound in ((True, lambda v: 2e-6 + 4e-6 * np.abs(v)), (False, lambda v: 1e-7 + 1.2e-7 * np.abs(v))):
        hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=np.float32, integrator=L.INTEGRATOR_NONE,
                         effectors=dsl.Program([draw], dsl.Pipe([]), []), columns={"s": seeds, "z": np.zeros((n, 3))}, fast_math=fast)
        hip.run(1)
        got = np.asarray(hip._aux["z"], dtype=np.float64)
        excess = np.abs(got - want) - bound(want)
        assert excess.max() <= 0.0, (fast, np.unravel_index(np.argmax(excess), excess.shape), excess.max())


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-13), (np.float32, 3e-6)])
def test_interp_over_a_long_evenly_spaced_table_is_exactly_searchsorted(dtype, tol):
    """Long evenly spaced tables are indexed by division (then corrected against the stored breakpoints) instead of
    bisected: queries ON breakpoints, between them, at both ends, outside, and for a table whose step is not exactly
    representable (0.1) must match np.interp."""
    xs = tuple(0.1 * k - 3.0 for k in range(101))
    fs = tuple(float(np.sin(1.7 * v) + 0.01 * k) for k, v in enumerate(xs))

    @dsl.system
    def look(q, r):
        return {"r": np_.array([np_.interp(q[0], xs, fs), np_.interp(q[1], xs, fs)])}
    rng = np.random.default_rng(9)
    n = 2048
    q = rng.uniform(-3.5, 7.6, (n, 2))
    q[:101, 0] = xs                                   # exactly on every breakpoint
    q[:100, 1] = np.nextafter(np.array(xs[1:]), -np.inf)   # just below every breakpoint
    q[101:105, 0] = [-1e30, 1e30, -3.0, 7.0]
    w = workloads.independent_bodies(n)
    hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=dtype, integrator=L.INTEGRATOR_NONE,
                     effectors=dsl.Program([look], dsl.Pipe([]), []), columns={"q": q, "r": np.zeros((n, 2))})
    hip.run(1)
    qq = q.astype(dtype).astype(np.float64)
    want = np.stack([np.interp(qq[:, 0], xs, fs), np.interp(qq[:, 1], xs, fs)], axis=1)
    assert "m_interp_uniform" in codegen_source_of(look)
    assert np.max(np.abs(np.asarray(hip._aux["r"], dtype=np.float64) - want)) < tol


def codegen_source_of(system):
    from elodin_amd import codegen
    tp = dsl.Program([system], dsl.Pipe([]), []).trace({"q": 2, "r": 2})
    return codegen.generate_source(tp, "float64", 2)


def test_while_loops_with_per_lane_trip_counts_on_the_gpu():
    """dsl.lax.while_loop becomes a real loop in the kernel; lanes leave it independently.  Kepler's equation by Newton
    (2..60 iterations depending on eccentricity) and the shape of the FSW's ballistic impact predictor (up to 2,400
    half-second steps) against numpy evaluating the same traced program, f64 and f32."""
    from tests.test_dsl_host import ballistic_impact, kepler

    @dsl.system
    def solve(q, r):
        E, iters = kepler(np_, q[0], q[1])
        x, k = ballistic_impact(np_, q[2], q[3], q[4], q[5])
        return {"r": np_.array([E, iters, x, k])}
    rng = np.random.default_rng(12)
    n = 4096
    q = np.stack([rng.uniform(0.0, 3.1, n), rng.uniform(0.0, 0.97, n), rng.uniform(200.0, 90_000.0, n), rng.uniform(-200.0, 1500.0, n),
                  rng.uniform(0.0, 1500.0, n), rng.uniform(1e-4, 3e-3, n)], axis=1)
    w = workloads.independent_bodies(n)
    prog = dsl.Program([solve], dsl.Pipe([]), [])
    tp = prog.trace({"q": 6, "r": 4})
    for dtype, tol in ((np.float64, 1e-11), (np.float32, 2e-3)):
        hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=dtype, integrator=L.INTEGRATOR_NONE, effectors=prog,
                         columns={"q": q, "r": np.zeros((n, 4))})
        hip.run(1)
        got = np.asarray(hip._aux["r"], dtype=np.float64)
        comps = {"q": q.astype(dtype).astype(np.float64), "r": np.zeros((n, 4))}
        pos, vel, inertia = (np.array(w[k], dtype=np.float64) for k in ("world_pos", "world_vel", "inertia"))
        dsl_numpy._run_systems(tp.pre, pos, vel, inertia, comps, tp.table, 1)
        want = comps["r"]
        if dtype == np.float64:
            # same trip counts, except where the 1e-13 stop test sits on a rounding boundary (FMA contraction on the GPU)
            assert np.abs(got[:, 1] - want[:, 1]).max() <= 1 and (got[:, 1] == want[:, 1]).mean() > 0.97
            assert np.array_equal(got[:, 3], want[:, 3])
            assert 2 <= got[:, 1].min() < got[:, 1].max() <= 60 and got[:, 3].min() < 100 < got[:, 3].max()
        assert np.max(np.abs(got[:, 0] - want[:, 0])) < tol * 10                                  # eccentric anomaly
        close = np.abs(got[:, 3] - want[:, 3]) <= (0 if dtype == np.float64 else 1)            # f32 may land one step apart
        assert close.mean() > 0.999
        assert np.max(np.abs(got[close, 2] - want[close, 2]) / np.maximum(np.abs(want[close, 2]), 1.0)) < max(tol, 1e-9) * 50


def test_common_subexpressions_are_not_reused_across_a_write():
    """The emitter shares temporaries across the systems of a tick only while no leaf they read has been written: here
    system A evaluates sin(x) and then rewrites x, system B evaluates the (structurally identical) sin(x) again and must
    see the new x; a third system, cadenced, reuses neither."""
    @dsl.system
    def a(x, y):
        return {"y": np_.sin(x) + np_.cos(x * 2.0), "x": x + 1.0}

    @dsl.system
    def b(x, z):
        return {"z": np_.sin(x) + np_.cos(x * 2.0)}

    @dsl.system(every=2)
    def c(x, z, u):
        return {"u": np_.sin(x) + z}
    n = 512
    x0 = np.random.default_rng(2).uniform(-2, 2, (n, 1))
    w = workloads.independent_bodies(n)
    prog = dsl.Program([a, b, c], dsl.Pipe([]), [])
    cols = {"x": x0, "y": np.zeros((n, 1)), "z": np.zeros((n, 1)), "u": np.zeros((n, 1))}
    hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.INTEGRATOR_NONE, effectors=prog, columns=cols)
    hip.run(3)
    x = x0[:, 0].copy()
    u = np.zeros(n)
    for tick in (1, 2, 3):
        y = np.sin(x) + np.cos(x * 2.0)
        x = x + 1.0
        z = np.sin(x) + np.cos(x * 2.0)
        if tick % 2 == 0:
            u = np.sin(x) + z
    for name, want in (("x", x), ("y", y), ("z", z), ("u", u)):
        assert np.allclose(hip._aux[name][:, 0], want, rtol=1e-13, atol=1e-15), name


def test_ball_example_end_to_end_from_the_seed():
    """examples/ball exactly as the reference builds it: the wind is NOT read from the golden CSV but sampled in the
    kernel by `sample_wind` from seed 0 with JAX's threefry2x32 generator (partitionable layout) + erfinv — it must equal
    ball.wind.csv, and the whole 100-tick trajectory the golden columns."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("ball_example", Path(__file__).resolve().parents[1] / "examples" / "ball.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = gu.load("ball")
    exec = mod.build(seed=0)
    worst = 0.0
    for r in range(1, 101):
        exec.run(1)
        assert np.allclose(exec.column_array("wind")[0], g["ball.wind"][r], rtol=1e-13, atol=0.0), r
        worst = max(worst, parity.pos_rel_err(exec.column_array("world_pos"), g["ball.world_pos"][r][None]),
                    parity.field_rel_err(exec.column_array("world_vel")[:, 3:], g["ball.world_vel"][r][None, 3:]),
                    parity.field_rel_err(exec.column_array("force")[:, 3:], g["ball.force"][r][None, 3:]))
    print("ball example from the seed: wind", exec.column_array("wind")[0], "worst rel err vs golden", worst)
    assert worst < parity.F64_RTOL


def test_random_streams_match_numpy_evaluation_and_differ_per_entity():
    """jax.random-shaped generators in per-entity code: every entity has its own seed column, fold_in(key, tick) gives a
    fresh key per tick (the pattern of the reference's sensor noise, examples/falcon9/sensors.py), uniform and normal."""
    @dsl.system
    def noise(seed, tick, sample):
        key = dsl.random.fold_in(dsl.random.key(seed), tick)
        z = dsl.random.normal(key, shape=(2,))
        u = dsl.random.uniform(dsl.random.fold_in(key, 3.0), shape=(2,), minval=-1.0, maxval=3.0)
        return {"sample": np_.concatenate([z, u])}
    n = 1000
    seeds = np.arange(n, dtype=np.float64)[:, None] * 7919.0 + 20170814.0
    w = workloads.independent_bodies(n)
    prog = dsl.Program([noise], dsl.Pipe([]), [])
    hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.INTEGRATOR_NONE, effectors=prog,
                     columns={"seed": seeds, "sample": np.zeros((n, 4))})
    tp = prog.trace()
    draws = []
    for tick in (1, 2, 3):
        hip.run(1)
        comps = {"seed": seeds.copy(), "sample": np.zeros((n, 4))}
        pos, vel, inertia = (np.array(w[k], dtype=np.float64) for k in ("world_pos", "world_vel", "inertia"))
        dsl_numpy._run_systems(tp.pre, pos, vel, inertia, comps, tp.table, tick)
        assert np.allclose(hip._aux["sample"], comps["sample"], rtol=1e-13, atol=1e-15)
        draws.append(hip._aux["sample"].copy())
    all_z = np.concatenate([d[:, :2].ravel() for d in draws])
    all_u = np.concatenate([d[:, 2:].ravel() for d in draws])
    assert abs(all_z.mean()) < 0.05 and abs(all_z.std() - 1.0) < 0.05 and -1.0 <= all_u.min() and all_u.max() < 3.0
    assert abs(all_u.mean() - 1.0) < 0.06 and len(np.unique(all_z)) == all_z.size       # independent streams per entity and tick
    # a float32 program: the generator runs in a double island (uint32 words do not fit a float32), the sample is rounded
    # on the way out — the same noise as the float64 program draws for the same seed and tick
    seeds32 = np.arange(n, dtype=np.float64)[:, None] * 7.0 + 1234.0               # exact in float32
    hip32 = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=np.float32, integrator=L.INTEGRATOR_NONE,
                       effectors=dsl.Program([noise], dsl.Pipe([]), []), columns={"seed": seeds32, "sample": np.zeros((n, 4))})
    for tick in (1, 2):
        hip32.run(1)
        comps = {"seed": seeds32.copy(), "sample": np.zeros((n, 4))}
        pos, vel, inertia = (np.array(w[k], dtype=np.float64) for k in ("world_pos", "world_vel", "inertia"))
        dsl_numpy._run_systems(tp.pre, pos, vel, inertia, comps, tp.table, tick)
        assert hip32._aux["sample"].dtype == np.float32
        assert np.allclose(hip32._aux["sample"], comps["sample"], rtol=3e-7, atol=1e-7), tick


def test_fast_math_bits_do_not_depend_on_the_launch_shape_and_a_program_object_is_built_once():
    """A fast-math program (products folded into their sums by the generator, hardware sqrt / rcp, a store-only column) gives
    the SAME bits with 1, 5 and 12 ticks per launch — fusion is decided per DAG node, not by what the compiler finds in one
    basic block — and the history ring sees the store-only column on every tick.  The three executors come from ONE program
    object: traced once, built once (HipExec keeps both on the object)."""
    @dsl.system
    def plant(x, y, derived):
        a, b, c = x[0], x[1], x[2]
        s = a * b + c
        t = c - b * b * 0.25
        u = np_.sqrt(np_.abs(s * t) + 1.0) / (np_.abs(a) + 0.5)
        return {"x": np_.array([a * 0.999 + u * 1e-3, b - a * 1e-3, c + np_.where(u > 1.2, s, t) * 1e-3]),
                "y": np_.array([s, t, u]) + y * 0.5,
                "derived": np_.array([np_.sin(a) * u, s - t])}                 # nobody reads `derived`: evaluated where stored
    prog = dsl.Program([plant], dsl.Pipe([]), [])
    n = 4096
    rng = np.random.default_rng(11)
    x0 = rng.uniform(-2.0, 2.0, (n, 3))
    w = workloads.independent_bodies(n)
    runs = []
    for k in (1, 5, 12):
        hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=np.float32, integrator=L.INTEGRATOR_NONE, effectors=prog,
                         columns={"x": x0, "y": np.zeros((n, 3)), "derived": np.zeros((n, 2))}, fast_math=True, ticks_per_launch=k,
                         reuse_trace=True)
        if k == 12:
            hip.enable_history(60)
        hip.run(60)
        runs.append({c: np.array(hip._aux[c]) for c in ("x", "y", "derived")})
        if k == 12:
            hist = hip.history("derived", 1, 60)
        hip.close()
    for r in runs[1:]:
        for c in ("x", "y", "derived"):
            assert np.array_equal(r[c], runs[0][c]), c
    assert np.isfinite(runs[0]["y"]).all() and np.abs(runs[0]["derived"]).max() > 0.0
    assert np.array_equal(hist[-1], runs[0]["derived"]) and not np.array_equal(hist[0], hist[-1])      # recorded on every tick
    memo = prog._exec_memo
    assert sum(1 for key in memo if key[0] == "trace") == 1 and sum(1 for key in memo if key[0] == "build") == 1


def test_a_program_object_is_traced_again_unless_the_caller_promises_it_did_not_change():
    """the trace memo is opt-in.  A system that closes over a Python gain is re-traced by every executor, so changing
    the gain between two executors changes what the second computes; with reuse_trace=True the FIRST trace is kept (the promise)."""
    gain = [2.0]

    @dsl.system
    def scale(x):
        return {"x": x * gain[0]}
    prog = dsl.Program([scale], dsl.Pipe([]), [])
    n = 64
    w = workloads.independent_bodies(n)
    x0 = np.arange(n, dtype=np.float64).reshape(n, 1) + 1.0

    def run(**kw):
        hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.INTEGRATOR_NONE, effectors=prog, columns={"x": x0}, **kw)
        hip.run(1)
        out = np.array(hip._aux["x"])
        hip.close()
        return out
    assert np.array_equal(run(), 2.0 * x0)
    gain[0] = 3.0
    assert np.array_equal(run(), 3.0 * x0) and "_exec_memo" not in prog.__dict__       # traced again: the new gain
    assert np.array_equal(run(reuse_trace=True), 3.0 * x0)
    gain[0] = 5.0
    assert np.array_equal(run(reuse_trace=True), 3.0 * x0)                              # the promise: the first kept trace
    assert np.array_equal(run(), 5.0 * x0)
