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
    for dtype, tol in ((np.float64, parity.F64_RTOL), (np.float32, 2e-4)):
        hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=dtype, simulation_time_step=workloads.DT_120HZ,
                         effectors=odd, columns={"cmd": cmd})
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
