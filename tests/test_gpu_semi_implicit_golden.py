"""The HIP path against the reference's cube-sat golden rows (SemiImplicit, torque != 0, I_diag non-uniform).

Same teacher-forced construction as tests/test_oracle_semi_implicit_golden.py — the recorded `force` row r drives a
one-tick step from row r-1 — through (a) the hand-written step kernel with the world-frame torque / force column ops
and (b) a generated program whose user-written effector reads the recorded wrench as a component column.
"""
import numpy as np
import pytest

import elodin_amd as ea
from elodin_amd import _lib as L
from elodin_amd import dsl
from elodin_amd.exec import Effector
from tests import parity
from tests.test_oracle_semi_implicit_golden import teacher_forced_world, worst_errors

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ticks_per_launch", [1, 4])
def test_handwritten_kernel_semi_implicit_vs_cube_sat_golden(ticks_per_launch):
    x, v, I, F, exp, dt = teacher_forced_world()
    hip = ea.HipExec(x, v, I, simulation_time_step=dt, integrator=L.SEMI_IMPLICIT, ticks_per_launch=ticks_per_launch,
                     effectors=[Effector(L.EFF_WORLD_TORQUE, (), "recorded_torque", F[:, :3]),
                                Effector(L.EFF_WORLD_FORCE, (), "recorded_force", F[:, 3:])])
    hip.run(1)
    err = worst_errors({c: getattr(hip, c) for c in exp}, exp)
    print("cube-sat teacher-forced, HIP step kernel:", err)
    assert err["force"] == 0.0
    assert max(err.values()) < parity.F64_RTOL, err
    assert max(err.values()) < 1e-13, err          # one tick: only the cheaper arithmetic forms separate the two


@dsl.effector(recorded_torque=3, recorded_force=3)
def replay(force, recorded_torque, recorded_force):
    return force + dsl.SpatialForce(torque=recorded_torque, linear=recorded_force)


def test_generated_effector_semi_implicit_vs_cube_sat_golden():
    x, v, I, F, exp, dt = teacher_forced_world()
    hip = ea.HipExec(x, v, I, simulation_time_step=dt, integrator=L.SEMI_IMPLICIT, effectors=replay,
                     columns={"recorded_torque": F[:, :3], "recorded_force": F[:, 3:]})
    hip.run(1)
    err = worst_errors({c: getattr(hip, c) for c in exp}, exp)
    print("cube-sat teacher-forced, generated effector:", err)
    assert max(err.values()) < 1e-13, err


def test_rk4_with_recorded_wrench_columns_matches_oracle():
    """The two column ops under RK4 (run-time interpreter pipe) against the oracle: 50 ticks, constant wrench rows."""
    from oracle import oracle as orc
    x, v, I, F, exp, dt = teacher_forced_world()
    hip = ea.HipExec(x, v, I, simulation_time_step=dt, integrator=L.RK4,
                     effectors=[Effector(L.EFF_WORLD_TORQUE, (), "recorded_torque", F[:, :3]),
                                Effector(L.EFF_WORLD_FORCE, (), "recorded_force", F[:, 3:])])
    ref = orc.OracleWorld(x, v, I, simulation_time_step=dt, integrator=orc.RK4,
                          ops=[(orc.EFF_WORLD_TORQUE, (), F[:, :3]), (orc.EFF_WORLD_FORCE, (), F[:, 3:])])
    hip.run(50)
    ref.step(50)
    errs = parity.state_errors(hip, ref)
    assert max(errs.values()) < parity.F64_RTOL, errs
