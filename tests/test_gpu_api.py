"""The reference's own pytest cases for this path (libs/nox-py/python/tests/test_all.py), written against
the mirrored API and executed by the HIP backend."""
import numpy as np
import pytest

import elodin_amd as el
from tests import golden_util as gu
from tests import parity

pytestmark = pytest.mark.gpu


def test_six_dof():  # test_all.py:67-83
    w = el.World()
    w.spawn(el.Body(world_pos=el.SpatialTransform(linear=np.array([0.0, 0.0, 0.0])),
                    world_vel=el.SpatialMotion(linear=np.array([1.0, 0.0, 0.0])),
                    inertia=el.SpatialInertia(1.0)), "e1")
    exec = w.build(el.six_dof(1.0 / 60.0))
    exec.run()
    x = exec.column_array("world_pos")[-1]
    assert np.allclose(x[:4], [0.0, 0.0, 0.0, 1.0])
    assert np.allclose(x[4:], [0.01666667, 0.0, 0.0])
    assert exec.tick == 1


@pytest.mark.parametrize("omega,q", [([0, 0, 1.0], [0.0, 0.0, 0.479425538604203, 0.8775825618903728]),
                                     ([0, 1.0, 0], [0.0, 0.479425538604203, 0.0, 0.8775825618903728]),
                                     ([1.0, 1.0, 0], [0.45936268493243, 0.45936268493243, 0.0, 0.76024459707606])])
def test_six_dof_ang_vel_int(omega, q):  # test_all.py:228-292, "value from Julia and Simulink"
    w = el.World()
    w.spawn(el.Body(world_vel=el.SpatialMotion(angular=np.array(omega)), inertia=el.SpatialInertia(1.0)), "e1")
    exec = w.build(el.six_dof(1.0 / 120.0))
    exec.run(120)
    assert np.isclose(exec.column_array("world_pos")[-1], q + [0.0, 0.0, 0.0], rtol=1e-5).all()


def test_six_dof_force():  # test_all.py:342-364, "values taken from simulink"
    w = el.World()
    w.spawn(el.Body(inertia=el.SpatialInertia(1.0)), "e1")
    exec = w.build(el.six_dof(1.0 / 120.0, el.constant_wrench(force=(1.0, 0.0, 0.0))))
    exec.run(120)
    assert np.isclose(exec.column_array("world_pos")[-1], [0.0, 0.0, 0.0, 1.0, 0.5, 0.0, 0.0], rtol=1e-5).all()


def test_three_body_example_script():
    """examples/three-body/main.py written against the mirror; checked against the reference's golden CSV."""
    G = 6.6743e-11
    w = el.World()
    a = w.spawn(el.Body(world_pos=el.SpatialTransform(linear=[0.8920281421, 0.0, 0.0]),
                        world_vel=el.SpatialMotion(linear=[0.0, 0.9957939373, 0.0]), inertia=el.SpatialInertia(1.0 / G)), name="A")
    b = w.spawn(el.Body(world_pos=el.SpatialTransform(linear=[-0.6628498947, 0.0, 0.0]),
                        world_vel=el.SpatialMotion(linear=[0.0, -1.6191613336, 0.0]), inertia=el.SpatialInertia(1.0 / G)), name="B")
    c = w.spawn(el.Body(world_pos=el.SpatialTransform(linear=[-0.2291782474, 0, 0]),
                        world_vel=el.SpatialMotion(linear=[0, 0.6233673964, 0.0]), inertia=el.SpatialInertia(1.0 / G)), name="C")
    for x, y in ((a, b), (b, a), (a, c), (b, c), (c, a), (c, b)):
        w.spawn(el.GravityEdge(x, y))
    exec = w.build(el.six_dof(sys=el.gravity_newton(G)), simulation_rate=120.0)
    exec.run(100)
    g = gu.load("three_body")
    for i, e in enumerate("abc"):
        assert parity.pos_rel_err(exec.column_array("world_pos")[i:i + 1], g[f"{e}.world_pos"][100][None]) < parity.F64_RTOL
        assert parity.field_rel_err(exec.column_array("world_vel")[i:i + 1, 3:], g[f"{e}.world_vel"][100][None, 3:]) < parity.F64_RTOL
    assert exec.entity_ids().tolist() == [1, 2, 3] and exec.tick == 100
    assert exec.profile()["real_time_factor"] > 0


def test_ball_example_with_telemetry_batches():
    """examples/ball (gravity | drag), telemetry_rate = simulation_rate / 4 -> 4 ticks per launch."""
    g = gu.load("ball")
    w = el.World()
    w.spawn([el.Body(world_pos=el.SpatialTransform(linear=[0.0, 0.0, 6.0])), el.C("wind", g["ball.wind"][1])], name="ball")
    exec = w.build(el.six_dof(sys=el.uniform_gravity() | el.ball_drag("wind")), simulation_rate=120.0, telemetry_rate=30.0)
    exec.run(100)
    assert parity.pos_rel_err(exec.column_array("world_pos"), g["ball.world_pos"][100][None]) < parity.F64_RTOL
    assert parity.field_rel_err(exec.column_array("world_vel")[:, 3:], g["ball.world_vel"][100][None, 3:]) < parity.F64_RTOL
