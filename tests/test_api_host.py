"""Host logic of the reference-shaped Python surface (no GPU): spawn/id/column semantics and error behaviour."""
import numpy as np
import pytest

import elodin_amd as el


def test_spawn_assigns_sequential_ids_from_one_and_appends_rows():
    w = el.World()
    a = w.spawn(el.Body(world_pos=el.SpatialTransform(linear=[1.0, 2.0, 3.0])), name="a")
    b = w.spawn([el.Body(inertia=el.SpatialInertia(2.0)), el.C("wind", [0.1, 0.2, 0.3])], name="b")
    e = w.spawn(el.GravityEdge(a, b))        # edge entities consume ids too
    c = w.spawn(el.Body())
    assert (a, b, e, c) == (1, 2, 3, 4)      # Globals = 0 (world.rs:174-196)
    pos, ids = w.column("world_pos")
    assert ids.dtype == np.uint64 and ids.tolist() == [1, 2, 4]
    assert pos.shape == (3, 7) and pos[0].tolist() == [0, 0, 0, 1, 1, 2, 3]
    inertia, _ = w.column("inertia")
    assert inertia[1].tolist() == [2, 2, 2, 0, 0, 0, 2]    # default inertia = ones*mass (spatial.rs:399-403)
    wind, wids = w.column("wind")
    assert wids.tolist() == [2] and wind.shape == (1, 3)
    with pytest.raises(KeyError):
        w.column("nope")


def test_body_defaults_and_spatial_types():
    b = el.Body()
    assert b.world_pos.arr.tolist() == [0, 0, 0, 1, 0, 0, 0] and b.inertia.mass() == 1.0
    q = el.Quaternion.from_axis_angle([0, 0, 2.0], np.pi)
    assert np.allclose(q.vector(), [0, 0, 1, 0])
    m = el.SpatialMotion(angular=[1, 2, 3], linear=[4, 5, 6])
    assert m.arr.tolist() == [1, 2, 3, 4, 5, 6]            # angular first
    f = el.SpatialForce(linear=[1, 0, 0])
    assert f.torque().tolist() == [0, 0, 0] and f.force().tolist() == [1, 0, 0]


def test_six_dof_signature_and_pipe():
    s = el.six_dof()
    assert s.time_step is None and s.integrator is el.Integrator.Rk4 and s.effectors.ops == []
    s = el.six_dof(1 / 60.0, el.uniform_gravity() | el.ball_drag("wind"), el.Integrator.SemiImplicit)
    assert [o.kind for o in s.effectors.ops] == [2, 5] and s.time_step == 1 / 60.0
    with pytest.raises(TypeError):
        el.six_dof(integrator="rk4")


def test_build_validates_rates_like_the_reference():
    w = el.World()
    w.spawn(el.Body())
    with pytest.raises(ValueError, match="simulation_rate must be > 0"):
        w.build(el.six_dof(), simulation_rate=0.0)
    with pytest.raises(ValueError, match="must evenly divide"):
        w.build(el.six_dof(), simulation_rate=120.0, telemetry_rate=50.0)
    with pytest.raises(ValueError, match="unknown backend"):
        w.build(el.six_dof(), backend="cranelift")
    with pytest.raises(KeyError):                      # effector column that was never spawned
        w.build(el.six_dof(sys=el.body_torque("rcs_torque")))


def test_value_size_mismatch_on_spawn():
    w = el.World()
    w.spawn(el.C("wind", [0.0, 0.0, 0.0]))
    with pytest.raises(ValueError, match="value size mismatch"):
        w.spawn(el.C("wind", [0.0, 0.0]))


def test_skew():  # test_all.py:367-378
    import elodin_amd as el
    assert np.isclose(el.skew(np.array([1.0, 2.0, 3.0])), np.array([[0.0, -3.0, 2.0], [3.0, 0.0, -1.0], [-2.0, 1.0, 0.0]])).all()
