"""Known-answer tests lifted from the reference's own unit tests."""
import numpy as np

from oracle import oracle as orc


def test_k1_quat_mult():  # libs/nox/src/quaternion.rs:353-361
    out = orc.quat_mul(orc.quat_from_axis_angle([1, 0, 0], 3.0), orc.quat_from_axis_angle([1, 0, 0], 1.0))
    assert np.array_equal(out, [0.9092974268256817, 0.0, 0.0, -0.4161468365471424])


def test_k2_quat_inverse():  # quaternion.rs:364-370
    out = orc.quat_inverse(orc.quat_from_axis_angle([1, 0, 0], 3.0))
    assert np.array_equal(out, [-0.9974949866040544, -0.0, -0.0, 0.0707372016677029])


def test_k3_quat_vec_mult():  # quaternion.rs:373-381 (eps 1e-6)
    out = orc.quat_rotate(orc.quat_from_axis_angle([1, 0, 0], 3.0), [1.0, 2.0, 3.0])
    assert np.allclose(out, [1.0, -2.4033450173804924, -2.6877374736816018], rtol=1e-6, atol=1e-6)


def test_k4_quat_convention():  # quaternion.rs:384-388: Quaternion::new(w,x,y,z); i*j = k
    out = orc.quat_mul([1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0])
    assert np.array_equal(out, [0.0, 0.0, 1.0, 0.0])


def test_k5_spatial_transform_mul():  # libs/nox/src/spatial.rs:604-628
    a = np.concatenate([orc.quat_from_axis_angle([0, 0, 1], np.radians(45.0)), [1.0, 0.0, 0.0]])
    b = np.concatenate([orc.quat_from_axis_angle([0, 0, 1], -np.radians(45.0)), [0.0, 2.0, 0.0]])
    out = orc.transform_mul(a, b)
    assert np.array_equal(out, [0.0, 0.0, 0.0, 1.0, -0.41421356237309515, 1.414213562373095, 0.0])


def test_k6_spatial_transform_add():  # spatial.rs:631-650
    out = orc.transform_add_motion([0, 0, 0, 1, 0, 0, 0], [0, 0, 1, 0, 0, 0])
    assert np.array_equal(out, [0.0, 0.0, 0.4472135954999579, 0.8944271909999159, 0.0, 0.0, 0.0])


def test_k7_spatial_transform_integrate():  # spatial.rs:653-676 (eps 1e-5)
    x = np.array([0, 0, 0, 1, 0, 0, 0], dtype=float)
    for _ in range(20):
        x = orc.transform_add_motion(x, [0, 0, 0.25 / 20.0, 0, 0, 0])
    assert np.allclose(x, [0, 0, 0.12467473338522769, 0.992197667229329, 0, 0, 0], atol=1e-5)


def _one_body(vel, **kw):
    return orc.OracleWorld([0, 0, 0, 1, 0, 0, 0], vel, [1, 1, 1, 0, 0, 0, 1],
                           simulation_time_step=orc.quantize_time_step(120.0), **kw)


def test_k8_six_dof_one_step():  # libs/nox-py/python/tests/test_all.py:67-83
    w = _one_body([0, 0, 0, 1, 0, 0], time_step=1.0 / 60.0).step(1)
    assert np.allclose(w.world_pos[0, :4], [0, 0, 0, 1])
    assert np.allclose(w.world_pos[0, 4:], [0.01666667, 0, 0])


def test_k8_six_dof_ang_vel_int():  # test_all.py:228-292 ("value from Julia and Simulink", rtol 1e-5)
    cases = [([0, 0, 1], [0.0, 0.0, 0.479425538604203, 0.8775825618903728]),
             ([0, 1, 0], [0.0, 0.479425538604203, 0.0, 0.8775825618903728]),
             ([1, 1, 0], [0.45936268493243, 0.45936268493243, 0.0, 0.76024459707606])]
    for omega, q in cases:
        w = _one_body(omega + [0, 0, 0], time_step=1.0 / 120.0).step(120)
        assert np.isclose(w.world_pos[0], q + [0, 0, 0], rtol=1e-5).all(), (omega, w.world_pos[0])


def test_k8_six_dof_force():  # test_all.py:342-364 ("values taken from simulink")
    w = _one_body([0] * 6, time_step=1.0 / 120.0,
                  ops=[(orc.EFF_CONST_WRENCH, (0, 0, 0, 1, 0, 0), None)]).step(120)
    assert np.isclose(w.world_pos[0], [0, 0, 0, 1, 0.5, 0, 0], rtol=1e-5).all()


def test_k8_spatial_integration_body_frame():  # test_all.py:86-114 (integrate_body, right-multiply)
    q = np.array([0, 0, 0, 1.0])
    for _ in range(2):
        q = orc.quat_integrate_body(q, [np.pi / 2, 0, 0])
    assert np.allclose(q, [0.97151626, 0.0, 0.0, 0.23697292])


def test_component_ids_and_output_slot_order():  # SURVEY App. B, impeller2/src/types.rs:39-44
    ids = {n: orc.component_id(n) for n in
           ("world_accel", "simulation_time_step", "tick", "world_vel", "world_pos", "inertia", "force")}
    assert ids["world_accel"] == 0x019091805BC057F4
    assert ids["force"] == 0x675AD8AFB3EEEBE4
    assert sorted(ids, key=ids.get) == ["world_accel", "simulation_time_step", "tick", "world_vel",
                                         "world_pos", "inertia", "force"]


def test_dt_quantisation():  # world_builder.rs:221: 120 Hz -> 0.008333333 as in the golden globals column
    assert orc.quantize_time_step(120.0) == 0.008333333
    assert orc.quantize_time_step(1000.0) == 0.001
    assert orc.quantize_time_step(1.0 / 3600.0) == 3600.0


def test_semi_implicit_restatement():  # integrator/semi_implicit.rs:17-31; unpinned by golden data
    w = _one_body([0] * 6, time_step=0.5, integrator=orc.SEMI_IMPLICIT,
                  ops=[(orc.EFF_CONST_WRENCH, (0, 0, 0, 2, 0, 0), None)]).step(2)
    # v1 = 1, x1 = .5 ; v2 = 2, x2 = 1.5
    assert np.array_equal(w.world_vel[0, 3:], [2, 0, 0]) and np.array_equal(w.world_pos[0, 4:], [1.5, 0, 0])
    assert np.array_equal(w.world_accel[0, 3:], [2, 0, 0])
