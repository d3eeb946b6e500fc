"""Falcon 9 ascent model (elodin_amd/models/falcon9.py, BASELINE config 5): the physics helpers against the known
answers the reference's own example tests assert (examples/falcon9/test_ladder.py, test_frames.py, test_propulsion.py,
test_aero.py), evaluated twice — the numpy instantiation and the traced DAG that becomes kernel code — plus the plan
sampler and the trace / code generation of the whole closed loop."""
import math

import numpy as np
import pytest

from elodin_amd import codegen, dsl
from elodin_amd.models import falcon9 as f9
from tests import dsl_numpy

PAD_LAT, PAD_LON = math.radians(f9.PAD_LAT_DEG), math.radians(f9.PAD_LON_DEG)


def both(fn, *args):
    """(numpy result, traced-DAG result), asserted equal to 1e-12 relative."""
    a = fn(np, *[np.asarray(x, dtype=np.float64) if np.ndim(x) else float(x) for x in args])
    b = dsl_numpy.trace_eval(fn, *args)

    def flat(x):
        return np.concatenate([np.ravel(np.asarray(v, dtype=np.float64)) for v in (x if isinstance(x, tuple) else (x,))])
    fa, fb = flat(a), flat(b)
    assert np.allclose(fa, fb, rtol=1e-12, atol=1e-9 * max(1.0, float(np.max(np.abs(fa))))), (fa, fb)
    return a


def test_us76_anchors():
    """test_ladder.py:43-53."""
    assert abs(both(f9.density, 0.0) - 1.2250) < 1e-3
    p11, t11 = both(f9.pressure_temperature_at_geopotential, 11_000.0)
    assert abs(p11 - 22_632.0) < 5.0 and abs(t11 - 216.65) < 1e-9
    assert abs(p11 / (f9.R_AIR * t11) - 0.3639) < 1e-3
    assert abs(both(f9.speed_of_sound, 0.0) - 340.29) < 0.1
    assert both(f9.density, 100_000.0) < 1e-5
    for h in (500.0, 15_000.0, 25_000.0, 40_000.0, 49_000.0, 60_000.0, 80_000.0, 90_000.0):   # every table layer
        both(f9.pressure_temperature, h)


def test_pad_and_landing_zone_ecef():
    """test_frames.py:35-42 (WHITEPAPER section 4 worked example)."""
    pad = both(f9.geodetic_to_ecef, PAD_LAT, PAD_LON, f9.PAD_ALT_M)
    assert np.allclose(np.asarray(pad) / 1000.0, [914.8, -5528.6, 3035.9], atol=0.1)
    lz1 = f9.geodetic_to_ecef(np, math.radians(28.48580), math.radians(-80.54440), 5.0)
    assert np.allclose(np.asarray(lz1) / 1000.0, [921.7, -5534.0, 3023.9], atol=0.1)
    assert abs(np.linalg.norm(pad - lz1) / 1000.0 - 14.8) < 0.1


def test_geodetic_roundtrip():
    """test_frames.py:45-53: 1e-9 deg, 1e-6 m over latitudes, longitudes and altitudes up to 200 km."""
    for lat_deg in (-75.0, -28.0, 0.0, 28.60839, 45.0, 89.0):
        for lon_deg in (-170.0, -80.60433, 0.0, 91.0):
            for alt in (0.0, 3.0, 8_700.0, 118_000.0, 200_000.0):
                r = f9.geodetic_to_ecef(np, math.radians(lat_deg), math.radians(lon_deg), alt)
                lat, lon, h = both(f9.ecef_to_geodetic, r) if (lat_deg, lon_deg) == (28.60839, -80.60433) else f9.ecef_to_geodetic(np, r)
                assert abs(math.degrees(lat) - lat_deg) < 1e-9 and abs(math.degrees(lon) - lon_deg) < 1e-9
                assert abs(h - alt) < 1e-6


def test_angle_free_geodetic_conversion_is_the_same_function():
    """geodetic_sincos (the f32 campaign builds' conversion) against ecef_to_geodetic over the test_frames.py:45-53 grid and
    along an ascent: sines / cosines to 1e-13, altitude to 1e-8 m — the recurrence is the reference's, carried on tangents."""
    pts = [f9.geodetic_to_ecef(np, math.radians(la), math.radians(lo), h)
           for la in (-75.0, -28.0, 0.0, 28.60839, 45.0, 89.0) for lo in (-170.0, -80.60433, 0.0, 91.0)
           for h in (0.0, 3.0, 8_700.0, 118_000.0, 200_000.0)]
    for k, r in enumerate(pts):
        lat, lon, h = f9.ecef_to_geodetic(np, r)
        got = both(f9.geodetic_sincos, r) if k == 47 else f9.geodetic_sincos(np, r)
        assert np.allclose(got[:4], [math.sin(lat), math.cos(lat), math.sin(lon), math.cos(lon)], rtol=0, atol=1e-13)
        assert abs(got[4] - h) < 1e-8
    n, e, d = f9.ned_rows(np, *f9.geodetic_sincos(np, pts[47])[:4])
    assert all(np.allclose(a, b, atol=1e-13) for a, b in zip((n, e, d), f9.ned_basis(np, *f9.ecef_to_geodetic(np, pts[47])[:2])))


def test_two_passes_of_the_geodetic_recurrence_are_converged():
    """The reference makes 4 fixed Bowring passes (frames.py:55-59); geodetic_sincos makes 2.  Over 200,000 random points
    (every latitude short of the poles, -100 m .. 400 km) in float64 the 2-pass result is the 4-pass one to 3e-16 in the sines /
    cosines and to the altitude's own cancellation noise (|p cos + z sin - a w| at 6.4e6 m: a few 1e-9 m, the same between 3 and
    4 passes); in float32 two passes are as close to four as three are (the rounding noise of a 6.4e6 m coordinate)."""
    rng = np.random.default_rng(0)
    n = 200_000
    lat, lon = np.radians(rng.uniform(-89.0, 89.0, n)), np.radians(rng.uniform(-180.0, 180.0, n))
    r = np.stack(f9.geodetic_to_ecef(np, lat, lon, rng.uniform(-100.0, 400e3, n)))
    assert f9.GEODETIC_SINCOS_PASSES == 2
    four, two, three = (f9.geodetic_sincos(np, r, passes=k) for k in (4, 2, 3))
    assert max(np.abs(a - b).max() for a, b in zip(two[:4], four[:4])) < 3e-16
    assert np.abs(two[4] - four[4]).max() < 5e-9 and np.abs(three[4] - four[4]).max() > 1e-9      # noise floor, not convergence
    r32 = r.astype(np.float32)
    f = {k: f9.geodetic_sincos(np, r32, passes=k) for k in (2, 3, 4)}
    assert all(v[4].dtype == np.float32 for v in f.values())
    # float32 holds an ECEF coordinate to 0.5 m: two passes are as close to four as three are (rounding noise, 2-3 ulp)
    assert np.abs(f[2][4] - f[4][4]).max() <= np.abs(f[3][4] - f[4][4]).max() <= 1.5
    assert max(np.abs(f[2][j] - f[4][j]).max() for j in range(4)) <= 1.2e-7


def test_ned_basis_and_ellipsoid_normal():
    """test_frames.py:56-66."""
    n, e, d = both(f9.ned_basis, PAD_LAT, PAD_LON)
    R = np.stack([n, e, d])
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.linalg.det(R) > 0.99
    up = -d
    base = f9.geodetic_to_ecef(np, PAD_LAT, PAD_LON, f9.PAD_ALT_M)
    assert np.allclose(f9.geodetic_to_ecef(np, PAD_LAT, PAD_LON, f9.PAD_ALT_M + 1.0) - base, up, atol=1e-9)
    assert abs(f9.ecef_to_geodetic(np, base + up * 100.0)[2] - (f9.PAD_ALT_M + 100.0)) < 1e-6


def test_rotating_frame_magnitudes_and_plumb_line():
    """test_frames.py:69-97."""
    pad = f9.pad_ecef()
    om = np.array([0.0, 0.0, f9.OMEGA_EARTH_RADPS])
    assert abs(np.linalg.norm(np.cross(om, pad)) - 408.6) < 0.5
    zero = np.zeros(3)
    assert abs(np.linalg.norm(both(f9.frame_accel, pad, zero)) - 0.0298) < 3e-4           # centrifugal at the pad
    assert abs(np.linalg.norm(both(f9.frame_accel, zero, [1656.0, 0.0, 0.0])) - 0.2415) < 1e-3   # Coriolis at MECO speed
    g_pad = np.linalg.norm(both(f9.gravity_accel, pad))
    assert abs(g_pad - 9.813) < 2e-3
    apo = pad * (1.0 + 118_000.0 / np.linalg.norm(pad))
    assert abs(np.linalg.norm(f9.gravity_accel(np, apo)) / g_pad - 0.964) < 1e-3
    g_app = both(f9.apparent_gravity, pad)
    misalign = math.degrees(math.acos(float(np.clip(-(g_app / np.linalg.norm(g_app)) @ f9.pad_up(), -1.0, 1.0))))
    assert misalign < 0.2 and abs(np.linalg.norm(g_app) - 9.79) < 0.02


def test_propulsion_anchors():
    """test_propulsion.py:45-71."""
    t_sl = both(f9.engine_thrust_per_engine, 1.0, f9.P_SL_PA)
    t_vac = both(f9.engine_thrust_per_engine, 1.0, 0.0)
    assert abs(t_sl - f9.ENGINE_T_SL_N) < 1.0 and abs(t_vac - f9.ENGINE_T_VAC_N) < 1.0
    assert abs((t_vac - t_sl) / f9.P_SL_PA - 0.681) < 1e-3
    mdot = both(f9.cluster_mdot, 1.0, 1.0)
    assert abs(mdot - f9.ENGINE_T_SL_N / (f9.ENGINE_ISP_SL_S * f9.G0)) < 0.5 and abs(mdot - 275.0) < 3.0
    total = 9 * mdot * 147.0 - 9 * mdot * 39.0 * 0.3 + 3 * mdot * 46.0 + 3 * mdot * 14.0 + mdot * 0.7 * 33.0
    assert 0.93 * f9.STAGE1_PROP_KG < total < 1.01 * f9.STAGE1_PROP_KG
    assert 1.2 < 9 * t_sl / (f9.LIFTOFF_MASS_KG * f9.G0) < 1.4


def test_actuator_exact_discretization():
    """test_propulsion.py:74-88."""
    x = 0.0
    for _ in range(7):
        x = both(lambda xp, v: f9.actuator_step(xp, v, 1.0, 0.001, 0.007), x)
    assert abs(x - (1.0 - math.exp(-1.0))) < 1e-9
    big = f9.actuator_step(np, 0.0, 1.0, 0.7, 0.007)
    assert 0.0 < big <= 1.0 and abs(big - 1.0) < 1e-9
    assert abs(both(lambda xp, v: f9.actuator_step(xp, v, 1.0, 0.001, 1e-6, rate_limit=10.0), 0.0) - 0.01) < 1e-12


def test_stack_mass_props():
    """test_propulsion.py:91-108."""
    mass, cg, inertia = both(f9.stack_mass_props, f9.LOX_LOAD_KG, f9.RP1_LOAD_KG)
    assert abs(mass - (f9.STAGE1_DRY_MASS_KG + f9.STAGE1_PROP_KG)) < 1.0 and 0.0 < cg < f9.STAGE1_LENGTH_M
    assert np.all(inertia > 0.0) and inertia[1] > 10.0 * inertia[0]
    cgs = [f9.stack_mass_props(np, f9.LOX_LOAD_KG * f, f9.RP1_LOAD_KG * f)[1] for f in (1.0, 0.6, 0.3)]
    assert cgs[0] > cgs[1] > cgs[2]
    cg_dry = f9.stack_mass_props(np, 0.0, 0.0)[1]
    assert abs(cg_dry - f9.DRY_CG_STATION_M) < 1e-6 and cgs[2] < cg_dry < cgs[0]


def test_aero_canonical_directions_damping_and_plume():
    """test_aero.py:77-100,122-142."""
    cg, qbar = 22.5, 20_000.0
    f, t = both(lambda xp, v: f9.body_aero_wrench(xp, v, 1.5, qbar, cg), [500.0, 0.0, 0.0])
    ca_ascent = np.interp(1.5, f9.MACH_PTS, f9.CA_ASCENT)
    assert f[0] < 0 and abs(f[1]) < 1e-9 and abs(f[2]) < 1e-9 and np.all(np.abs(t) < 1e-9)
    assert abs(f[0] + qbar * f9.S_REF_M2 * ca_ascent) < 1e-2
    f, _ = both(lambda xp, v: f9.body_aero_wrench(xp, v, 1.5, qbar, cg), [-500.0, 0.0, 0.0])
    assert f[0] > 0 and abs(f[0] - qbar * f9.S_REF_M2 * np.interp(1.5, f9.MACH_PTS, f9.CA_DESCENT)) < 1e-2
    v = [-400.0, 0.0, 0.0]
    _, t0 = both(lambda xp, a, w: f9.body_aero_wrench(xp, a, 1.5, 40_000.0, cg, omega_body=w), v, [0.0, 0.0, 0.0])
    _, t1 = both(lambda xp, a, w: f9.body_aero_wrench(xp, a, 1.5, 40_000.0, cg, omega_body=w), v, [0.0, 0.5, 0.0])
    assert t1[1] < t0[1] - 1e3
    assert both(f9.plume_dominance, 0.0, 30_000.0) == 0.0
    assert 0.85 < both(f9.plume_dominance, 2.3e6, 30_000.0) < 0.95
    assert both(f9.plume_dominance, 5.0e5, 40_000.0) > 0.5


def test_quat_between_x_matches_axis_angle():
    """math.rs:122-138 for from = +X, including both degenerate branches; upright_attitude (sim.py:1204-1211)."""
    for to in ([0.0, 1.0, 0.0], [0.6, 0.0, 0.8], list(f9.pad_up()), [1.0, 0.0, 0.0], [-1.0, 0.0, 0.0]):
        q = both(f9.quat_between_x, to)
        assert abs(np.linalg.norm(q) - 1.0) < 1e-12
        assert np.allclose(f9.quat_rotate(np, q, np.array([1.0, 0.0, 0.0])), to, atol=1e-12)
    assert np.allclose(f9.quat_between_x(np, f9.pad_up()), f9.upright_attitude(), atol=1e-15)
    q = f9.upright_attitude()
    assert np.allclose(f9.quat_mul(np, q, f9.quat_inverse(np, q)), [0, 0, 0, 1], atol=1e-15)


def test_spec_plan_is_the_reference_sampler_on_the_full_variable_table():
    a, b = f9.sample_params(24), f9.sample_params(24)
    assert a.shape == (24, 16) and np.array_equal(a, b)
    for name, (lo, hi) in f9.SPEC_RANGES.items():
        if name in f9.P:
            col = a[:, f9.P[name]]
            assert lo <= col.min() and col.max() <= hi
            # Latin hypercube: exactly one sample per stratum
            assert sorted(np.floor((col - lo) / (hi - lo) * 24).astype(int)) == list(range(24))


def test_closed_loop_traces_and_generates_for_both_dtypes():
    prog = f9.build_program(origin=f9.pad_ecef())
    cols = f9.initial_columns(f9.default_param_row()[None, :], origin=f9.pad_ecef())
    tp = prog.trace({k: v.shape[1] for k, v in cols.items()})
    names = dict(tp.columns)
    assert names["engine_spool"] == 9 and names["valve_state"] == 8 and names["params"] == 16 and len(names) <= 64
    assert tp.writes_inertia and tp.reads_velocity
    for dtype, fast in (("float64", False), ("float32", False), ("float32", True)):
        src = codegen.generate_source(tp, dtype, 1, fast_math=fast)
        assert "m_pow(" in src and "m_interp<" in src and "fsw_ascent" in src and ("#define SIXDOF_FAST_MATH" in src) == fast
    with pytest.raises(ValueError):
        codegen.generate_source(tp, "float64", 1, fast_math=True)
    mass = cols["inertia"][0, 6]
    assert abs(mass - (f9.STAGE1_DRY_MASS_KG + f9.DEFAULT_PARAMS["lox_kg"] + f9.DEFAULT_PARAMS["rp1_kg"] + f9.UPPER_KG)) < 1e-6


def test_rcs_allocation_and_fin_mixing():
    """test_aero.py:24-74: axis purity / authority of the cold-gas allocation, force-free roll, fin mixing axis purity."""
    cg = 22.0
    b = np.stack(f9.rcs_torque_authority(np, cg), axis=1)                      # 3 x 8 torque rows of the effectiveness matrix
    authority = [np.abs(b[axis]).sum() / 2.0 for axis in range(3)]
    for axis in range(3):
        for sign in (1.0, -1.0):
            cmd = [0.0, 0.0, 0.0]
            cmd[axis] = sign * 0.5 * authority[axis]
            levels = both(lambda xp, c: f9.allocate_torque(xp, c, cg), cmd)
            _, torque = both(lambda xp, lv: f9.rcs_wrench(xp, lv, cg), list(levels))
            assert abs(torque[axis] - cmd[axis]) < 1e-6 * abs(cmd[axis]) + 1e-9
            assert np.all(np.abs(np.delete(torque, axis)) < 1e-6 * abs(cmd[axis]) + 1e-6)
    for tx in (1e4, -1e4):
        force, _ = f9.rcs_wrench(np, f9.allocate_torque(np, np.array([tx, 0.0, 0.0]), cg), cg)
        assert np.all(np.abs(force) < 1e-9)
    for axis, cmd in ((1, [0.1, 0.0, 0.0]), (2, [0.0, 0.1, 0.0]), (0, [0.0, 0.0, 0.1])):     # pitch -> My, yaw -> Mz, roll -> Mx
        deltas = both(f9.fin_mix, cmd)
        _, torque = both(lambda xp, d: f9.fin_wrench(xp, d, 2.0, 30_000.0, 20.0), list(deltas))
        assert int(np.argmax(np.abs(torque))) == axis
        assert np.all(np.abs(np.delete(torque, axis)) < 1e-9 * max(1.0, abs(torque[axis])))
    force, _ = f9.fin_wrench(np, f9.fin_mix(np, np.array([0.0, 0.0, 0.2])), 2.0, 30_000.0, 20.0)
    assert np.all(np.abs(force) < 1e-9)


def test_tracing_the_campaign_program_twice_in_one_process_gives_one_text():
    """bench.py builds the campaign executor twice in one process (an untimed warm-up, then the campaign): the second trace must
    generate the SAME source — loop-carried leaves are named after a counter that restarts with every top-level trace — or it
    misses the JIT cache and the timed campaign pays a 19 s hipcc run (measured on the GPU box in round 4: 6.3 s of 7.6)."""
    from elodin_amd import codegen
    cols = f9.initial_columns(f9.default_param_row()[None, :], origin=f9.pad_ecef())
    widths = {k: v.shape[1] for k, v in cols.items() if k not in ("world_pos", "world_vel", "inertia")}
    texts = [codegen.generate_source(f9.build_program(origin=f9.pad_ecef()).trace(widths), "float32", 1, fast_math=True) for _ in range(2)]
    assert texts[0] == texts[1] and "for (int it_" in texts[0]        # it does hold loops (branch_cond)
