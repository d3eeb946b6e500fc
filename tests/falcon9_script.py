"""Open-loop command scripts for the Falcon 9 plant parity tests, written once against an array namespace `xp`.

The SAME function drives (a) the reference's own plant systems executed on numpy by tests/golden/make_falcon9_fixtures.py
(xp = the jax.numpy shim) and (b) this repo's generated program on the GPU / the numpy DAG interpreter (xp = dsl.np),
so both sides see bit-identical commands.  A script replaces the flight software (an external Rust process in the
reference, examples/falcon9/controller): it writes the `external_control` components of sim.py:97-137,285-300.

CASES: initial conditions + campaign knobs of each 10 s window (see the generator for how the reference is spawned).
"""
import math

# name -> window description.  `aloft`: spawn state as (geodetic alt m, speed m/s, pitch-from-vertical deg, lox kg, rp1 kg),
# turned into ECEF numbers by the generator with the reference's frames.py and stored in the fixture;
# engines_running: spool / armed / valve state pre-set as after a nominal ignition (sim.py:372-432 state machine).
CASES = {
    "pad": dict(ticks=10_000, aloft=None, engines_running=False, thrust_scale=1.0, isp_scale=1.0, ca_scale=1.0,
                cn_scale=1.0, wind_ned=(0.0, 0.0, 0.0), upper_kg=0.0),
    "maxq": dict(ticks=10_000, aloft=(11_500.0, 420.0, 35.0, 160_000.0, 68_000.0), engines_running=True,
                 thrust_scale=1.02, isp_scale=0.99, ca_scale=1.1, cn_scale=0.9, wind_ned=(12.0, -7.0, 0.0),
                 upper_kg=116_000.0),
    "coast": dict(ticks=10_000, aloft=(78_000.0, 1_600.0, 62.0, 9_000.0, 4_000.0), engines_running=False,
                  thrust_scale=1.0, isp_scale=1.0, ca_scale=1.0, cn_scale=1.0, wind_ned=(0.0, 0.0, 0.0), upper_kg=0.0),
}
CHECKPOINT_EVERY = 500


def _quat_mul(xp, l, r):   # Hamilton product, scalar last (script detail, identical on both sides)
    li, lj, lk, lw = l[0], l[1], l[2], l[3]
    ri, rj, rk, rw = r[0], r[1], r[2], r[3]
    return xp.array([lw * ri + li * rw + lj * rk - lk * rj, lw * rj - li * rk + lj * rw + lk * ri,
                     lw * rk + li * rj - lj * ri + lk * rw, lw * rw - li * ri - lj * rj - lk * rk])


def make_script(name, base_attitude):
    """-> script(xp, t) -> dict of command columns.  `base_attitude`: the spawn attitude quaternion (4 floats)."""
    base = tuple(float(v) for v in base_attitude)

    def tilt(xp, t, amp, rate):   # spawn attitude tilted about body +Y then body +Z by slowly varying angles
        a, b = amp * xp.sin(rate * t), 0.6 * amp * xp.sin(0.7 * rate * t + 0.4)
        qy = xp.array([0.0, xp.sin(0.5 * a), 0.0, xp.cos(0.5 * a)])
        qz = xp.array([0.0, 0.0, xp.sin(0.5 * b), xp.cos(0.5 * b)])
        return _quat_mul(xp, _quat_mul(xp, xp.array(base), qy), qz)

    if name == "pad":
        def script(xp, t):
            lit = xp.where(t >= 0.2, 1.0, 0.0)
            u = 1.0 - 0.35 * xp.clip((t - 6.0) / 2.0, 0.0, 1.0)                    # throttle down 6..8 s
            outer = xp.where(t < 9.0, 1.0, 0.0)                                    # outer six cut at 9 s (shutdown tau)
            eng = xp.array([u, u, u] + [u * outer] * 6) * lit
            valves = xp.array([1.0, 0.0, 1.0, xp.where(t > 5.0, 1.0, 0.0), 1.0, 1.0, xp.where(t < 3.0, 1.0, 0.0), 0.0])
            return {"engine_cmd": eng, "valve_cmd": valves, "attitude_setpoint": tilt(xp, t, 0.04, 0.9),
                    "ctrl_enable": xp.array([1.0, 1.0]), "fin_cmd": xp.array([0.0, 0.0, 0.0]), "fsw_phase": xp.array([1.0])}
    elif name == "maxq":
        def script(xp, t):
            u = 0.72 + 0.28 * xp.clip((t - 3.0) / 1.5, 0.0, 1.0)                   # bucket, then throttle back up
            valves = xp.array([1.0, 0.0, 1.0, 0.0, 1.0, 1.0, 0.0, 0.0])
            fins = 0.12 * xp.array([xp.sin(1.3 * t), -0.5 * xp.cos(0.9 * t), 0.3 * xp.sin(2.1 * t)])
            return {"engine_cmd": xp.ones(9) * u, "valve_cmd": valves, "attitude_setpoint": tilt(xp, t, 0.08, 0.6),
                    "ctrl_enable": xp.array([1.0, 1.0]), "fin_cmd": fins, "fsw_phase": xp.array([3.0])}
    elif name == "coast":
        def script(xp, t):
            relight = xp.where(t >= 6.0, 0.8, 0.0)                                 # three-engine relight at 6 s
            eng = xp.array([relight, relight, relight] + [0.0] * 6)
            valves = xp.array([1.0, 0.0, 1.0, 0.0, 1.0, 1.0, 1.0, xp.where(t < 2.0, 1.0, 0.0)])
            fins = 0.25 * xp.array([xp.sin(0.8 * t), xp.sin(0.5 * t + 1.0), 0.2 * xp.cos(t)])
            return {"engine_cmd": eng, "valve_cmd": valves, "attitude_setpoint": tilt(xp, t, 0.35, 0.5),
                    "ctrl_enable": xp.array([xp.where(t >= 6.0, 1.0, 0.0), 1.0]), "fin_cmd": fins,
                    "fsw_phase": xp.array([5.0])}
    else:
        raise KeyError(name)
    return script
