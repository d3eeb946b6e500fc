"""How the drone example's components are compared with the reference's CI baseline (tests/golden/drone.json <-
scripts/ci/baseline/drone-csv): shared by the CPU walk of the unmodified script (tests/test_compat_reference_scripts.py) and the
GPU run of its generated kernel (tests/test_gpu_drone.py)."""
import numpy as np

# What a drone component is compared against: its own largest baseline value in the row, but not less than 1e-3 rad/s (rad) for
# the angular quantities, which start at 1e-11 on the first ticks.
DRONE_FLOORS = {"ang_vel_setpoint": 1e-3, "body_ang_vel": 1e-3, "gyro": 1e-3, "rate_pid_state": 1e-3, "gyro_lpf_delay": 1e-3,
                "torque": 1e-3, "euler_rate_target": 1e-3, "angle_desired": 1e-3, "attitude_estimate_error": 1e-3}


# Through tick 6 every component agrees to 1e-17; on ticks 7-9 the attitude error is an angle of ~4e-6 rad taken through
# arccos of a quaternion component 1e-12 below 1 (examples/drone/control.py), where one ulp of the argument is 1e-10 rad of the
# result: ang_vel_setpoint differs by 1.1e-10 rad/s there, the rate PID's derivative term carries it to 6e-9 of its state, and
# the closed loop damps it again (6.6e-12 rad/s by tick 100).  Those five columns are held to 5e-7 of their scale, everything
# else to 2e-9; the reference's own CI accepts 1e-4 on this baseline.
DRONE_RATE_CHAIN = ("ang_vel_setpoint", "body_ang_vel", "rate_pid_state", "gyro", "gyro_lpf_delay")


def drone_verdict(worst, rate_chain=5e-7, rest=2e-9):
    assert len(worst) >= 27, sorted(worst)
    bad = {k: v for k, v in worst.items() if not v < (rate_chain if k in DRONE_RATE_CHAIN else rest)}
    assert not bad, bad


def drone_errors(gold, row, cur, worst):
    for name, rows in gold["rows"].items():
        if name not in cur:
            continue
        ref, got = np.asarray(rows[row], dtype=np.float64).reshape(-1), np.asarray(cur[name], dtype=np.float64).reshape(-1)
        scale = max(float(np.max(np.abs(ref))), DRONE_FLOORS.get(name, 1e-9))
        worst[name] = max(worst.get(name, 0.0), float(np.max(np.abs(got - ref))) / scale)
