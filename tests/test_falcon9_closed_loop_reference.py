"""The Falcon 9 CLOSED LOOP of elodin_amd/models/falcon9.py — plant, sensors, navigator, guidance — against ascents flown by
the reference's own code with an independent restatement of its flight software (tests/falcon9_closed_loop_util.py says what
is on the other side).  Here: the traced program stepped on the CPU (tests/dsl_numpy.program_tick) through the first three
seconds — navigator initialisation on the first GPS fix, the ignition command, liftoff, the radar altimeter feeding the
navigator.  tests/test_gpu_falcon9_closed_loop.py flies the whole ascents through the generated kernel."""
import numpy as np
import pytest

from elodin_amd import _lib as L
from elodin_amd.models import falcon9 as f9
from tests import dsl_numpy, falcon9_closed_loop_util as cu

CPU_TICKS = 3000


@pytest.mark.skipif(not cu.FLIGHTS, reason="closed-loop fixture not generated")
@pytest.mark.parametrize("row", ["0", "1"])
def test_traced_closed_loop_follows_the_reference_flight(row):
    flight = cu.FLIGHTS[row]
    params, cols = cu.initial_columns(flight)
    tp = f9.build_program().trace({k: v.shape[1] for k, v in cols.items()})
    pos, vel, inertia = (cols[k].copy() for k in ("world_pos", "world_vel", "inertia"))
    acc = np.zeros((1, 6))
    comps = {name: cols[name].copy() for name, _ in tp.columns}
    worst, seen = {}, 0
    cps = {c["tick"]: c for c in flight["checkpoints"] if c["tick"] <= CPU_TICKS}
    for tick in range(1, CPU_TICKS + 1):
        dsl_numpy.program_tick(tp, pos, vel, acc, inertia, comps, tick, f9.SIM_TIME_STEP, L.SEMI_IMPLICIT)
        if tick in cps:
            body = {"world_pos": pos, "world_vel": vel, "world_accel": acc, "inertia": inertia}
            for k, e in cu.compare(flight, cps[tick], lambda name: body[name] if name in body else comps[name]).items():
                worst[k] = max(worst.get(k, 0.0), e)
            seen += 1
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print(f"row {row}: worst of {len(worst)} quantities at {seen} checkpoints in {CPU_TICKS} ticks:", ", ".join(f"{k} {e:.1e}" for k, e in top))
    assert seen >= 8 and len(worst) >= 55
    assert max(worst.values()) < 1e-9, top
