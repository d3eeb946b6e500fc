"""The Falcon 9 plant of elodin_amd/models/falcon9.py against trajectories produced by the REFERENCE'S OWN SYSTEMS.

tests/golden/falcon9_plant.json holds three 10 s windows flown by examples/falcon9/sim.py's @el.map functions themselves
(imported unmodified under tests/golden/refshim.py, chained in build_powered's pipe order around the pinned oracle's
semi-implicit step) under the open-loop command scripts of tests/falcon9_script.py.  Here the model's traced program is
stepped on the CPU (tests/dsl_numpy.program_tick: the DAG codegen.py turns into kernel code) through the first three seconds
of each window; tests/test_gpu_falcon9_plant.py flies the whole windows through the generated kernel.  This is the oracle
that is NOT the product: nothing of the model is on the reference side of the comparison."""
import numpy as np
import pytest

from elodin_amd import _lib as L
from elodin_amd.models import falcon9 as f9
from tests import dsl_numpy, falcon9_plant_util as pu

CPU_TICKS = 3000


@pytest.mark.parametrize("case", sorted(pu.PLANT))
def test_traced_program_follows_the_reference_plant(case):
    params, cols = pu.initial_columns(case)
    program = f9.build_program(fsw=False, scripted=pu.script(case))
    tp = program.trace({k: v.shape[1] for k, v in cols.items()})
    pos, vel, inertia = (cols[k].copy() for k in ("world_pos", "world_vel", "inertia"))
    acc = np.zeros((1, 6))
    comps = {name: cols[name].copy() for name, _ in tp.columns}
    worst = {}
    for tick in range(1, CPU_TICKS + 1):
        dsl_numpy.program_tick(tp, pos, vel, acc, inertia, comps, tick, f9.SIM_TIME_STEP, L.SEMI_IMPLICIT)
        if tick in (1, 2, 10) or tick % 500 == 0:
            body = {"world_pos": pos, "world_vel": vel, "world_accel": acc, "inertia": inertia}

            def get(name):
                if name in body:
                    return body[name]
                return comps[name]
            for k, e in pu.compare(case, tick, get).items():
                worst[k] = max(worst.get(k, 0.0), e)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"{case}: worst over {len(worst)} columns after {CPU_TICKS} ticks:", ", ".join(f"{k} {e:.1e}" for k, e in top))
    assert len(worst) >= 30
    assert max(worst.values()) < 1e-9, top
