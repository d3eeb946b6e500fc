"""The generated Falcon 9 kernel against trajectories flown by the REFERENCE'S OWN plant systems (see
tests/test_falcon9_plant_reference.py for how tests/golden/falcon9_plant.json was made): three whole 10 s windows,
every column at every checkpoint, f64 to 1e-9; and the f32 instantiation config 5 runs, with its stated bound."""
import numpy as np
import pytest

from elodin_amd.models import falcon9 as f9
from tests import falcon9_plant_util as pu

pytestmark = pytest.mark.gpu


def _fly(case, dtype, ticks_per_launch):
    params, cols = pu.initial_columns(case)
    local = np.dtype(dtype) == np.float32
    if local:   # f32 state integrates pad-relative coordinates (an f32 ECEF metre has a 0.5 m ulp)
        cols["world_pos"] = cols["world_pos"].copy()
        cols["world_pos"][:, 4:] -= f9.pad_ecef()
    ex = f9.AscentExec(params, dtype=dtype, local_origin=local, fsw=False, scripted=pu.script(case), columns=cols,
                       ticks_per_launch=ticks_per_launch)
    worst, done = {}, 0
    for cp in pu.PLANT[case]["checkpoints"]:
        ex.run(cp["tick"] - done)
        done = cp["tick"]

        def get(name):
            if name == "world_pos" and local:
                x = np.asarray(ex.column("world_pos"), dtype=np.float64).copy()
                x[:, 4:] += ex.origin
                return x
            return ex.column(name)
        for k, e in pu.compare(case, cp["tick"], get, pu.FLOORS_F32 if local else None).items():
            worst[k] = max(worst.get(k, 0.0), e)
    ex.close()
    return worst


@pytest.mark.parametrize("ticks_per_launch", [1, 250])
@pytest.mark.parametrize("case", sorted(pu.PLANT))
def test_generated_kernel_follows_the_reference_plant_f64(case, ticks_per_launch):
    worst = _fly(case, np.float64, ticks_per_launch)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"{case} f64 K={ticks_per_launch}: worst of {len(worst)} columns over 10,000 ticks:", ", ".join(f"{k} {e:.1e}" for k, e in top))
    assert len(worst) >= 30
    assert max(worst.values()) < 1e-9, top


@pytest.mark.parametrize("case", sorted(pu.PLANT))
def test_generated_kernel_f32_tracks_the_reference_plant(case):
    """Config 5's arithmetic type.  The reference has no f32 six_dof (six_dof.rs:12-14), so the bound is this build's:
    f32 state over 10,000 ticks of a feedback loop stays within 1e-2 of the f64 reference flight on every column
    (relative to the column's scale; floors falcon9_plant_util.FLOORS_F32, i.e. the f64 floors x 1e3 and the
    half-metre resolution of an f32 position for the quantities derived from it)."""
    worst = _fly(case, np.float32, 250)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"{case} f32: worst of {len(worst)} columns over 10,000 ticks:", ", ".join(f"{k} {e:.1e}" for k, e in top))
    assert max(worst.values()) < 1e-2, top
