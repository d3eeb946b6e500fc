"""`python -m elodin_amd.compat script.py`: a script in the reference's spelling, run as it is, end to end on the GPU
(trace -> generate -> hipcc / cache -> step -> history).  The script is this repo's own (tests/scripts/ref_style_probe.py); its
physics has closed forms: velocity under gravity + linear drag relaxes as exp(-k t), a torque-free spin about a principal
axis keeps its rate, and the attitude is the rotation by w t about z."""
import json
import math
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_reference_style_script_runs_unmodified_on_the_gpu():
    ticks = 360
    r = subprocess.run([sys.executable, "-m", "elodin_amd.compat", str(ROOT / "tests" / "scripts" / "ref_style_probe.py"), str(ticks)],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    pos, vel = np.array(out["world_pos"]), np.array(out["world_vel"])
    assert out["history_rows"] == ticks + 1
    # (1) the same script traced in this process and walked with numpy (the reference's integrator restated, tests/np_sixdof.py)
    import importlib.util
    import elodin_amd.compat as compat
    from tests import dsl_numpy
    before = set(sys.modules)
    compat.install(run="record")
    try:
        spec = importlib.util.spec_from_file_location("ref_style_probe", ROOT / "tests" / "scripts" / "ref_style_probe.py")
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        import elodin as el
        plan = m.world().build(m.count_distance | el.six_dof(sys=m.gravity | m.drag), simulation_rate=m.SIM_RATE, _dry=True)
        tp = plan["effectors"].trace()
    finally:
        compat.uninstall()
        for name in set(sys.modules) - before:
            del sys.modules[name]
    body = plan["body"]
    x, v, a, inertia = (np.array(body[k], dtype=np.float64).copy() for k in ("world_pos", "world_vel", "world_accel", "inertia"))
    comps = {n: np.array(plan["columns"][n], dtype=np.float64).reshape(1, -1).copy() for n, _ in tp.columns}
    for tick in range(1, ticks + 1):
        dsl_numpy.program_tick(tp, x, v, a, inertia, comps, tick, plan["dt"], plan["integrator"])
    assert np.allclose(pos, x[0], rtol=1e-11, atol=1e-11) and np.allclose(vel, v[0], rtol=1e-11, atol=1e-11)
    assert abs(out["odometer"] - comps["odometer"][0, 0]) < 1e-9
    # (2) physics: velocity under gravity + linear drag relaxes as exp(-k t); the reference's RK4 stages are not the textbook
    # ones (WorldVel sits in both the state and the derivative group, six_dof.rs:28-130), so closed forms hold to ~1e-7, not 1e-12
    t, k, g = ticks / 120.0, 0.35, 9.80665
    e = math.exp(-k * t)
    assert np.allclose(vel[3:], [12.0 * e, 0.0, -(g / k) * (1.0 - e)], rtol=1e-6, atol=1e-9)
    assert np.allclose(pos[4:], [12.0 / k * (1.0 - e), 0.0, 1000.0 - (g / k) * (t - (1.0 - e) / k)], rtol=1e-6, atol=1e-6)
    assert np.allclose(vel[:3], [0.0, 0.0, 2.0], atol=1e-12)                      # torque-free spin about a principal axis
    assert abs(pos[0]) < 1e-12 and abs(pos[1]) < 1e-12 and abs(math.atan2(pos[2], pos[3]) * 2.0 - 2.0 * t) < 1e-3
