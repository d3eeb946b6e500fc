"""The runnable examples (examples/*.py) stay working on the GPU."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EX = Path(__file__).resolve().parents[1] / "examples"


def _load(name):
    spec = importlib.util.spec_from_file_location(name, EX / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_three_body_example():
    from tests import golden_util as gu
    exec = _load("three_body").main(100)
    g = gu.load("three_body")
    assert exec.tick == 100
    for i, e in enumerate("abc"):      # the reference's own regression baseline after 100 ticks
        assert np.allclose(exec.column_array("world_pos")[i], g[f"{e}.world_pos"][100], rtol=1e-9, atol=1e-12)


def test_ball_example_bounces():
    exec, lowest = _load("ball").main(600)           # 5 s: falls 6 m in ~1.1 s, then bounces
    assert -0.3 < lowest < 0.3 and exec.column_array("world_pos")[0, 6] > 0.0


def test_nbody_example_small():
    exec, mom = _load("nbody").main(512, 5)
    assert np.all(np.abs(mom) < 1e-12)


def test_apollo_campaign_example(tmp_path):
    import csv
    import json
    res = _load("apollo_campaign").main(256, tmp_path / "camp")
    assert res.shape == (256, 12) and res[:, 8].mean() == 1.0 and res[:, 9].mean() > 0.6
    rows = list(csv.DictReader(open(tmp_path / "camp" / "results.csv")))
    summary = json.loads((tmp_path / "camp" / "summary.json").read_text())
    assert len(rows) == 256 and summary["passed"] == int(res[:, 9].sum()) == sum(r["passed"] == "true" for r in rows)
    assert abs(float(rows[7]["touchdown_speed_mps"]) - res[7, 0]) < 1e-12


def test_falcon9_ascent_example():
    res = _load("falcon9_ascent").main(128, "f32")
    assert res.shape == (128, 8) and np.all(res[:, 3] > 100.0) and np.all(res[:, 0] > 14_000.0)   # all reach MECO past Max-Q


def test_plain_c_host_over_the_abi(tmp_path):
    """examples/c_host.c: gcc-built host, no Python in the loop; its output must match the reference's golden row."""
    import subprocess
    from tests import golden_util as gu
    root = EX.parent
    exe = tmp_path / "c_host"
    subprocess.run(["gcc", "-O2", f"-I{root / 'include'}", str(EX / "c_host.c"), f"-L{root / 'elodin_amd'}", "-lsixdof_hip",
                    f"-Wl,-rpath,{root / 'elodin_amd'}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), "100"], check=True, capture_output=True, text=True).stdout.splitlines()
    g = gu.load("three_body")
    for line, e in zip(out[:3], "abc"):
        got = np.array([float(x) for x in line.split()[2:]])
        assert np.allclose(got, g[f"{e}.world_pos"][100][4:], rtol=1e-9, atol=1e-12)
    assert out[3].startswith("tick 100, 10 launches")
