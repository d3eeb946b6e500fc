"""examples/monte-carlo of the reference on the GPU as one executor (elodin_amd/vectorize.py): the respelled script
(examples/monte_carlo_sitl.py — byte for byte the program the unmodified sim.py generates, tests/test_monte_carlo_example.py),
its drag table gathered from device memory, main.py's own post_step called per run on the server loop's cadence through a
StepContext; against runs flown by the reference's sim.py / main.py (tests/golden/monte_carlo_example.json)."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

from elodin_amd import vectorize

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"


def example(probe_rows, grid=4096):
    os.environ["ELODIN_MONTE_CARLO_GRID_SIZE"] = str(grid)
    os.environ["ELODIN_MONTE_CARLO_PROBE_ROWS"] = str(probe_rows)
    sys.path.insert(0, str(ROOT))
    from examples import monte_carlo_sitl as ex
    return ex


@pytest.mark.parametrize("probe_rows", [0, 64])
def test_campaign_with_the_scripts_own_post_step_lands_on_the_reference_runs(probe_rows):
    ex = example(probe_rows)
    doc = json.loads((GOLDEN / "monte_carlo_example.json").read_text())
    runs = [r for r in doc["runs"] if r["probe_rows"] == probe_rows]
    reps = 15                                                   # 120-135 rows: full waves and a ragged one
    c = vectorize.Campaign(ex.build, vectorize.plan_of([r["params"] for r in runs] * reps), ex.PARAMS, simulation_rate=ex.SIMULATION_RATE_HZ)
    worst, seen = [0.0], [0]

    def post_step(tick, ctx):
        ex.post_step(tick, ctx)                                 # main.py:88-106: reads, the PD law, the write, the result record
        for row in (x for x in runs[ctx.run_index % len(runs)]["rows"] if x[0] == tick + 1):
            got = [float(ctx.read_component("vehicle." + cn)[0]) for cn in ("position", "velocity", "command", "specific_force")]
            worst[0] = max(worst[0], max(abs(g - w) / max(1.0, abs(w)) for g, w in zip(got, row[1:])))
            seen[0] += 1

    c.run(doc["max_ticks"], post_step=post_step)
    assert seen[0] == len(runs) * reps * len(runs[0]["rows"])
    assert worst[0] < 1e-11, worst[0]
    res = c.result_table(["final_position", "target", "error"])
    for k in range(len(runs) * reps):
        want = runs[k % len(runs)]["result"]
        assert np.allclose(res[k], [want["final_position"], want["target"], want["error"]], rtol=1e-11, atol=1e-11), (k, res[k], want)
    print(f"monte-carlo example, {len(runs) * reps} runs x {doc['max_ticks']} ticks, probe rows {probe_rows}: worst {worst[0]:.1e} vs the reference's runs")


def test_unmodified_style_world_run_with_step_callbacks():
    """compat's World.run(post_step=..., pre_step=...) — the reference's signature and server-loop cadence — on one run."""
    ex = example(0)
    doc = json.loads((GOLDEN / "monte_carlo_example.json").read_text())
    run = next(r for r in doc["runs"] if r["run_id"] == "defaults")
    from elodin_amd import compat, monte_carlo as mc
    world, system = ex.build(mc.Params(dict(run["params"])))
    w2 = compat._make_elodin().World()
    for eid, comps in vectorize._entities(world).items():
        w2.spawn([vectorize._api.C(c_, row) for c_, row in comps.items()], name=world._names[eid])
    pre_ticks = []
    rec = {}
    saved, mode = mc._active_result[0], compat._RUN_MODE[0]
    mc._active_result[0], compat._RUN_MODE[0] = rec, "execute"          # (another test may have left the shim in record mode)
    try:
        w2.run(system, ex.SIMULATION_RATE_HZ, False, None, 1.0, doc["max_ticks"], pre_step=lambda t, ctx: pre_ticks.append((t, ctx.tick)),
               post_step=ex.post_step, interactive=False)
        sink = w2.compat_exec.compat_sink
        with pytest.raises(ValueError, match="max_ticks"):      # no editor to stop the loop: callbacks need a tick budget
            w2.run(system, ex.SIMULATION_RATE_HZ, post_step=ex.post_step)
    finally:
        mc._active_result[0], compat._RUN_MODE[0] = saved, mode
    assert pre_ticks[:3] == [(0, 0), (1, 1), (2, 2)] and len(pre_ticks) == doc["max_ticks"]
    assert abs(rec["final_position"] - run["result"]["final_position"]) < 1e-11 and abs(rec["error"] - run["result"]["error"]) < 1e-11
    # the run went through the commit path's hand-off (elodin_amd.telemetry.Sink): one sample of every pair per tick + the spawned
    # state, stamped start + tick / 120 Hz; `command` is an external control (its component says so): only the callback's writes
    ts, pos = sink.series("vehicle.position")
    assert len(ts) == doc["max_ticks"] + 1 and ts[0] == 0 and ts[-1] == int(round((doc["max_ticks"] - 1) / 120.0 * 1e6))
    assert abs(pos[-1, 0] - run["result"]["final_position"]) < 1e-11
    assert sink.sample_count("vehicle.command") == doc["max_ticks"] + 1 and sink.sample_count("vehicle.target") == doc["max_ticks"] + 1
