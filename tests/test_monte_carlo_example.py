"""examples/monte-carlo of the reference (a point-mass plant that gathers its drag coefficient from a large lookup table, a
post_step controller, 100 runs) as ONE executor: elodin_amd/vectorize.py + the traced gather (dsl.gather).

CPU side: the unmodified script under elodin_amd.compat generates byte for byte the program of the respelling the GPU box runs
(examples/monte_carlo_sitl.py); the numpy walk of that program, with main.py's control law, lands on runs flown by the
reference's own sim.py / main.py (tests/golden/monte_carlo_example.json <- make_monte_carlo_fixture.py); parameters that reach
the code through host arithmetic are refused."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

from elodin_amd import monte_carlo as mc
from elodin_amd import vectorize
from tests import dsl_numpy

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"
REF = Path("/root/reference/examples/monte-carlo")


def example(probe_rows=0, grid=4096):
    os.environ["ELODIN_MONTE_CARLO_GRID_SIZE"] = str(grid)
    os.environ["ELODIN_MONTE_CARLO_PROBE_ROWS"] = str(probe_rows)
    sys.path.insert(0, str(ROOT))
    from examples import monte_carlo_sitl as ex
    return ex


def fixture_runs(probe_rows):
    doc = json.loads((GOLDEN / "monte_carlo_example.json").read_text())
    return doc, [r for r in doc["runs"] if r["probe_rows"] == probe_rows]


@pytest.mark.parametrize("probe_rows", [0, 64])
def test_numpy_walk_of_the_campaign_program_lands_on_the_reference_runs(probe_rows):
    ex = example(probe_rows)
    doc, runs = fixture_runs(probe_rows)
    c = vectorize.Campaign(ex.build, vectorize.plan_of([r["params"] for r in runs]), ex.PARAMS, simulation_rate=ex.SIMULATION_RATE_HZ, dry=True)
    tp, comps, dt = c.traced()
    assert {"mc:mass", "mc:thrust_gain"} <= {n for n, _ in tp.columns}            # the baked parameters became per-run columns
    n = len(runs)
    pos, vel, acc, inertia = np.zeros((n, 7)), np.zeros((n, 6)), np.zeros((n, 6)), np.ones((n, 7))
    target = c.world.column("target")[0][:, 0]
    worst = 0.0
    for tick in range(doc["max_ticks"]):
        dsl_numpy.program_tick_systems_only(tp, pos, vel, acc, inertia, comps, tick + 1)
        p, v = comps["position"][:, 0], comps["velocity"][:, 0]
        comps["command"][:, 0] = np.clip((target - p) * 1.2 - v * 0.35, -20.0, 20.0)         # main.py:98 (controller off)
        for k, r in enumerate(runs):
            for row in (x for x in r["rows"] if x[0] == tick + 1):
                got = [comps[cn][k, 0] for cn in ("position", "velocity", "command", "specific_force")]
                worst = max(worst, max(abs(g - w) / max(1.0, abs(w)) for g, w in zip(got, row[1:])))
    assert worst < 1e-12, worst
    for k, r in enumerate(runs):
        assert abs(comps["position"][k, 0] - r["result"]["final_position"]) < 1e-12 * max(1.0, abs(r["result"]["final_position"]))


@pytest.mark.skipif(not REF.exists(), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize("probe_rows", [0, 64])
def test_unmodified_script_generates_the_program_of_the_respelling(probe_rows):
    ex = example(probe_rows)
    plan = vectorize.plan_of([{"mass": 1.2, "target_x": 25.0, "thrust_gain": 0.9, "wind": 0.3}, {"mass": 1.9, "target_x": 35.0, "thrust_gain": 1.1, "wind": -0.2}])
    mine = vectorize.Campaign(ex.build, plan, ex.PARAMS, simulation_rate=120.0, dry=True)
    from elodin_amd import compat
    compat.install(run="record")
    try:
        sys.path.insert(0, str(REF))
        sys.modules.pop("sim", None)
        import sim
        theirs = vectorize.Campaign(sim.build, plan, sim.PARAMS, simulation_rate=sim.SIMULATION_RATE_HZ, dry=True)
    finally:
        sys.path.remove(str(REF))
        sys.modules.pop("sim", None)
        compat.uninstall()
    assert mine.sources == theirs.sources
    for name in ("position", "velocity", "target", "mc:mass", "mc:thrust_gain"):
        assert np.array_equal(mine.world.column(name)[0], theirs.world.column(name)[0]), name
    src = next(iter(mine.sources.values()))
    # the drag coefficient + every probe row: loads, not select chains — once in the tick, and once more where the
    # specific_force column (which nothing reads) is evaluated for storing
    assert src.count("m_gather<T>(gtab0") == 2 * (1 + probe_rows) and src.count("// store-only columns of") == 1


def test_a_parameter_that_reaches_the_code_through_host_arithmetic_is_refused():
    import elodin_amd.frontend as el
    ex = example(0)

    def build(params):
        world, _ = ex.build(params)
        k = 2.0 * float(params.get("mass", 1.5))               # derived on the host: the tracer only ever sees the product

        @el.map
        def plant(pos: ex.Position, vel: ex.Velocity) -> ex.Position:
            return pos + vel * (1.0 / k)
        return world, plant
    with pytest.raises(NotImplementedError, match="host arithmetic"):
        vectorize.Campaign(build, vectorize.plan_of([{"mass": 1.0}, {"mass": 2.0}]), ex.PARAMS, dry=True)


def test_spawned_worlds_come_from_the_plan_table_or_from_one_build_per_run():
    """A spawned element that IS a parameter is filled from the plan row; one derived from a parameter on the host makes the
    campaign call build(params_i) for every run — same world either way."""
    import elodin_amd.frontend as el
    ex = example(0)
    rows = [{"mass": 1.0 + 0.1 * k, "target_x": 20.0 + k, "thrust_gain": 1.0, "wind": 0.1 * k - 0.2} for k in range(5)]
    direct = vectorize.Campaign(ex.build, vectorize.plan_of(rows), ex.PARAMS, dry=True)
    assert not direct.per_run_builds

    def build_derived(params):
        world, plant = ex.build(params)
        world.spawn([el.C(ex.Target, np.array([2.0 * float(params.get("target_x", 30.0))]))], name="beacon")     # host arithmetic on a parameter
        return world, plant
    derived = vectorize.Campaign(build_derived, vectorize.plan_of(rows), ex.PARAMS, dry=True)
    assert derived.per_run_builds and derived.entities_per_run == 2
    t = derived.world.column("target")[0][:, 0]
    assert np.array_equal(t[0::2], [r["target_x"] for r in rows]) and np.array_equal(t[1::2], [2.0 * r["target_x"] for r in rows])
    assert np.array_equal(direct.world.column("velocity")[0][:, 0], [r["wind"] for r in rows])
    assert np.array_equal(direct.world.column("mc:mass")[0][:, 0], [r["mass"] for r in rows])
    assert direct.entity_names[3] == {"vehicle": 4} and derived.entity_names[2] == {"vehicle": 5, "beacon": 6}


def test_host_control_flow_on_a_parameter_is_refused_even_when_both_sentinels_take_the_same_branch():
    """two mid-range sentinels take the same side of `if wind > 0:` / give the same `int(p)`, so comparing their two
    traces proves nothing; the campaign also traces at both ends of every parameter's range and on real plan rows."""
    import elodin_amd.frontend as el
    ex = example(0)

    def build_branch(params):
        world, _ = ex.build(params)
        wind = float(params.get("wind", 0.0))                     # PARAMS: wind in [-1, 1] — the sentinels sit on one side of 0.9

        @el.map
        def plant(pos: ex.Position, vel: ex.Velocity) -> ex.Position:
            return pos + vel * (0.5 if wind > 0.9 else 0.25)      # a Python branch on a parameter: per-run code
        return world, plant
    rows = [{"mass": 1.5, "target_x": 30.0, "thrust_gain": 1.0, "wind": w} for w in (-0.5, 0.2, 0.95)]
    with pytest.raises(NotImplementedError, match="different path"):
        vectorize.Campaign(build_branch, vectorize.plan_of(rows), ex.PARAMS, dry=True)

    def build_int(params):
        world, _ = ex.build(params)
        n = int(float(params.get("target_x", 30.0)) / 10.0)       # truncation on the host: 2 / 3 / 4 copies of a term

        @el.map
        def plant(pos: ex.Position, vel: ex.Velocity) -> ex.Position:
            out = pos
            for _ in range(n):
                out = out + vel * 0.125
            return out
        return world, plant
    rows = [{"mass": 1.5, "target_x": t, "thrust_gain": 1.0, "wind": 0.1} for t in (21.0, 30.0, 44.0)]
    with pytest.raises(NotImplementedError, match="cannot share one program"):
        vectorize.Campaign(build_int, vectorize.plan_of(rows), ex.PARAMS, dry=True)


def test_a_spawn_that_branches_on_a_parameter_falls_back_to_one_build_per_run():
    import elodin_amd.frontend as el
    ex = example(0)

    def build(params):
        world, plant = ex.build(params)
        side = 1.0 if float(params.get("wind", 0.0)) > 0.9 else -1.0        # constant under both sentinels, not over the plan
        world.spawn([el.C(ex.Target, np.array([side]))], name="marker")
        return world, plant
    rows = [{"mass": 1.5, "target_x": 30.0, "thrust_gain": 1.0, "wind": w} for w in (-0.5, 0.2, 0.95)]
    c = vectorize.Campaign(build, vectorize.plan_of(rows), ex.PARAMS, dry=True)
    assert c.per_run_builds
    assert np.array_equal(c.world.column("target")[0][1::2, 0], [-1.0, -1.0, 1.0])
