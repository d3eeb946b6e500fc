"""Shared by the CPU and GPU whole-world StableHLO tests: the assembled world ticks (tests/golden/hlo_world_builder.py), their
initial columns from the reference's golden data, and the comparison with it.  TEST INFRASTRUCTURE."""
import numpy as np

from elodin_amd import dsl
from elodin_amd import stablehlo as sh
from tests import golden_util as gu
from tests.golden import hlo_world_builder as hb

BODY = (("world_pos", 7), ("world_vel", 6), ("world_accel", 6), ("force", 6), ("inertia", 7))


def three_body(mode="world", arith="reference"):
    """-> (system, manifest, widths, {column: one row of initial values}, golden)"""
    text, slots = hb.three_body_world()
    system, manifest = sh.world_system(text, slots, mode=mode, name="three_body_world", arith=arith)
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    g = gu.load("three_body")
    row = {"hlo_tick": np.zeros(1), "hlo_simulation_time_step": np.array([g["globals.simulation_time_step"][0, 0]])}
    for c, _ in BODY:
        row["hlo_" + c] = np.concatenate([g[f"{e}.{c}"][0] for e in "abc"])
    return system, manifest, widths, row, g


def three_body_errors(columns, g, r, lane=0):
    """Worst relative error (per row vector, SURVEY §8(d)'s vector-scaled form) and worst element-wise error of tick r."""
    worst, worst_elem = 0.0, 0.0
    for c, w in BODY[:4]:
        for i, e in enumerate("abc"):
            got, ref = np.asarray(columns["hlo_" + c][lane, i * w:(i + 1) * w], dtype=np.float64), g[f"{e}.{c}"][r]
            worst = max(worst, gu.rel_err(got, ref))
            scale = np.maximum(np.abs(ref), 1e-12 * max(np.max(np.abs(ref)), 1e-300))
            worst_elem = max(worst_elem, float(np.max(np.abs(got - ref) / scale)))
    return worst, worst_elem


def independent_bodies(n, seed=5):
    """configs[1] as a whole-world module over n bodies -> (text, slots, {column: [n, w] initial rows}, oracle world factory)"""
    text, slots = hb.independent_bodies_world(n)
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    cols = {"world_pos": np.concatenate([q, rng.uniform(-10, 10, (n, 3))], axis=1),
            "world_vel": rng.uniform(-1, 1, (n, 6)), "world_accel": np.zeros((n, 6)), "force": np.zeros((n, 6)),
            "inertia": np.concatenate([rng.uniform(0.5, 2.0, (n, 3)), np.zeros((n, 3)), rng.uniform(1.0, 5.0, (n, 1))], axis=1),
            "torque": rng.uniform(-0.1, 0.1, (n, 3))}
    return text, slots, cols


def ball(mode="auto", arith="reference"):
    """examples/ball's singleton world as a whole-world module -> (system, manifest, widths, {column: initial row}, golden)"""
    text, slots = hb.ball_world()
    system, manifest = sh.world_system(text, slots, mode=mode, name="ball_world", arith=arith)
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    g = gu.load("ball")
    row = {"hlo_tick": np.zeros(1), "hlo_seed": np.array([float(g["ball.seed"][0, 0])]), "hlo_wind": g["ball.wind"][0].copy(),
           "hlo_simulation_time_step": np.array([g["globals.simulation_time_step"][0, 0]])}
    for c, _ in BODY:
        row["hlo_" + c] = g[f"ball.{c}"][0].copy()
    return system, manifest, widths, row, g


def ball_errors(columns, g, r, lane=0):
    worst = 0.0
    for c in ("world_pos", "world_vel", "world_accel", "force", "wind"):
        worst = max(worst, gu.rel_err(np.asarray(columns["hlo_" + c][lane], dtype=np.float64), g[f"ball.{c}"][r]))
    return worst


def strided_world_columns(g, names, stride, worlds, extra_fill=None):
    """Columns of `worlds` copies of a golden world laid out one lane per entity: a world = `stride` consecutive rows, its entities first,
    padding rows (identity pose, unit inertia) behind them."""
    n = len(names)
    rows = stride * worlds
    comps = {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), g["globals.simulation_time_step"][0, 0])}
    for c, w in BODY:
        a = np.zeros((rows, w))
        if c == "world_pos":
            a[:, 3] = 1.0
        if c == "inertia":
            a[:] = 1.0
        for wd in range(worlds):
            for i, e in enumerate(names):
                a[wd * stride + i] = g[f"{e}.{c}"][0]
        comps["hlo_" + c] = a
    return comps
