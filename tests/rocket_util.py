"""Shared by the rocket golden tests (CPU walker and GPU): the reference's rocket-csv baseline rows and the comparison.
TEST INFRASTRUCTURE."""
import json
from pathlib import Path

import numpy as np

GOLDEN = json.loads((Path(__file__).parent / "golden" / "rocket.json").read_text())
RTOL = 1e-9
# every golden column, scaled by its own largest magnitude over the 100 ticks (the PID columns start at 1e-18)
SCALE = {name: max(float(np.abs(np.array(rows)).max()), 1e-300) for name, rows in GOLDEN["rows"].items()}


def check_row(tick: int, got: dict, worst: dict):
    """got: {column: [w] values after `tick` ticks}.  Records the worst scaled deviation per column."""
    for name, rows in GOLDEN["rows"].items():
        want = np.array(rows[tick])
        dev = float(np.abs(np.asarray(got[name], dtype=np.float64).ravel() - want).max())
        err = dev / SCALE[name] if SCALE[name] > 1e-300 else dev      # an all-zero golden column must stay exactly zero
        worst[name] = max(worst.get(name, 0.0), err)


def assert_all_columns(worst: dict, rtol: float = RTOL):
    assert set(worst) == set(GOLDEN["rows"]) and len(worst) == 24
    bad = {k: v for k, v in worst.items() if not v < rtol}
    assert not bad, bad
