"""The fuzz tests' independent check of the TRACER.  tests/test_gpu_fuzz.py compares generated kernels with the numpy walk of
the SAME traced DAG (tests/dsl_numpy.py): that pins the code generator, but a bug in elodin_amd/dsl.py passes on both sides.
tests/fuzz_gen.py therefore evaluates every random program a second way — plain numpy closures built from the same seed by
code that imports nothing of the tracer's arithmetic.  Here, on the CPU: the two agree on every seed the GPU fuzz flies, and a
seeded tracer bug (two operands swapped in dsl.py) is caught."""
import numpy as np
import pytest

from elodin_amd import dsl, workloads
from tests import dsl_numpy
from tests import fuzz_gen as fg


def walk(prog, cols, ticks):
    n = len(cols["x"])
    tp = prog.trace({k: v.shape[1] for k, v in cols.items()})
    w = workloads.independent_bodies(n)
    pos, vel, inertia = (np.array(w[k], dtype=np.float64) for k in ("world_pos", "world_vel", "inertia"))
    want = {k: v.copy() for k, v in cols.items()}
    for t in range(1, ticks + 1):
        dsl_numpy._run_systems(tp.pre, pos, vel, inertia, want, tp.table, t)
        dsl_numpy._run_systems(tp.post, pos, vel, inertia, want, tp.table, t)
    return want


def worst(a, b):
    return max(float(np.max(np.abs(a[k] - b[k]) / np.maximum(np.abs(b[k]), 1.0))) for k in ("a", "b", "c", "x"))


@pytest.mark.parametrize("seed,depth", [(s, 4) for s in range(6)] + [(101, 3)])
def test_numpy_twin_agrees_with_the_walk_of_the_traced_program(seed, depth):
    cols = fg.columns(seed, 512)
    twin = fg.twin_run(seed, cols, 2, depth)
    traced = walk(fg.make_program(seed, depth), cols, 2)
    assert all(np.isfinite(twin[k]).all() for k in twin)
    assert worst(traced, twin) < 1e-11, worst(traced, twin)
    assert np.abs(twin["c"]).max() > 0.0 and not np.array_equal(twin["x"], cols["x"])


@pytest.mark.parametrize("bug", ["sub", "arctan2", "where", "maximum"])
def test_a_seeded_tracer_bug_is_caught(bug, monkeypatch):
    """Swap two operands inside the tracer: the traced program (and with it the walker AND any kernel generated from it) now
    computes something else, the twin does not — the comparison the GPU fuzz makes against the twin fails."""
    if bug == "sub":
        real = dsl.Expr.__sub__
        monkeypatch.setattr(dsl.Expr, "__sub__", lambda self, o: real(dsl._lift(o), self) if not isinstance(o, dsl.Vec) else NotImplemented)
    elif bug == "arctan2":
        real2 = dsl._Np.arctan2
        monkeypatch.setattr(dsl._Np, "arctan2", staticmethod(lambda y, x: real2(x, y)))
    elif bug == "where":
        real3 = dsl._Np.where
        monkeypatch.setattr(dsl._Np, "where", staticmethod(lambda c, a, b: real3(c, b, a)))
    else:
        monkeypatch.setattr(dsl._Np, "maximum", staticmethod(dsl._Np.minimum))
    caught = 0
    for seed in range(6):
        cols = fg.columns(seed, 256)
        if worst(walk(fg.make_program(seed), cols, 2), fg.twin_run(seed, cols, 2)) > 1e-6:
            caught += 1
    assert caught >= (1 if bug == "maximum" else 5), (bug, caught)      # `maximum` is drawn rarely: some programs have none
