"""CPU side of stand-alone folds inside a program: the trace splits the tick into a chain of launches, the generated source
holds one PIPE struct per link and bakes the CSR, and the numpy walker (the GPU tests' reference) folds in spawn order reading
the values from before the fold ran."""
import numpy as np
import pytest

from elodin_amd import codegen, dsl
from tests import dsl_numpy


@dsl.system
def double(x):
    return {"x": x * 2.0}


@dsl.graph_fold("e", left=("x",), right=("x",), out="x", init=5.0)
def fold_test(x, a, b):
    return x + a + b


@dsl.system
def add_one(x, n):
    return {"x": x + 1.0, "n": n + 1.0}


EDGES = {"e": ([0, 0, 1], [1, 2, 2])}          # rows: e1 -> e2, e1 -> e3, e2 -> e3 (test_all.py:117-142's graph)


def test_trace_and_generated_chain():
    tp = dsl.Program([double, fold_test, add_one], dsl.pipe(), []).trace({"x": 1, "n": 1}, fold_edges=EDGES)
    fs = tp.fold_stages[0]
    assert (fs.src_rows, fs.row_start, fs.dst) == ([0, 1], [0, 2, 3], [1, 2, 2])
    assert [n for n, _ in tp.columns] == ["x", "x#fold0", "n"] and fs.scratch_name == "x#fold0"
    src = codegen.generate_source(tp, "float64", 2)
    assert "// tick = [double] | fold:fold_test | [add_one]" in src
    assert src.count("struct PipeSeg") == 2 and "fold0_kernel<double>" in src and "fold0_commit<double>" in src
    assert "__device__ const uint32_t fold0_dst[3] = {1, 2, 2};" in src
    assert "qs.hist_ring = 0;" in src                     # only the last link records the tick
    plain = codegen.generate_source(dsl.Program([double, add_one], dsl.pipe(), []).trace({"x": 1, "n": 1}), "float64", 2)
    assert "struct PipeCustom" in plain and "PipeSeg" not in plain and "fold0" not in plain


def test_walker_folds_in_spawn_order_on_pre_fold_values():
    tp = dsl.Program([double, fold_test, add_one], dsl.pipe(), []).trace({"x": 1, "n": 1}, fold_edges=EDGES)
    pos = np.tile([0.0, 0, 0, 1, 0, 0, 0], (3, 1))
    vel, inertia, acc = np.zeros((3, 6)), np.ones((3, 7)), np.zeros((3, 6))
    comps = {"x": np.array([[1.0], [2.0], [2.0]]), "n": np.zeros((3, 1)), "x#fold0": np.zeros((3, 1))}
    x = np.array([1.0, 2.0, 2.0])
    for tick in range(1, 4):
        dsl_numpy.program_tick_systems_only(tp, pos, vel, acc, inertia, comps, tick)
        x = x * 2.0
        x = np.array([5.0 + (x[0] + x[1]) + (x[0] + x[2]), 5.0 + (x[1] + x[2]), x[2]]) + 1.0
        assert np.array_equal(comps["x"][:, 0], x)


def test_fold_stage_guards():
    with pytest.raises(ValueError, match="no edges given"):
        dsl.Program([fold_test], dsl.pipe(), []).trace({"x": 1})
    big = {"e": (list(range(70000)), list(range(70000)))}
    with pytest.raises(ValueError, match="bake their edges"):
        dsl.Program([fold_test], dsl.pipe(), []).trace({"x": 1}, fold_edges=big)

    @dsl.graph_fold("e", left=("x",), right=("x",), out="world_pos", init=[0.0] * 7)
    def bad(acc, a, b):
        return acc
    with pytest.raises(TypeError, match="plain component"):
        dsl.Program([bad], dsl.pipe(), []).trace({"x": 1}, fold_edges=EDGES)


def test_additive_folds_are_recognised_for_the_hub_path():
    """pair_kernel.hpp 2c sums partial folds of a hub's out-edges: only sound when every component of the fold is
    acc +/- g(a, b) or the constant 0 — decided on the traced DAG (codegen._fold_is_additive)."""
    np_ = dsl.np

    @dsl.edge_fold
    def newton(acc, a_pos, a_inertia, b_pos, b_inertia):          # three-body's gravity_fn: Force(linear = acc.f - f), torque zeroed
        r = a_pos.linear() - b_pos.linear()
        n = np_.linalg.norm(r)
        return dsl.SpatialForce(linear=acc.force() - r * (a_inertia.mass() * b_inertia.mass() / (n * n * n)))

    @dsl.edge_fold
    def damped(acc, a_pos, a_inertia, b_pos, b_inertia):          # acc * 0.5 + g: order matters, stays sequential
        return dsl.SpatialForce(linear=acc.force() * 0.5 + (b_pos.linear() - a_pos.linear()))

    @dsl.edge_fold
    def offset(acc, a_pos, a_inertia, b_pos, b_inertia):          # a non-zero constant would be counted once per partial
        return dsl.SpatialForce(torque=np_.array([1.0, 0.0, 0.0]), linear=acc.force() + b_pos.linear())
    assert codegen._fold_is_additive(newton.trace())
    assert not codegen._fold_is_additive(damped.trace())
    assert not codegen._fold_is_additive(offset.trace())
    assert "kAdditive = true" in codegen.generate_pair_source(newton.trace())
    assert "kAdditive = false" in codegen.generate_pair_source(damped.trace())


def test_fold_stage_csr_keeps_spawn_order_per_source():
    tp = dsl.Program([fold_test], dsl.pipe(), []).trace({"x": 1}, fold_edges={"e": ([2, 0, 2, 0, 2], [1, 2, 0, 1, 2])})
    fs = tp.fold_stages[0]
    assert fs.src_rows == [0, 2] and fs.row_start == [0, 2, 5] and fs.dst == [2, 1, 1, 0, 2]


def test_only_plain_sums_are_folded_by_a_wave_per_source():
    """dsl.GraphFold.wave_fold asks for the wave-per-source fold kernel (lane partials + a shuffle tree: another association of the
    sum).  codegen grants it only when every component is acc_k +- g, acc_k itself or the constant 0 (_graph_fold_kinds); a fold that
    is not a plain sum — a running maximum, a product, an accumulator that feeds its own update — keeps the sequential kernel."""
    from elodin_amd import codegen
    np_ = dsl.np

    def program(fn, init):
        @dsl.system(x=3)
        def produce(x, y):
            return {"y": np_.array([x[0] * 2.0, x[1] + x[2]])}
        fold = dsl.GraphFold(fn, "e", ("y",), ("y",), "z", init)
        fold.wave_fold = True
        prog = dsl.Program([produce, fold], dsl.Pipe([]), [])
        n = 6
        edges = ([s_ for s_ in range(n) for _ in range(n - 1)], [t for s_ in range(n) for t in range(n) if t != s_])
        tp = prog.trace({"x": 3, "y": 2, "z": 2}, fold_edges={"e": edges})
        return tp, codegen.generate_source(tp, "float64", 2)

    tp, src = program(lambda acc, a, b: np_.array([acc[0] + a[0] * b[1], acc[1] - b[0]]), [0.0, 1.5])
    assert codegen._graph_fold_kinds(tp.fold_stages[0].traced.outputs, 2) == ["sum", "sum"]
    assert "one WAVE per source" in src and "T(1.5) + v1" in src                       # the initial value joins after the tree
    tp, src = program(lambda acc, a, b: np_.array([np_.maximum(acc[0], b[0]), acc[1] + b[1]]), [0.0, 0.0])
    assert codegen._graph_fold_kinds(tp.fold_stages[0].traced.outputs, 2) is None      # a running maximum is not a sum
    assert "one WAVE per source" not in src and "fold0_kernel" in src
    tp, src = program(lambda acc, a, b: np_.array([acc[0] * b[0], acc[1]]), [1.0, 0.0])
    assert codegen._graph_fold_kinds(tp.fold_stages[0].traced.outputs, 2) is None and "one WAVE per source" not in src
    tp, src = program(lambda acc, a, b: np_.array([acc[0] + acc[1] * b[0], acc[1] + 1.0]), [0.0, 0.0])      # g depends on the accumulator
    assert codegen._graph_fold_kinds(tp.fold_stages[0].traced.outputs, 2) is None and "one WAVE per source" not in src
