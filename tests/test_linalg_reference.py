"""Small dense linear algebra in traced user code (elodin_amd/dsl_mat.py) against the reference's own data: its linalg
example (examples/linalg/sim.py — 2-, 3- and 6-state Kalman filters built on solve / inv / cholesky / qr / det / slogdet /
svd / eigh, a matrix-RHS solve, Cholesky variants, an int64 mode selector with a traced scatter index) written against the DSL
(tests/linalg_dsl.py) and stepped 100 ticks must land on the rows of its CI baseline scripts/ci/baseline/linalg/*.csv
(tests/golden/linalg.json).  The reference's CI accepts 1e-4 (scripts/ci/baseline/tolerances.json); here 1e-9.
Plus known answers of every factorisation against numpy's LAPACK on random matrices, evaluated from the traced DAG."""
import json
from pathlib import Path

import numpy as np
import pytest

from elodin_amd import dsl, dsl_mat as M
from tests import dsl_numpy, linalg_dsl as S

GOLD = json.loads((Path(__file__).parent / "golden" / "linalg.json").read_text())["rows"]
# chol_res_norms: Frobenius norms of reconstruction residuals — 0.0 exactly in the reference's LAPACK run, a few ulp of the
# 9.0-scale matrices here (L L^T re-multiplied in another order): compared absolutely
ABS = {"chol_res_norms": 1e-14}


def initial_columns():
    cols = {}
    for comps in S.INITIAL.values():
        for k, v in comps.items():
            cols[k] = np.asarray(v, dtype=np.float64).reshape(1, -1)
    widths = {k: (S.SHAPES[k] if k in S.SHAPES else v.shape[1]) for k, v in cols.items()}
    return cols, widths


def worst_errors(get, tick, worst):
    for name, rows in GOLD.items():
        ref, got = np.asarray(rows[tick]), np.asarray(get(name), dtype=np.float64).reshape(-1)
        e = float(np.max(np.abs(got - ref))) / (1.0 if name in ABS else max(float(np.max(np.abs(ref))), 1e-300))
        worst[name] = max(worst.get(name, 0.0), e / (ABS[name] / 1e-9 if name in ABS else 1.0))


def test_linalg_example_lands_on_the_reference_baseline_rows():
    cols, widths = initial_columns()
    tp = dsl.Program(S.SYSTEMS, dsl.pipe(), []).trace(widths)
    assert dict(tp.columns)["ekf6_cov"] == 36 and tp.table.mats["ekf6_cov"] == (6, 6) and tp.table.mats["mrhs_state"] == (3, 2)
    pos, vel, acc, inertia = np.array([[0, 0, 0, 1.0, 0, 0, 0]]), np.zeros((1, 6)), np.zeros((1, 6)), np.ones((1, 7))
    comps = {n: cols[n].copy() for n, _ in tp.columns}
    worst = {}
    for tick in range(1, 101):
        dsl_numpy.program_tick_systems_only(tp, pos, vel, acc, inertia, comps, tick)
        worst_errors(lambda name: comps[name], tick, worst)
    print("linalg example vs reference baseline (100 ticks), worst per component:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert len(worst) == 11 and max(worst.values()) < 1e-9, worst
    assert np.array_equal(comps["mode_state"][0], GOLD["mode_state"][100])        # the integer surface: exact


def _ev(fn, *args):
    return dsl_numpy.trace_eval(lambda xp, *a: fn(*a), *args)


@pytest.mark.parametrize("n", [2, 3, 4, 6, 8])
def test_factorisations_agree_with_lapack(n):
    rng = np.random.default_rng(100 + n)
    la = dsl.np.linalg
    A, B, b = rng.normal(size=(n, n)), rng.normal(size=(n, 3)), rng.normal(size=n)
    A[0, 0] = 1e-9                                                           # forces a row exchange on the first pivot
    spd = A @ A.T + np.eye(n)
    mat = lambda flat, r=n, c=n: M.reshape(flat, (r, c))
    close = lambda got, want, tol=1e-11: np.max(np.abs(np.asarray(got) - want)) <= tol * max(1.0, np.max(np.abs(want)))
    assert close(_ev(lambda a, v: la.solve(mat(a), v), A.ravel(), b), np.linalg.solve(A, b))
    assert close(_ev(lambda a, m: la.solve(mat(a), mat(m, n, 3)).flatten(), A.ravel(), B.ravel()).reshape(n, 3), np.linalg.solve(A, B))
    assert close(_ev(lambda a: la.inv(mat(a)).flatten(), A.ravel()).reshape(n, n), np.linalg.inv(A), 1e-10)
    assert close(_ev(lambda a: la.det(mat(a)), A.ravel()), np.linalg.det(A))
    sgn, logabs = np.linalg.slogdet(A)
    got = _ev(lambda a: dsl.np.array(list(la.slogdet(mat(a)))), A.ravel())
    assert got[0] == sgn and abs(got[1] - logabs) < 1e-11
    assert close(_ev(lambda a: la.cholesky(mat(a)).flatten(), spd.ravel()).reshape(n, n), np.linalg.cholesky(spd))
    assert close(_ev(lambda a: M.cholesky(mat(a), lower=False).flatten(), spd.ravel()).reshape(n, n), np.linalg.cholesky(spd).T)
    qr = _ev(lambda a: dsl.np.concatenate([m.flatten() for m in la.qr(mat(a))]), A.ravel())
    qn, rn = np.linalg.qr(A)                                                 # same Householder sign convention as dgeqrf
    assert close(qr[:n * n].reshape(n, n), qn) and close(qr[n * n:].reshape(n, n), rn)
    w = _ev(lambda a: dsl.np.concatenate([la.eigh(mat(a))[0], la.eigh(mat(a))[1].flatten()]), spd.ravel())
    assert close(w[:n], np.linalg.eigvalsh(spd)) and close(spd @ w[n:].reshape(n, n), w[n:].reshape(n, n) * w[:n], 1e-10)
    usv = _ev(lambda a: dsl.np.concatenate([la.svd(mat(a))[0].flatten(), la.svd(mat(a))[1], la.svd(mat(a))[2].flatten()]), A.ravel())
    u, s, vh = usv[:n * n].reshape(n, n), usv[n * n:n * n + n], usv[n * n + n:].reshape(n, n)
    assert close(s, np.linalg.svd(A)[1]) and close((u * s) @ vh, A) and close(u.T @ u, np.eye(n)) and np.all(np.diff(s) <= 0)
    assert close(_ev(lambda a: la.pinv(mat(a)).flatten(), spd.ravel()).reshape(n, n), np.linalg.pinv(spd), 1e-10)
    assert close(_ev(lambda a: la.norm(mat(a)), A.ravel()), np.linalg.norm(A))


def test_pinv_of_a_rank_deficient_matrix_cuts_like_jax():
    a = np.outer([1.0, 2.0, 3.0], [0.5, -1.0, 2.0])                           # rank 1
    got = _ev(lambda m: dsl.np.linalg.pinv(M.reshape(m, (3, 3))).flatten(), a.ravel()).reshape(3, 3)
    assert np.max(np.abs(got - np.linalg.pinv(a))) < 1e-13


def test_matrix_surface():
    np_ = dsl.np
    a = np_.array([[1.0, 2.0], [3.0, 4.0]])
    v = np_.array([1.0, -1.0])
    val = lambda e: float(dsl_numpy._eval([dsl._lift(e)], {}, 1)[0][0])
    assert isinstance(a, M.Mat) and a.shape == (2, 2) and isinstance(a @ v, dsl.Vec) and isinstance(v @ a, dsl.Vec)
    assert [val(x) for x in (a @ v)] == [-1.0, -1.0] and [val(x) for x in (v @ a)] == [-2.0, -2.0] and val(v @ v) == 2.0
    assert [val(x) for x in a.T[0]] == [1.0, 3.0] and [val(x) for x in a[:, 1]] == [2.0, 4.0] and val(a[1, 0]) == 3.0
    b = np_.block([[a, np_.zeros((2, 2))], [np_.eye(2), a.T]])
    assert b.shape == (4, 4) and val(b[3, 3]) == 4.0 and val(b[2, 0]) == 1.0 and val(np_.trace(b)) == 10.0
    c = np_.zeros((3, 3)).at[0:2, 1:3].set(a).at[2, 0].set(7.0)
    assert [[val(x) for x in r] for r in c] == [[0, 1, 2], [0, 3, 4], [7, 0, 0]]
    assert [val(x) for x in np_.diag(np_.diag(v))] == [1.0, -1.0] and [val(x) for x in a.flatten()] == [1, 2, 3, 4]
    assert [val(x) for x in np_.zeros(4).at[np_.array([2.0])[0]].set(1.0)] == [0, 0, 1, 0]
    assert [[val(x) for x in r] for r in M.skew(np_.array([1.0, 2.0, 3.0]))] == [[0, -3, 2], [3, 0, -1], [-2, 1, 0]]
