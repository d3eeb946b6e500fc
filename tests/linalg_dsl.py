"""The reference's linalg example (examples/linalg/sim.py:183-390) written against elodin_amd.dsl — the same expressions in
the same order, `jnp` / `la` / `jsl` spelled `np_` / `np_.linalg` / dsl_mat.  Six single-component-set entities, one system
each, piped in the reference's order (sim.py:413-414).  Used by the golden-CSV tests (scripts/ci/baseline/linalg ->
tests/golden/linalg.json).  TEST INFRASTRUCTURE."""
from elodin_amd import dsl, dsl_mat

np_ = dsl.np
lax = dsl.lax
la = np_.linalg

DT = 1.0 / 120.0
F3 = np_.array([[1.0, DT, 0.0], [0.0, 1.0, DT], [0.0, 0.0, 1.0]])
Q3, H3, R3 = np_.eye(3) * 0.01, np_.eye(3), np_.eye(3) * 0.1
F2 = np_.array([[1.0, DT], [0.0, 1.0]])
Q2, H2, R2 = np_.eye(2) * 0.01, np_.eye(2), np_.eye(2) * 0.1
F6 = np_.block([[np_.eye(3), np_.eye(3) * DT], [np_.zeros((3, 3)), np_.eye(3)]])
Q6, H6, R6 = np_.eye(6) * 0.01, np_.eye(6), np_.eye(6) * 0.1
CHOL_A_3X3 = np_.array([[4.0, 2.0, 3.0], [2.0, 8.0, 1.0], [3.0, 1.0, 9.0]])
CHOL_B_3X3 = np_.array([[9.0, 3.0, 1.0], [3.0, 6.0, 2.0], [1.0, 2.0, 5.0]])

INITIAL = {                                                                        # sim.py:374-410
    "tracker3": {"kf3_state": [0.0, 1.0, 0.0], "kf3_cov": [[10.0, 0, 0], [0, 10.0, 0], [0, 0, 10.0]], "kf3_info": [0.0] * 5},
    "tracker6": {"ekf6_state": [0.0, 0.0, 100.0, 10.0, 0.0, -5.0], "ekf6_cov": [[100.0 if i == j else 0.0 for j in range(6)] for i in range(6)],
                 "ekf6_info": [0.0] * 4},
    "mat_rhs": {"mrhs_state": [[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]]},
    "small2": {"sm2_state": [1.0, 0.5], "sm2_cov": [[5.0, 0.0], [0.0, 5.0]]},
    "mode_sel": {"mode_state": [0.0, 0.0, 0.0, 0.0]},
    "chol_variants": {"chol_res_norms": [0.0, 0.0, 0.0]},
}
SHAPES = {"kf3_cov": (3, 3), "ekf6_cov": (6, 6), "sm2_cov": (2, 2), "mrhs_state": (3, 2)}


def _safe_matrix_inverse(matrix, tolerance=1e-12):                                  # sim.py:172-176
    u, s, vh = la.svd(matrix)
    s_inv = np_.where(s > tolerance, 1.0 / s, 0.0)
    return np_.transpose(vh) @ np_.diag(s_inv) @ np_.transpose(u)


@dsl.system(mrhs_state=(3, 2))
def mat_rhs_step(mrhs_state):                                                       # sim.py:182-185
    A = F3 + np_.eye(3) * 0.01
    return {"mrhs_state": la.solve(A, mrhs_state)}


@dsl.system(sm2_state=2, sm2_cov=(2, 2))
def small2_step(sm2_state, sm2_cov):                                                # sim.py:191-217
    state, cov = sm2_state, sm2_cov
    x_pred = F2 @ state
    P_pred = F2 @ cov @ F2.T + Q2
    z = x_pred + np_.ones(2) * 0.01
    y = z - H2 @ x_pred
    S = H2 @ P_pred @ H2.T + R2
    K = la.solve(S.T, (P_pred @ H2.T).T).T
    x_upd = x_pred + K @ y
    IKH = np_.eye(2) - K @ H2
    P_upd = IKH @ P_pred @ IKH.T + K @ R2 @ K.T
    should_refine = np_.logical_and(la.norm(y) < 50.0, state[0] > -1e6)
    x_upd = lax.cond(should_refine, lambda _: x_upd + la.solve(S + np_.eye(2) * 1e-3, y) * 1e-12, lambda _: x_upd, operand=None)
    return {"sm2_state": x_upd, "sm2_cov": P_upd}


@dsl.system(kf3_state=3, kf3_cov=(3, 3), kf3_info=5)
def kf3_step(kf3_state, kf3_cov, kf3_info):                                         # sim.py:223-287
    state, cov = kf3_state, kf3_cov
    x_pred = F3 @ state
    P_pred = F3 @ cov @ F3.T + Q3
    z = x_pred + np_.ones(3) * 0.01
    y = z - H3 @ x_pred
    S = H3 @ P_pred @ H3.T + R3
    K = la.solve(S.T, (P_pred @ H3.T).T).T
    x_upd = x_pred + K @ y
    IKH = np_.eye(3) - K @ H3
    P_upd = IKH @ P_pred @ IKH.T + K @ R3 @ K.T
    Q_f, R_f = la.qr(P_upd)
    P_upd = Q_f @ R_f
    d = la.det(S)
    sign, logdet = la.slogdet(S)
    S_inv_y = la.solve(S, y)
    log_lik = (np_.log(2.0 * np_.pi) * 3.0 + logdet + y @ S_inv_y) * -0.5

    def _heavy_cond_branch(_):
        solve_vec = la.solve(S + np_.eye(3) * 1e-3, y + np_.ones(3) * 1e-3)
        v = solve_vec + x_upd
        for _k in range(12):
            yaw = np_.arctan2(v[1], v[0] + 1e-9)
            pitch = np_.arctan2(v[2], np_.sqrt(v[0] * v[0] + v[1] * v[1]) + 1e-9)
            c0, s0, c1, s1 = np_.cos(yaw), np_.sin(yaw), np_.cos(pitch), np_.sin(pitch)
            v = np_.array([v[0] * c0 - v[1] * s0 + 0.01 * s1, v[0] * s0 + v[1] * c0 + 0.01 * c1, v[2] * c1 + 0.01 * (s0 * c0)])
        return x_upd + v * 1e-12

    armed = np_.logical_and(state[0] > 0.5, state[1] > -1e3)
    trigger = np_.logical_and(armed, la.norm(x_upd) < 1e8)
    x_upd = lax.cond(trigger, _heavy_cond_branch, lambda _: x_upd, operand=None)
    info_out = np_.array([log_lik, d, sign, la.norm(x_upd), la.norm(K[:, 0])])
    return {"kf3_state": x_upd, "kf3_cov": P_upd, "kf3_info": info_out}


@dsl.system(ekf6_state=6, ekf6_cov=(6, 6), ekf6_info=4)
def ekf6_step(ekf6_state, ekf6_cov, ekf6_info):                                     # sim.py:293-330
    state, cov = ekf6_state, ekf6_cov
    x_pred = F6 @ state
    P_pred = F6 @ cov @ F6.T + Q6
    z = x_pred + np_.ones(6) * 0.001
    y = z - H6 @ x_pred
    S = H6 @ P_pred @ H6.T + R6
    S_pinv = _safe_matrix_inverse(S)
    K = P_pred @ H6.T @ S_pinv
    x_upd = x_pred + K @ y
    IKH = np_.eye(6) - K @ H6
    P_upd = IKH @ P_pred @ IKH.T + K @ R6 @ K.T
    eigvals, _eigvecs = la.eigh(P_upd)
    should_correct = np_.logical_and(la.norm(y) < 100.0, eigvals[0] > 0.0)
    x_upd = lax.cond(should_correct, lambda _: x_upd + la.solve(P_upd + np_.eye(6) * 1e-3, y) * 1e-12, lambda _: x_upd, operand=None)
    info_out = np_.array([la.norm(y), np_.max(eigvals), np_.min(eigvals), la.norm(x_upd[:3])])
    return {"ekf6_state": x_upd, "ekf6_cov": P_upd, "ekf6_info": info_out}


@dsl.system(chol_res_norms=3)
def chol_variants_step(chol_res_norms):                                             # sim.py:343-360
    U = dsl_mat.cholesky(CHOL_A_3X3, lower=False)                                   # jsl.cholesky(A, lower=False)
    L = dsl_mat.cholesky(CHOL_A_3X3, lower=True)
    Lb = [la.cholesky(CHOL_A_3X3), la.cholesky(CHOL_B_3X3)]                         # jnp.linalg.cholesky of the [2,3,3] batch
    upper_res = U.T @ U - CHOL_A_3X3
    lower_res = L @ L.T - CHOL_A_3X3
    batch = [Lb[0] @ Lb[0].T - CHOL_A_3X3, Lb[1] @ Lb[1].T - CHOL_B_3X3]
    batch_norm = np_.sqrt(la.norm(batch[0]) ** 2 + la.norm(batch[1]) ** 2)         # la.norm of the whole [2,3,3] residual
    return {"chol_res_norms": np_.array([la.norm(upper_res), la.norm(lower_res), batch_norm]) + chol_res_norms * 0.0}


@dsl.system(mode_state=4)
def mode_step(mode_state):                                                          # sim.py:364-375 (int64 component, exact in f64)
    active = np_.logical_and(mode_state[0] > 1, np_.equal(mode_state[1], 0))
    seed = lax.cond(active, lambda _: mode_state + np_.array([1.0, 0.0, 0.0, 0.0]), lambda _: mode_state, operand=None)
    idx = np_.remainder(seed[0], 4.0)
    return {"mode_state": np_.zeros(4).at[idx].set(1.0)}


SYSTEMS = [mat_rhs_step, small2_step, kf3_step, ekf6_step, mode_step, chol_variants_step]      # sim.py:413-414
