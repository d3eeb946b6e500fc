"""Pin semi-implicit Euler and the torque-bearing half of calc_accel on the reference's OWN data.

scripts/ci/baseline/cube-sat-csv is a SemiImplicit run (examples/cube-sat/main.py:699-710) whose satellite carries a
non-zero torque and a non-uniform inertia diagonal; `earth` is a pure rotation.  Teacher forcing: row r of the
recorded `force` column is what the effectors produced on tick r (evaluated at row r-1's state), so

    a_r = calc_accel(F_r, I, x_{r-1});  v_r = v_{r-1} + dt*a_r;  x_r = x_{r-1} (+) dt*v_r
    (libs/nox-py/src/integrator/semi_implicit.rs:17-62, six_dof.rs:137-146)

must reproduce rows r of world_accel / world_vel / world_pos with no effector of the example restated.  All 100
transitions of both entities run as ONE 200-entity one-tick world (entity = transition).
"""
import numpy as np

from oracle import oracle as orc
from tests import golden_util as gu

ENTITIES = ("ore_sat", "earth")


def teacher_forced_world():
    """-> (x_prev [200,7], v_prev [200,6], inertia [200,7], F [200,6], expected dict, dt)"""
    g = gu.load("cube_sat")
    dt = float(g["globals.simulation_time_step"][0, 0])
    assert dt == 0.008333333
    assert g["globals.tick"][:, 0].tolist() == list(range(101))
    xs, vs, Is, Fs, exp = [], [], [], [], {"world_pos": [], "world_vel": [], "world_accel": [], "force": []}
    for e in ENTITIES:
        xs.append(g[f"{e}.world_pos"][:-1]); vs.append(g[f"{e}.world_vel"][:-1])
        Is.append(g[f"{e}.inertia"][1:]); Fs.append(g[f"{e}.force"][1:])
        for c in exp:
            exp[c].append(g[f"{e}.{c}"][1:])
    cat = lambda a: np.ascontiguousarray(np.concatenate(a))
    return cat(xs), cat(vs), cat(Is), cat(Fs), {c: cat(v) for c, v in exp.items()}, dt


def worst_errors(got, exp):
    out = {}
    for c, sl in (("world_accel", (slice(0, 3), slice(3, 6))), ("world_vel", (slice(0, 3), slice(3, 6))),
                  ("world_pos", (slice(0, 4), slice(4, 7))), ("force", (slice(0, 3), slice(3, 6)))):
        out[c] = max(gu.rel_err(got[c][:, s], exp[c][:, s]) for s in sl)
    return out


def test_golden_has_torque_and_nonuniform_inertia():
    x, v, I, F, exp, dt = teacher_forced_world()
    sat = slice(0, 100)
    assert np.all(np.abs(F[sat, :3]).max(axis=1) > 0)            # torque on every tick
    assert len({I[0, 0], I[0, 1], I[0, 2]}) == 3                 # Ixx != Iyy != Izz
    assert np.abs(exp["world_accel"][sat, :3]).max() > 0.1       # angular acceleration actually exercised


def test_oracle_semi_implicit_matches_cube_sat_golden():
    x, v, I, F, exp, dt = teacher_forced_world()
    w = orc.OracleWorld(x, v, I, simulation_time_step=dt, integrator=orc.SEMI_IMPLICIT,
                        ops=[(orc.EFF_WORLD_TORQUE, (), F[:, :3]), (orc.EFF_WORLD_FORCE, (), F[:, 3:])])
    w.step(1)
    err = worst_errors({c: getattr(w, c) for c in exp}, exp)
    print("cube-sat teacher-forced worst rel err:", err)
    assert err["force"] == 0.0
    for c, e in err.items():
        assert e < 1e-12, (c, e)


def test_oracle_calc_accel_matches_golden_rows_directly():
    """orc_calc_accel alone (six_dof.rs:137-146) on (F_r, I, x_{r-1}) -> world_accel row r."""
    x, v, I, F, exp, dt = teacher_forced_world()
    for i in range(100):   # the satellite rows: torque != 0, I_diag non-uniform, attitude far from identity later on
        a = orc.calc_accel(F[i], I[i], x[i])
        assert gu.rel_err(a[:3], exp["world_accel"][i, :3]) < 1e-12
        assert gu.rel_err(a[3:], exp["world_accel"][i, 3:]) < 1e-12
