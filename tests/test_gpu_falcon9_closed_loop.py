"""Whole Falcon 9 ascents, closed loop, through the generated gfx950 kernel against flights flown by the REFERENCE's own plant,
sensors and bridge with an independent C restatement of its Rust flight software (tests/falcon9_closed_loop_util.py,
tests/golden/make_falcon9_closed_loop.py): the calibrated default row and rows of the example's own LHS plan, pad to
MECO + 3 s (150,000-170,000 ticks).  f64: every phase transition on the SAME tick, every component and the navigator's
private state at every checkpoint to 1e-9 of its scale.  f32 (config 5's arithmetic): a stated bound."""
import numpy as np
import pytest

from elodin_amd.models import falcon9 as f9
from tests import falcon9_closed_loop_util as cu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not cu.FLIGHTS, reason="closed-loop fixture not generated")]
ROWS = sorted(cu.FLIGHTS, key=int)


def _exec(flights, dtype, ticks_per_launch, fast_math=False):
    local = np.dtype(dtype) == np.float32
    built = [cu.initial_columns(fl) for fl in flights]
    params = np.concatenate([p for p, _ in built], axis=0)
    cols = {k: np.concatenate([c[k] for _, c in built], axis=0) for k in built[0][1]}
    if local:   # f32 state integrates pad-relative coordinates (an f32 ECEF metre has a 0.5 m ulp)
        cols["world_pos"] = cols["world_pos"].copy()
        cols["world_pos"][:, 4:] -= f9.pad_ecef()
    return f9.AscentExec(params, dtype=dtype, local_origin=local, columns=cols, ticks_per_launch=ticks_per_launch, fast_math=fast_math)


def _getter(ex, i):
    return lambda name: np.asarray(ex.column(name), dtype=np.float64)[i:i + 1]


@pytest.mark.parametrize("ticks_per_launch", [1000, 1])
def test_generated_kernel_flies_the_reference_ascents_f64(ticks_per_launch):
    """All fixture rows as ONE executor (one lane each).  K = 1 (every tick its own launch) flies the first 3 s only
    (navigator initialisation, ignition, liftoff, the radar altimeter: the f64 program is the slow scratch-image build)."""
    flights = [cu.FLIGHTS[r] for r in ROWS]
    ex = _exec(flights, np.float64, ticks_per_launch)
    horizon = 3_000 if ticks_per_launch == 1 else max(fl["ticks"] for fl in flights)
    # every tick any row has something to check on: checkpoints, and the tick before each phase transition
    stops = sorted({c["tick"] for fl in flights for c in fl["checkpoints"] if c["tick"] <= horizon}
                   | {t - 1 for fl in flights for t in fl["transitions"].values() if t - 1 <= horizon})
    worst, done, n_cp, n_tr = {}, 0, 0, 0
    for stop in stops:
        ex.run(stop - done)
        done = stop
        for i, fl in enumerate(flights):
            if stop > fl["ticks"]:
                continue
            for ph, t in fl["transitions"].items():      # the transition happens on exactly this tick: not one exchange earlier
                if t - 1 == stop:
                    assert ex.column("fsw_state")[i, 0] == float(ph) - 1.0, (ROWS[i], ph, stop)
                    n_tr += 1
            for cp in fl["checkpoints"]:
                if cp["tick"] == stop:
                    errs = cu.compare(fl, cp, _getter(ex, i))
                    assert errs["fsw.phase"] == 0.0, (ROWS[i], stop, cp["state"]["fsw"]["phase"], ex.column("fsw_state")[i, 0])
                    for k, e in errs.items():
                        if e > worst.get(k, (0.0,))[0]:
                            worst[k] = (e, ROWS[i], stop, cu.DETAIL.get(k.replace("fsw.", "")))
                    n_cp += 1
    ex.close()
    top = sorted(worst.items(), key=lambda kv: -kv[1][0])[:12]
    print(f"f64 K={ticks_per_launch}: {len(flights)} flights, {n_cp} checkpoints, {n_tr} transitions on their tick; worst of {len(worst)} quantities:",
          ", ".join(f"{k} {e:.1e} (row {r} tick {t}; abs, scale, part {d})" for k, (e, r, t, d) in top))
    assert n_cp >= (60 if ticks_per_launch > 1 else 35) and n_tr >= (4 * len(flights) if ticks_per_launch > 1 else len(flights))
    assert max(v[0] for v in worst.values()) < 1e-9, top


@pytest.mark.parametrize("fast_math", [False, True])
def test_generated_kernel_f32_tracks_the_reference_ascents(fast_math):
    """Config 5's arithmetic.  The reference has no f32 six_dof, so the bound is this build's: f32 state through a 150 s
    closed loop reaches every phase within 0.25 s of the reference flight's transition and arrives at MECO within 1 % in
    altitude and speed and 0.5 deg in flight-path angle."""
    flights = [cu.FLIGHTS[r] for r in ROWS]
    ex = _exec(flights, np.float32, 1000, fast_math=fast_math)
    horizon = max(fl["ticks"] for fl in flights)
    seen = [dict() for _ in flights]
    done = 0
    while done < horizon:
        ex.run(10)                                        # one guidance exchange at a time: catch the transition ticks
        done += 10
        ph = ex.column("fsw_state")[:, 0]
        for i in range(len(flights)):
            seen[i].setdefault(int(ph[i]), done - 9)       # the exchange of this batch ran on tick done - 9
    res = ex.result
    out = []
    for i, fl in enumerate(flights):
        for p, t in fl["transitions"].items():
            got = seen[i].get(int(p))
            assert got is not None and abs(got - t) <= 250, (ROWS[i], p, t, got)      # measured: <= 20 ticks (two exchanges)
        meco_cp = next(c for c in fl["checkpoints"] if c["tick"] == fl["transitions"]["4"])["state"]
        alt_ref, v_ref = meco_cp["altitude_geodetic"][0], meco_cp["ground_speed"][0]
        # ascent_metrics latch MECO on the tick after the cutoff command: [3] t, [4] altitude, [5] speed
        assert abs(res[i, 4] - alt_ref) / alt_ref < 1e-2 and abs(res[i, 5] - v_ref) / v_ref < 1e-2, (ROWS[i], res[i], alt_ref, v_ref)
        out.append(f"row {ROWS[i]}: MECO tick {seen[i].get(4)} vs {fl['transitions']['4']}, alt {res[i, 4] / 1e3:.2f} vs {alt_ref / 1e3:.2f} km")
    ex.close()
    print(f"f32 fast_math={fast_math}: " + "; ".join(out))
