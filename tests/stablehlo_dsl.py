"""The reference's StableHLO coverage example (examples/stablehlo/sim.py:120-345) written against elodin_amd.dsl — same
expressions in the same order (the int64 bitwise system included: integer components are integral values in float columns); static shape manipulation (broadcast / concat / slice / reshape
/ transpose / flip) is done on the traced vectors directly.  Used by the golden-CSV tests.  TEST INFRASTRUCTURE."""
from elodin_amd import dsl

np_ = dsl.np
lax = dsl.lax

INITIAL = {"math_state": [0.5, 1.0, -0.3, 2.0], "sort_state": [3.0, 1.0, 4.0, 1.5, 2.0, 5.0, 0.5, 2.5],
           "shape_state": [1.0, 2.0, 3.0, 4.0], "control_state": [5.0, 1.0, -0.5, 0.0], "linalg_state": [1.0, 2.0, 3.0, 4.0],
           "convert_state": [1.5, -2.7, 0.0, 100.0], "linalg2_state": [4.0, 2.0, 2.0, 3.0],
           "bitwise_state": [float(0xA5), float(0x3C), float(0xFF), float(0x01)]}                    # sim.py:72-113 defaults


@dsl.system
def math_step(math_state):                                        # sim.py:120-165
    x = math_state
    r = np_.zeros(4)
    r = r + np_.sin(x) + np_.cos(x)
    r = r + np_.tanh(x)
    r = r + dsl.Vec([np_.arctan2(a, 1.0) for a in x])
    r = r + np_.exp(x * 0.1)
    r = r + np_.log(np_.abs(x) + 1.0)
    r = r + np_.log1p(np_.abs(x))
    r = r + np_.expm1(x * 0.01)
    r = r + np_.sqrt(np_.abs(x) + 1.0)
    r = r + lax.rsqrt(np_.abs(x) + 1.0)
    r = r + np_.cbrt(np_.abs(x) + 1.0)
    r = r + np_.power(np_.abs(x) + 1.0, 0.5)
    r = r + np_.floor(x) + np_.ceil(x)
    r = r + dsl.Vec([np_.sign(a) for a in x]) + np_.round(x)
    r = r + np_.abs(x)
    safe_x = np_.clip(x * 0.1, -0.99, 0.99)
    r = r + np_.arcsin(safe_x)
    r = r + np_.arccos(safe_x)
    r = r + np_.arctan(x * 0.1)
    r = r + np_.sinh(x * 0.1) + np_.cosh(x * 0.1)
    r = r + np_.erfc(x * 0.1)
    r = r + np_.clip(x, -2.0, 2.0)
    mask = np_.where(np_.isfinite(r), np_.ones(4), np_.zeros(4))
    return {"math_state": r * mask * 0.01}


@dsl.system
def sort_step(sort_state):                                        # sim.py:173-176
    return {"sort_state": np_.sort(sort_state) * 0.99 + 0.01}


@dsl.system
def shape_step(shape_state):                                      # sim.py:184-207
    x = shape_state
    s = x + x + x                                                 # sum over the broadcast (3, 4) matrix, axis 0
    c = np_.concatenate([s, s[:2]])
    sl = c[1:5]
    t = dsl.Vec([sl[0], sl[2], sl[1], sl[3]])                     # reshape(2, 2) -> transpose -> flatten
    r = np_.flip(t)
    return {"shape_state": r * 0.5 + np_.arange(4) * 0.01}


@dsl.system
def control_step(control_state):                                  # sim.py:215-237
    state = control_state
    result, _ = lax.while_loop(lambda c: c[1] < 5.0, lambda c: (c[0] * 0.9 + 0.1, c[1] + 1.0), (state[0], 0.0 * state[0]))
    idx = np_.trunc(np_.abs(state[1]) % 3.0)
    branch = lax.switch(idx, [lambda: state * 0.95, lambda: state * 1.05, lambda: state + 0.01])
    return {"control_state": np_.array([result, branch[0], branch[1], state[3] + 0.01])}


@dsl.system
def linalg_step(linalg_state):                                    # sim.py:260-275
    x = linalg_state
    mat = np_.outer(x[:2], x[2:])
    mv = np_.matvec(mat, x[2:])
    s, mx, mn = np_.sum(x), np_.max(x), np_.min(x)
    rem = np_.remainder(x, np_.array([1.5, 1.5, 1.5, 1.5]))
    return {"linalg_state": np_.array([mv[0] * 0.01 + s * 0.001, mx, mn, rem[0]])}


@dsl.system
def convert_step(convert_state):                                  # sim.py:283-304
    x = convert_state
    back = np_.trunc(x)                                           # f64 -> i32 -> f64 round trip
    selected = np_.where(x > 0.0, x, -x)
    updated = x.set(0, selected[1]).set(2, back[3])
    combined = np_.minimum(np_.maximum(-x, updated), np_.ones(4) * 50.0)
    return {"convert_state": combined * 0.99}


@dsl.system
def linalg2_step(linalg2_state):                                  # sim.py:312-323: 2x2 Cholesky + lower-triangular solve
    st = linalg2_state
    a00, a01, a11 = np_.abs(st[0]) + 1.0, st[1] * 0.1, np_.abs(st[2]) + 1.0
    l00 = np_.sqrt(a00)
    l10 = a01 / l00
    l11 = np_.sqrt(a11 - l10 * l10)
    x0 = st[3] / l00
    x1 = (1.0 - l10 * x0) / l11
    return {"linalg2_state": np_.array([l00, l11, x0, x1])}


@dsl.system
def bitwise_step(bitwise_state):                                  # sim.py:243-251
    r = np_.bitwise_xor(bitwise_state, 255.0)
    r = np_.bitwise_or(r, 15.0)
    r = np_.bitwise_and(r, 4095.0)
    r = np_.left_shift(r, 1.0)
    return {"bitwise_state": lax.shift_right_logical(r, 2.0)}


SYSTEMS = [math_step, sort_step, shape_step, control_step, bitwise_step, linalg_step, convert_step, linalg2_step]    # sim.py:343-353 order
