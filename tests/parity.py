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


# parity = max_i |s_i - s_i^ref| / max(|s_i^ref|, tiny), element by element.  `tiny` is written down here: an
# ABSOLUTE floor of ELEMENT_FLOOR x the largest component of that entity's field vector — the form of the reference's own CI
# compare, math.isclose(rel_tol, abs_tol) (scripts/ci/compare_baseline_csv.py:209-214; its tolerances.json: 1e-4 / 1e-4), at
# rel_tol = 1e-9 and abs_tol = 1e-11 x the vector's size.  A floor is needed because a component crossing zero has no relative
# error of its own: measured at 65,536 bodies x 1,000 ticks, the vector-scaled error is 1.6e-13 while a quaternion component that
# happens to be 1e-4 of its vector shows 1.1e-9 "relative to itself" (gpurun r06b) — the same absolute error, divided by nothing.
ELEMENT_FLOOR = 1e-2
REPORT_FLOOR = 1e-12      # the near-floorless figure, reported beside the gated one (it grows with how close to zero a component gets)


def elementwise_rel_err(got, ref, floor=ELEMENT_FLOOR):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), floor * scale))) if got.size else 0.0


def state_errors_elementwise(hip, oracle_world, floor=ELEMENT_FLOOR):
    errs = {}
    for f in FIELDS:
        g, r = getattr(hip, f), getattr(oracle_world, f)
        cut = 4 if f == "world_pos" else 3
        errs[f] = max(elementwise_rel_err(g[:, :cut], r[:, :cut], floor), elementwise_rel_err(g[:, cut:], r[:, cut:], floor))
    return errs




class Worst:
    """Running worst case over checkpoints of SURVEY §8(d)'s parity figures, and the gate the BASELINE-size tests share, on ALL
    FOUR columns (world_pos, world_vel, world_accel, force):

      * per entity and field vector: |d| <= 1e-9 x the vector's largest component — `vector`;
      * element by element: |s_i - ref_i| <= 1e-9 x max(|ref_i|, ELEMENT_FLOOR x vector scale) — `element` (§8(d)'s formula, its
        `tiny` as above);
      * `element_raw`: the same with a 1e-12 floor, REPORTED: it measures how close to zero some component of 65,536 x 25 happens to
        be (world_accel reaches 4.5e-4 on a cancelling w x Iw term that is 1e-11 of its vector), not how far the paths are apart."""

    def __init__(self):
        self.vector, self.element, self.element_raw, self.checkpoints = {}, {}, {}, []

    def update(self, hip, ref, tick=None):
        for into, errs in ((self.vector, state_errors(hip, ref)), (self.element, state_errors_elementwise(hip, ref)),
                           (self.element_raw, state_errors_elementwise(hip, ref, REPORT_FLOOR))):
            for k, v in errs.items():
                into[k] = max(into.get(k, 0.0), v)
        self.checkpoints.append(tick)
        return self

    def check(self, what=""):
        line = (f"{what}: {len(self.checkpoints)} checkpoints {self.checkpoints}; vector-scaled {self.vector}; element-wise (floor {ELEMENT_FLOOR:g} x "
                f"vector) {self.element}; element-wise (floor {REPORT_FLOOR:g}, reported) {self.element_raw}")
        print(line)
        from pathlib import Path
        out = Path(__file__).resolve().parent.parent / "gpurun_out"
        if out.is_dir():                          # the GPU box's scratch: the measured figures come back with the run
            with open(out / "parity_figures.txt", "a") as f:
                f.write(line + "\n")
        assert max(self.vector.values()) < F64_RTOL, (what, self.vector)
        assert max(self.element.values()) < F64_RTOL, (what, "element-wise", self.element)


def to_oracle_ops(effectors):
    return [(e.kind, tuple(e.p), e.aux) for e in effectors]
