"""StableHLO text ingestion through the generated gfx950 kernel: the reference's op-test known answers
(libs/cranelift-mlir/tests/ops.rs -> tests/golden/stablehlo_ops.json) parsed by elodin_amd/stablehlo.py, every module's @main a
per-entity system, batched a few dozen systems to a program (one hipcc run each), stepped one tick on the GPU.  The CPU twin
(tests/test_stablehlo_ingest.py) walks the same DAGs with numpy."""
import numpy as np
import pytest

import elodin_amd as ea
from elodin_amd import _lib as L
from elodin_amd import dsl, workloads
from tests import stablehlo_util as U

pytestmark = pytest.mark.gpu
RUNNABLE = [c for c in U.CASES if c["name"] not in U.UNSUPPORTED and c["name"] not in U.BEYOND_F64_INTEGERS]
GROUP = 24
GROUPS = [RUNNABLE[k:k + GROUP] for k in range(0, len(RUNNABLE), GROUP)]


@pytest.mark.parametrize("group", range(len(GROUPS)))
def test_reference_op_test_known_answers_through_the_generated_kernel(group):
    systems, columns, expects = [], {}, []
    n = 70                                                           # a full wave and a ragged one, every row the same case data
    for k, case in enumerate(GROUPS[group]):
        system, values, expect = U.build(case, prefix=f"k{k}_")
        systems.append(system)
        for nm, v in values.items():
            columns[nm] = np.tile(v.reshape(1, -1), (n, 1))
        for nm, (w, _) in expect.items():
            columns[nm] = np.zeros((n, w))
        expects.append((case["name"], expect))
    assert len(columns) <= dsl.MAX_PROGRAM_COLUMNS
    w = workloads.independent_bodies(n)
    hip = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.INTEGRATOR_NONE, effectors=dsl.Program(systems, dsl.Pipe([]), []),
                     columns=columns)
    hip.run(1)
    for name, expect in expects:
        for nm, (width, exp) in expect.items():
            got = np.asarray(hip._aux[nm], dtype=np.float64)
            U.check(name, got[0], width, exp, 1e-9)
            assert np.array_equal(got, np.repeat(got[:1], n, axis=0), equal_nan=True), (name, nm)      # lanes do not interact
    print(f"StableHLO ingestion, group {group}: {len(expects)} of the reference's op tests through one generated kernel")
    hip.close()
