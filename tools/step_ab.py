"""A/B of the K=1 step launch at BASELINE config-2 sizes (VERDICT r1 #5a): cache policy of the loads x the stores.

    python tools/step_ab.py [n ...]      # default 65536
Prints us/launch (HIP events around graph-replayed batches, best of 5) and algorithmic GB/s (384 B x n per launch).
Policy code = load * 8 + store (csrc/step_kernel.hpp); the product library ships 0, 1 and 9 and ignores every other code.
The rest of the matrix (incl. the UNSAFE sc1 stores) exists in the A/B library only: `make -C elodin_amd/csrc ab`, which
this script loads through SIXDOF_LIBRARY.
"""
import os
import sys

_AB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "elodin_amd", "libsixdof_hip_ab.so")
if not os.path.exists(_AB):
    sys.exit("build the A/B library first: make -C elodin_amd/csrc ab")
os.environ.setdefault("SIXDOF_LIBRARY", _AB)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import elodin_amd as ea  # noqa: E402
from elodin_amd import workloads  # noqa: E402

LOADS = {0: "ld plain", 1: "ld nt"}
STORES = {0: "st plain", 1: "st nt", 2: "st sc1", 3: "st sc0 sc1", 4: "st sc1 nt"}
for n in [int(x) for x in sys.argv[1:]] or (65536,):
    w = workloads.independent_bodies(n)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    base = None
    print(f"# n={n}: us/launch (GB/s algorithmic)")
    print(f"{'':10s}" + "".join(f"{v:>20}" for v in STORES.values()))
    for ld, ld_name in LOADS.items():
        cells = []
        for st in STORES:
            os.environ["SIXDOF_STREAMING"] = str(ld * 8 + st)
            ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                            effectors=eff, use_graph=True)
            reps = 2048 if n <= (1 << 18) else 64
            ex.invoke_batch(reps // 8)
            best = min(ex.invoke_batch(reps).kernel_device_ms / reps for _ in range(5))
            ex.download()
            sig = (ex.world_pos.copy(), ex.world_vel.copy(), ex.world_accel.copy(), ex.force.copy())
            if base is None:
                base = sig
            same = all(np.array_equal(a, b) for a, b in zip(base, sig))   # every variant must give the same bits
            cells.append(f"{best * 1e3:8.2f} ({384 * n / best / 1e6:5.0f}){'' if same else ' DIFF'}")
            ex.close()
        print(f"{ld_name:10s}" + "".join(f"{c:>20}" for c in cells))
    # early per-column stores (the shipped form) vs one flush after the tick, at the shipped policy 9
    for late in (0, 256):
        os.environ["SIXDOF_STREAMING"] = str(9 + late)
        ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                        effectors=eff, use_graph=True)
        reps = 2048 if n <= (1 << 18) else 64
        ex.invoke_batch(reps // 8)
        best = min(ex.invoke_batch(reps).kernel_device_ms / reps for _ in range(5))
        print(f"policy 9, {'one flush after the tick' if late else 'columns stored as they complete'}: {best * 1e3:.2f} us ({384 * n / best / 1e6:.0f} GB/s)")
        ex.close()
os.environ.pop("SIXDOF_STREAMING", None)
