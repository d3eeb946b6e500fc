"""A/B of the K=1 step launch: one entity per lane (64 rows per wave) vs half-filled waves (32 rows per wave, twice the waves).

    python tools/step_half_waves_ab.py [n ...]      # default 16384 32768 65536 131072 262144
At 65,536 bodies a full-width launch is 1,024 waves = ONE per SIMD: every wave walks load -> math -> store alone and the SIMD
idles while its only wave waits.  With 32 rows per wave the same launch is 2,048 waves, two per SIMD, which overlap each
other's waits.  us/launch from HIP events around graph-replayed batches (best of 5); both variants must give the same bits.
Needs the A/B library (`make -C elodin_amd/csrc ab`): bit 9 of SIXDOF_STREAMING selects the half-wave instantiation there.
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

for n in [int(x) for x in sys.argv[1:]] or (16384, 32768, 65536, 131072, 262144):
    w = workloads.independent_bodies(n)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    out, sig = [], []
    for name, code in (("64 rows/wave", 1), ("32 rows/wave", 1 | 512)):
        os.environ["SIXDOF_STREAMING"] = str(code)
        ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=eff, use_graph=True)
        ex.invoke_batch(256)
        best = min(ex.invoke_batch(2048).kernel_device_ms / 2048 for _ in range(5))
        ex.download()
        sig.append((ex.world_pos.copy(), ex.world_vel.copy(), ex.world_accel.copy(), ex.force.copy()))
        out.append(f"{name}: {best * 1e3:6.2f} us ({384 * n / best / 1e6:5.0f} GB/s)")
        ex.close()
    same = all(np.array_equal(a, b) for a, b in zip(*sig))
    print(f"n={n:7d}  " + "   ".join(out) + ("   same bits" if same else "   DIFFERENT RESULTS"))
os.environ.pop("SIXDOF_STREAMING", None)
