#!/usr/bin/env python3
"""configs[1] at one tick per launch, graph replay, 4,096-launch batches, interleaved passes: the hand-written pipe (PipeStatic<gravity,
body_torque>) against the same effectors written as user code and generated into the step kernel (tools/bench_legs.py generated_leg),
and their final states against each other.   gpurun -- 'python tools/k1_generated_ab.py'"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import elodin_amd as ea
from elodin_amd import dsl, workloads

n = 65536
np_ = dsl.np


@dsl.effector
def gravity(force, inertia):
    return force + dsl.SpatialForce(linear=np_.array([0.0, 0.0, -9.81]) * inertia.mass())


@dsl.effector(body_torque=3)
def rcs(force, pos, body_torque):
    return force + dsl.SpatialForce(torque=pos.angular() @ body_torque)


w = workloads.independent_bodies(n)
hand = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], entity_ids=w["entity_ids"], simulation_time_step=workloads.DT_120HZ,
                  effectors=workloads.gravity_torque_effectors(w["body_torque"]), use_graph=True)
gen = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                 effectors=gravity | rcs, columns={"body_torque": w["body_torque"]}, use_graph=True)
for ex in (hand, gen):
    ex.prepare(4096)
    ex.invoke_batch(512)
res = {"hand-written": [], "generated": []}
for _ in range(5):
    for name, ex in (("hand-written", hand), ("generated", gen)):
        t = ex.invoke_batch(4096)
        res[name].append(t.kernel_device_ms / 4096 * 1e3)
for name, v in res.items():
    print(f"{name}: us per tick " + " ".join(f"{x:.3f}" for x in v) + f"   min {min(v):.3f}")
hand.download()
gen.download()
for c in ("world_pos", "world_vel", "world_accel", "force"):
    a, b = np.asarray(getattr(hand, c)), np.asarray(getattr(gen, c))
    print(c, "bit-identical" if np.array_equal(a, b) else f"max abs diff {np.max(np.abs(a - b)):.3e}")
hand.close()
gen.close()
