#!/usr/bin/env python3
"""BASELINE configs[1] as a whole-world StableHLO module, one lane per entity: the reference's arithmetic operation for operation
against stablehlo.world_system(arith="relaxed") (finite values assumed, one division per denominator, a * b + c contracted) — time per
tick at one tick per launch (graph replay, best of 5 batches of 1,024) and at 64 ticks per launch.  Their errors against the oracle:
tests/test_gpu_stablehlo_world.py::test_relaxed_arithmetic_module_on_the_gpu.  gpurun -- 'python tools/world_relaxed_ab.py out.json'"""
import json
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import elodin_amd as ea
from elodin_amd import _lib as L
from elodin_amd import dsl, workloads
from elodin_amd import stablehlo as sh
from tests.golden import hlo_world_builder as hb


def columns(w, n):
    return {"hlo_tick": np.zeros((n, 1)), "hlo_simulation_time_step": np.full((n, 1), workloads.DT_120HZ), "hlo_world_pos": w["world_pos"].copy(),
            "hlo_world_vel": w["world_vel"].copy(), "hlo_world_accel": np.zeros((n, 6)), "hlo_force": np.zeros((n, 6)), "hlo_inertia": w["inertia"].copy(),
            "hlo_torque": w["body_torque"].copy()}


def executor(arith, n, one_world=False, **kw):
    text, slots = hb.independent_bodies_world(n)
    system, manifest = sh.world_system(text, slots, mode="lane", arith=arith, one_world=one_world)
    w = workloads.independent_bodies(n)
    prog = dsl.Program([system], dsl.Pipe([]), [])
    ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.INTEGRATOR_NONE, effectors=prog, columns=columns(w, n), **kw)
    return ex, w, manifest


def timing(arith, one_world=False, n=65536):
    ex, _, manifest = executor(arith, n, one_world=one_world, use_graph=True)
    ex.prepare(1024)
    ex.invoke_batch(1024)
    batches = [ex.invoke_batch(1024) for _ in range(5)]
    k1 = min(t.kernel_device_ms for t in batches) / 1024 * 1e3
    ex.set_ticks_per_launch(64)
    ex.invoke_batch(64)
    k64 = min(ex.invoke_batch(64 * 32).kernel_device_ms for _ in range(3)) / (64 * 32) * 1e3
    ex.close()
    return {"us_per_tick_k1": round(k1, 3), "entity_steps_per_s_k1": round(n / k1 * 1e6, 1), "us_per_tick_k64": round(k64, 4),
            "entity_steps_per_s_k64": round(n / k64 * 1e6, 1), "graph_launches": int(batches[0].graph_launches)}


def main():
    out = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "entities": 65536}
    variants = [("reference", {}), ("reference_one_world", {}), ("relaxed", {}), ("relaxed_one_world", {}), ("relaxed_ieee_reciprocals", {"SIXDOF_RELAXED_IEEE_RCP": "1"})]
    for name, env in variants:
        os.environ.pop("SIXDOF_RELAXED_IEEE_RCP", None)
        os.environ.update(env)
        arith = "reference" if name.startswith("reference") else "relaxed"
        out[name] = timing(arith, name.endswith("one_world"))
        print(name, json.dumps(out[name]), flush=True)
    os.environ.pop("SIXDOF_RELAXED_IEEE_RCP", None)
    path = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "world_relaxed_ab.json"
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
