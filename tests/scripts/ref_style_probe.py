"""A sim script in the reference's own spelling (`import elodin as el`, `import jax.numpy as jnp`, its decorators, archetypes
and `world.build / exec.run / exec.history`) — written for this repo's tests, not taken from the reference: a probe falling
through linear drag while a torque-free rotor spins, with one plain component integrated beside the Body.  Run as
`python -m elodin_amd.compat tests/scripts/ref_style_probe.py [ticks]`; prints one JSON line with the final state and the
recorded history length.  tests/test_gpu_compat_run.py checks it against closed forms."""
import json
import sys
import typing as ty
from dataclasses import dataclass

import elodin as el
import jax
import jax.numpy as jnp

SIM_RATE = 120.0
DRAG_PER_KG = 0.35            # 1 / s
G = 9.80665

Odometer = ty.Annotated[jax.Array, el.Component("odometer", el.ComponentType(el.PrimitiveType.F64, (1,)))]


@dataclass
class Instruments(el.Archetype):
    odometer: Odometer


@el.map
def gravity(f: el.Force, inertia: el.Inertia) -> el.Force:
    return f + el.SpatialForce(linear=inertia.mass() * jnp.array([0.0, 0.0, -G]))


@el.map
def drag(f: el.Force, v: el.WorldVel, inertia: el.Inertia) -> el.Force:
    return f + el.SpatialForce(linear=-DRAG_PER_KG * inertia.mass() * v.linear())


@el.map
def count_distance(o: Odometer, v: el.WorldVel) -> Odometer:
    return o + jnp.linalg.norm(v.linear()) * (1.0 / SIM_RATE)


def world() -> el.World:
    w = el.World()
    w.spawn([el.Body(world_pos=el.SpatialTransform(linear=jnp.array([0.0, 0.0, 1000.0])),
                     world_vel=el.SpatialMotion(angular=jnp.array([0.0, 0.0, 2.0]), linear=jnp.array([12.0, 0.0, 0.0])),
                     inertia=el.SpatialInertia(3.0, jnp.array([0.2, 0.2, 0.4]))),
             Instruments(jnp.array([0.0]))], name="probe")
    return w


if __name__ == "__main__":
    ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    w = world()
    exec = w.build(count_distance | el.six_dof(sys=gravity | drag), simulation_rate=SIM_RATE)
    exec.run(ticks)
    hist = exec.history(["probe.world_pos"])
    print(json.dumps({"ticks": ticks, "world_pos": [float(x) for x in exec.column_array("world_pos")[0]],
                      "world_vel": [float(x) for x in exec.column_array("world_vel")[0]],
                      "odometer": float(exec.column_array("odometer")[0][0]), "history_rows": len(hist["probe.world_pos"])}))
