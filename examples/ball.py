#!/usr/bin/env python3
"""The reference's bouncing-ball script (examples/ball/sim.py) on the HIP backend, written against
`elodin_amd.frontend` — the same decorators, queries and archetypes as `import elodin as el`; only jax.numpy /
jax.lax / jax.random are spelled el.np / el.lax / el.random.  The effectors are compiled into the RK4 stage loop,
sample_wind and bounce into the same kernel ahead of it.  python examples/ball.py [ticks]"""
import sys
import typing
from dataclasses import field
from pathlib import Path

import numpy

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import elodin_amd.frontend as el  # noqa: E402

jnp, la = el.np, el.np.linalg
BALL_RADIUS, BOUNCINESS = 0.2, 0.85

Wind = typing.Annotated[el.Array, el.Component("wind", el.ComponentType(el.PrimitiveType.F64, (3,)),
                                               metadata={"element_names": "x,y,z"})]


@el.dataclass
class WindData(el.Archetype):
    seed: el.Seed = field(default_factory=lambda: numpy.int64(0))
    wind: Wind = field(default_factory=lambda: numpy.zeros(3))


@el.map
def sample_wind(s: el.Seed, _w: Wind) -> Wind:            # jax's own generator, bit for bit (threefry2x32 + erfinv)
    return el.random.normal(el.random.key(s), shape=(3,))


@el.map
def bounce(p: el.WorldPos, v: el.WorldVel) -> el.WorldVel:
    below_and_falling = el.lax.max(p.linear()[2], v.linear()[2]) < 0.0
    return el.lax.cond(below_and_falling,
                       lambda _: el.SpatialMotion(linear=v.linear() * jnp.array([1.0, 1.0, -1.0]) * BOUNCINESS),
                       lambda _: v, operand=None)


@el.map
def gravity(f: el.Force, inertia: el.Inertia) -> el.Force:
    return f + el.SpatialForce(linear=jnp.array([0.0, 0.0, -9.81]) * inertia.mass())


@el.map
def apply_drag(w: Wind, v: el.WorldVel, f: el.Force) -> el.Force:
    fluid = w - v.linear()
    speed = la.norm(fluid)
    cd, rho, area = 0.5, 1.225, 2 * 3.1415 * BALL_RADIUS ** 2
    drag = 0.5 * (cd * rho * speed ** 2 * area)
    return el.SpatialForce(linear=f.force() + drag * (fluid / speed))


def system():
    return sample_wind | bounce | el.six_dof(sys=gravity | apply_drag)


def world(seed: int = 0) -> el.World:
    w = el.World()
    w.spawn([el.Body(world_pos=el.SpatialTransform(linear=numpy.array([0.0, 0.0, 6.0]))), WindData(seed=numpy.int64(seed))],
            name="ball")
    return w


def build(seed=0):
    return world(seed).build(system(), simulation_rate=120.0)


def main(ticks=600):
    exec = build()
    exec.run(ticks)
    track = exec.history("ball.world_pos")["ball.world_pos"]          # one row per tick, the spawned state first
    lowest = float(track[::20, 6].min())
    print("ball at", track[-1, 4:], "lowest z", lowest, "rows", len(track))
    return exec, lowest


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 600)
