#!/usr/bin/env python3
"""examples/ball of the reference with its effectors written as user code (elodin_amd.dsl), compiled into the fused
step kernel at build time; `bounce` runs as a generated pre-system.  python examples/ball.py [ticks]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import elodin_amd as el  # noqa: E402
from elodin_amd import dsl  # noqa: E402

np = dsl.np
BALL_RADIUS, BOUNCINESS = 0.2, 0.85


@dsl.effector
def gravity(force, inertia):
    return force + dsl.SpatialForce(linear=np.array([0.0, 0.0, -9.81]) * inertia.mass())


@dsl.effector
def apply_drag(wind, vel, force):
    fluid = wind - vel.linear()
    speed = np.linalg.norm(fluid)
    drag = 0.5 * (0.5 * 1.225 * speed ** 2 * (2 * 3.1415 * BALL_RADIUS ** 2))
    return dsl.SpatialForce(linear=force.force() + drag * (fluid / speed))


@dsl.system
def bounce(pos, vel):                                   # examples/ball/sim.py:65-73, same lax.cond
    return {"world_vel": dsl.lax.cond(
        dsl.lax.max(pos.linear()[2], vel.linear()[2]) < 0.0,
        lambda _: dsl.SpatialMotion(linear=vel.linear() * np.array([1.0, 1.0, -1.0]) * BOUNCINESS),
        lambda _: vel,
        operand=None)}


@dsl.system
def sample_wind(seed, wind):                            # examples/ball/sim.py:92-94: jax's own generator, same bits
    return {"wind": dsl.random.normal(dsl.random.key(seed), shape=(3,))}


def build(seed=0):
    """examples/ball/sim.py:120-133: WindData(seed) + Body at 6 m; sample_wind | bounce | six_dof(gravity | apply_drag)."""
    w = el.World()
    w.spawn([el.Body(world_pos=el.SpatialTransform(linear=[0.0, 0.0, 6.0])), el.C("seed", [float(seed)]),
             el.C("wind", [0.0, 0.0, 0.0])], name="ball")
    return w.build(sample_wind | bounce | el.six_dof(sys=gravity | apply_drag), simulation_rate=120.0)


def main(ticks=600):
    exec = build()
    lowest = 1e9
    for _ in range(ticks // 20):
        exec.run(20)
        lowest = min(lowest, exec.column_array("world_pos")[0, 6])
    print("ball at", exec.column_array("world_pos")[0, 4:], "lowest z", lowest)
    return exec, lowest


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 600)
