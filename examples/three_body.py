#!/usr/bin/env python3
"""examples/three-body/main.py of the reference, on the HIP backend: three bodies, six gravity edges, RK4 @120 Hz.
Run on an MI355X:  python examples/three_body.py [ticks]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import elodin_amd as el  # noqa: E402
from elodin_amd import dsl  # noqa: E402

np = dsl.np
G = 6.6743e-11


@dsl.edge_fold
def gravity_fn(force, a_pos, a_inertia, b_pos, b_inertia):
    """The reference example's fold function (examples/three-body/main.py:61-70), compiled into the pair kernels at
    build(); `el.gravity_newton(G)` is the equivalent built-in op."""
    r = a_pos.linear() - b_pos.linear()
    m = a_inertia.mass()
    M = b_inertia.mass()
    norm = np.linalg.norm(r)
    f = G * M * m * r / (norm * norm * norm)
    return dsl.SpatialForce(linear=force.force() - f)


def build(builtin: bool = False):
    w = el.World()
    a = w.spawn(el.Body(world_pos=el.SpatialTransform(linear=[0.8920281421, 0.0, 0.0]),
                        world_vel=el.SpatialMotion(linear=[0.0, 0.9957939373, 0.0]),
                        inertia=el.SpatialInertia(1.0 / G)), name="A")
    b = w.spawn(el.Body(world_pos=el.SpatialTransform(linear=[-0.6628498947, 0.0, 0.0]),
                        world_vel=el.SpatialMotion(linear=[0.0, -1.6191613336, 0.0]),
                        inertia=el.SpatialInertia(1.0 / G)), name="B")
    c = w.spawn(el.Body(world_pos=el.SpatialTransform(linear=[-0.2291782474, 0.0, 0.0]),
                        world_vel=el.SpatialMotion(linear=[0.0, 0.6233673964, 0.0]),
                        inertia=el.SpatialInertia(1.0 / G)), name="C")
    for x, y in ((a, b), (b, a), (a, c), (b, c), (c, a), (c, b)):   # spawn order = fold order
        w.spawn(el.GravityEdge(x, y))
    return w.build(el.six_dof(sys=el.gravity_newton(G) if builtin else gravity_fn), simulation_rate=120.0)


def main(ticks=1000):
    exec = build()
    exec.run(ticks)
    for name, row in zip("ABC", exec.column_array("world_pos")):
        print(name, row[4:])
    print("profile:", {k: round(v, 4) for k, v in exec.profile().items()})
    return exec


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
