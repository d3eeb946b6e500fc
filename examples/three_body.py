#!/usr/bin/env python3
"""examples/three-body/main.py of the reference, on the HIP backend: three bodies, six gravity edges, RK4 @120 Hz, the
gravity system written the reference's way (an edge_fold over a user edge component) with `elodin_amd.frontend`.  The
fold function is compiled into the pair kernels at build(); `build(builtin=True)` uses the built-in Newton functor.
Run on an MI355X:  python examples/three_body.py [ticks]"""
import sys
from pathlib import Path

import numpy

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import elodin_amd as builtin_api  # noqa: E402
import elodin_amd.frontend as el  # noqa: E402

la = el.np.linalg
G = 6.6743e-11

GravityEdge = el.Annotated[el.Edge, el.Component("gravity_edge", el.ComponentType.Edge)]


@el.dataclass
class GravityConstraint(el.Archetype):
    a: GravityEdge

    def __init__(self, a: el.EntityId, b: el.EntityId):
        self.a = GravityEdge(a, b)


@el.system
def gravity(graph: el.GraphQuery[GravityEdge], query: el.Query[el.WorldPos, el.Inertia]) -> el.Query[el.Force]:
    def gravity_fn(force, a_pos, a_inertia, b_pos, b_inertia):     # Newton's law on the edge a -> b, folded into a's force
        r = a_pos.linear() - b_pos.linear()
        m, M = a_inertia.mass(), b_inertia.mass()
        norm = la.norm(r)
        f = G * M * m * r / (norm * norm * norm)
        return el.Force(linear=force.force() - f)

    return graph.edge_fold(left_query=query, right_query=query, return_type=el.Force, init_value=el.Force(), fold_fn=gravity_fn)


BODIES = (("A", 0.8920281421, 0.9957939373), ("B", -0.6628498947, -1.6191613336), ("C", -0.2291782474, 0.6233673964))


def world_and_system(builtin: bool = False):
    w = el.World()
    a, b, c = (w.spawn(el.Body(world_pos=el.SpatialTransform(linear=numpy.array([x, 0.0, 0.0])),
                               world_vel=el.SpatialMotion(linear=numpy.array([0.0, vy, 0.0])),
                               inertia=el.SpatialInertia(1.0 / G)), name=name) for name, x, vy in BODIES)
    for x, y in ((a, b), (b, a), (a, c), (b, c), (c, a), (c, b)):   # spawn order = fold order
        w.spawn(GravityConstraint(x, y))
    return w, (builtin_api.six_dof(sys=builtin_api.gravity_newton(G)) if builtin else el.six_dof(sys=gravity))


def build(builtin: bool = False, history: bool = True):
    w, sys_ = world_and_system(builtin)
    return w.build(sys_, simulation_rate=120.0, history=history)


def main(ticks=1000):
    exec = build(history=False)
    exec.run(ticks)
    for name, row in zip("ABC", exec.column_array("world_pos")):
        print(name, row[4:])
    print("profile:", {k: round(v, 4) for k, v in exec.profile().items()})
    return exec


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
