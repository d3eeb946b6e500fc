#!/usr/bin/env python3
"""All-pairs softened gravity (examples/n-body/sim.py:344-369 fold) on N bodies — the size the reference cannot
reach (it materialises O(N^2) gathered operands).  python examples/nbody.py [bodies] [ticks]"""
import sys
from pathlib import Path

import numpy

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import elodin_amd as el  # noqa: E402

K_SQUARED = 2.9591220828e-4 / (86_400.0 * 86_400.0)   # AU^3 / (solar mass * s^2)
SOFTENING_AU2 = 1.0e-10


def build(n=16384, seed=7):
    rng = numpy.random.default_rng(seed)
    u = rng.uniform(0.05, 0.95, n)
    d = rng.normal(size=(n, 3))
    p = d / numpy.linalg.norm(d, axis=1, keepdims=True) * (1.0 / numpy.sqrt(u ** (-2.0 / 3.0) - 1.0))[:, None]
    m = rng.uniform(1e-9, 1e-3, n)
    w = el.World()
    for k in range(n):
        w.spawn(el.Body(world_pos=el.SpatialTransform(linear=p[k]), inertia=el.SpatialInertia(m[k])))
    # edge_component=None: the complete graph without materialising N(N-1) edge entities
    return w.build(el.six_dof(sys=el.gravity_softened(K_SQUARED, SOFTENING_AU2, edge_component=None)),
                   simulation_rate=1.0 / 3600.0), m


def main(n=16384, ticks=24):
    exec, m = build(n)
    exec.run(ticks)
    prof = exec.profile()
    mom = (m[:, None] * exec.column_array("world_vel")[:, 3:]).sum(axis=0)
    print(f"{n} bodies, {ticks} ticks of 1 h: {prof['tick']:.3f} ms/tick, net momentum {mom}")
    return exec, mom


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 16384, int(sys.argv[2]) if len(sys.argv) > 2 else 24)
