"""Synthetic worlds for the BASELINE.json configs (inputs only — no compute here).

Config 2: N independent 6DOF bodies under constant world-frame gravity
(`f += (0,0,-9.81) m`, examples/ball/sim.py:57-59) plus a constant body-frame torque
(`tau += q @ tau_b`, the pattern of examples/apollo-lander/sim.py:396-398), RK4 f64, dt = 1/120 s
ns-quantised.  Seeded with numpy PCG64(0x5EED) so every rank / test regenerates identical rows.
"""
from __future__ import annotations

import numpy as np

from . import _lib as L
from .exec import Effector

SEED = 0x5EED
DT_120HZ = 0.008333333  # Duration::from_secs_f64(1/120).as_secs_f64(), world_builder.rs:221


def independent_bodies(n: int, seed: int = SEED, dtype=np.float64, first_row: int = 0):
    """Rows [first_row, first_row+n) of the config-2 world.  Row i depends only on (seed, i),
    so shards generated on different ranks concatenate to the single-GPU world bit-for-bit."""
    rows = np.arange(first_row, first_row + n, dtype=np.uint64)
    # draw in fixed blocks of 4096 rows, each block seeded by (seed, block index), so a shard's rows
    # do not depend on how the world is partitioned.
    B = 4096
    b0, b1 = first_row // B, (first_row + n + B - 1) // B
    chunks = []
    for b in range(b0, b1):
        g = np.random.Generator(np.random.PCG64([seed, b]))
        q = g.normal(size=(B, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        p = g.uniform(-1.0e3, 1.0e3, size=(B, 3))
        w = g.normal(0.0, 0.5, size=(B, 3))
        v = g.normal(0.0, 10.0, size=(B, 3))
        m = g.uniform(1.0, 1.0e3, size=(B, 1))
        I = g.uniform(0.1, 10.0, size=(B, 3)) * m
        tb = g.uniform(-1.0, 1.0, size=(B, 3))
        chunks.append(np.concatenate([q, p, w, v, I, np.zeros((B, 3)), m, tb], axis=1))
    allrows = np.concatenate(chunks, axis=0)
    sel = allrows[first_row - b0 * B: first_row - b0 * B + n]
    world_pos = np.ascontiguousarray(sel[:, 0:7], dtype=dtype)
    world_vel = np.ascontiguousarray(sel[:, 7:13], dtype=dtype)
    inertia = np.ascontiguousarray(sel[:, 13:20], dtype=dtype)
    body_torque = np.ascontiguousarray(sel[:, 20:23], dtype=dtype)
    entity_ids = rows + np.uint64(1)  # id 0 = Globals (world.rs:193-196)
    return dict(world_pos=world_pos, world_vel=world_vel, inertia=inertia, body_torque=body_torque,
                entity_ids=entity_ids)


def gravity_torque_effectors(body_torque):
    return [Effector(L.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -9.81)),
            Effector(L.EFF_BODY_TORQUE, (), aux_name="body_torque", aux=body_torque)]
