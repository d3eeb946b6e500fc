"""Loaders for the committed golden fixtures (tests/golden/*.csv, made by make_golden.py)."""
import csv
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
SUFFIX = {"world_pos": 7, "world_vel": 6, "world_accel": 6, "force": 6, "inertia": 7, "wind": 3}


def load(name: str):
    """-> dict prefix ('a.world_pos', 'globals.tick', ...) -> float64 array [rows, width]."""
    with open(GOLDEN / f"{name}.csv", newline="", encoding="utf-8") as f:
        rows = list(csv.reader(f))
    header = rows[0][1:]
    data = rows[1:]
    groups, order = {}, []
    for j, h in enumerate(header):
        ent, rest = h.split(".", 1)
        comp = None
        for c in sorted(SUFFIX, key=len, reverse=True):
            if rest.startswith(c + "_"):
                comp = c
        if comp is None:
            comp = rest.rsplit("_", 1)[0] if rest.startswith("gravity_edge") else rest
        key = f"{ent}.{comp}"
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(j)
    out = {}
    for key in order:
        cols = groups[key]
        if key.endswith((".tick", ".seed", ".gravity_edge")):
            out[key] = np.array([[int(r[1 + j]) for j in cols] for r in data], dtype=np.uint64)
        else:
            out[key] = np.array([[float(r[1 + j]) for j in cols] for r in data], dtype=np.float64)
    return out


def rel_err(got, ref):
    """max |got-ref| / max(|ref|_inf of the vector, tiny) per row-vector."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    scale = np.maximum(np.max(np.abs(ref), axis=-1, keepdims=True), 1e-300)
    return float(np.max(np.abs(got - ref) / scale)) if got.size else 0.0
