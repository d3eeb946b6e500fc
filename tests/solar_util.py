"""Solar-system sanity case ('N from planets_truth.csv'): world setup of examples/n-body/sim.py
build_world (sun at rest at the origin + the planets at their day-0 truth state, complete gravity graph) and the
accuracy coefficient of examples/n-body/accuracy_report.py:76-100.  TEST INFRASTRUCTURE."""
import json
from pathlib import Path

import numpy as np

K_SQUARED = 2.9591220828e-4 / (86_400.0 * 86_400.0)      # AU^3 / (solar mass * s^2), sim.py:14-17
SOFTENING_AU2 = 1.0e-10
DT = 3600.0                                               # 1 tick per hour
TICKS_PER_DAY = 24


def load():
    d = json.loads((Path(__file__).parent / "golden" / "solar_system.json").read_text())
    n = len(d["bodies"]) + 1
    pos = np.tile([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], (n, 1))
    vel = np.zeros((n, 6))
    pos[1:, 4:] = np.array(d["pos0_au"])
    vel[1:, 3:] = np.array(d["vel0_au_per_day"]) / 86_400.0
    mass = np.concatenate([[1.0], d["mass_solar"]])
    inertia = np.concatenate([np.tile(mass[:, None], (1, 3)), np.zeros((n, 3)), mass[:, None]], axis=1)   # SpatialInertia(mass)
    return d, pos, vel, inertia


def accuracy(sim_pos, truth_pos):
    """sim_pos, truth_pos [bodies, samples, 3] -> (global rms AU, accuracy coefficient), accuracy_report.py:76-100."""
    err = np.linalg.norm(sim_pos - truth_pos, axis=2)
    rms = np.sqrt(np.mean(err * err, axis=1))
    radius = np.median(np.linalg.norm(truth_pos, axis=2), axis=1)
    global_rms = float(np.sqrt(np.mean(rms * rms)))
    return global_rms, 1.0 / (1.0 + global_rms / max(float(np.median(radius)), 1e-12)), rms
