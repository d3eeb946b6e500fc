#!/usr/bin/env python3
"""The compute-bound kernels of the path, a few launches each, for a rocprofv3 --pmc pass: allpairs_kernel<3> (16,384-body n-body
   tick), the generated Falcon 9 program (BASELINE configs[4]: 32,768 rollouts, 1000 ticks per launch) and the Apollo rollout kernel
   (configs[3]: 8,192 rollouts, 1000 ticks per launch).  TICKS_PER_LAUNCH is what profiles/summarize_compute.py divides by."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import elodin_amd as ea
from elodin_amd import _lib as L
from elodin_amd.models import falcon9 as f9

n = 16384
rng = np.random.default_rng(7)
u = rng.uniform(0.05, 0.95, n)
d = rng.normal(size=(n, 3))
p = d / np.linalg.norm(d, axis=1, keepdims=True) * (1.0 / np.sqrt(u ** (-2.0 / 3.0) - 1.0))[:, None]
m = rng.uniform(1e-9, 1e-3, n)
pos = np.concatenate([np.tile([0, 0, 0, 1.0], (n, 1)), p], axis=1)
vel = np.concatenate([np.zeros((n, 3)), rng.normal(scale=1e-7, size=(n, 3))], axis=1)
inertia = np.concatenate([np.tile(m[:, None], (1, 3)), np.zeros((n, 3)), m[:, None]], axis=1)
ex = ea.HipExec(pos, vel, inertia, simulation_time_step=3600.0,
                effectors=[ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, (2.9591220828e-4 / 86400.0 ** 2, 1.0e-10))])
ex.invoke_batch(4)
ex.close()
TICKS_PER_LAUNCH = 1000
fx = f9.AscentExec(f9.sample_params(32768), dtype=np.float32, ticks_per_launch=TICKS_PER_LAUNCH, fast_math=True)
fx.hip.invoke_batch(3 * TICKS_PER_LAUNCH)
fx.close()
from elodin_amd import monte_carlo as mc
from elodin_amd.models import apollo
spec = mc.load_spec(Path(__file__).resolve().parents[1] / "tests" / "golden" / "plans" / "apollo.toml")
spec["monte_carlo"]["n_samples"] = 8192
ax = apollo.ApolloExec(mc.materialize(spec).table(), ticks_per_launch=TICKS_PER_LAUNCH)
ax.invoke_batch(3 * TICKS_PER_LAUNCH)
ax.close()
print("done")
