import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pathlib import Path
from elodin_amd import monte_carlo as mc
from elodin_amd.models import apollo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
spec = mc.load_spec(Path(__file__).resolve().parents[1] / "tests/golden/plans/apollo.toml"); spec["monte_carlo"]["n_samples"] = n
t0 = time.perf_counter(); P = mc.materialize(spec).table(); t1 = time.perf_counter()
for K in (120, 1000):
    ex = apollo.ApolloExec(P, ticks_per_launch=K)
    ex.invoke_batch(K)
    t = time.perf_counter(); tm = ex.invoke_batch(ticks); dt = time.perf_counter() - t
    print(f"n={n} ticks={ticks} K={K}: wall {dt*1e3:.1f} ms device {tm.kernel_device_ms:.1f} ms -> {n*ticks/dt:.3e} rollout-steps/s, {tm.launches} launches; plan {1e3*(t1-t0):.0f} ms")
    ex.close()
