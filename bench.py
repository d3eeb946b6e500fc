#!/usr/bin/env python3
"""bench.py — entity-steps/s of the MI355X six_dof path (BASELINE.json's metric on configs[1]).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = one RK4 tick of the hot path over one rank's batch of 65,536 synthetic bodies (BASELINE configs[1]: constant
gravity + body torque, f64).  Inputs are resident in HBM when the timed region starts.  Weak scaling: every rank owns its
own 65,536-row shard of one larger world (rows [rank*65536, (rank+1)*65536)); the path has no per-step exchange, so the data
path uses no collective — torch.distributed is only the launcher's barrier and the max-over-ranks of the time.

Rank 0's LAST stdout line is ONE compact JSON object (< 4 KB, gated by tests/test_bench_line.py): the contract's keys,
`roofline` (the step kernel at one tick per launch: algorithmic 384 B per entity-step against the 8 TB/s HBM peak, launch
duration from HIP events around the timed region), `parity` (max |dState| vs the oracle on a 4,096-row world, outside the
timed region — the second half of BASELINE.json's metric) and `cpu_baseline` (the CPU oracle on the host cores, one thread,
bounded sample, N = 1 only); at N > 1 also `rccl` (what the process group is) and `campaigns` (BASELINE configs[3] / [4] as
whole Monte-Carlo campaigns over the same ranks, four numbers — run LAST and under a watchdog, `guarded_campaigns`: a rank
failing alone in there costs the line `campaigns`, never the headline).  What the reference's own profile prints is as small
(libs/nox-py/src/profile.rs:14-59: build, h2d, kernel_invoke, d2h, tick, real-time factor).

Everything else — the 4 M-body HBM roofline, fused ticks, n-body, sparse edges, telemetry commit, the campaigns on one GPU,
whole-world StableHLO ticks, build times — lives in tools/bench_legs.py and runs only under `--extras`, into a sidecar file
(`--extras-out`); a leg that fails there makes this script exit non-zero.

`--dry-run` (CPU, no GPU, fake timings; the line says so in `data`) exercises argument handling, the process group, the
max-over-ranks and the composition of the line: it is what the CPU tests parse.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

ENTITIES = 65536
# read pos 56 + vel 48 + inertia 56, write pos 56 + vel 48 + accel 48 + force 48 = 360 B per f64
# entity-step, plus 24 B for the per-entity body-torque effector column this workload reads = 384 B.
BYTES_PER_ENTITY_STEP_F64 = 360 + 24
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
PMC_FILE = ROOT / "profiles" / "pmc_traffic.json"  # written by profiles/collect.sh from separate --pmc passes
MAX_LINE_BYTES = 4096            # the driver parses the last stdout line; r05's 20.7 KB line came back `parsed: null`
CAMPAIGN_TOTALS = {"apollo": 8192, "falcon9": 32768}      # BASELINE configs[3] / configs[4]: rollouts of the WHOLE campaign

REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "parity")


def make_exec(n, first_row, device, ticks_per_launch, use_graph):
    import elodin_amd as ea
    from elodin_amd import workloads
    w = workloads.independent_bodies(n, first_row=first_row)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    return ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], entity_ids=w["entity_ids"],
                      simulation_time_step=workloads.DT_120HZ, effectors=eff, device=device,
                      ticks_per_launch=ticks_per_launch, use_graph=use_graph), w, eff


STEP_KERNEL_SOURCES = ("step_kernel.hpp", "effectors.hpp", "spatial.hpp", "kernels.hpp", "sixdof_kernels.hip")


def step_kernel_hash():
    """Content hash of the sources the hand-written step kernel is compiled from: what a PMC collection is valid for."""
    import hashlib
    h = hashlib.sha256()
    for name in STEP_KERNEL_SOURCES:
        h.update((ROOT / "elodin_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def pmc_traffic(n):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KiB).  The
    file is stamped (profiles/summarize.py) with the hash of the kernel sources it was collected on; when the kernel has
    changed since — or no profile of this entity count exists — the figure would be stale and None is returned."""
    try:
        doc = json.loads(PMC_FILE.read_text())
        if doc.get("_stamp", {}).get("step_kernel_hash") != step_kernel_hash():
            return None
        rec = doc.get(str(n))
        return rec and rec["hbm_bytes_per_launch"]
    except Exception:
        return None


def roofline_from(avg_launch_ms, n, launches, how):
    """SURVEY §8(d) / DESIGN §4: ALGORITHMIC bytes per launch (384 B x the entities one launch steps) / the launch's average
    duration, against the 8 TB/s spec peak."""
    bytes_per_launch = BYTES_PER_ENTITY_STEP_F64 * n
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    traffic = pmc_traffic(n)
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic and round(traffic),
            "traffic_source": f"profiles/pmc_traffic.json (rocprofv3 --pmc passes; valid for step-kernel sources {step_kernel_hash()})"
                              if traffic is not None else None,
            "kernel": "sixdof_step_kernel<double, rk4, gravity|body_torque>",
            "avg_launch_us": round(avg_launch_ms * 1e3, 3), "algorithmic_bytes_per_launch": bytes_per_launch,
            "launches_timed": int(launches), "timing": how}


def rccl_attestation(torch, dist, world, local_rank, shared_gpu):
    """What the process group actually is: backend, world size as the group reports it, and that every rank computes on its
    own device (PCI bus id + uuid gathered over the group itself; asserted distinct unless the dry-run switch is set)."""
    props = torch.cuda.get_device_properties(local_rank)
    try:
        bus = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{getattr(props, 'pci_device_id', 0):02x}"
    except Exception:  # noqa: BLE001
        bus = os.environ.get("HIP_VISIBLE_DEVICES", "") + f"#{local_rank}"
    mine = (bus, str(getattr(props, "uuid", "")))
    if dist is None:
        return {"backend": None, "world_size": 1, "devices": [bus]}
    box = [None] * world
    dist.all_gather_object(box, mine)
    distinct = len(set(box)) == len(box)
    if not distinct and not shared_gpu:
        raise SystemExit(f"bench.py --gpus {world}: ranks share a device {box} (set SIXDOF_BENCH_SHARED_GPU=1 for a one-GPU dry run)")
    ver = None
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        pass
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": ver, "distinct_devices": distinct,
            "devices": [b for b, _ in box]}


def cpu_quota():
    """CPUs this process may actually burn: the cgroup CPU quota (cpu.max / cfs_quota), which sched_getaffinity does not
    show — 256 visible cores under a quota of 16 CPUs scale like 16."""
    try:
        q, p = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        return None if q == "max" else round(int(q) / int(p), 2)
    except Exception:  # noqa: BLE001
        pass
    try:
        q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        p = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        return None if q <= 0 else round(q / p, 2)
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(w, eff, target_seconds=10.0):
    """The CPU oracle (a port of the reference arithmetic; the reference itself needs rustc + jax) timed on this host's
    cores on a bounded sample of the same workload.  `value` is the SINGLE-THREAD figure — the like-for-like stand-in for
    the reference's tick, which is one native function call on one thread (cranelift_exec.rs:163-165); the OpenMP figure
    over entity blocks is reported beside it with threads = min(visible cores, cgroup CPU quota)."""
    from oracle import oracle as orc
    visible = len(os.sched_getaffinity(0))
    quota = cpu_quota()
    threads = max(1, min(visible, int(quota))) if quota else visible
    ops = [(e.kind, tuple(e.p), e.aux) for e in eff]
    n = w["world_pos"].shape[0]

    def run(ticks, th):
        o = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=0.008333333, ops=ops)
        t0 = time.perf_counter()
        o.step(ticks, threads=th)
        return time.perf_counter() - t0

    # two builds of the same source: the portable -O2 one the tests use and the one made ON THIS HOST
    # (-O3 -march=native -fno-tree-vectorize, oracle/Makefile says why); the faster single-thread build is the one timed
    st_ticks = max(2, int(min(256, target_seconds * 0.25 / max(1e-9, run(1, 1)))))
    builds = {"-O2": n * st_ticks / run(st_ticks, 1)}
    portable = orc.LIB_PATH
    lib_path = orc.use_native_build()
    if lib_path.name.endswith("_native.so"):
        builds["-O3 -march=native -fno-tree-vectorize"] = n * st_ticks / run(st_ticks, 1)
        if builds["-O3 -march=native -fno-tree-vectorize"] < builds["-O2"]:
            orc.LIB_PATH, orc._lib = portable, None
    flags = max(builds, key=builds.get)
    st = run(st_ticks, 1)
    out = {"value": round(n * st_ticks / st, 1), "unit": "entity-steps/s", "cores": 1, "kind": "port",
           "sample": f"{n} bodies x {st_ticks} RK4 ticks, oracle/sixdof_oracle.c {flags} -ffp-contract=off, one thread "
                     f"(the reference's tick is single-threaded)"}
    if threads > 1:
        probe = run(4, threads)
        ticks = int(min(4096, max(8, target_seconds * 0.5 / (probe / 4))))
        dt = run(ticks, threads)
        out["multi_thread"] = {"value": round(n * ticks / dt, 1), "cores": threads, "ticks": ticks,
                               "threads_rule": "OpenMP over entity blocks, min(visible cores, cgroup CPU quota)"}
    return out


def parity_figure(device, rows=4096, ticks=16):
    """BASELINE.json's metric ends in "max |dState| vs ref": the same workload at `rows` bodies stepped `ticks` RK4 ticks
    through the HIP path and through the oracle (the checker; pinned bit-exact on the reference's golden CSVs), OUTSIDE
    the timed region.  The figures tests/test_gpu_parity.py gates at BASELINE size (tests/parity.py says why the floor):
      max_rel_err               per entity and field, |d| / the field vector's largest component, all four columns
      max_rel_err_elementwise   SURVEY §8(d)'s form, all four columns: |s_i - ref_i| / max(|ref_i|, 1e-2 x vector scale) — the
                                reference's own CI compare (math.isclose: rel_tol, abs_tol) at 1e-9 / 1e-11 x the vector's size
      elementwise_floor_1e-12   the same with a near-zero floor, REPORTED: it measures how close to zero a component happens to be"""
    from oracle import oracle as orc
    import elodin_amd as ea
    from elodin_amd import workloads
    w = workloads.independent_bodies(rows)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    hip = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], entity_ids=w["entity_ids"],
                     simulation_time_step=workloads.DT_120HZ, effectors=eff, device=device)
    hip.run(ticks)
    ref = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                          ops=[(e.kind, tuple(e.p), e.aux) for e in eff]).step(ticks)
    worst, worst_elem, worst_raw = {}, {}, {}
    for f in ("world_pos", "world_vel", "world_accel", "force"):
        g, r = getattr(hip, f), getattr(ref, f)
        e, ee, er = 0.0, 0.0, 0.0
        for sl in ((slice(0, 4), slice(4, 7)) if f == "world_pos" else (slice(0, 3), slice(3, 6))):
            scale = np.maximum(np.max(np.abs(r[:, sl]), axis=1, keepdims=True), 1e-300)
            d = np.abs(g[:, sl] - r[:, sl])
            e = max(e, float(np.max(d / scale)))
            ee = max(ee, float(np.max(d / np.maximum(np.abs(r[:, sl]), 1e-2 * scale))))
            er = max(er, float(np.max(d / np.maximum(np.abs(r[:, sl]), 1e-12 * scale))))
        worst[f], worst_elem[f], worst_raw[f] = e, ee, er
    # integer surface: the gather rows the C ABI resolved for the joined entity ids (identity here) — bit-exact or wrong
    ids_equal = bool(np.array_equal(hip.join_rows("world_pos"), np.arange(rows, dtype=np.uint32)))
    hip.close()
    sig = lambda d: {k: float(f"{v:.3e}") for k, v in d.items()}
    return {"max_rel_err": float(f"{max(worst.values()):.3e}"), "by_column": sig(worst),
            "max_rel_err_elementwise": float(f"{max(worst_elem.values()):.3e}"), "by_column_elementwise": sig(worst_elem),
            "elementwise_floor_1e-12": sig(worst_raw), "tolerance": 1e-9, "entity_rows_bit_exact": ids_equal,
            "rows": rows, "ticks": ticks, "vs": "oracle/sixdof_oracle.c (bit-exact on the reference's golden CSVs)"}


def campaign_numbers(rank, world, local_rank, comm, barrier):
    """BASELINE configs[3] / configs[4] as whole campaigns over the SAME ranks (plan broadcast, flight of the rank's run-id
    block, result gather): rollout-steps/s, `strong` = the stated totals split over the ranks, `weak` = the totals per GPU.
    Four numbers; the full lines are `--campaign apollo|falcon9`."""
    from tools import bench_legs
    out = {"unit": "rollout-steps/s", "totals": CAMPAIGN_TOTALS}
    for which in ("apollo", "falcon9"):
        for scaling in (("strong",) if world == 1 else ("strong", "weak")):
            line = bench_legs.campaign_bench(which, rank, world, local_rank, comm, barrier, None, scaling)
            out.setdefault(which, {})[scaling] = line["value"]
            out[which][scaling + "_seconds"] = line["campaign_seconds"]
    return out


def guarded_campaigns(rank, leg, finish, line):
    """The campaign leg at N > 1 under a watchdog.  It is the one part of the line that runs collectives inside library code
    (plan broadcast, result gather) on ranks that have never met on real hardware: if a rank fails alone, the others wait in a
    collective for ever and the WHOLE line — the headline the driver's scaling curve is made of — would be lost with it.  So:
    an exception on this rank is recorded instead of raised, and when leg + closing barrier have not finished within
    SIXDOF_BENCH_CAMPAIGN_TIMEOUT seconds (default 300) rank 0 prints the headline with `campaigns: {"error": ...}` and every
    rank leaves.  `line(campaigns)` -> the text rank 0 prints; `finish()` = closing barrier + group teardown.
    Returns the campaigns object (numbers, or {"error": ...})."""
    import threading
    limit = float(os.environ.get("SIXDOF_BENCH_CAMPAIGN_TIMEOUT", "300"))
    state = {"error": None}

    def bail():
        why = state["error"] or f"the campaign leg did not finish within {limit:.0f} s on every rank"
        print(f"bench.py: rank {rank}: {why}; the headline is printed without the campaign numbers", file=sys.stderr, flush=True)
        if rank == 0:
            print(line({"error": why}), flush=True)
        sys.stdout.flush()
        os._exit(0)             # the other ranks sit in a collective that will never complete: no orderly teardown exists

    timer = threading.Timer(limit, bail)
    timer.daemon = True
    timer.start()
    try:
        out = leg()
    except Exception as e:  # noqa: BLE001 — recorded in the line and on stderr, never fatal to the headline
        state["error"] = f"{type(e).__name__}: {e}"[:300]
        print(f"bench.py: rank {rank}: campaign leg FAILED: {state['error']}", file=sys.stderr, flush=True)
        out = {"error": state["error"]}
    try:
        finish()                # still under the watchdog: a rank that failed alone leaves the others inside a collective
    except Exception as e:  # noqa: BLE001 — a peer that left early breaks the closing barrier; the line is still owed
        print(f"bench.py: rank {rank}: closing barrier after the campaign leg failed: {type(e).__name__}: {e}"[:400], file=sys.stderr, flush=True)
        if isinstance(out, dict) and "error" not in out:
            out = dict(out, closing_barrier_error=f"{type(e).__name__}"[:80])
    timer.cancel()
    return out


def compose_line(args, world, n, K, elapsed, elapsed_incl, tm, *, roofline, parity, cpu=None, rccl=None, campaigns=None,
                 shared_gpu=False, data="synthetic"):
    """The one object the driver parses.  Kept flat and small; `fit_line` enforces the size."""
    value = n * world * args.steps / elapsed
    out = {
        "metric": "entity-steps/s (6DOF RK4)", "value": round(value, 1), "unit": "entity-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 6), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": data,
        "config": {"workload": ("65536 independent 6DOF bodies (constant gravity + body torque), RK4 f64 (BASELINE configs[1])"
                                if n == ENTITIES else f"{n} independent 6DOF bodies (constant gravity + body torque), RK4 f64"),
                   "entities_per_gpu": n, "ticks_per_launch": K, "dt": 0.008333333,
                   "graph_replay": bool(tm["launches"] and tm["graph_launches"] == tm["launches"]), "launches": tm["launches"],
                   "parallelism": f"entity shards x{world}, no collective" + (" — all ranks share GPU 0 (gloo group): not a scaling point" if shared_gpu else ""),
                   "sync": "barrier, device synchronize, t0, the steps + hipStreamSynchronize of their stream, t1; MAX over ranks of t1 - t0",
                   "excluded": "H2D upload and the final D2H download of the columns (0.3 ms over PCIe, amortised over the config's 10,000 ticks)"},
        "ms_per_step_incl_closing_sync_and_barrier": round(elapsed_incl / args.steps * 1e3, 6),
        "device_ms_per_step": round(tm["kernel_device_ms"] / max(1, args.steps), 6),
        "roofline": roofline, "parity": parity,
    }
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if world > 1:
        out["rccl"] = rccl
        out["n1_reference_value"] = round(value / world, 1)     # per-GPU value: what `value` at N = 1 should read under weak scaling
    if campaigns is not None:
        out["campaigns"] = campaigns
    return out


def fit_line(out):
    """Serialise; when the text would outgrow what the driver parses, drop the optional detail (never a contract key)."""
    text = json.dumps(out, separators=(", ", ": "))
    for path in (("parity", "elementwise_floor_1e-12"), ("parity", "by_column_elementwise"), ("parity", "by_column"), ("roofline", "timing"), ("roofline", "traffic_source"),
                 ("config", "sync"), ("config", "excluded"), ("rccl", "devices"), ("cpu_baseline", "multi_thread"), ("campaigns",)):
        if len(text) < MAX_LINE_BYTES:
            break
        d = out
        for k in path[:-1]:
            d = d.get(k) or {}
        d.pop(path[-1], None)
        text = json.dumps(out, separators=(", ", ": "))
    if len(text) >= MAX_LINE_BYTES:
        raise SystemExit(f"bench.py: the line is {len(text)} bytes (limit {MAX_LINE_BYTES})")
    json.loads(text)
    return text


def extras_path(args):
    if args.extras_out:
        return Path(args.extras_out)
    d = ROOT / "gpurun_out"
    return (d if d.is_dir() else ROOT) / "bench_extras.json"


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--entities", type=int, default=ENTITIES, help="bodies per GPU")
    ap.add_argument("--ticks-per-launch", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="leave the oracle legs (parity, cpu_baseline) out")
    ap.add_argument("--extras", action="store_true", help="also run the side legs (tools/bench_legs.py) into the sidecar file; N = 1 only")
    ap.add_argument("--extras-out", default=None, help="sidecar path (default gpurun_out/bench_extras.json when that directory exists)")
    ap.add_argument("--only-legs", default="", help="with --extras: comma-separated legs to run")
    ap.add_argument("--skip-legs", default="", help="with --extras: comma-separated legs to leave out")
    ap.add_argument("--campaigns", action="store_true", help="add the four campaign numbers at N = 1 too (they are in the line at N > 1)")
    ap.add_argument("--no-campaigns", action="store_true", help="leave the campaign numbers out at N > 1")
    ap.add_argument("--capi-comm", action="store_true",
                    help="with --campaign: plan broadcast / result gather through the C ABI's own RCCL entry points "
                         "(sixdof_comm_*), the id shipped over the launcher's process group, instead of torch collectives")
    ap.add_argument("--campaign", choices=("apollo", "falcon9"), default=None,
                    help="instead of the config-2 step: one whole Monte-Carlo campaign (BASELINE configs[3] / configs[4]) "
                         "sharded over the ranks, plan broadcast + result gather over RCCL; --steps/--warmup are ignored")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="with --campaign: `strong` = BASELINE's rollout total (8,192 / 32,768) split over the ranks; `weak` = that "
                         "count per GPU")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: fake timings through the same process-group and line code (CPU tests)")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Dry run of the N > 1 code path on a box with ONE GPU (SIXDOF_BENCH_SHARED_GPU=1 under torch.distributed.run): every
    # rank uses device 0 and the process group is gloo (RCCL refuses two ranks on one device).  The timed window is the same
    # code as on N GPUs; the value means nothing as a scaling point (the ranks time-share one GPU) and the line says so.
    shared_gpu = os.environ.get("SIXDOF_BENCH_SHARED_GPU", "") == "1"
    if shared_gpu:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = "WORLD_SIZE" in os.environ and "RANK" in os.environ   # launched by torch.distributed.run
    if args.gpus != world and distributed:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    n = args.entities
    K = max(1, args.ticks_per_launch)

    import torch
    if args.dry_run:
        return dry_run(args, rank, world, distributed, n, K)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the six_dof product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rank_device = "cpu" if shared_gpu else torch.device("cuda", local_rank)      # where the ranks' scalars are reduced
    attest = rccl_attestation(torch, dist, world, local_rank, shared_gpu)

    def barrier():
        if distributed:
            dist.barrier() if shared_gpu else dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    def finish():
        if distributed:
            dist.barrier() if shared_gpu else dist.barrier(device_ids=[local_rank])
            dist.destroy_process_group()

    if args.campaign:
        from tools import bench_legs
        comm = rank_device if distributed else "cpu"
        capi = None
        if args.capi_comm:
            from elodin_amd import shard
            box = [shard.CapiComm.unique_id() if (rank == 0 and world > 1) else None]
            if distributed:
                dist.broadcast_object_list(box, src=0, device=rank_device)
            capi = shard.CapiComm(box[0], world, rank, local_rank)
        line = bench_legs.campaign_bench(args.campaign, rank, world, local_rank, comm, barrier, capi, args.scaling)
        if capi is not None:
            capi.close()
        finish()
        if rank == 0:
            print(json.dumps(line), flush=True)
        return 0

    ex, w, eff = make_exec(n, rank * n, local_rank, K, not args.no_graph)

    # ---- timed region: W warmup steps, then exactly `steps` steps between barriers --------------------
    # set-up, not stepping: capture the launch chains a batch of `steps` ticks replays (like the JIT at build time), so
    # even a 20-step timed region is steady-state device work and not eager launches racing the host
    if not args.no_graph:
        ex.prepare(args.steps)
    ex.invoke_batch(args.warmup)
    if not args.no_graph:
        ex.prepare(args.steps)            # a no-op when the chains are cached (they are: captured above)
    # ONE rule at every N (`barrier` is only a device synchronise without a process group):  barrier + device synchronise |
    # t0 | the steps + hipStreamSynchronize of the stream they ran on (inside sixdof_step: nothing else is in flight on this
    # device) | t1 | device synchronise + barrier | t2 ;  MAX over ranks of (t1 - t0) AFTER the clocks have stopped.  `value`
    # uses t1 - t0; the window with the closing device-wide synchronise and barrier (host bookkeeping with nothing left to
    # wait for, profiles/r02_short_batch_ab.txt) is reported beside it as `ms_per_step_incl_closing_sync_and_barrier`.
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tm = ex.invoke_batch(args.steps)      # enqueues every launch and SYNCHRONISES the stream they run on
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    barrier()
    t2 = time.perf_counter()
    elapsed, elapsed_incl = t1 - t0, t2 - t0
    if distributed:
        from elodin_amd import shard
        elapsed = shard.max_over_ranks(elapsed, device=rank_device)        # MAX over ranks
        elapsed_incl = shard.max_over_ranks(elapsed_incl, device=rank_device)
    tmd = {"launches": int(tm.launches), "graph_launches": int(tm.graph_launches), "kernel_device_ms": float(tm.kernel_device_ms)}

    roofline = None
    if rank == 0:
        if K == 1:  # the timed region itself: HIP events bracket exactly the `steps` launches on the launch stream
            roofline = roofline_from(tm.kernel_device_ms / max(1, tm.launches), n, tm.launches,
                                     "HIP events around the timed region on the launch stream / launches")
        else:       # K > 1: the dominant kernel is still priced at one tick per launch, outside the timed region
            ex.set_ticks_per_launch(1)
            ex.invoke_batch(args.warmup)
            t = ex.invoke_batch(args.steps)
            roofline = roofline_from(t.kernel_device_ms / max(1, t.launches), n, t.launches,
                                     "HIP events around a batch of single-tick launches on the launch stream / launches (outside the timed region)")
        if world == 1 and K == 1 and args.steps < 2048:
            # A short timed region carries the batch's start-up (~20 us of device time before the first launch of a replayed
            # chain runs at speed, profiles/r02_short_batch_ab.txt) in its average.  The same kernel, same handle, same
            # measurement over a long batch, reported beside it — not instead of it.
            if not args.no_graph:
                ex.prepare(4096)
            ex.invoke_batch(256)
            st = ex.invoke_batch(4096)
            us = st.kernel_device_ms / max(1, st.launches) * 1e3
            roofline["long_batch"] = {"launches": int(st.launches), "avg_launch_us": round(us, 3),
                                      "frac": round(BYTES_PER_ENTITY_STEP_F64 * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}
    ex.close()

    extras_errors = {}
    if args.extras:
        if world != 1:
            raise SystemExit("bench.py --extras: the side legs are single-GPU (run them at --gpus 1)")
        from tools import bench_legs
        only = tuple(x for x in args.only_legs.split(",") if x)
        skip = tuple(x for x in args.skip_legs.split(",") if x)
        doc, extras_errors = bench_legs.run_legs(bench_legs.single_gpu_legs(local_rank, n), only, skip)
        path = extras_path(args)
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(json.dumps(doc, indent=1) + "\n")
        print(f"bench.py: side legs written to {path}", file=sys.stderr, flush=True)

    parity = cpu = None
    if rank == 0:
        # the oracle legs (the checker and the reported CPU baseline; nothing timed above touches the oracle).  A failure INSIDE them
        # (the oracle library cannot be built or loaded on this host) is said in the line and on stderr — it makes no parity claim,
        # and it does not take the measured headline with it
        def oracle_leg(name, fn):
            try:
                return fn()
            except Exception as e:  # noqa: BLE001
                why = f"{type(e).__name__}: {e}"[:300]
                print(f"bench.py: the `{name}` leg FAILED: {why}", file=sys.stderr, flush=True)
                return {"error": why}
        parity = {"skipped": "--no-cpu-baseline"} if args.no_cpu_baseline else oracle_leg("parity", lambda: parity_figure(local_rank))
        if world == 1 and not args.no_cpu_baseline:
            cpu = oracle_leg("cpu_baseline", lambda: cpu_baseline(w, eff))

    def line(campaigns):
        return fit_line(compose_line(args, world, n, K, elapsed, elapsed_incl, tmd, roofline=roofline, parity=parity, cpu=cpu, rccl=attest,
                                     campaigns=campaigns, shared_gpu=shared_gpu))

    # the campaign numbers come LAST and under a watchdog: the headline above is complete before any of their collectives run
    if (world > 1 and not args.no_campaigns) or args.campaigns:
        campaigns = guarded_campaigns(
            rank, lambda: campaign_numbers(rank, world, local_rank, rank_device if distributed else "cpu", barrier), finish, line)
    else:
        campaigns = None
        finish()
    if rank == 0:
        print(line(campaigns), flush=True)
    if extras_errors:
        print("bench.py --extras: FAILED legs: " + json.dumps(extras_errors), file=sys.stderr, flush=True)
        return 1
    return 0


def dry_run(args, rank, world, distributed, n, K):
    """No GPU: the launcher's process group (gloo), the barrier, the max-over-ranks and the composition of the line with
    fake timings (6 us per step, rank r 1 % slower than rank r - 1), the CPU oracle on a tiny sample as `cpu_baseline`.  Not a
    measurement; `data` says so."""
    dist = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    elapsed = 6.0e-6 * args.steps * (1.0 + 0.01 * rank)
    elapsed_incl = elapsed + 1.5e-4
    if distributed:
        from elodin_amd import shard
        elapsed = shard.max_over_ranks(elapsed, device="cpu")
        elapsed_incl = shard.max_over_ranks(elapsed_incl, device="cpu")
    tmd = {"launches": args.steps // K, "graph_launches": 0 if args.no_graph else args.steps // K, "kernel_device_ms": 5.6e-3 * args.steps}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from elodin_amd import workloads
        w = workloads.independent_bodies(512)
        cpu = cpu_baseline(w, workloads.gravity_torque_effectors(w["body_torque"]), target_seconds=0.2)
    if distributed:
        box = [None] * world
        dist.all_gather_object(box, f"dry-run#{rank}")
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": None, "distinct_devices": False, "devices": box}
    else:
        rccl = {"backend": None, "world_size": 1, "devices": ["dry-run"]}
    parity = {"max_rel_err": 0.0, "by_column": {f: 0.0 for f in ("world_pos", "world_vel", "world_accel", "force")},
              "max_rel_err_elementwise": 0.0, "by_column_elementwise": {f: 0.0 for f in ("world_pos", "world_vel", "world_accel", "force")},
              "tolerance": 1e-9, "entity_rows_bit_exact": None, "rows": 0, "ticks": 0, "vs": "DRY RUN: nothing was compared"}

    def line(campaigns):
        return fit_line(compose_line(args, world, n, K, elapsed, elapsed_incl, tmd,
                                     roofline=roofline_from(tmd["kernel_device_ms"] / max(1, tmd["launches"]), n, tmd["launches"], "DRY RUN: fake timings"),
                                     parity=parity, cpu=cpu, rccl=rccl, campaigns=campaigns, data="DRY RUN (no GPU, fake timings): not a measurement"))

    def finish():
        if distributed:
            dist.barrier()
            dist.destroy_process_group()

    def fake_campaigns():
        # SIXDOF_BENCH_DRYRUN_FAULT = "hang:<rank>" | "raise:<rank>": what a rank failing alone inside the leg looks like
        kind, _, who = os.environ.get("SIXDOF_BENCH_DRYRUN_FAULT", "").partition(":")
        if kind and int(who) == rank:
            if kind == "raise":
                raise RuntimeError("dry-run fault injected on this rank")
            time.sleep(3600)
        if distributed:
            dist.barrier()          # the leg's collectives
        camp = {"unit": "rollout-steps/s", "totals": CAMPAIGN_TOTALS}
        for which in CAMPAIGN_TOTALS:
            for s in (("strong",) if world == 1 else ("strong", "weak")):
                camp.setdefault(which, {})[s] = 0.0
                camp[which][s + "_seconds"] = 0.0
        return camp

    if (world > 1 and not args.no_campaigns) or args.campaigns:
        camp = guarded_campaigns(rank, fake_campaigns, finish, line)
    else:
        camp = None
        finish()
    if rank == 0:
        print(line(camp), flush=True)
    return 0

if __name__ == "__main__":
    sys.exit(main())
