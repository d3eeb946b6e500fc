"""Apollo-lander Monte-Carlo rollouts on the GPU (BASELINE config 4).

Host-side mirror of examples/apollo-lander: `initial_columns` restates `build(params)`
(sim.py:225-310: spawn of the `lander` entity), `ApolloCampaign` plays the role of
`elodin monte-carlo run` for this sim — plan rows become rows of the entity axis, sharded over
ranks (libs/monte-carlo runs them as one OS process each).  All stepping happens in
elodin_amd/csrc/apollo_kernels.hip through the C ABI.
"""
from __future__ import annotations

import csv
import ctypes as C
import math
from pathlib import Path
from typing import Dict, Mapping

import numpy as np

from .. import _lib as L
from ..exec import HipExec, _raise

DATA = Path(__file__).resolve().parents[1] / "data" / "apollo_reference.csv"

SIMULATION_RATE_HZ, GUIDANCE_RATE_HZ, TELEMETRY_RATE_HZ = 120.0, 24.0, 40.0   # sim.py:15-18
DPS_FTP_THROTTLE = 0.925
# sorted names = column order of a plan table (sample.py:112) = include/sixdof_apollo.h APOLLO_P_*
PARAM_NAMES = ["attitude_gain", "dry_mass_kg", "gravity_scale", "horizontal_gain", "init_altitude_m",
               "init_crossrange_speed_mps", "init_downrange_offset_m", "init_downrange_speed_mps", "init_pitch_deg",
               "init_vertical_speed_mps", "isp_s", "propellant_kg", "rcs_propellant_kg", "throttle_response_hz",
               "thrust_scale", "track_gain", "vertical_gain"]
RESULT_NAMES = ["touchdown_speed", "horizontal_speed", "fuel_remaining", "rcs_fuel_remaining", "traj_rmse",
                "pitch_rmse", "downrange_miss", "upright_dot", "landed", "soft_landing", "tick", "reserved"]
N_STATE, N_PARAMS, N_GUIDANCE, N_SCORE, N_RESULT = 16, 17, 8, 4, 12


class Tables(C.Structure):  # sixdof_apollo_tables
    _fields_ = [("time_s", C.c_void_p), ("altitude_m", C.c_void_p), ("descent_rate_mps", C.c_void_p),
                ("pitch_deg", C.c_void_p), ("horizontal_speed_mps", C.c_void_p), ("downrange_m", C.c_void_p),
                ("n", C.c_uint32), ("guidance_period_ticks", C.c_uint32), ("max_ticks", C.c_uint64),
                ("ticks_per_telemetry", C.c_uint32), ("reserved", C.c_uint32)]


def load_reference() -> Dict[str, np.ndarray]:
    with open(DATA, newline="") as f:
        rows = list(csv.reader(f))
    data = np.array([[float(x) for x in r] for r in rows[1:]], dtype=np.float64)
    return {name: np.ascontiguousarray(data[:, j]) for j, name in enumerate(rows[0])}


def default_params(ref: Mapping[str, np.ndarray]) -> Dict[str, float]:
    """PARAMS defaults of sim.py:69-106 (main.py reads track_gain with default 0.06; the spec always sets it)."""
    return dict(attitude_gain=0.040, dry_mass_kg=6853.0, gravity_scale=1.0, horizontal_gain=0.05,
                init_altitude_m=float(ref["altitude_m"][0]), init_crossrange_speed_mps=0.0,
                init_downrange_offset_m=0.0, init_downrange_speed_mps=float(ref["horizontal_speed_mps"][0]),
                init_pitch_deg=-77.0, init_vertical_speed_mps=float(ref["descent_rate_mps"][0]), isp_s=311.0,
                propellant_kg=3950.0, rcs_propellant_kg=240.0, throttle_response_hz=3.0, thrust_scale=1.0,
                track_gain=0.04, vertical_gain=0.45)


def max_ticks(ref: Mapping[str, np.ndarray]) -> int:
    """DEFAULT_MAX_TICKS, sim.py:55."""
    return int(math.ceil((float(ref["time_s"][-1]) + 20.0) * SIMULATION_RATE_HZ)) + 1


def initial_columns(params: np.ndarray, ref: Mapping[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Columns of the `lander` entity for every rollout (sim.py:258-310), one row per plan row."""
    P = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, N_PARAMS)
    n = P.shape[0]
    col = {name: P[:, j] for j, name in enumerate(PARAM_NAMES)}
    half = np.deg2rad(col["init_pitch_deg"]) / 2.0           # from_axis_angle([0,1,0], deg2rad(pitch))
    q = np.stack([np.zeros(n), np.sin(half), np.zeros(n), np.cos(half)], axis=1)
    downrange = float(ref["downrange_m"][0]) + col["init_downrange_offset_m"]
    world_pos = np.concatenate([q, downrange[:, None], np.zeros((n, 1)), col["init_altitude_m"][:, None]], axis=1)
    world_vel = np.concatenate([np.zeros((n, 3)), col["init_downrange_speed_mps"][:, None],
                                col["init_crossrange_speed_mps"][:, None], col["init_vertical_speed_mps"][:, None]], axis=1)
    total_mass = col["dry_mass_kg"] + col["propellant_kg"] + col["rcs_propellant_kg"]
    inertia = np.concatenate([np.tile([78000.0, 72000.0, 45000.0], (n, 1)), np.zeros((n, 3)), total_mass[:, None]], axis=1)
    state = np.zeros((n, N_STATE))
    state[:, 0] = DPS_FTP_THROTTLE           # throttle
    state[:, 1] = DPS_FTP_THROTTLE           # throttle_cmd
    state[:, 2:6] = q                        # attitude_setpoint
    state[:, 6] = col["propellant_kg"]
    state[:, 7] = col["rcs_propellant_kg"]
    state[:, 15] = np.abs(col["init_pitch_deg"])   # pitch telemetry is unsigned tilt (sim.py:287-288)
    guidance = np.zeros((n, N_GUIDANCE))     # main.py:113-116 + ThrottleLogic{ftp_latched: true}
    guidance[:, 0] = DPS_FTP_THROTTLE
    guidance[:, 1:5] = q
    guidance[:, 6] = 1.0
    return dict(world_pos=world_pos, world_vel=world_vel, inertia=inertia, apollo_state=state,
                apollo_params=P.copy(), apollo_guidance=guidance, apollo_score=np.zeros((n, N_SCORE)),
                apollo_result=np.zeros((n, N_RESULT)))


class ApolloExec(HipExec):
    """HipExec with the Apollo rollout model selected (sixdof_set_model_apollo)."""

    MODEL_COLUMNS = ("apollo_state", "apollo_params", "apollo_guidance", "apollo_score", "apollo_result")

    def __init__(self, params: np.ndarray, *, ref=None, ticks_per_launch: int = 120, device: int = 0,
                 max_ticks_override: int | None = None, first_row: int = 0, ticks_per_telemetry: int | None = None):
        self.ref = ref or load_reference()
        cols = initial_columns(params, self.ref)
        n = cols["world_pos"].shape[0]
        ids = np.arange(first_row + 1, first_row + n + 1, dtype=np.uint64)
        super().__init__(cols["world_pos"], cols["world_vel"], cols["inertia"], entity_ids=ids,
                         simulation_time_step=float(L.lib().sixdof_quantize_time_step(SIMULATION_RATE_HZ)),
                         integrator=L.SEMI_IMPLICIT, ticks_per_launch=ticks_per_launch, device=device)
        self.model = {k: np.ascontiguousarray(cols[k]) for k in self.MODEL_COLUMNS}
        self._bind([(k, v) for k, v in self.model.items()])
        t = Tables()
        self._tab = [np.ascontiguousarray(self.ref[k]) for k in ("time_s", "altitude_m", "descent_rate_mps", "pitch_deg",
                                                                "horizontal_speed_mps", "downrange_m")]
        (t.time_s, t.altitude_m, t.descent_rate_mps, t.pitch_deg, t.horizontal_speed_mps, t.downrange_m) = \
            [a.ctypes.data for a in self._tab]
        t.n = len(self._tab[0])
        t.guidance_period_ticks = max(1, round(SIMULATION_RATE_HZ / GUIDANCE_RATE_HZ))
        t.max_ticks = max_ticks_override if max_ticks_override is not None else max_ticks(self.ref)
        # world.run(simulation_rate=120, telemetry_rate=40) (main.py:274-283): the server loop runs 3 ticks per batch and
        # calls post_step once per batch (impeller2_server.rs:553-678) — guidance, RMSE samples, result check happen there
        t.ticks_per_telemetry = (ticks_per_telemetry if ticks_per_telemetry is not None
                                 else max(1, round(SIMULATION_RATE_HZ / TELEMETRY_RATE_HZ)))
        fn = self._lib.sixdof_set_model_apollo
        fn.argtypes, fn.restype = [C.c_void_p, C.POINTER(Tables)], C.c_int
        rc = fn(self._h, C.byref(t))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_set_model_apollo")
        self.upload()

    def download(self, mask: int = L.COL_ALL):
        super().download(mask)
        for name in self.MODEL_COLUMNS:
            if name != "apollo_params":
                rc = self._lib.sixdof_download_column(self._h, L.component_id(name))
                if rc != L.OK:
                    _raise(self._h, rc, "sixdof_download_column")
        return self

    # the reference's component names, as views of the packed state column
    @property
    def throttle(self): return self.model["apollo_state"][:, 0]
    @property
    def propellant(self): return self.model["apollo_state"][:, 6]
    @property
    def landed(self): return self.model["apollo_state"][:, 12]
    @property
    def result(self): return self.model["apollo_result"]

    def results(self) -> Dict[str, np.ndarray]:
        """el.monte_carlo.result(...) fields of main.py:259-271, one array per field."""
        return {name: self.model["apollo_result"][:, j] for j, name in enumerate(RESULT_NAMES[:-1])}


def result_record(row: Mapping[str, float]) -> Dict[str, object]:
    """One RESULT_NAMES row as the `result.json` the example's sim writes at touchdown (flags as booleans, the tick as an
    integer), which `hooks/score.py` reads back (`landed`, `soft_landing`, `touchdown_speed`, ...)."""
    rec: Dict[str, object] = {k: float(v) for k, v in row.items() if k != "reserved"}
    rec["landed"], rec["soft_landing"] = bool(row["landed"]), bool(row["soft_landing"])
    rec["tick"] = int(row["tick"])
    return rec


def run_campaign(plan_table: np.ndarray | None, n_runs: int, n_ticks: int, *, ticks_per_launch: int = 1000,
                 device: int = 0, comm_device="cpu", make_exec=None, comm=None) -> np.ndarray:
    """One Monte-Carlo campaign across the ranks of the current torch.distributed group (or one process).

    Rank 0 holds the plan table ([n_runs, 17], plan.table()); it is broadcast (RCCL over xGMI on "nccl"),
    every rank flies its contiguous block of run ids with no per-step exchange, and the result rows are
    all-gathered back into run-id order — the GPU counterpart of `elodin monte-carlo run`
    (libs/monte-carlo/src/lib.rs:1066-1140: one process per run, results.csv per campaign).
    `make_exec(params_block, first_row)` builds the executor (default: ApolloExec on `device`).
    `comm`: a shard.CapiComm — the same broadcast / gather through the C ABI's RCCL entry points instead of torch."""
    from .. import shard
    import torch.distributed as dist
    if comm is not None:
        world, rank = comm.world, comm.rank
        table = comm.broadcast_table(plan_table, (n_runs, N_PARAMS))
    else:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        table = shard.broadcast_table(plan_table, (n_runs, N_PARAMS), device=comm_device)
    lo, hi = shard.shard_range(n_runs, world, rank)
    if make_exec is None:
        make_exec = lambda block, first_row: ApolloExec(block, ticks_per_launch=ticks_per_launch, device=device,
                                                         first_row=first_row)
    ex = make_exec(table[lo:hi], lo)
    ex.run(n_ticks)
    local = np.ascontiguousarray(ex.result)
    if hasattr(ex, "close"):
        ex.close()
    return comm.gather_rows(local, n_runs) if comm is not None else shard.gather_rows(local, n_runs, device=comm_device)
