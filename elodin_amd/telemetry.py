"""The commit path either side of the stepper: where a batch's columns go, and where a callback's writes come from.

The reference's server loop (libs/nox-py/src/impeller2_server.rs:553-678) is, per batch of ticks_per_telemetry ticks,

    pre_step(tick, ctx) -> copy_db_to_world -> world.run() -> commit_world_head(batch_end_timestamp) -> post_step(end_tick, ctx)

with the database as the hand-off: `commit_world_head` pushes every row of every component column into the time series of its
(entity, component) pair, `copy_db_to_world` overwrites the world's rows with the pairs' latest samples, and a
`StepContext.write_component` from a callback is a push into a pair's series.  `Sink` is that hand-off (native:
csrc/telemetry_sink.cpp behind sixdof_sink_*), without the database behind it: the GPU path's column download lands in it,
callbacks read samples by timestamp from it, external controls flow back through it.

    sink = telemetry.Sink.attach(exec_, world, start_timestamp)    # registers every named entity's pairs
    sink.commit(timestamp_us)                                       # after exec_.run(batch): commit_world_head
    sink.copy_to_world()                                            # before the next batch: copy_db_to_world (uploads if dirty)
    sink.latest("drone.world_pos"), sink.at("drone.world_pos", t), sink.series("drone.world_pos")
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib as L


class TimeTravel(RuntimeError):
    """elodin-db's Error::TimeTravel: a sample older than the last one of its pair (libs/db/src/time_series.rs:206-222)."""


class Sink:
    def __init__(self):
        self._lib = L.lib()
        self._s = C.c_void_p(self._lib.sixdof_sink_create())
        self._width: Dict[int, int] = {}
        self._columns: List[Tuple[str, np.ndarray, callable]] = []     # (component, pair ids per row, rows getter)
        self._ex = None
        self.excluded: set = set()                                     # components the simulation does not commit (external controls)

    def __del__(self):
        try:
            if getattr(self, "_s", None) and self._s.value:
                self._lib.sixdof_sink_destroy(self._s)
                self._s = C.c_void_p()
        except Exception:
            pass

    # ---- pairs -----------------------------------------------------------------------------------------------------------
    @staticmethod
    def pair_id(pair_name: str) -> int:
        entity, _, component = pair_name.rpartition(".")
        return int(L.lib().sixdof_pair_id(entity.encode(), component.encode()))

    def _fail(self, rc, what):
        msg = self._lib.sixdof_sink_last_error(self._s).decode()
        if rc == L.ERR_TIME_TRAVEL:
            raise TimeTravel(msg)
        if rc == L.ERR_VALUE_SIZE_MISMATCH:
            raise ValueError(msg)
        raise RuntimeError(f"{what}: {msg} (status {rc})")

    def register(self, pair_name: str, width: int) -> int:
        pid = self.pair_id(pair_name)
        rc = self._lib.sixdof_sink_register(self._s, pid, int(width) * 8, pair_name.encode())
        if rc != L.OK:
            self._fail(rc, "sixdof_sink_register")
        self._width[pid] = int(width)
        return pid

    def _pid(self, pair_name: str) -> int:
        pid = self.pair_id(pair_name)
        if pid not in self._width:
            raise RuntimeError(f"component {pair_name!r} does not exist")
        return pid

    def push(self, pair_name: str, data, timestamp_us: int) -> None:
        pid = self._pid(pair_name)
        buf = np.ascontiguousarray(data, dtype=np.float64).reshape(-1)
        if buf.size != self._width[pid]:
            raise ValueError(f"component {pair_name!r}: {buf.size} values for a component of {self._width[pid]}")
        rc = self._lib.sixdof_sink_push(self._s, pid, int(timestamp_us), buf.ctypes.data, buf.nbytes)
        if rc != L.OK:
            self._fail(rc, "sixdof_sink_push")

    def latest(self, pair_name: str) -> Tuple[int, np.ndarray]:
        pid = self._pid(pair_name)
        out, ts = np.empty(self._width[pid]), C.c_int64()
        if self._lib.sixdof_sink_latest(self._s, pid, C.byref(ts), out.ctypes.data, out.nbytes) != L.OK:
            raise RuntimeError(f"component {pair_name!r} has no samples")
        return int(ts.value), out

    def at(self, pair_name: str, timestamp_us: int) -> Tuple[int, np.ndarray]:
        """The sample with the greatest timestamp <= timestamp_us (past the last write: the latest)."""
        pid = self._pid(pair_name)
        out, ts = np.empty(self._width[pid]), C.c_int64()
        if self._lib.sixdof_sink_at(self._s, pid, int(timestamp_us), C.byref(ts), out.ctypes.data, out.nbytes) != L.OK:
            raise RuntimeError(f"component {pair_name!r} has no sample at or before {timestamp_us}")
        return int(ts.value), out

    def sample_count(self, pair_name: str) -> int:
        return int(self._lib.sixdof_sink_sample_count(self._s, self.pair_id(pair_name)))

    def series(self, pair_name: str) -> Tuple[np.ndarray, np.ndarray]:
        """(timestamps [k] int64, samples [k, w]) — the two append logs, copied."""
        pid = self._pid(pair_name)
        tp, dp, n, eb = C.POINTER(C.c_int64)(), C.POINTER(C.c_uint8)(), C.c_uint64(), C.c_uint32()
        if self._lib.sixdof_sink_series(self._s, pid, C.byref(tp), C.byref(dp), C.byref(n), C.byref(eb)) != L.OK:
            raise RuntimeError(f"component {pair_name!r} does not exist")
        k, w = int(n.value), self._width[pid]
        if k == 0:
            return np.zeros(0, dtype=np.int64), np.zeros((0, w))
        ts = np.ctypeslib.as_array(tp, shape=(k,)).copy()
        data = np.frombuffer(C.string_at(dp, k * int(eb.value)), dtype=np.float64).reshape(k, w).copy()
        return ts, data

    def truncate(self) -> None:
        self._lib.sixdof_sink_truncate(self._s)

    # ---- the executor's columns ------------------------------------------------------------------------------------------
    @classmethod
    def attach(cls, ex, world, timestamp_us: int = 0, external: Tuple[str, ...] = ()) -> "Sink":
        """Register the pairs of every component column the executor holds (+ the world's spawned components no system
        touches), named like the reference names them — `<entity name>.<component>`; entities without a name are skipped,
        like an entity without metadata is (impeller2_server.rs:418-420) — and commit the spawned state at `timestamp_us`."""
        self = cls()
        self._ex, self._world = ex, world
        self.excluded = set(external)
        names = dict(world._names)
        names.update({eid: nm for nm, eid in world.entity_ids_by_name.items()})       # spawn(id=...): the database name wins
        comps = list(dict.fromkeys(["world_pos", "world_vel", "world_accel", "force", "inertia"] + list(ex._hip._aux) + list(world._components)))
        for comp in comps:
            if comp.startswith(("has:", "mc:")) or comp.endswith("#head") or "#fold" in comp:
                continue
            try:
                rows = np.asarray(ex.column_array(comp))
                ids = np.asarray(ex.column_ids(comp))
                getter = (lambda c=comp: np.asarray(self._ex.column_array(c), dtype=np.float64))
                target = comp
            except KeyError:
                try:
                    rows, ids = world.column(comp)
                except KeyError:
                    continue
                static = np.array(rows, dtype=np.float64)
                getter = (lambda a=static: a)
                target = None
            rows = rows.reshape(len(ids), -1)
            pids = np.zeros(len(ids), dtype=np.uint64)
            for k, eid in enumerate(ids):
                nm = names.get(int(eid))
                if nm is not None:
                    pids[k] = self.register(f"{nm}.{comp}", rows.shape[1])
            self._columns.append((comp, pids, getter, target))
        self.commit(timestamp_us, include_external=True)
        return self

    def commit(self, timestamp_us: int, include_external: bool = False) -> None:
        """commit_world_head: every row of every column -> its pair's series at `timestamp_us` (external controls excepted:
        the database is their source of truth, impeller2_server.rs:423-427)."""
        for comp, pids, getter, _ in self._columns:
            if comp in self.excluded and not include_external:
                continue
            rows = np.ascontiguousarray(getter(), dtype=np.float64).reshape(len(pids), -1)
            rc = self._lib.sixdof_sink_commit_rows(self._s, pids.ctypes.data_as(C.POINTER(C.c_uint64)), rows.ctypes.data, len(pids),
                                                   rows.shape[1] * 8, int(timestamp_us))
            if rc != L.OK:
                self._fail(rc, "sixdof_sink_commit_rows")

    def _real_host_array(self, name: str):
        """The array the executor uploads from for `name` (None: the column is assembled on read)."""
        hip = self._ex._hip
        if name in ("world_pos", "world_vel", "world_accel", "force", "inertia"):
            return getattr(hip, name, None)
        if name in getattr(hip, "_windows", {}) or name in getattr(self._ex, "_partial", {}):
            return None
        return hip._aux.get(name)

    def copy_to_world(self) -> bool:
        """copy_db_to_world: the latest sample of every pair overwrites its row of the executor's host columns; uploads when a
        byte changed.  Returns whether anything did."""
        dirty = False
        for comp, pids, getter, target in self._columns:
            if target is None:
                continue
            host = self._ex._main_column_array(target)
            real = self._real_host_array(target)
            direct = (real is not None and host.shape[0] == len(pids) and host.dtype == np.float64 and host.flags.c_contiguous
                      and np.shares_memory(host, real))
            # what the executor hands out may be a COPY (stand-in rows filtered out by fancy indexing, a reshaped window, a
            # densified partial column, an f32 column): the latest samples go into an f64 staging copy, and a change is
            # scattered into the real host array where that is possible and REFUSED where it is not — never dropped
            stage = host if direct else np.ascontiguousarray(np.asarray(host, dtype=np.float64).reshape(len(pids), -1))
            changed = C.c_int()
            self._lib.sixdof_sink_copy_to_rows(self._s, pids.ctypes.data_as(C.POINTER(C.c_uint64)), stage.ctypes.data, len(pids),
                                               stage.shape[1] * 8, C.byref(changed))
            if changed.value and not direct:
                rows = getattr(self._ex, "_body_rows", None)
                if real is not None and rows is not None and real.dtype == np.float64 and real[rows].shape == stage.shape:
                    real[rows] = stage                      # Body column of an executor with stand-in rows: scatter through the row map
                else:
                    raise NotImplementedError(
                        f"a write to component {target!r} reached the telemetry sink, but its column cannot be written back to the "
                        "executor (a window, a component on fewer entities than the row set, or a float32 column): same refusal as "
                        "StepContext.write_component on the direct path")
            dirty = dirty or bool(changed.value)
        if dirty:
            self._ex._hip.upload()
        return dirty
