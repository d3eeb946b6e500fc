"""HipExec — the backend object a `WorldExec::Hip` variant would wrap.

Mirrors the contract of the reference backends (libs/nox-py/src/exec.rs:53-93,
cranelift_exec.rs:129-195 `invoke_batch(world, n, detailed) -> TickTimings`), over columns
held as numpy arrays in the reference's row-major layout.  All compute goes through the C ABI.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import os
from pathlib import Path

import numpy as np

from . import _lib as L


def _raise(h, rc: int, what: str):
    msg = L.lib().sixdof_last_error(h)
    msg = msg.decode() if msg else ""
    if rc == L.ERR_COMPONENT_NOT_FOUND:
        raise KeyError(f"{what}: {msg}")           # Error::ComponentNotFound -> ValueError-class in PyO3
    if rc in (L.ERR_VALUE_SIZE_MISMATCH, L.ERR_INVALID_ARGUMENT, L.ERR_ENTITY_MISMATCH, L.ERR_UNSUPPORTED):
        raise ValueError(f"{what}: {msg}")
    if rc == L.ERR_OUT_OF_MEMORY:
        raise MemoryError(f"{what}: {msg}")
    raise L.BackendError(f"{what}: {msg} (status {rc})")


@dataclass
class Effector:
    """One op of the effector pipe (include/sixdof_hip.h sixdof_effector_kind)."""
    kind: int
    p: Sequence[float] = ()
    aux_name: Optional[str] = None          # component name of a per-entity [n,3] column
    aux: Optional[np.ndarray] = None        # its data


@dataclass
class TickTimings:  # profile.rs TickTimings
    h2d_upload_ms: float = 0.0
    kernel_invoke_ms: float = 0.0
    d2h_download_ms: float = 0.0
    kernel_device_ms: float = 0.0
    launches: int = 0
    ticks: int = 0
    kernel_sum_ms: float = 0.0
    graph_launches: int = 0    # how many of `launches` were replayed from a captured hipGraph


class HipExec:
    def __init__(self, world_pos, world_vel, inertia, *, world_accel=None, force=None, entity_ids=None,
                 simulation_time_step: float = 1.0 / 120.0, time_step: Optional[float] = None,
                 integrator: int = L.RK4, dtype=np.float64, effectors: Sequence[Effector] = (),
                 edges=None, ticks_per_launch: int = 1, use_graph: bool = False, device: int = 0,
                 tick: int = 0, column_entity_ids=None, columns=None, fast_math: bool = False, graph_edges=None,
                 graph_replicas=None, guard_selects: Optional[bool] = None, reuse_trace: bool = False):
        lib = L.lib()
        self._lib = lib
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("dtype must be float64 or float32")
        f = lambda a, w: np.array(a, dtype=self.dtype, order="C").reshape(-1, w)
        self.world_pos = f(world_pos, 7)
        self.world_vel = f(world_vel, 6)
        self.inertia = f(inertia, 7)
        nv = self.world_vel.shape[0]
        self.world_accel = np.zeros((nv, 6), self.dtype) if world_accel is None else f(world_accel, 6)
        self.force = np.zeros((nv, 6), self.dtype) if force is None else f(force, 6)
        # ids are sequential from 1 (0 = Globals) unless given: world.rs:193-196.  `column_entity_ids` gives a
        # column its own id vector when components live on different entity sets (six_dof then runs on the
        # intersection, query.rs:136-208)
        self.entity_ids = (np.arange(1, self.world_pos.shape[0] + 1, dtype=np.uint64) if entity_ids is None
                           else np.ascontiguousarray(entity_ids, dtype=np.uint64))
        self._column_ids = {k: np.ascontiguousarray(v, dtype=np.uint64) for k, v in (column_entity_ids or {}).items()}
        n = self.world_pos.shape[0] if not self._column_ids else 0   # 0 = let the library size the join
        # rows the device really steps: with per-column id vectors six_dof runs on the INTERSECTION of the Body columns' entity
        # sets (csrc/sixdof_capi.cpp sixdof_bind_columns, query.rs:136-208) — every build-time choice below that depends on the
        # row count (one-launch pair kernel, cache policy, column layout, whole worlds per executor) uses this, not world_pos's
        self._n_rows = self.world_pos.shape[0]
        if self._column_ids:
            joined = None
            for name in ("world_pos", "world_vel", "world_accel", "force", "inertia"):
                ids = self._column_ids.get(name, self.entity_ids)
                joined = set(ids.tolist()) if joined is None else joined & set(ids.tolist())
            self._n_rows = len(joined)
        self._aux = {}
        self._windows = {}
        self._window_soa = False
        self._column_soa = False
        self._soa = {}            # program column name -> the element-major staging array bound to the C ABI
        d = L.Desc()
        d.struct_size = C.sizeof(L.Desc)
        d.device_ordinal = device
        d.integrator = integrator
        d.dtype = L.F64 if self.dtype == np.float64 else L.F32
        d.n_entities = n
        d.simulation_time_step = simulation_time_step
        d.has_time_step = 0 if time_step is None else 1
        d.time_step = 0.0 if time_step is None else float(time_step)
        d.ticks_per_launch = ticks_per_launch
        d.flags = L.FLAG_USE_GRAPH if use_graph else 0
        self._h = C.c_void_p()
        try:
            cols = [("world_pos", self.world_pos), ("world_vel", self.world_vel), ("world_accel", self.world_accel),
                    ("force", self.force), ("inertia", self.inertia)]
            custom = None
            from . import dsl as _dsl
            if isinstance(effectors, _dsl.Effector):
                effectors = _dsl.pipe(effectors)
            self._program_columns = []
            if isinstance(effectors, (_dsl.Pipe, _dsl.Program)):
                # user-written effectors / systems: trace -> generate HIP -> hipcc -> sixdof_set_custom_pipe
                from . import codegen
                # [n, w] columns give their row width; an [n, rows, w] column is a window component (dsl.Window)
                widths = {k: (tuple(int(x) for x in np.shape(v)[1:]) if np.ndim(v) == 3 else int(np.atleast_2d(np.asarray(v)).shape[-1]))
                          for k, v in (columns or {}).items()}
                fold_rows = None
                # A Pipe / Program keeps its first trace (`_traced`).  An executor starts from a FRESH one unless the caller promises
                # (reuse_trace=True) that nothing a trace reads has changed since: the program's systems, and every Python value their
                # functions close over (gains, host tables, parameter sentinels).  A frozen program has no functions to trace again.
                if not reuse_trace and not isinstance(effectors, _dsl.FrozenProgram) and getattr(effectors, "_traced", None) is not None:
                    effectors._traced = None
                if isinstance(effectors, _dsl.Program) and effectors.folds:
                    # stand-alone folds inside the program: their edges as row pairs of this executor (spawn order)
                    if self._column_ids:
                        raise ValueError("a program with stand-alone folds needs every column on the executor's own entity ids")
                    row_of = {int(e): k for k, e in enumerate(self.entity_ids)}
                    fold_rows = {}
                    for name, (frm, to) in (graph_edges or {}).items():
                        if isinstance(frm, str) and frm == "complete":      # the complete graph over a world's rows: nothing to resolve
                            fold_rows[name] = (frm, int(to))
                            continue
                        try:
                            fold_rows[name] = ([row_of[int(a)] for a in frm], [row_of[int(b)] for b in to])
                        except KeyError as e:
                            raise KeyError(f"graph_edges[{name!r}]: edge endpoint {e} is not an entity of this executor") from None
                    if graph_replicas is not None and int(graph_replicas[0]) * int(graph_replicas[1]) != len(self.entity_ids):
                        raise ValueError("graph_replicas=(count, rows per replica) must cover the executor's rows exactly")
                    custom = effectors.trace(widths, fold_edges=fold_rows, fold_replicas=graph_replicas)
                    columns = dict(columns or {})
                    for fs in custom.fold_stages:       # scratch rows: a fold reads the values from before it ran
                        columns.setdefault(fs.scratch_name, np.zeros((self.world_pos.shape[0], fs.out[2])))
                else:
                    # executors built again and again from ONE program object (a campaign service: one per block of runs)
                    # trace it and generate its source once per column layout; the objects themselves are cached on disk
                    # — OPT-IN (`reuse_trace=True`): the memo returns the FIRST trace, so the caller promises that neither the
                    # program's systems nor any Python value its functions close over (gains, host tables, parameter
                    # sentinels) changes between the executors.  Without the promise every executor traces again; the object
                    # is still found in the on-disk cache by the digest of the generated source.
                    memo = (effectors.__dict__.setdefault("_exec_memo", {}) if reuse_trace and hasattr(effectors, "__dict__") else None)
                    wkey = (tuple(sorted((k, v) for k, v in widths.items())),
                            tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("SIXDOF_"))),
                            tuple(sorted((str(k), repr(v)) for k, v in getattr(_dsl, "PARAM_SENTINELS", {}).items())))
                    custom = memo.get(("trace", wkey)) if memo is not None else None
                    if custom is None:
                        custom = effectors.trace(widths)
                        if memo is not None:
                            memo[("trace", wkey)] = custom
                rows_multiple = int(getattr(custom, "rows_multiple", 0) or codegen.lane_stride(custom))
                if getattr(custom, "exact_rows", 0) and self._n_rows != int(custom.exact_rows):
                    raise ValueError(f"this object's fold stages were generated for {int(custom.exact_rows)} rows (manifest 'row_count'); the executor has {self._n_rows}")
                if rows_multiple > 1 and self._n_rows % rows_multiple:
                    raise ValueError(f"this program exchanges data between the entities of a world laid out as {rows_multiple} consecutive rows "
                                     f"(a whole-world StableHLO tick in lane mode, manifest 'rows_per_world'): the executor's {self._n_rows} "
                                     f"rows are not a whole number of worlds")
                if isinstance(effectors, _dsl.Program):
                    self._program_columns = [n for n, _ in custom.columns]
                    self._windows = {name: (rows, width) for name, (_, rows, width) in custom.windows.items()}
                    self._mats = dict(custom.table.mats)       # small 2-D components held as register matrices
                    columns = dict(columns or {})
                    for name in self._windows:       # the ring's head (physical index of the oldest row): starts at 0
                        columns.setdefault(name + "#head", np.zeros((np.shape(columns[name])[0], 1)) if name in columns else None)
                self._window_soa = bool(self._windows) and self._n_rows >= codegen.WINDOW_SOA_MIN_ROWS
                # register columns of a large program executor: element-major on the device (codegen.COLUMN_SOA_MIN_ROWS); the
                # host-facing arrays (self._aux) stay in the reference's [n, w] rows, upload / download transpose
                soa_env = os.environ.get("SIXDOF_COLUMN_SOA")
                self._column_soa = (isinstance(effectors, _dsl.Program) and not getattr(custom, "fold_stages", None)
                                    and not self._column_ids and (soa_env == "1" or (soa_env != "0" and
                                                                  self._n_rows >= codegen.COLUMN_SOA_MIN_ROWS)))
                if getattr(custom, "frozen_source", None) is not None or getattr(custom, "prebuilt_so", None) is not None:      # generated for ONE device layout
                    self._column_soa, self._window_soa = bool(custom.column_soa), False
                memo = effectors.__dict__.get("_exec_memo") if reuse_trace and hasattr(effectors, "__dict__") else None
                bkey = ("build", id(custom), self.dtype.name, integrator, bool(fast_math), self._window_soa, self._column_soa, guard_selects,
                        self._n_rows * self.dtype.itemsize * (32 + sum(int(w_) for _, w_ in custom.columns)) <= (768 << 20),
                        tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("SIXDOF_"))))
                so = memo.get(bkey) if memo is not None else None
                if so is None or not Path(so).exists():
                    row_elems = 32 + sum(int(w_) for _, w_ in custom.columns)       # what fill_step_params counts (csrc/sixdof_capi.cpp)
                    so = codegen.build(custom, self.dtype.name, integrator, fast_math=fast_math, window_soa=self._window_soa,
                                       column_soa=self._column_soa, guard_selects=guard_selects,
                                       policy=codegen.policy_for(self._n_rows, row_elems, self.dtype.itemsize))
                    if memo is not None:
                        memo[bkey] = so
                for name, width in custom.columns:
                    if columns is None or columns.get(name) is None:
                        raise KeyError(f"effector reads component {name!r} which was not provided")
                    arr = np.array(columns[name], dtype=self.dtype, order="C").reshape(-1, width)
                    if name in self._windows and self._window_soa:
                        arr = np.ascontiguousarray(arr.T).reshape(arr.shape)   # large executors: element-major [rows*width][n] (codegen.py)
                    self._aux[name] = arr
                    if self._column_soa and name not in self._windows and width > 1:
                        self._soa[name] = np.ascontiguousarray(arr.T).reshape(arr.shape)    # what the device holds
                        cols.append((name, self._soa[name]))
                    else:
                        cols.append((name, arr))
                effectors = ()
            pair_so = None
            effectors = list(effectors)
            if effectors and isinstance(effectors[-1], _dsl.EdgeFold):
                # user-written edge_fold function: trace -> generate the PAIR functor -> hipcc -> sixdof_set_custom_pair
                from . import codegen
                if self.dtype != np.float64:
                    raise ValueError("edge_fold effectors are float64 only")
                if edges is None:
                    raise ValueError("an edge_fold effector needs edges=(from_ids, to_ids)")
                # the object is built for THIS executor: its integrator, and the one-launch small-graph kernel or the multi-kernel
                # tick (pack once per batch, then the fused fold-and-integrate launch per tick) — the rule csrc/sixdof_capi.cpp applies
                # at step time (kPairSmallMax = 256; SIXDOF_PAIR_SMALL=0 forces the multi-kernel path)
                pair_small = self._n_rows <= 256 and os.environ.get("SIXDOF_PAIR_SMALL", "")[:1] != "0"
                pair_so = codegen.build_pair(effectors.pop().trace(), integrator=integrator, small=pair_small)
            if any(isinstance(e, _dsl.EdgeFold) for e in effectors):
                raise ValueError("an edge_fold effector must be last in the pipe")
            ops = (L.EffectorOp * max(1, len(effectors)))()
            for k, e in enumerate(effectors):
                ops[k].kind = e.kind
                for j, v in enumerate(e.p):
                    ops[k].p[j] = float(v)
                if e.aux is not None:
                    name = e.aux_name or f"effector_aux_{k}"
                    arr = np.array(e.aux, dtype=self.dtype, order="C").reshape(-1, 3)
                    self._aux[name] = arr
                    cols.append((name, arr))
                    ops[k].aux_component_id = L.component_id(name)
            # user code is traced and compiled above, before the device is touched: a program that cannot be built fails
            # without a context, and a machine without a GPU can fill the JIT cache (elodin_amd/_jit) for one that has it
            rc = lib.sixdof_create(C.byref(d), C.byref(self._h))
            if rc != L.OK:
                _raise(None, rc, "sixdof_create")
            self._bind(cols)
            rc = lib.sixdof_set_effectors(self._h, ops, len(effectors))
            if rc != L.OK:
                _raise(self._h, rc, "sixdof_set_effectors")
            if custom is not None:
                ids = (C.c_uint64 * max(1, len(custom.columns)))(*[L.component_id(n) for n, _ in custom.columns])
                rc = lib.sixdof_set_custom_pipe(self._h, str(so).encode(), ids, len(custom.columns))
                if rc != L.OK:
                    _raise(self._h, rc, "sixdof_set_custom_pipe")
            if pair_so is not None:
                rc = lib.sixdof_set_custom_pair(self._h, str(pair_so).encode())
                if rc != L.OK:
                    _raise(self._h, rc, "sixdof_set_custom_pair")
            if edges is not None:
                frm = np.ascontiguousarray(edges[0], dtype=np.uint64)
                to = np.ascontiguousarray(edges[1], dtype=np.uint64)
                u64p = C.POINTER(C.c_uint64)
                rc = lib.sixdof_set_edges(self._h, frm.ctypes.data_as(u64p), to.ctypes.data_as(u64p), len(frm))
                if rc != L.OK:
                    _raise(self._h, rc, "sixdof_set_edges")
            lib.sixdof_set_tick(self._h, tick)
            self.upload()
        except Exception:
            self.close()
            raise

    def _bind(self, named_arrays):
        cols = (L.Column * len(named_arrays))()
        prim = L.PRIM_F64 if self.dtype == np.float64 else L.PRIM_F32
        for c, (name, arr) in zip(cols, named_arrays):
            c.component_id = L.component_id(name)
            c.prim_type = prim
            c.ndim = 1
            c.dims[0] = arr.shape[1]
            c.n_rows = arr.shape[0]
            ids = self._column_ids.get(name, self.entity_ids)
            if len(ids) != arr.shape[0]:
                raise ValueError(f"column {name}: {arr.shape[0]} rows but {len(ids)} entity ids")
            c.entity_ids = ids.ctypes.data_as(C.POINTER(C.c_uint64))
            c.host_ptr = arr.ctypes.data
        rc = self._lib.sixdof_bind_columns(self._h, cols, len(named_arrays))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_bind_columns")
        m = C.c_size_t()
        self._lib.sixdof_get_join_rows(self._h, L.component_id("world_pos"), None, 0, C.byref(m))
        self.n = int(m.value)   # size of the joined Body entity set

    # -- reference-shaped surface -------------------------------------------------------------------
    def upload(self):
        for name, dev in self._soa.items():      # [n, w] rows -> [w][n]
            a = self._aux[name]
            dev.reshape(a.shape[1], a.shape[0])[...] = a.T
        rc = self._lib.sixdof_upload(self._h)
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_upload")

    def invoke_batch(self, n_ticks: int) -> TickTimings:
        """cranelift_exec.rs:129-195: run n ticks back to back; state stays in HBM."""
        t = L.Timings()
        rc = self._lib.sixdof_step(self._h, int(n_ticks), C.byref(t))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_step")
        return TickTimings(t.h2d_upload_ms, t.kernel_invoke_ms, t.d2h_download_ms, t.kernel_device_ms,
                           int(t.launches), int(t.ticks), t.kernel_sum_ms, int(t.graph_launches))

    def prepare(self, n_ticks: int):
        """Capture the replay graphs a later invoke_batch(n_ticks) uses (use_graph=True), outside any timed region."""
        rc = self._lib.sixdof_prepare_step(self._h, int(n_ticks))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_prepare_step")

    def download(self, mask: int = L.COL_ALL):
        rc = self._lib.sixdof_download(self._h, mask)
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_download")
        for name in getattr(self, "_program_columns", ()):   # components a generated program writes
            rc = self._lib.sixdof_download_column(self._h, L.component_id(name))
            if rc != L.OK:
                _raise(self._h, rc, "sixdof_download_column")
        for name, dev in self._soa.items():      # [w][n] -> [n, w] rows
            a = self._aux[name]
            a[...] = dev.reshape(a.shape[1], a.shape[0]).T
        return self

    def download_column(self, name: str) -> np.ndarray:
        """D2H of ONE bound column (a Body column or a program component): what StepContext.read_component costs."""
        rc = self._lib.sixdof_download_column(self._h, L.component_id(name))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_download_column")
        if name in self._soa:
            a = self._aux[name]
            a[...] = self._soa[name].reshape(a.shape[1], a.shape[0]).T
        return getattr(self, name) if name in ("world_pos", "world_vel", "world_accel", "force", "inertia") else self._aux[name]

    def upload_column(self, name: str) -> None:
        """H2D of ONE bound column from its host array (StepContext.write_component; copy_db_to_world's per-component copy): the
        other columns — which the host may never have downloaded — are left alone."""
        if name in self._soa:
            a = self._aux[name]
            self._soa[name].reshape(a.shape[1], a.shape[0])[...] = a.T
        rc = self._lib.sixdof_upload_column(self._h, L.component_id(name))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_upload_column")

    def component(self, name: str) -> np.ndarray:
        """A generated program's component column as the reference lays it out.  Plain columns: the [n, w] host array
        itself.  Window components (dsl.Window) are kept as a ring on the device: un-rotated here to [n, rows, w], oldest
        row first — what `concatenate((buffer[1:], row))` leaves in the reference's column."""
        if name in getattr(self, "_mats", {}):
            return self._aux[name].reshape(self._aux[name].shape[0], *self._mats[name])
        if name not in self._windows:
            return self._aux[name]
        rows, width = self._windows[name]
        n = self._aux[name].shape[0]
        ring = (self._aux[name].reshape(rows * width, n).T if self._window_soa else self._aux[name]).reshape(n, rows, width)
        head = self._aux[name + "#head"][:, 0].astype(np.int64)
        idx = (head[:, None] + np.arange(rows)[None, :]) % rows
        return ring[np.arange(ring.shape[0])[:, None], idx]

    def run(self, ticks: int = 1) -> TickTimings:
        """PyExec.run (exec.rs:110-172) without the DB commit: step, then refresh the host columns."""
        t = self.invoke_batch(ticks)
        self.download()
        return t

    @property
    def tick(self) -> int:
        v = C.c_uint64()
        self._lib.sixdof_get_tick(self._h, C.byref(v))
        return int(v.value)

    def set_ticks_per_launch(self, k: int):
        rc = self._lib.sixdof_set_ticks_per_launch(self._h, int(k))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_set_ticks_per_launch")

    def nonfinite_rows(self) -> np.ndarray:
        """Failure sentinel: boolean [n], True where a row's world_pos / world_vel holds NaN or Inf."""
        flags = np.zeros(self.n, dtype=np.uint8)
        cnt = C.c_uint64()
        rc = self._lib.sixdof_count_nonfinite(self._h, C.byref(cnt), flags.ctypes.data)
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_count_nonfinite")
        assert int(cnt.value) == int(flags.sum())
        return flags.astype(bool)

    def checkpoint(self) -> dict:
        """Sim state = the Body columns + tick (World is `Serialize` in the reference, world.rs:41-46)."""
        self.download()
        state = {k: getattr(self, k).copy() for k in ("world_pos", "world_vel", "world_accel", "force", "inertia")}
        state.update({f"aux:{k}": v.copy() for k, v in self._aux.items()})
        state["tick"] = self.tick
        return state

    def restore(self, state: dict):
        for k in ("world_pos", "world_vel", "world_accel", "force", "inertia"):
            getattr(self, k)[...] = state[k]
        for k, v in self._aux.items():
            v[...] = state[f"aux:{k}"]
        self._lib.sixdof_set_tick(self._h, int(state["tick"]))
        self.upload()

    def last_timings(self) -> TickTimings:
        t = L.Timings()
        self._lib.sixdof_last_timings(self._h, C.byref(t))
        return TickTimings(t.h2d_upload_ms, t.kernel_invoke_ms, t.d2h_download_ms, t.kernel_device_ms,
                           int(t.launches), int(t.ticks), t.kernel_sum_ms, int(t.graph_launches))

    def enable_history(self, ring_ticks: int):
        """Record every tick's world_pos/world_vel/world_accel/force in a device ring (sixdof_set_history)."""
        rc = self._lib.sixdof_set_history(self._h, int(ring_ticks))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_set_history")

    def history(self, name: str, first_tick: int, last_tick: int) -> np.ndarray:
        """exec.history() analogue: [last-first+1, n, w] block of component `name`, row k = state after tick first+k."""
        if name in self._windows:
            raise ValueError(f"{name} is a window component: it is its own history (HipExec.component), the ring does not copy it per tick")
        w = 7 if name == "world_pos" else (self._aux[name].shape[1] if name in self._aux else 6)   # program columns too
        out = np.empty((last_tick - first_tick + 1, self.n, w), dtype=self.dtype)
        for k, tick in enumerate(range(first_tick, last_tick + 1)):
            rc = self._lib.sixdof_history_read(self._h, L.component_id(name), tick, out[k].ctypes.data)
            if rc != L.OK:
                _raise(self._h, rc, "sixdof_history_read")
        return out

    def set_flags(self, flags: int):
        self._lib.sixdof_set_flags(self._h, int(flags))

    # ---- telemetry commit overlapped with the next batch -------------------------------------------------------
    def download_async(self, mask: int = L.COL_ALL & ~L.COL_INERTIA):
        rc = self._lib.sixdof_download_async(self._h, int(mask))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_download_async")

    def download_wait(self):
        rc = self._lib.sixdof_download_wait(self._h)
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_download_wait")

    def sync(self):
        rc = self._lib.sixdof_sync(self._h)
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_sync")

    def stream_history(self, n_batches: int, ticks_per_batch: int, consume=None, columns=("world_pos", "world_vel", "world_accel", "force"),
                       flags: int = 0) -> float:
        """EVERY tick of every batch to the host, overlapped with the stepper: the kernel records each tick into the
        device ring (two batches deep), sixdof_history_stream copies batch i's [ticks, n, w] blocks into one of two
        page-locked host buffers on the copy stream while batch i+1 computes.  `consume(batch_index, first_tick,
        {column: array [ticks, n, w]})` sees a batch once it has landed (the arrays are reused two batches later).
        Returns the wall time in seconds."""
        import time
        names = ("world_pos", "world_vel", "world_accel", "force")
        rc = self._lib.sixdof_set_history(self._h, 2 * int(ticks_per_batch))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_set_history")
        bufs = [{c: np.empty((ticks_per_batch, self.n, 7 if c == "world_pos" else 6), dtype=self.dtype) for c in columns}
                for _ in range(2)]
        ptrs = [(C.c_void_p * 4)(*[b[c].ctypes.data if c in b else None for c in names]) for b in bufs]
        self.set_flags(flags | L.FLAG_ASYNC_STEP)
        t0 = time.perf_counter()
        try:
            first = []
            for i in range(n_batches):
                first.append(self.tick + 1)
                self.invoke_batch(ticks_per_batch)                       # enqueue batch i (records into ring half i % 2)
                if i > 0:
                    self.download_wait()                                 # batch i-1 has landed in bufs[(i-1) % 2]
                    if consume is not None:
                        consume(i - 1, first[i - 1], bufs[(i - 1) % 2])
                rc = self._lib.sixdof_history_stream(self._h, first[i], ticks_per_batch, ptrs[i % 2])
                if rc != L.OK:
                    _raise(self._h, rc, "sixdof_history_stream")
            self.download_wait()
            if consume is not None and n_batches:
                consume(n_batches - 1, first[-1], bufs[(n_batches - 1) % 2])
        finally:
            self.set_flags(flags)
            self.sync()          # also drops the page locks on `bufs` before they go out of scope
        return time.perf_counter() - t0

    def run_streaming(self, n_batches: int, ticks_per_batch: int, consume=None, flags: int = 0) -> float:
        """The run loop of exec.rs:110-172 (`run(ticks)` = batches of ticks_per_telemetry ticks, each followed by a
        commit of the output columns) with the commit off the critical path: batch i+1 is enqueued before batch i's
        columns are waited for, and they travel to the (page-locked) host columns on a second stream meanwhile.
        `consume(batch_index)` is called when the host columns hold that batch's state (the DB commit goes there).
        Returns the wall time in seconds."""
        import time
        self.set_flags(flags | L.FLAG_ASYNC_STEP)
        t0 = time.perf_counter()
        try:
            for i in range(n_batches):
                self.invoke_batch(ticks_per_batch)          # enqueue batch i (returns once batch i-1 has computed)
                if i > 0:
                    self.download_wait()                    # batch i-1 is in the host columns
                    if consume is not None:
                        consume(i - 1)
                self.download_async()                       # snapshot of batch i; its copy overlaps batch i+1
            self.download_wait()
            if consume is not None and n_batches:
                consume(n_batches - 1)
            self.sync()
        finally:
            self.set_flags(flags)
        return time.perf_counter() - t0

    def join_rows(self, name: str) -> np.ndarray:
        """Row of joined entity j inside column `name` (the reference's constant u32 gather indices)."""
        rows = np.zeros(self.n, dtype=np.uint32)
        m = C.c_size_t()
        rc = self._lib.sixdof_get_join_rows(self._h, L.component_id(name), rows.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            self.n, C.byref(m))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_get_join_rows")
        return rows

    def edge_rows(self):
        n = C.c_size_t()
        self._lib.sixdof_get_edge_rows(self._h, None, None, 0, C.byref(n))
        src = np.zeros(n.value, dtype=np.uint32)
        dst = np.zeros(n.value, dtype=np.uint32)
        u32p = C.POINTER(C.c_uint32)
        rc = self._lib.sixdof_get_edge_rows(self._h, src.ctypes.data_as(u32p), dst.ctypes.data_as(u32p), n.value,
                                            C.byref(n))
        if rc != L.OK:
            _raise(self._h, rc, "sixdof_get_edge_rows")
        return src, dst

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.sixdof_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
