"""ctypes loader for the CPU oracle (oracle/sixdof_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by
anything under elodin_amd/.  It is the checker, not the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libsixdof_oracle.so"
MAX_OPS = 8

# sixdof_effector_kind (include/sixdof_hip.h)
EFF_CONST_WRENCH = 1
EFF_UNIFORM_GRAVITY = 2
EFF_BODY_TORQUE = 3
EFF_BODY_FORCE = 4
EFF_BALL_DRAG = 5
EFF_EDGE_GRAVITY_NEWTON = 6
EFF_EDGE_GRAVITY_SOFTENED = 7
EFF_ALLPAIRS_GRAVITY_SOFTENED = 8
EFF_WORLD_TORQUE = 10
EFF_WORLD_FORCE = 11
RK4, SEMI_IMPLICIT = 0, 1


class EffectorOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("aux_component_id", C.c_uint64),
                ("p", C.c_double * 6)]


class _World(C.Structure):
    _fields_ = [
        ("n", C.c_uint64),
        ("world_pos", C.c_void_p), ("world_vel", C.c_void_p), ("world_accel", C.c_void_p),
        ("force", C.c_void_p), ("inertia", C.c_void_p),
        ("tick", C.c_uint64),
        ("simulation_time_step", C.c_double), ("time_step", C.c_double),
        ("has_time_step", C.c_int32), ("integrator", C.c_int32),
        ("n_ops", C.c_uint32), ("pair_threads", C.c_uint32),
        ("ops", EffectorOp * MAX_OPS),
        ("aux", C.c_void_p * MAX_OPS),
        ("edge_src", C.c_void_p), ("edge_dst", C.c_void_p), ("n_edges", C.c_uint64),
    ]


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (no-FMA).  Building the checker is not using it."""
    newest = max((HERE / f).stat().st_mtime for f in ("sixdof_oracle.c", "apollo_oracle.c", "falcon9_fsw.c", "sixdof_oracle.h", "apollo_oracle.h", "falcon9_fsw.h"))
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < newest:
        subprocess.run(["make", "-C", str(HERE), "-B", "libsixdof_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def use_native_build() -> Path:
    """bench.py's cpu_baseline: (re)build the oracle -O3 -march=native -fno-tree-vectorize ON THIS MACHINE (oracle/Makefile `native`) and bind
    to it from here on.  Returns the library path actually in use (the portable -O2 build if the native one fails)."""
    global _lib, LIB_PATH
    native = HERE / "_native" / "libsixdof_oracle_native.so"
    try:
        subprocess.run(["make", "-C", str(HERE), "native"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        LIB_PATH, _lib = native, None
    except (OSError, subprocess.CalledProcessError):
        pass
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        L = C.CDLL(str(LIB_PATH))
        dp = C.POINTER(C.c_double)
        for name, n in (("orc_quat_mul", 3), ("orc_quat_inverse", 2), ("orc_quat_normalize", 2),
                        ("orc_quat_rotate", 3), ("orc_quat_integrate_body", 3),
                        ("orc_transform_add_motion", 3), ("orc_transform_mul", 3)):
            getattr(L, name).argtypes = [dp] * n
            getattr(L, name).restype = None
        L.orc_quat_from_axis_angle.argtypes = [dp, C.c_double, dp]
        L.orc_quat_from_axis_angle.restype = None
        L.orc_calc_accel.argtypes = [dp] * 4
        L.orc_calc_accel.restype = None
        L.orc_component_id.argtypes = [C.c_char_p]
        L.orc_component_id.restype = C.c_uint64
        L.orc_quantize_time_step.argtypes = [C.c_double]
        L.orc_quantize_time_step.restype = C.c_double
        L.orc_step.argtypes = [C.POINTER(_World), C.c_uint64]
        L.orc_step.restype = C.c_int
        L.orc_step_omp.argtypes = [C.POINTER(_World), C.c_uint64, C.c_int]
        L.orc_step_omp.restype = C.c_int
        u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        L.orc_resolve_edges.argtypes = [u64p, C.c_uint64, u64p, u64p, C.c_uint64, u32p, u32p]
        L.orc_resolve_edges.restype = C.c_int
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vec(fn, n_out, *ins):
    arrs = [np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in ins]
    out = np.zeros(n_out)
    fn(*[_dp(a) for a in arrs], _dp(out))
    return out


def quat_mul(l, r): return _vec(lib().orc_quat_mul, 4, l, r)
def quat_inverse(q): return _vec(lib().orc_quat_inverse, 4, q)
def quat_normalize(q): return _vec(lib().orc_quat_normalize, 4, q)
def quat_rotate(q, v): return _vec(lib().orc_quat_rotate, 3, q, v)
def quat_integrate_body(q, d): return _vec(lib().orc_quat_integrate_body, 4, q, d)
def transform_add_motion(x, m): return _vec(lib().orc_transform_add_motion, 7, x, m)
def transform_mul(a, b): return _vec(lib().orc_transform_mul, 7, a, b)
def calc_accel(F, I, x): return _vec(lib().orc_calc_accel, 6, F, I, x)


def quat_from_axis_angle(axis, angle):
    a = np.ascontiguousarray(np.asarray(axis, dtype=np.float64))
    out = np.zeros(4)
    lib().orc_quat_from_axis_angle(_dp(a), float(angle), _dp(out))
    return out


def component_id(name: str) -> int:
    return int(lib().orc_component_id(name.encode()))


def quantize_time_step(rate_hz: float) -> float:
    return float(lib().orc_quantize_time_step(float(rate_hz)))


def resolve_edges(body_ids, from_ids, to_ids):
    body_ids = np.ascontiguousarray(body_ids, dtype=np.uint64)
    f = np.ascontiguousarray(from_ids, dtype=np.uint64)
    t = np.ascontiguousarray(to_ids, dtype=np.uint64)
    src = np.zeros(len(f), dtype=np.uint32)
    dst = np.zeros(len(f), dtype=np.uint32)
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    rc = lib().orc_resolve_edges(body_ids.ctypes.data_as(u64p), len(body_ids), f.ctypes.data_as(u64p),
                                 t.ctypes.data_as(u64p), len(f), src.ctypes.data_as(u32p),
                                 dst.ctypes.data_as(u32p))
    if rc != 0:
        raise KeyError("edge endpoint is not a Body entity")
    return src, dst


class OracleWorld:
    """Body columns in the reference layout + effector list; stepped in place on the CPU."""

    def __init__(self, world_pos, world_vel, inertia, *, world_accel=None, force=None,
                 simulation_time_step=1.0 / 120.0, time_step=None, integrator=RK4,
                 ops=(), edges=None, tick=0):
        self.world_pos = np.array(world_pos, dtype=np.float64, order="C").reshape(-1, 7)
        n = self.world_pos.shape[0]
        self.n = n
        self.world_vel = np.array(world_vel, dtype=np.float64, order="C").reshape(n, 6)
        self.inertia = np.array(inertia, dtype=np.float64, order="C").reshape(n, 7)
        self.world_accel = (np.zeros((n, 6)) if world_accel is None
                            else np.array(world_accel, dtype=np.float64, order="C").reshape(n, 6))
        self.force = (np.zeros((n, 6)) if force is None
                      else np.array(force, dtype=np.float64, order="C").reshape(n, 6))
        self._w = _World()
        w = self._w
        w.n = n
        for name in ("world_pos", "world_vel", "world_accel", "force", "inertia"):
            setattr(w, name, getattr(self, name).ctypes.data)
        w.tick = tick
        w.simulation_time_step = simulation_time_step
        w.has_time_step = 0 if time_step is None else 1
        w.time_step = 0.0 if time_step is None else float(time_step)
        w.integrator = integrator
        self._keep = []
        assert len(ops) <= MAX_OPS
        w.n_ops = len(ops)
        for k, op in enumerate(ops):
            kind, p, aux = op  # (kind, params tuple, aux [n,3] array or None)
            w.ops[k].kind = kind
            for j, v in enumerate(p):
                w.ops[k].p[j] = float(v)
            if aux is not None:
                a = np.ascontiguousarray(np.asarray(aux, dtype=np.float64).reshape(n, 3))
                self._keep.append(a)
                w.aux[k] = a.ctypes.data
        if edges is not None:
            src, dst = edges
            self._src = np.ascontiguousarray(src, dtype=np.uint32)
            self._dst = np.ascontiguousarray(dst, dtype=np.uint32)
            w.edge_src, w.edge_dst, w.n_edges = self._src.ctypes.data, self._dst.ctypes.data, len(self._src)

    @property
    def tick(self) -> int:
        return int(self._w.tick)

    def step(self, n_ticks: int = 1, threads: int = 1):
        self._w.pair_threads = max(1, int(threads))
        if threads > 1:
            rc = lib().orc_step_omp(C.byref(self._w), n_ticks, threads)
        else:
            rc = lib().orc_step(C.byref(self._w), n_ticks)
        if rc != 0:
            raise MemoryError("oracle step failed")
        return self
