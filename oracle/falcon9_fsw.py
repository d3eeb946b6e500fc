"""ctypes wrapper of oracle/falcon9_fsw.c — the C restatement of the Falcon 9 example's Rust flight software
(examples/falcon9/controller/src/{main,math,profile}.rs), ascent phases.  TEST INFRASTRUCTURE ONLY: imported by
tests/ and by tests/golden/make_falcon9_closed_loop.py, never by anything under elodin_amd/."""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path

import numpy as np

from . import oracle as orc

STATE_FLOATS, CMD_FLOATS = 49, 27          # main.py:44-45 / main.rs:14-15
BEYOND_ASCENT = 1
PEEK = ("phase", "phase_t0", "purge_until", "t_liftoff", "initialized", "last_gps_count", "radar_alt_m")


def _lib():
    L = orc.lib()
    if not getattr(L, "_f9fsw_bound", False):
        dp = C.POINTER(C.c_double)
        L.f9fsw_new.restype = C.c_void_p
        L.f9fsw_new.argtypes = []
        L.f9fsw_free.argtypes = [C.c_void_p]
        L.f9fsw_free.restype = None
        L.f9fsw_load_profile.argtypes = [C.c_void_p, dp, dp, dp, C.c_size_t]
        L.f9fsw_load_profile.restype = C.c_int
        L.f9fsw_load_table.argtypes = [C.c_void_p, dp, dp, dp, dp, C.c_size_t]
        L.f9fsw_load_table.restype = C.c_int
        L.f9fsw_profile_table.argtypes = [C.c_void_p, dp, dp, dp, dp, C.c_size_t]
        L.f9fsw_profile_table.restype = C.c_size_t
        L.f9fsw_step.argtypes = [C.c_void_p, dp, dp]
        L.f9fsw_step.restype = C.c_int
        L.f9fsw_peek.argtypes = [C.c_void_p, dp]
        L.f9fsw_peek.restype = None
        L.f9fsw_ecef_to_geodetic.argtypes = [dp, dp]
        L.f9fsw_ecef_to_geodetic.restype = None
        L.f9fsw_quat_between.argtypes = [dp, dp, dp]
        L.f9fsw_quat_between.restype = None
        L.f9fsw_density.argtypes = [C.c_double]
        L.f9fsw_density.restype = C.c_double
        L._f9fsw_bound = True
    return L


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def read_raw_profile(path) -> tuple:
    """The JSON parse of AscentProfile::load (profile.rs:9-14,41-43): the three columns the FSW reads."""
    raw = json.loads(Path(path).read_text())
    return tuple(np.ascontiguousarray(raw[k], dtype=np.float64) for k in ("time", "velocity", "altitude"))


class Fsw:
    """One flight-software process: `cmd = fsw.step(state)` per exchange."""

    def __init__(self, profile=None, table=None):
        """profile: the raw (time, velocity, altitude_km) columns, resampled like profile.rs does; table: the resampled
        (time, speed, alt_m, vspeed) columns themselves (elodin_amd.models.falcon9.ascent_profile(): bit-equal to the former)."""
        self._L = _lib()
        self._h = C.c_void_p(self._L.f9fsw_new())
        self.beyond_ascent = False
        if table is not None:
            cols = [np.ascontiguousarray(x, dtype=np.float64) for x in table]
            if self._L.f9fsw_load_table(self._h, *[_p(c) for c in cols], cols[0].size) != 0:
                raise ValueError("f9fsw_load_table failed")
        if profile is not None:
            t, v, a = (np.ascontiguousarray(x, dtype=np.float64) for x in profile)
            if self._L.f9fsw_load_profile(self._h, _p(t), _p(v), _p(a), t.size) != 0:
                raise ValueError("f9fsw_load_profile failed")

    def step(self, state) -> np.ndarray:
        s = np.ascontiguousarray(state, dtype=np.float64)
        assert s.size == STATE_FLOATS
        cmd = np.zeros(CMD_FLOATS)
        rc = self._L.f9fsw_step(self._h, _p(s), _p(cmd))
        if rc < 0:
            raise RuntimeError("f9fsw_step failed")
        self.beyond_ascent = self.beyond_ascent or rc == BEYOND_ASCENT
        return cmd

    def peek(self) -> dict:
        out = np.zeros(23)
        self._L.f9fsw_peek(self._h, _p(out))
        d = {k: float(out[i]) for i, k in enumerate(PEEK)}
        d.update(nav_pos=out[7:10].copy(), nav_vel=out[10:13].copy(), nav_att=out[13:17].copy(), up_pad=out[17:20].copy(),
                 track_dir=out[20:23].copy())
        return d

    def profile_table(self):
        n = self._L.f9fsw_profile_table(self._h, None, None, None, None, 0)
        cols = [np.zeros(n) for _ in range(4)]
        self._L.f9fsw_profile_table(self._h, *[_p(c) for c in cols], n)
        return cols      # time, speed, alt_m, vspeed

    def close(self):
        if self._h:
            self._L.f9fsw_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def ecef_to_geodetic(r):
    out = np.zeros(3)
    _lib().f9fsw_ecef_to_geodetic(_p(np.ascontiguousarray(r, dtype=np.float64)), _p(out))
    return out


def quat_between(a, b):
    out = np.zeros(4)
    _lib().f9fsw_quat_between(_p(np.ascontiguousarray(a, dtype=np.float64)), _p(np.ascontiguousarray(b, dtype=np.float64)), _p(out))
    return out


def density(alt_m: float) -> float:
    return float(_lib().f9fsw_density(float(alt_m)))
