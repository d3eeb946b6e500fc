"""ctypes wrapper of oracle/apollo_oracle.c (CPU restatement of the Apollo-lander rollout).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import oracle as orc

N_STATE, N_PARAMS, N_GUIDANCE, N_SCORE, N_RESULT = 16, 17, 8, 4, 12


class _World(C.Structure):
    _fields_ = ([("n", C.c_uint64)] +
                [(k, C.c_void_p) for k in ("world_pos", "world_vel", "world_accel", "force", "inertia", "throttle",
                                           "throttle_cmd", "attitude_setpoint", "propellant", "rcs_propellant", "thrust",
                                           "rcs_torque", "landed", "touchdown_speed", "touchdown_horizontal_speed",
                                           "altitude", "vertical_speed", "horizontal_speed", "pitch", "params",
                                           "guidance", "score", "result", "ref_time", "ref_altitude", "ref_rate",
                                           "ref_pitch", "ref_hspeed", "ref_downrange")] +
                [("n_ref", C.c_uint32), ("guidance_period", C.c_uint32), ("max_ticks", C.c_uint64),
                 ("tick", C.c_uint64), ("simulation_time_step", C.c_double), ("ticks_per_telemetry", C.c_uint32)])


class ApolloOracle:
    """Takes the same initial columns as the product (packed apollo_state etc.), unpacks them into the
    reference's separate components, steps on the CPU, and re-packs for comparison."""

    def __init__(self, cols, ref, *, max_ticks, guidance_period=5, simulation_time_step=0.008333333, ticks_per_telemetry=3):
        c = lambda a: np.array(a, dtype=np.float64, order="C")
        self.world_pos, self.world_vel, self.inertia = c(cols["world_pos"]), c(cols["world_vel"]), c(cols["inertia"])
        n = self.n = self.world_pos.shape[0]
        self.world_accel, self.force = np.zeros((n, 6)), np.zeros((n, 6))
        st = c(cols["apollo_state"])
        self.throttle, self.throttle_cmd = c(st[:, 0]), c(st[:, 1])
        self.attitude_setpoint = c(st[:, 2:6])
        self.propellant, self.rcs_propellant, self.thrust = c(st[:, 6]), c(st[:, 7]), c(st[:, 8])
        self.rcs_torque = c(st[:, 9:12])
        self.landed, self.touchdown_speed, self.touchdown_horizontal_speed = c(st[:, 12]), c(st[:, 13]), c(st[:, 14])
        self.pitch = c(st[:, 15])
        self.altitude = c(self.world_pos[:, 6])
        self.vertical_speed = c(self.world_vel[:, 5])
        self.horizontal_speed = np.hypot(self.world_vel[:, 3], self.world_vel[:, 4])
        self.params, self.guidance = c(cols["apollo_params"]), c(cols["apollo_guidance"])
        self.score, self.result = c(cols["apollo_score"]), c(cols["apollo_result"])
        self._ref = [c(ref[k]) for k in ("time_s", "altitude_m", "descent_rate_mps", "pitch_deg",
                                         "horizontal_speed_mps", "downrange_m")]
        w = self._w = _World()
        w.n = n
        for k in ("world_pos", "world_vel", "world_accel", "force", "inertia", "throttle", "throttle_cmd",
                  "attitude_setpoint", "propellant", "rcs_propellant", "thrust", "rcs_torque", "landed",
                  "touchdown_speed", "touchdown_horizontal_speed", "altitude", "vertical_speed", "horizontal_speed",
                  "pitch", "params", "guidance", "score", "result"):
            setattr(w, k, getattr(self, k).ctypes.data)
        for k, a in zip(("ref_time", "ref_altitude", "ref_rate", "ref_pitch", "ref_hspeed", "ref_downrange"), self._ref):
            setattr(w, k, a.ctypes.data)
        w.n_ref = len(self._ref[0])
        w.guidance_period, w.max_ticks, w.tick = guidance_period, max_ticks, 0
        w.simulation_time_step = simulation_time_step
        w.ticks_per_telemetry = ticks_per_telemetry
        lib = orc.lib()
        lib.apollo_step.argtypes = [C.POINTER(_World), C.c_uint64, C.c_int]
        lib.apollo_step.restype = C.c_int
        self._lib = lib

    @property
    def tick(self):
        return int(self._w.tick)

    def step(self, n_ticks, threads=1):
        self._lib.apollo_step(C.byref(self._w), int(n_ticks), int(threads))
        return self

    @property
    def apollo_state(self):
        return np.concatenate([self.throttle[:, None], self.throttle_cmd[:, None], self.attitude_setpoint,
                               self.propellant[:, None], self.rcs_propellant[:, None], self.thrust[:, None],
                               self.rcs_torque, self.landed[:, None], self.touchdown_speed[:, None],
                               self.touchdown_horizontal_speed[:, None], self.pitch[:, None]], axis=1)
