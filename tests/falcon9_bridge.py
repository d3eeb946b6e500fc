"""The exchange between the Falcon 9 sim and its flight software, live: what examples/falcon9/main.py's `post_step` does on the server
loop's cadence (main.py:284-347) — read the sensor components, pack the 49-double state packet the controller's UDP socket expects
(controller/src/main.rs reads the same offsets), hand it to the flight software, unpack the 27-double command into the command
components — with oracle/falcon9_fsw.c (the C restatement of the Rust sidecar) behind it instead of the socket.

main.py cannot travel to the GPU box, so the packet layout is restated here as two DATA tables (offsets, not code), and the
script-level constants it packs (guidance parameters, fin bandwidth, caps, the upper-stage mass, the exchange period) come from the
fixture tests/golden/falcon9_main_program.json, where make_falcon9_main_program.py recorded them from main.py's own module.  The
bridge is PINNED on main.py: tests/test_compat_reference_scripts.py runs both side by side over the 1,000-tick window in the build
container and requires identical writes, tick for tick.  TEST INFRASTRUCTURE."""
import numpy as np

STATE_FLOATS, CMD_FLOATS = 49, 27
# packet offset <- component (main.py:293-308)
STATE_LAYOUT = (("imu_accel", 1, 3), ("imu_gyro", 4, 3), ("gps_pos", 7, 3), ("gps_vel", 10, 3), ("gps_count", 13, 1), ("radar_range", 14, 1),
                ("pressure_meas", 16, 4), ("propellant_lox", 20, 1), ("propellant_rp1", 21, 1), ("landed", 43, 1))
# component <- command slice (main.py:338-346)
CMD_LAYOUT = (("engine_cmd", 0, 9), ("valve_cmd", 9, 8), ("attitude_setpoint", 17, 4), ("ctrl_enable", 21, 2), ("fin_cmd", 23, 3))


class Exchange:
    def __init__(self, constants: dict, fsw):
        """constants: the fixture's "exchange" record; fsw: an object with step(state[49]) -> cmd[27] (oracle.falcon9_fsw.Fsw)."""
        self.c, self.fsw = constants, fsw
        self.separated = False
        self.exchanges = 0

    def post_step(self, tick: int, ctx) -> None:
        """The server loop calls this after every tick with the tick's index (impeller2_server.rs:553-678; ticks_per_telemetry = 1)."""
        c = self.c
        if tick % c["period_ticks"] != 0:
            return
        reads = ctx.component_batch_operation(reads=["booster." + name for name, _, _ in STATE_LAYOUT])
        state = np.zeros(STATE_FLOATS)
        state[0] = tick * c["sim_time_step"]
        for name, at, n in STATE_LAYOUT:
            state[at:at + n] = np.asarray(reads["booster." + name], dtype=np.float64).reshape(-1)[:n]
        gv = c["guidance_values"]
        state[22:22 + len(gv)] = gv
        state[46], state[47], state[48] = c["fin_wn"], c["divert_speed_cap"], c["steer_tilt_cap"]
        cmd = np.asarray(self.fsw.step(state), dtype=np.float64)
        self.exchanges += 1
        phase = float(cmd[26])
        if not self.separated and phase >= 5.0:
            self.separated = True
        writes = {"booster." + name: cmd[at:at + n] for name, at, n in CMD_LAYOUT}
        writes["booster.fsw_phase"] = np.array([phase])
        writes["booster.upper_mass"] = np.array([0.0 if self.separated else c["upper_kg"]])
        ctx.component_batch_operation(writes=writes)
