"""The commit path's hand-off: csrc/telemetry_sink.cpp behind sixdof_sink_*, elodin_amd/telemetry.py and the server
loop of elodin_amd/compat.py around it — host code, no GPU.  The semantics are the reference's: PairId = the component id of
"entity.component" (impeller2 types.rs:54-59), push refuses time travel and appends equal timestamps (time_series.rs:201-230),
reads by timestamp are floor / clamp-to-latest (elodin.pyi:63-88), commit_world_head pushes every named row at the batch's
END timestamp and skips external controls, copy_db_to_world brings the latest samples back (impeller2_server.rs:320-438)."""
import numpy as np
import pytest

from elodin_amd import _lib as L
from elodin_amd import telemetry


def test_pair_ids_and_series_semantics():
    assert telemetry.Sink.pair_id("a.world_pos") == L.component_id("a.world_pos") != L.component_id("world_pos")
    assert telemetry.Sink.pair_id("a.world_pos") < (1 << 63)
    s = telemetry.Sink()
    s.register("drone.world_pos", 7)
    s.register("drone.world_pos", 7)                                # idempotent
    with pytest.raises(ValueError):
        s.register("drone.world_pos", 6)                            # same pair, another element size
    with pytest.raises(RuntimeError, match="does not exist"):
        s.push("drone.gyro", np.zeros(3), 0)
    with pytest.raises(RuntimeError, match="no samples"):
        s.latest("drone.world_pos")
    for ts, v in ((100, 0.0), (200, 1.0), (200, 2.0), (350, 3.0)):
        s.push("drone.world_pos", np.full(7, v), ts)
    assert s.sample_count("drone.world_pos") == 4
    assert s.latest("drone.world_pos")[0] == 350 and s.latest("drone.world_pos")[1][0] == 3.0
    assert s.at("drone.world_pos", 150) == (100, pytest.approx(np.zeros(7)))
    assert s.at("drone.world_pos", 200)[1][0] == 2.0                # the LAST of equal timestamps
    assert s.at("drone.world_pos", 10 ** 12)[0] == 350              # past the last write: the latest
    with pytest.raises(RuntimeError, match="at or before"):
        s.at("drone.world_pos", 99)
    with pytest.raises(telemetry.TimeTravel):
        s.push("drone.world_pos", np.zeros(7), 349)
    with pytest.raises(ValueError):
        s.push("drone.world_pos", np.zeros(6), 400)
    ts, data = s.series("drone.world_pos")
    assert ts.tolist() == [100, 200, 200, 350] and data[:, 0].tolist() == [0.0, 1.0, 2.0, 3.0] and data.shape == (4, 7)
    s.truncate()
    assert s.sample_count("drone.world_pos") == 0
    s.push("drone.world_pos", np.ones(7), 5)                        # the schema survived, time starts over
    assert s.latest("drone.world_pos")[0] == 5


class _FakeExec:
    """What the sink and the server loop need of an executor: host columns, their entity ids, run / tick, upload."""

    def __init__(self, cols, ids):
        self.cols, self.ids, self.tick, self.uploads, self.seen_command = cols, ids, 0, 0, []
        outer = self

        class Hip:
            _aux = cols
            def upload(self_inner): outer.uploads += 1
        self._hip = Hip()

    def column_array(self, name): return self.cols[name]
    def _main_column_array(self, name): return self.cols[name]
    def column_ids(self, name): return self.ids[name]

    def run(self, n):
        for _ in range(n):
            self.tick += 1
            self.seen_command.append(float(self.cols["command"][0, 0]))
            self.cols["position"][:, 0] += self.cols["command"][:, 0]          # the "plant"
            self.cols["world_pos"][:, 4] = self.cols["position"][:, 0]


def test_server_loop_commits_at_batch_end_and_brings_callback_writes_back(monkeypatch):
    from elodin_amd import api, compat
    from elodin_amd import frontend as fe
    world = api.World()
    a = world.spawn([api.C("position", [0.0]), api.C("command", [0.0])], name="vehicle")
    b = world.spawn([api.C("position", [10.0]), api.C("command", [0.0])])            # an entity without a name: no pairs
    cols = {"position": np.array([[0.0], [10.0]]), "command": np.zeros((2, 1)), "world_pos": np.zeros((2, 7)),
            "world_vel": np.zeros((2, 6)), "world_accel": np.zeros((2, 6)), "force": np.zeros((2, 6)), "inertia": np.ones((2, 7))}
    ids = {k: np.array([int(a), int(b)], dtype=np.uint64) for k in cols}
    ex = _FakeExec(cols, ids)
    monkeypatch.setitem(fe.COMPONENT_METADATA, "command", {"external_control": "true"})
    calls = []

    def pre_step(tick, ctx):
        calls.append(("pre", tick, ctx.tick, ctx.timestamp))

    def post_step(tick, ctx):
        calls.append(("post", tick, ctx.tick, ctx.timestamp))
        pos = float(ctx.read_component("vehicle.position")[0])
        ctx.write_component("vehicle.command", np.array([1.0 if pos < 3.0 else 0.0]))
        if tick == 5:
            assert float(ctx.read_component("vehicle.position", timestamp=ctx.timestamp - 1)[0]) < pos       # floor: the batch before
            with pytest.raises(RuntimeError):
                ctx.read_component("nobody.position")
    t0 = 1_000_000
    compat.run_stepwise(ex, world, simulation_rate=100.0, telemetry_rate=50.0, max_ticks=9, pre_step=pre_step, post_step=post_step,
                        start_timestamp=t0)
    # batches of 2 ticks (100 Hz / 50 Hz), the last one cut at max_ticks; post_step sees the batch's LAST tick and its timestamp
    assert [c[:3] for c in calls if c[0] == "pre"] == [("pre", k, k) for k in (0, 2, 4, 6, 8)]
    assert [c[1:] for c in calls if c[0] == "post"] == [(k, k, t0 + k * 10_000) for k in (1, 3, 5, 7, 8)]
    sink = ex.compat_sink
    ts, pos = sink.series("vehicle.position")
    assert ts.tolist() == [t0] + [t0 + k * 10_000 for k in (1, 3, 5, 7, 8)]           # the spawned state, then one commit per batch
    assert pos[:, 0].tolist() == [0.0, 0.0, 2.0, 4.0, 4.0, 4.0]
    # the command a callback wrote reaches the plant in the NEXT batch (copy_db_to_world), and the simulation never commits an
    # external control over it: its series holds the spawn value and the callbacks' writes only
    assert ex.seen_command == [0.0, 0.0, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0]
    cts, cmd = sink.series("vehicle.command")
    assert cts.tolist() == [t0] + [t0 + k * 10_000 for k in (1, 3, 5, 7, 8)] and cmd[:, 0].tolist() == [0.0, 1.0, 1.0, 0.0, 0.0, 0.0]
    assert ex.uploads == 2                                                             # only when a byte really changed (0 -> 1, 1 -> 0)
    assert sink.sample_count("vehicle.world_pos") == 6 and sink.latest("vehicle.world_pos")[1][4] == 4.0
    with pytest.raises(RuntimeError):
        sink.latest(f"{int(b)}.position")                                              # the unnamed entity has no pairs


def test_copy_to_world_reaches_the_real_host_array_or_refuses():
    """the column an executor hands out may be a fancy-indexed COPY (stand-in rows of plain entities filtered out of a
    Body column) — a sample pushed into the sink must land in the array the executor uploads from, and a column that cannot be
    written back (float32 here) must refuse instead of dropping the write."""
    from elodin_amd import api
    world = api.World()
    world.spawn([api.C("gain", [1.0])], name="a")
    world.spawn([api.C("gain", [2.0])], name="b")
    real = np.zeros((3, 7))                                                       # row 1 is a stand-in the world does not know
    real[:, 4] = [1.0, -1.0, 2.0]
    gain32 = np.array([[1.0], [2.0]], dtype=np.float32)
    ex = _FakeExec({"world_pos": real, "gain": gain32}, {"world_pos": np.array([1, 2], dtype=np.uint64), "gain": np.array([1, 2], dtype=np.uint64)})
    ex._body_rows = np.array([0, 2])
    ex._hip.world_pos = real
    ex._main_column_array = lambda name: real[ex._body_rows] if name == "world_pos" else ex.cols[name]
    ex.column_array = ex._main_column_array
    sink = telemetry.Sink().attach(ex, world, 0)
    assert sink.copy_to_world() is False and ex.uploads == 0                       # nothing newer than what was committed
    sink.push("b.world_pos", np.array([0, 0, 0, 1, 7.5, 0, 0.0]), 10)
    assert sink.copy_to_world() is True and ex.uploads == 1
    assert real[2, 4] == 7.5 and real[1, 4] == -1.0 and real[0, 4] == 1.0        # scattered through the row map, stand-in untouched
    sink.push("a.gain", np.array([5.0]), 20)
    with pytest.raises(NotImplementedError, match="cannot be written back"):
        sink.copy_to_world()
