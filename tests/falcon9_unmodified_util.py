"""examples/falcon9/sim.py — the plant of BASELINE configs[4]'s example — imported UNMODIFIED under elodin_amd.compat and built
the way the reference's own tests build it (test_propulsion.py:111-124: `build_powered(..., extra_systems=script)` with a
tick-driven open-loop command script written as an @el.system over el.SimulationTick).  Build container only.

The windows, their spawn states, campaign knobs and command scripts are those of tests/golden/falcon9_plant.json, i.e. of the
trajectories the reference's own system functions flew under tests/golden/refshim.py (tests/golden/make_falcon9_fixtures.py):
the same fixtures that pin this repo's own model of the vehicle (tests/test_falcon9_plant_reference.py) pin what the front end
makes of the reference's script.  TEST INFRASTRUCTURE."""
import importlib.util
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/examples/falcon9")


def load_sim():
    """sim.py (and the modules it imports from its directory) under an installed compat layer."""
    if "ref_falcon9_sim" in sys.modules:
        return sys.modules["ref_falcon9_sim"]
    sys.path.insert(0, str(REF))
    spec = importlib.util.spec_from_file_location("ref_falcon9_sim", REF / "sim.py")
    sim = importlib.util.module_from_spec(spec)
    sys.modules["ref_falcon9_sim"] = sim
    spec.loader.exec_module(sim)
    return sim


def build(case):
    """-> (plan of World.build(_dry=True), traced program, initial arrays) of window `case` ('pad' / 'maxq' / 'coast')."""
    import elodin as el
    import jax.numpy as jnp
    from tests import falcon9_plant_util as pu, falcon9_script as fs
    sim = load_sim()
    c, init = fs.CASES[case], pu.PLANT[case]["init"]
    script = fs.make_script(case, pu.PLANT[case]["base_attitude"])
    outs = (sim.EngineCmd, sim.ValveCmd, sim.AttitudeSetpoint, sim.CtrlEnable, sim.FinCmd, sim.FswPhase)

    @el.system
    def commands(tick: el.Query[el.SimulationTick], q: el.Query[sim.EngineCmd]) -> el.Query[outs]:
        cmd = script(jnp, tick[0] * 0.001)
        return q.map(outs, lambda _c: (cmd["engine_cmd"], cmd["valve_cmd"], el.Quaternion(cmd["attitude_setpoint"]), cmd["ctrl_enable"],
                                       cmd["fin_cmd"], cmd["fsw_phase"]))
    x0, v0 = np.asarray(init["world_pos"]), np.asarray(init["world_vel"])
    world, system = sim.build_powered(
        jnp.asarray(x0[4:]), jnp.asarray(v0[3:]), init_attitude=el.Quaternion(jnp.asarray(x0[:4])), lox_kg=init["propellant_lox"][0],
        rp1_kg=init["propellant_rp1"][0], upper_kg=init["upper_mass"][0], thrust_scale=c["thrust_scale"], isp_scale=c["isp_scale"],
        ca_scale=c["ca_scale"], cn_scale=c["cn_scale"], wind_north_mps=c["wind_ned"][0], wind_east_mps=c["wind_ned"][1],
        wind_down_mps=c["wind_ned"][2], extra_systems=commands)
    plan = world.build(system, simulation_rate=1.0 / sim.SIM_TIME_STEP, _dry=True)
    tp = plan["effectors"].trace()
    body, cols = plan["body"], plan["columns"]
    arrays = {k: np.array(body[k], dtype=np.float64).copy() for k in ("world_pos", "world_vel", "world_accel", "inertia")}
    arrays.update({n: np.array(cols[n], dtype=np.float64).reshape(1, -1).copy() for n, _ in tp.columns})
    for name, v in init.items():          # the fixture's spawn state: the aloft / engines-running windows preset more than build_powered's arguments
        if name in arrays:
            arrays[name][:] = np.asarray(v, dtype=np.float64).reshape(1, -1)
    return plan, tp, arrays
