#!/usr/bin/env python3
"""The reference's Monte-Carlo SITL example (examples/monte-carlo/{sim,main}.py) on the HIP backend as ONE GPU job, written
against `elodin_amd.frontend` — the same decorators, components and `build(params) -> (world, system)` shape as the reference
script; only jax.numpy is spelled el.np.  The drag table (sim.py:49-52) lives in device memory once and is gathered per
vehicle; a campaign's runs are the rows of one executor (elodin_amd.vectorize.Campaign), the script's own post_step — the
saturated PD law of main.py with its external controller switched off — is called per run on the server loop's cadence.
python examples/monte_carlo_sitl.py [runs]"""
import os
import sys
import typing as ty
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import elodin_amd.frontend as el  # noqa: E402

jnp = el.np
SIMULATION_RATE_HZ = 120.0
DEFAULT_MAX_TICKS = 360
DEFAULT_GRID_SIZE = 262_144
GRID_SIZE_ENV = "ELODIN_MONTE_CARLO_GRID_SIZE"
PROBE_ROWS_ENV = "ELODIN_MONTE_CARLO_PROBE_ROWS"

PARAMS = el.monte_carlo.params_spec(                                              # sim.py:18-23
    mass=el.monte_carlo.Param(float, default=1.5, min=0.5, max=5.0),
    target_x=el.monte_carlo.Param(float, default=30.0, min=5.0, max=100.0),
    thrust_gain=el.monte_carlo.Param(float, default=1.0, min=0.1, max=4.0),
    wind=el.monte_carlo.Param(float, default=0.0, min=-5.0, max=5.0),
)

F1 = el.ComponentType(el.PrimitiveType.F64, (1,))
Position = ty.Annotated[el.Array, el.Component("position", F1)]
Velocity = ty.Annotated[el.Array, el.Component("velocity", F1)]
Command = ty.Annotated[el.Array, el.Component("command", F1, metadata={"external_control": "true"})]
Target = ty.Annotated[el.Array, el.Component("target", F1)]
SpecificForce = ty.Annotated[el.Array, el.Component("specific_force", F1)]


def lookup_table(size: int) -> np.ndarray:                                        # sim.py:49-52
    x = np.linspace(0.0, 1.0, size, dtype=np.float64)
    return np.stack([1.0 + 0.05 * np.sin(x * 20.0), 0.1 + x * 0.01], axis=1)


def build(params):                                                                 # sim.py:55-112
    world = el.World()
    grid_size = int(os.environ.get(GRID_SIZE_ENV, str(DEFAULT_GRID_SIZE)))
    probe_rows_count = int(os.environ.get(PROBE_ROWS_ENV, "0"))
    table = el.table(lookup_table(grid_size))            # jnp.asarray(...) of the reference: a constant the kernel gathers from
    probe_base = (np.linspace(0, grid_size - 1, min(probe_rows_count, grid_size), dtype=np.int32).astype(np.float64)
                  if probe_rows_count > 0 else None)
    mass = float(params.get("mass", 1.5))
    target_x = float(params.get("target_x", 30.0))
    wind = float(params.get("wind", 0.0))
    thrust_gain = float(params.get("thrust_gain", 1.0))
    world.spawn([el.C(Position, np.array([0.0])), el.C(Velocity, np.array([wind])), el.C(Command, np.array([0.0])),
                 el.C(Target, np.array([target_x])), el.C(SpecificForce, np.array([0.0]))], name="vehicle")
    dt = 1.0 / SIMULATION_RATE_HZ

    @el.map
    def point_mass(pos: Position, vel: Velocity, command: Command) -> tuple[Position, Velocity, SpecificForce]:
        idx = jnp.clip(jnp.abs(vel[0] * 1000.0).astype(jnp.int32), 0, table.shape[0] - 1)
        drag_coeff = table[idx, 0]
        if probe_base is None:
            probe_sum = 0.0
        else:
            probe_rows = (probe_base + idx) % table.shape[0]
            probe_sum = jnp.sum(table[probe_rows, 0])
        drag = drag_coeff * vel[0] * jnp.abs(vel[0]) * 0.02
        acc = (command[0] * thrust_gain - drag) / mass
        acc = acc + probe_sum * 1e-300
        new_vel = vel + jnp.array([acc * dt])
        new_pos = pos + new_vel * dt
        return new_pos, new_vel, jnp.array([acc])

    return world, point_mass


def post_step(tick: int, ctx) -> None:                                             # main.py:88-106, controller switched off
    position = float(ctx.read_component("vehicle.position")[0])
    velocity = float(ctx.read_component("vehicle.velocity")[0])
    target = float(ctx.read_component("vehicle.target")[0])
    command = max(min((target - position) * 1.2 - velocity * 0.35, 20.0), -20.0)
    ctx.write_component("vehicle.command", np.array([command], dtype=np.float64))
    if tick >= DEFAULT_MAX_TICKS - 1:
        el.monte_carlo.result(final_position=position, target=target, error=abs(target - position))


@el.map
def pd_controller(pos: Position, vel: Velocity, target: Target) -> Command:
    """main.py's control law (:98, external controller switched off) as a system piped behind the plant: the same campaign
    without a host callback per run and tick — `campaign.run(ticks)` is then launches only."""
    return jnp.clip((target - pos) * 1.2 - vel * 0.35, -20.0, 20.0)


SPEC = {"monte_carlo": {"n_samples": 100, "seed": 42, "method": "lhs",                                   # spec.toml
                        "variables": {"mass": {"dist": "uniform", "min": 1.0, "max": 2.0},
                                      "target_x": {"dist": "uniform", "min": 20.0, "max": 40.0},
                                      "thrust_gain": {"dist": "uniform", "min": 0.8, "max": 1.2},
                                      "wind": {"dist": "normal", "mean": 0.0, "std": 0.5}}}}


def build_closed_loop(params):
    """build(params) with the control law on the device: plant | pd_controller."""
    world, plant = build(params)
    return world, plant | pd_controller


def main(runs=100):
    from elodin_amd import monte_carlo as mc
    from elodin_amd import vectorize
    os.environ.setdefault(GRID_SIZE_ENV, "4096")
    spec = {"monte_carlo": dict(SPEC["monte_carlo"], n_samples=int(runs))}
    plan = mc.materialize(spec)
    campaign = vectorize.Campaign(build, plan, PARAMS, simulation_rate=SIMULATION_RATE_HZ)
    campaign.run(DEFAULT_MAX_TICKS, post_step=post_step)
    res = campaign.result_table(["final_position", "target", "error"])
    captured = float(np.mean(res[:, 2] < 8.5))                                     # hooks/score.py: capture radius 8.5 m
    print(f"{len(plan)} runs x {DEFAULT_MAX_TICKS} ticks as one executor: captured {captured:.2f}, mean error {res[:, 2].mean():.3f} m")
    return campaign, captured


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100)
