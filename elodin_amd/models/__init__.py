"""Rollout models: example sims of the reference whose systems around six_dof are fused into one kernel."""
