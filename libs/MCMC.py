"""Sampler driver at the reference's module path (libs/MCMC.py); the
reference's own, unmodified libs/MCMC.py can be used instead."""
from bnpc_amd.mcmc import MCMC, Chain, Chain_steps, Chain_time  # noqa: F401
