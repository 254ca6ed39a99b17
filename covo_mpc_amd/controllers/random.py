"""Gaussian random baseline: quadjax/controllers/random.py:15-16 (host numpy)."""
from .. import random as crandom
from .base import BaseController


class RandomController(BaseController):
    def __call__(self, obs, state, env_params, rng_act, control_params, info=None):
        return crandom.normal(rng_act, (self.env.action_dim,)), control_params, None
