"""Auto-resetting step/reset wrapper: quadjax/envs/base.py:15-50 (gymnax-style, host plumbing)."""
from .. import random as crandom


class BaseEnvironment:
    @property
    def default_params(self):
        raise NotImplementedError

    def step(self, key, state, action, params=None):
        """base.py:15-40: step_env, and select a fresh reset state/info/obs when done."""
        if params is None:
            params = self.default_params
        key, key_reset = crandom.split(key)
        obs_st, state_st, reward, done, info = self.step_env(key, state, action, params)
        if done:  # lax.select(done, reset, stepped) leaf-wise (base.py:33-39)
            obs_re, info_re, state_re = self.reset_env(key_reset, params)
            return obs_re, state_re, reward, done, info_re
        return obs_st, state_st, reward, done, info

    def reset(self, key, params=None):
        """base.py:42-50."""
        if params is None:
            params = self.default_params
        return self.reset_env(key, params)
