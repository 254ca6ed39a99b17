"""Controller protocol: quadjax/controllers/base.py:5-19 (same names, arguments and defaults)."""


class BaseController:
    def __init__(self, env, control_params) -> None:
        self.env = env
        self.init_control_params = control_params

    def update_params(self, env_params, control_params):
        return control_params

    def _params_c(self, env_params):
        """struct covo_env_params of `env_params` (+ the env's rollover-termination switch, quadrotor.py:486), cached on
        object identity (frozen dataclass: never mutated)."""
        cache = getattr(self, "_params_c_cache", None)
        if cache is None or cache[0] is not env_params:
            roll = not getattr(self.env, "disable_rollover_terminate", True)
            cache = (env_params, env_params.to_c(rollover_terminate=roll))
            self._params_c_cache = cache
        return cache[1]

    def reset(self, env_state=None, env_params=None, control_params=None, key=None):
        return self.init_control_params

    def __call__(self, obs, state, env_params, rng_act, control_params, env_info=None):
        raise NotImplementedError
