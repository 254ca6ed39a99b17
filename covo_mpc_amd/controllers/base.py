"""The controller protocol every quadjax caller relies on (quadjax/controllers/base.py:5-19): construct with
(env, control_params), `reset(...)` -> control_params, `__call__(obs, state, env_params, rng_act, control_params, info)`
-> (action, control_params', info').  Subclasses here add one thing the reference has no need for: the C struct of the
env parameters their kernels read."""


class BaseController:
    def __init__(self, env, control_params) -> None:
        self.env, self.init_control_params = env, control_params
        self._c_params = (None, None)  # (EnvParams3D it was built from, struct covo_env_params)

    def __call__(self, obs, state, env_params, rng_act, control_params, env_info=None):
        raise NotImplementedError(f"{type(self).__name__} does not define a control law")

    def reset(self, env_state=None, env_params=None, control_params=None, key=None):
        """Episode start: stateless controllers hand back the parameters they were built with."""
        return self.init_control_params

    def update_params(self, env_params, control_params):
        """Hook of the reference's RL-side wrappers; the sampling controllers have nothing to refresh."""
        return control_params

    def _params_c(self, env_params):
        """struct covo_env_params for `env_params` plus the env's rollover-termination switch (quadrotor.py:486), rebuilt
        only when a different (frozen, hence never mutated) parameter object comes in."""
        if self._c_params[0] is not env_params:
            roll_on = not getattr(self.env, "disable_rollover_terminate", True)
            self._c_params = (env_params, env_params.to_c(rollover_terminate=roll_on))
        return self._c_params[1]
