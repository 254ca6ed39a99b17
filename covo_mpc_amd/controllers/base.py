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
        """struct covo_env_params for `env_params` plus what the kernels need to know about the ENV object: its
        rollover-termination switch (quadrotor.py:486), which reward function env.reward_fn is (quadrotor.py:49-84) and its
        disturbance model (quadrotor.py:35,87-89).  Rebuilt only when a different (frozen, hence never mutated) parameter
        object comes in."""
        if self._c_params[0] is not env_params:
            self._c_params = (env_params, env_model_params_c(self.env, env_params))
        return self._c_params[1]


def env_reward_kind(env) -> str:
    """Which of the kernels' rewards `env.reward_fn` is.  The fused rollout, the Hessian and the env-step kernel evaluate the
    reward themselves (csrc/rollout_pipe.hpp, quad_model.hpp): an env bound to any OTHER function would make the controller
    plan against one objective while env.step pays another -- refuse instead of diverging silently."""
    from ..dynamics import utils
    fn = getattr(env, "reward_fn", None)
    if fn is None or fn is utils.tracking_penyaw_reward_fn:
        return "penyaw"
    if fn is utils.tracking_realworld_reward_fn:
        return "realworld"
    raise NotImplementedError(f"env.reward_fn={getattr(fn, '__name__', fn)!r} is not one of the rewards the kernels evaluate "
                              "(tracking_penyaw_reward_fn, tracking_realworld_reward_fn: quadjax/envs/quadrotor.py:49-84)")


def env_model_params_c(env, env_params, auto_reset: bool = False):
    """struct covo_env_params of `env` + `env_params`; auto_reset: the device env resets a finished episode as
    BaseEnvironment.step does (quadjax/envs/base.py:22-40) with the env's own trajectory generator."""
    from .._lib import DISTURB_KINDS
    disturb = getattr(env, "disturb_type", "none")
    if disturb not in DISTURB_KINDS:
        raise NotImplementedError(f"disturb_type={disturb!r}")
    c = env_params.to_c(rollover_terminate=not getattr(env, "disable_rollover_terminate", True),
                        reward=env_reward_kind(env), disturb_type=disturb,
                        reset_task=getattr(env, "task", None) if auto_reset else None)
    if auto_reset:  # generate_traj is bound to the env's DEFAULT dt (quadrotor.py:50-80), whatever the instance's parameters
        c.reset_dt = float(env.default_params.dt)
    return c
