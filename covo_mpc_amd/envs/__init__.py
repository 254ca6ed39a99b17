from .quadrotor import Quad3D, eval_env, get_controller, Args, main  # noqa: F401
