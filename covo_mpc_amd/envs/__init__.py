from .quadrotor import Quad3D, eval_env, eval_env_device, DeviceEpisode, get_controller, Args, main  # noqa: F401
