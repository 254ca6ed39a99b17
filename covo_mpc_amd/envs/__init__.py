from .quadrotor import Quad3D, eval_env, eval_env_device, eval_env_batched, DeviceEpisode, BatchedDeviceEpisode, get_controller, Args, main  # noqa: F401
