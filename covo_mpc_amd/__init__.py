"""covo_mpc_amd -- MI355X (gfx950) native sampling-MPC inner loop behind quadjax's controller API.

Drop-in for ONE path of LeCAR-Lab/CoVO-MPC: `quadjax.controllers.{MPPIController, CoVOController}
.__call__` (one MPC control step) and the `quadjax.envs.quadrotor` plumbing that feeds it.  The
compute path is hand-written HIP behind a C ABI (include/covo_hip.h, csrc/libcovo_hip.so); PyTorch
only owns device memory, streams and torch.distributed.  There is NO CPU fallback: importing the
controllers without the built library, or calling them without a GPU, raises.
"""
from . import controllers, dynamics, envs, random  # noqa: F401
from ._lib import lib_path, load_library  # noqa: F401

__version__ = "0.1.0"


def get_package_path():
    """quadjax/__init__.py:7-8."""
    import os
    return os.path.dirname(os.path.abspath(__file__))
