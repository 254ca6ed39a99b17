"""quadjax/controllers/__init__.py:1-7 re-exports."""
from .base import BaseController  # noqa: F401
from .random import RandomController  # noqa: F401
from .pid import PIDController, PIDParams  # noqa: F401
from .mppi import MPPIController, MPPIParams  # noqa: F401
from .covo import CoVOController, CoVOParams  # noqa: F401
from .batched import BatchedCoVOController  # noqa: F401
