"""quadjax/dynamics/__init__.py:1-4 re-exports."""
from .dataclass import Action3D, DeviceState, EnvParams3D, EnvState3D, as_device_state  # noqa: F401
from . import geom, utils  # noqa: F401
from .free import get_quadrotor_1st_order_dyn  # noqa: F401
