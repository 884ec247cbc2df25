"""MI355X-native rigid-body-dynamics code generator (drop-in for A2R-Lab/GRiDCodeGenerator's hot path)."""
from .GRiDCodeGenerator import GRiDCodeGenerator  # noqa: F401
from .robot import RobotModel, load_urdf  # noqa: F401
