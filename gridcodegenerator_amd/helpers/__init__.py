from ._code_generation_helpers import *
from ._spatial_algebra_helpers import *
from ._topology_helpers import *
