from ._inverse_dynamics import *
from ._direct_minv import *
from ._forward_dynamics import *
from ._inverse_dynamics_gradient import *
from ._forward_dynamics_gradient import *
