"""Alias so that ``from LB_D2Q9.dimensionless import opencl_dim as lb`` keeps working unchanged:
the classes are the HIP implementations of ``hip_dim``."""
from .hip_dim import *          # noqa: F401,F403
from .hip_dim import (Pipe_Flow, Pipe_Flow_Cylinder, Pipe_Flow_PeriodicBC_VelocityInlet,  # noqa: F401
                      get_divisible_global)
