"""Alias: the reference's ``python_dim`` is the pure-Python twin of ``cython_dim`` (identical arithmetic,
LB_D2Q9/dimensionless/python_dim.py:22-666); here both names give the GPU-backed Cython-path classes."""
from .cython_dim import *          # noqa: F401,F403
from .cython_dim import Pipe_Flow, Pipe_Flow_Cylinder  # noqa: F401
