"""Physical-units front ends (``hip_dim``; ``opencl_dim`` is an alias for drop-in imports)."""
