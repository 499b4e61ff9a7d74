"""MI355X-native D2Q9 lattice-Boltzmann engine behind the API of the reference package
``LB_D2Q9`` (latticeboltzmann/2d-lb).  ``LB_D2Q9.dimensionless.hip_dim`` (alias
``opencl_dim``) offers ``Pipe_Flow`` / ``Pipe_Flow_Cylinder``; ``LB_D2Q9.simulation.Simulation``
is the generic lattice with ``step()``.  All compute runs in hand-written HIP kernels
(``liblbhip.so``); there is no CPU fallback."""
