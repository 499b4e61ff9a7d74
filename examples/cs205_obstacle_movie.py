#!/usr/bin/env python3
"""docs/cs205_movie.ipynb through the drop-in classes: a pipe flow past a cylinder whose obstacle is then replaced by an
image (the reference reads docs/CS205_obstacle_4.tif with tifffile and rescales it with skimage; here LB_D2Q9.masks does
both), re-initialised the way the notebook does it (`sim.obstacle_mask_host = ...; init_hydro(); update_feq();
init_pop()`), and rendered frame by frame (dimensionless horizontal velocity, colour range +-3, as cell 23).

    python examples/cs205_obstacle_movie.py [out_dir] [frames] [steps_per_frame]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "2d-lb_amd"))


def build(N=25, verbose=False):
    from LB_D2Q9.dimensionless import opencl_dim as lb
    from LB_D2Q9.masks import obstacle_mask_from_tiff
    D = 1.
    sim = lb.Pipe_Flow_Cylinder(diameter=D, rho=1., viscosity=1., pressure_grad=-100., pipe_length=3 * D, N=N,
                                time_prefactor=1., cylinder_center=[3 * D / 4, D / 2], cylinder_radius=D / 10,
                                verbose=verbose)                                              # cell 7
    tif = os.path.join(ROOT, "tests", "golden", "CS205_obstacle_4.tif")
    mask = obstacle_mask_from_tiff(tif, (sim.nx, sim.ny))                                     # cells 11-14
    mask[0, :] = mask[-1, :] = 0
    sim.obstacle_mask_host = np.asfortranarray(mask.astype(np.int32))                         # cell 16
    sim.init_hydro()
    sim.update_feq()
    sim.init_pop()
    return sim


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "cs205_frames"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    from LB_D2Q9.frames import Frame_Dumper
    sim = build()
    os.makedirs(out, exist_ok=True)
    scale = sim.delta_x / sim.delta_t                         # get_nondim_fields: u * dx/dt
    d = Frame_Dumper(sim, sim.u, num_steps_per_draw=per, scaling_factor=scale, max_magnitude=3., render_folder=out)
    d.run(frames)
    print("%d frames of a %d x %d lattice (%d solid cells) in %s" % (len(d.frames_written), sim.nx, sim.ny,
                                                                    int(np.asarray(sim.obstacle_mask_host).sum()), out))


if __name__ == "__main__":
    main()
