"""Headless frame dumper: the reference's main consumer of ``run()`` without OpenGL.

``Field_Visualizer_Canvas.on_draw`` (LB_D2Q9/field_visualizer.py:146-161) does
``sim.run(num_steps_per_draw)`` -> ``field.get()`` -> texture -> optional PNG.  ``Frame_Dumper`` keeps
that loop and its constructor vocabulary (``sim``, ``sim_field_to_draw``, ``num_steps_per_draw``,
``scaling_factor``, ``max_magnitude``, ``save_images``, ``render_folder``, ``run_func``) and writes
``.png`` (own zlib encoder, no matplotlib/vispy needed) or ``.npy`` frames.
"""
import os
import struct
import zlib

import numpy as np


def vorticity(u, v):
    """Central-difference curl of the lattice velocity, (nx, ny) arrays, one-sided at the edges."""
    return np.gradient(np.asarray(v, np.float64), axis=0) - np.gradient(np.asarray(u, np.float64), axis=1)


def diverging_rgb(a, max_magnitude):
    """Blue-white-red map of a in [-max_magnitude, max_magnitude] -> uint8 (..., 3)."""
    t = np.clip(np.asarray(a, np.float64) / float(max_magnitude), -1., 1.)
    pos, neg = np.clip(t, 0, 1), np.clip(-t, 0, 1)
    rgb = np.stack([1. - neg, 1. - np.maximum(pos, neg), 1. - pos], axis=-1)
    return (255. * rgb + 0.5).astype(np.uint8)


def write_png(path, rgb):
    """Minimal 8-bit RGB PNG writer; rgb is (rows, cols, 3) uint8."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    rows, cols, _ = rgb.shape
    raw = np.concatenate([np.zeros((rows, 1), np.uint8), rgb.reshape(rows, cols * 3)], axis=1).tobytes()

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n")
        fh.write(chunk(b"IHDR", struct.pack(">IIBBBBB", cols, rows, 8, 2, 0, 0, 0)))
        fh.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        fh.write(chunk(b"IEND", b""))


class Frame_Dumper(object):
    def __init__(self, sim, sim_field_to_draw, num_steps_per_draw=1, scaling_factor=1.0, max_magnitude=1.0,
                 save_images=True, render_folder='./', run_func=None, image_format='png'):
        """
        :param sim: anything with ``run(n)`` (Pipe_Flow, Pipe_Flow_Cylinder, Simulation, DistributedSlab...).
        :param sim_field_to_draw: an object with ``.get()`` (e.g. ``sim.u`` of the drop-in classes, as in the
               reference), a callable returning an (nx, ny) array, or one of 'rho', 'u', 'v', 'speed',
               'vorticity' (read through ``sim.get_fields()``).
        :param image_format: 'png' (diverging colour map of scaling_factor*field clipped to
               +-max_magnitude, drawn with y upwards like the reference's quad) or 'npy' (raw float32).
        """
        self.sim = sim
        self.sim_field_to_draw = sim_field_to_draw
        self.num_steps_per_draw = num_steps_per_draw
        self.scaling_factor = scaling_factor
        self.max_magnitude = max_magnitude
        self.save_images = save_images
        self.render_folder = render_folder
        self.run_func = run_func
        self.image_format = image_format
        self.total_num_steps = 0
        self.frames_written = []
        self.I = self._fetch()

    def _fetch(self):
        f = self.sim_field_to_draw
        if hasattr(f, "get"):
            a = f.get()
        elif callable(f):
            a = f()
        else:
            g = self.sim.get_fields() if f in ("rho", "u", "v") else None
            if g is None:
                g = self.sim.get_fields()
            a = {"rho": lambda: g["rho"], "u": lambda: g["u"], "v": lambda: g["v"],
                 "speed": lambda: np.hypot(g["u"], g["v"]),
                 "vorticity": lambda: vorticity(g["u"], g["v"])}[f]()
        return np.asarray(a, np.float32)

    def on_draw(self, event=None):
        """One pass of the reference's draw handler: advance, read back, optionally write a frame."""
        if self.run_func is None:
            self.sim.run(self.num_steps_per_draw)
        else:
            self.run_func(self.num_steps_per_draw)
        self.total_num_steps += self.num_steps_per_draw
        self.I = self._fetch()
        if self.save_images:
            os.makedirs(self.render_folder, exist_ok=True)
            stem = os.path.join(self.render_folder, "%08d" % self.total_num_steps)
            if self.image_format == "npy":
                path = stem + ".npy"
                np.save(path, self.I)
            else:
                path = stem + ".png"
                write_png(path, diverging_rgb(self.scaling_factor * self.I.T[::-1], self.max_magnitude))
            self.frames_written.append(path)
        return self.I

    def run(self, num_frames):
        for _ in range(num_frames):
            self.on_draw()
        return self.frames_written
