"""Obstacle-mask ingestion: disc generator, minimal TIFF reader, nearest-neighbour rescale.

Replaces the third-party calls the reference uses around its obstacle path:
  * ``skimage.draw.circle(r, c, radius)``     opencl_dim.py:474 / cython_dim.pyx:430
  * ``tifffile.imread`` + ``skimage.transform.resize``   docs/cs205_movie.ipynb:220, 264
scikit-image / tifffile are un-pinned in the reference (setup.py:29) and absent from this image,
so these are restatements of their published behaviour, not bit-level ports.
"""
import struct

import numpy as np


def disc_pixels(xc, yc, radius, shape=None):
    """Index arrays (xs, ys) of the pixels with (x-xc)^2 + (y-yc)^2 < radius^2 - the strict
    inequality scikit-image's ``circle``/``disk`` uses - clipped to ``shape`` when given."""
    x_lo, x_hi = int(np.floor(xc - radius)), int(np.ceil(xc + radius)) + 1
    y_lo, y_hi = int(np.floor(yc - radius)), int(np.ceil(yc + radius)) + 1
    if shape is not None:
        x_lo, y_lo = max(x_lo, 0), max(y_lo, 0)
        x_hi, y_hi = min(x_hi, shape[0]), min(y_hi, shape[1])
    xs, ys = np.mgrid[x_lo:x_hi, y_lo:y_hi]
    keep = ((xs - xc) ** 2 + (ys - yc) ** 2) < float(radius) ** 2
    return xs[keep], ys[keep]


def read_tiff_u8(path):
    """Read a baseline, uncompressed, 8-bit single-channel TIFF (what ImageJ wrote for
    docs/CS205_obstacle_*.tif: big-endian, BlackIsZero, one strip) -> uint8 (rows, cols)."""
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:2] == b"II":
        bo = "<"
    elif data[:2] == b"MM":
        bo = ">"
    else:
        raise ValueError("%s: not a TIFF file" % path)
    magic, ifd = struct.unpack(bo + "HI", data[2:8])
    if magic != 42:
        raise ValueError("%s: bad TIFF magic %d" % (path, magic))
    (n_entries,) = struct.unpack(bo + "H", data[ifd:ifd + 2])
    type_size = {1: 1, 2: 1, 3: 2, 4: 4}
    tags = {}
    for i in range(n_entries):
        off = ifd + 2 + 12 * i
        tag, typ, count = struct.unpack(bo + "HHI", data[off:off + 8])
        if typ not in type_size:
            continue
        nbytes = type_size[typ] * count
        start = off + 8 if nbytes <= 4 else struct.unpack(bo + "I", data[off + 8:off + 12])[0]
        fmt = {1: "B", 2: "B", 3: "H", 4: "I"}[typ]
        tags[tag] = struct.unpack(bo + fmt * count, data[start:start + nbytes])
    width, height = tags[256][0], tags[257][0]
    if tags.get(258, (1,))[0] != 8 or tags.get(277, (1,))[0] != 1:
        raise ValueError("%s: only 8-bit single-channel TIFFs are supported" % path)
    if tags.get(259, (1,))[0] != 1:
        raise ValueError("%s: compressed TIFFs are not supported" % path)
    offsets = tags[273]
    counts = tags.get(279, (width * height,))
    raw = b"".join(data[o:o + c] for o, c in zip(offsets, counts))
    img = np.frombuffer(raw[:width * height], dtype=np.uint8).reshape(height, width)
    if tags.get(262, (1,))[0] == 0:      # WhiteIsZero
        img = 255 - img
    return img.copy()


def resize_nearest(a, shape):
    """Nearest-neighbour rescale of a 2-d array to ``shape`` (pixel-centre convention)."""
    a = np.asarray(a)
    ix = np.minimum(((np.arange(shape[0]) + 0.5) * a.shape[0] / shape[0]).astype(np.int64), a.shape[0] - 1)
    iy = np.minimum(((np.arange(shape[1]) + 0.5) * a.shape[1] / shape[1]).astype(np.int64), a.shape[1] - 1)
    return a[np.ix_(ix, iy)]


def obstacle_mask_from_tiff(path, shape):
    """The notebook recipe: image -> bool -> transpose to (x, y) -> rescale to the grid ->
    int32 F-ordered mask ready for ``obstacle_mask_host`` (docs/cs205_movie.ipynb:220, 264, 292)."""
    img = read_tiff_u8(path).astype(bool).T
    return np.asfortranarray(resize_nearest(img, shape).astype(np.int32))
