"""Host-side logic of the drop-in classes, without a GPU: parameter derivation against the constants
the reference's notebooks print, mask ingestion, slab arithmetic."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden


def dry(cls):
    """The real constructor code with the device hooks stubbed out (no GPU in this test)."""
    class Dry(cls):
        def init_hip(self):
            self._sim = None

        def init_hydro(self):
            self.inlet_rho, self.outlet_rho = self._boundary_densities()

        def update_feq(self):
            pass

        def init_pop(self, amplitude=.001):
            pass
    return Dry


def test_pipe_flow_constants_match_reference_prints():
    """opencl Pipe_Flow(D=1.5, rho=10, nu=5, gradP=-100, len=3, N=10/50/200):
    docs/opencl_dimensionless_verification.ipynb:94-98, 115, 168-172, 189, 221-225, 242."""
    from LB_D2Q9.dimensionless import opencl_dim as lb
    expect = {10: dict(ulb=0.1, glob=(32, 32), rin=1.063),
              50: dict(ulb=0.02, glob=(128, 64), rin=1.002424),
              200: dict(ulb=0.005, glob=(416, 224), rin=1.000150375)}
    for N, e in expect.items():
        s = dry(lb.Pipe_Flow)(diameter=1.5, rho=10., viscosity=5., pressure_grad=-100., pipe_length=3., N=N,
                              verbose=False)
        assert s.T == pytest.approx(0.387298334621, rel=1e-11)
        assert s.W == pytest.approx(1.16189500386, rel=1e-11)
        assert s.omega == pytest.approx(0.324465802203, rel=1e-11)
        assert s.ulb == pytest.approx(e["ulb"]) and s.two_d_global_size == e["glob"]
        assert s.inlet_rho == pytest.approx(e["rin"], rel=1e-12) and s.outlet_rho == 1.
        assert (s.nx, s.ny) == (2 * N + 1, N + 1) and (s.lx, s.ly) == (2 * N, N)
        assert s.three_d_global_size == e["glob"] + (9,)


def test_pipe_flow_constants_equal_oracle_restatement(oracle):
    from LB_D2Q9.dimensionless import hip_dim as lb
    kw = dict(diameter=0.7, rho=2.5, viscosity=0.3, pressure_grad=-4., pipe_length=2.2, N=37, time_prefactor=1.3)
    s = dry(lb.Pipe_Flow)(verbose=False, **kw)
    p = oracle.opencl_pipe_parameters(**kw)
    for k in ("L", "T", "W", "delta_x", "delta_t", "ulb", "lb_viscosity", "omega", "lx", "ly", "nx", "ny",
              "inlet_rho", "outlet_rho"):
        assert getattr(s, k) == p[k], k
    c = dry(lb.Pipe_Flow_Cylinder)(cylinder_center=[.6, .35], cylinder_radius=.1, verbose=False, **kw)
    q = oracle.opencl_pipe_parameters(cylinder_radius=.1, **kw)
    for k in ("L", "T", "W", "omega", "nx", "ny", "inlet_rho"):
        assert getattr(c, k) == q[k], k
    want = oracle.disc_mask(q["nx"], q["ny"], 37 * .6 / .1, 37 * .35 / .1, 37)
    assert c.obstacle_mask_host.dtype == np.int32 and c.obstacle_mask_host.flags.f_contiguous
    assert np.array_equal(c.obstacle_mask_host.astype(bool), want)


def test_constructor_rejects_unstable_omega():
    from LB_D2Q9.dimensionless import hip_dim as lb
    with pytest.raises(AssertionError):      # reference: `assert self.omega < 2.` (opencl_dim.py:120)
        dry(lb.Pipe_Flow)(diameter=1., rho=1., viscosity=-0.1, pressure_grad=-1., pipe_length=1., N=8, verbose=False)


def test_module_constants_and_helpers():
    from LB_D2Q9.dimensionless import opencl_dim as lb
    assert lb.get_divisible_global((3751, 1251), (32, 32)) == (3776, 1280)       # comparison notebook :233
    assert lb.get_divisible_global((64, 32, 9), (32, 32, 1)) == (64, 32, 9)
    assert lb.NUM_JUMPERS == 9 and lb.w.dtype == np.float32 and lb.cx.dtype == np.int32
    assert list(lb.cx) == [0, 1, 0, -1, 0, 1, -1, -1, 1] and list(lb.cy) == [0, 0, 1, 0, -1, 1, 1, -1, -1]
    assert float(lb.w.sum()) == pytest.approx(1.0, abs=1e-7)
    assert lb.cs2 == pytest.approx(1. / 3.) and lb.two_cs4 == pytest.approx(2. / 9.)


# ---- masks --------------------------------------------------------------------------------------------
def test_disc_pixels_is_the_strict_disc(oracle):
    from LB_D2Q9.masks import disc_pixels
    d = golden("o1_cyl_61x41")                 # mask built by the imported reference class
    m = np.zeros((61, 41), bool)
    xs, ys = disc_pixels(4 * .4 / .1, 4 * .5 / .1, 4, m.shape)
    m[xs, ys] = True
    assert np.array_equal(m, d["mask"])
    for xc, yc, r, shape in ((10.5, 7.25, 3.3, (30, 20)), (2., 2., 5., (12, 9)), (0., 0., 1., (4, 4))):
        m = np.zeros(shape, bool)
        xs, ys = disc_pixels(xc, yc, r, shape)
        m[xs, ys] = True
        assert np.array_equal(m, oracle.disc_mask(shape[0], shape[1], xc, yc, r))


def test_tiff_reader_on_the_reference_obstacle_image():
    """docs/CS205_obstacle_4.tif (ImageJ, big-endian, 8-bit, one strip) committed as input data."""
    from LB_D2Q9.masks import obstacle_mask_from_tiff, read_tiff_u8
    path = os.path.join(GOLDEN, "CS205_obstacle_4.tif")
    img = read_tiff_u8(path)
    assert img.shape == (400, 1600) and img.dtype == np.uint8
    assert set(np.unique(img)) <= {0, 255}
    assert 0.005 < (img > 0).mean() < 0.02                       # SURVEY: 1.08 % solid
    PIL = pytest.importorskip("PIL.Image")
    assert np.array_equal(img, np.asarray(PIL.open(path)))
    mask = obstacle_mask_from_tiff(path, (1600, 400))            # identity rescale, transposed to (x, y)
    assert mask.dtype == np.int32 and mask.flags.f_contiguous and np.array_equal(mask.astype(bool), (img > 0).T)
    big = obstacle_mask_from_tiff(path, (4096, 4096))
    assert big.shape == (4096, 4096) and abs(big.mean() - (img > 0).mean()) < 2e-3


def test_tiff_reader_little_endian_multi_strip(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    from LB_D2Q9.masks import read_tiff_u8, resize_nearest
    rng = np.random.default_rng(0)
    a = (rng.random((37, 53)) < 0.3).astype(np.uint8) * 255
    p = str(tmp_path / "m.tif")
    PIL.fromarray(a).save(p)
    assert np.array_equal(read_tiff_u8(p), a)
    assert np.array_equal(resize_nearest(a, a.shape), a)
    up = resize_nearest(a, (74, 106))
    assert np.array_equal(up[::2, ::2], a) and np.array_equal(up[1::2, 1::2], a)
    with pytest.raises(ValueError):
        read_tiff_u8(__file__)


# ---- slab arithmetic -------------------------------------------------------------------------------------
def test_partition_rows_and_neighbours():
    from LB_D2Q9.slabs import neighbours, partition_rows
    for ny, n in ((8192, 8), (8192, 3), (50, 7), (5, 5)):
        parts = partition_rows(ny, n)
        assert parts[0][0] == 0 and sum(h for _, h in parts) == ny
        assert all(a[0] + a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert max(h for _, h in parts) - min(h for _, h in parts) <= 1
    assert partition_rows(8192, 8) == [(1024 * r, 1024) for r in range(8)]
    with pytest.raises(ValueError):
        partition_rows(4, 5)
    assert neighbours(0, 4, False) == (-1, 1) and neighbours(3, 4, False) == (2, -1)
    assert neighbours(0, 4, True) == (3, 1) and neighbours(3, 4, True) == (2, 0)
    assert neighbours(0, 1, True) == (0, 0) and neighbours(0, 1, False) == (-1, -1)
    assert neighbours(1, 2, True) == (0, 0)


def test_mask_halo_rows_of_a_slab_match_the_header():
    """The obstacle-mask rows a slab keeps of its neighbours: LB_MASK_HALO_ROWS of them per side (the halo cycle
    recomputes seven of the neighbour's rows and reads the mask six rows beyond), nearest last below / nearest
    first above, wrapped in a periodic box, absent at a wall, empty outside a walled box."""
    import re
    from LB_D2Q9 import _native
    from LB_D2Q9.slabs import MASK_HALO_ROWS, _SlabSet
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "lb_hip.h")).read()
    assert MASK_HALO_ROWS == _native.LB_MASK_HALO_ROWS == int(re.search(r"#define LB_MASK_HALO_ROWS (\d+)", hdr).group(1))
    d, nx, ny = MASK_HALO_ROWS, 5, 40
    mask = np.zeros((nx, ny), bool)
    mask[2, :] = np.arange(ny) % 3 == 0                        # row y is solid at x=2 iff y % 3 == 0
    rows = lambda ys: np.array([[(y % ny) % 3 == 0 if x == 2 else False for x in range(nx)] for y in ys])
    south, north = _SlabSet._mask_halo_rows(mask, 14, 12, ny, False)
    assert south.shape == north.shape == (d, nx)
    assert np.array_equal(south, rows(range(14 - d, 14))) and np.array_equal(north, rows(range(26, 26 + d)))
    south, north = _SlabSet._mask_halo_rows(mask, 0, 12, ny, False)          # bottom slab of a walled box
    assert south is None and np.array_equal(north, rows(range(12, 12 + d)))
    south, north = _SlabSet._mask_halo_rows(mask, 0, 12, ny, True)           # periodic: wraps to the top rows
    assert np.array_equal(south, rows(range(ny - d, ny)))
    south, north = _SlabSet._mask_halo_rows(mask, ny - 4, 4, ny, True)
    assert np.array_equal(north, rows(range(0, d)))
    south, north = _SlabSet._mask_halo_rows(mask, ny - 12, 9, ny, False)     # 3 rows below the top wall: the rest is empty
    assert north[:3].any() == rows(range(ny - 3, ny)).any() and not north[3:].any()


# ---- headless frame dumper (reference: field_visualizer.py:146-161) -------------------------------------------
class _FakeSim(object):
    """run()/get_fields() provider without a GPU: a rigid rotation whose angle grows with the step count."""
    nx, ny = 24, 16

    def __init__(self):
        self.steps = 0

    def run(self, n):
        self.steps += n

    def get_fields(self):
        x = np.arange(self.nx)[:, None] - self.nx / 2.
        y = np.arange(self.ny)[None, :] - self.ny / 2.
        w = 1e-3 * self.steps
        return {"rho": np.ones((self.nx, self.ny), np.float32), "u": (-w * y * np.ones_like(x)).astype(np.float32),
                "v": (w * x * np.ones_like(y)).astype(np.float32)}


def test_frame_dumper_loop_and_png(tmp_path):
    from LB_D2Q9.frames import Frame_Dumper, vorticity
    sim = _FakeSim()
    d = Frame_Dumper(sim, "vorticity", num_steps_per_draw=5, max_magnitude=0.02, render_folder=str(tmp_path))
    frames = d.run(3)
    assert sim.steps == 15 and d.total_num_steps == 15 and len(frames) == 3
    assert [os.path.basename(f) for f in frames] == ["00000005.png", "00000010.png", "00000015.png"]
    g = sim.get_fields()
    assert np.allclose(vorticity(g["u"], g["v"]), 2 * 1e-3 * 15, atol=1e-9)       # curl of a rigid rotation = 2 w
    PIL = pytest.importorskip("PIL.Image")
    img = np.asarray(PIL.open(frames[-1]))
    assert img.shape == (sim.ny, sim.nx, 3)                                         # rows = y, drawn upwards
    assert img[0, 0, 0] == 255 and img[0, 0, 2] < 255                               # positive vorticity -> red side
    d2 = Frame_Dumper(sim, lambda: sim.get_fields()["u"], num_steps_per_draw=1, render_folder=str(tmp_path / "n"),
                      image_format="npy", run_func=lambda n: sim.run(2 * n))
    d2.on_draw()
    assert sim.steps == 17 and np.load(d2.frames_written[0]).shape == (sim.nx, sim.ny)


def test_cython_style_constants_match_reference_prints(oracle):
    """LB_D2Q9.dimensionless.cython_dim (GPU-backed): the comparison notebook's cylinder case prints
    L=0.1, T=0.08, Re=1.5625, omega=0.413223140496, inlet rho 1.00368738304 on a 3751x1251 grid
    (docs/python_cython_opencl_comparison.ipynb cells 10-13); any other input must equal the oracle's
    restatement of the reference constructor."""
    from LB_D2Q9.dimensionless import cython_dim as lb

    def dry_c(cls):
        class Dry(cls):
            def init_hydro(self):
                self.inlet_rho, self.outlet_rho = self._boundary_densities()

            def update_feq(self):
                pass

            def init_pop(self, amplitude=.001):
                pass
        return Dry

    c = dry_c(lb.Pipe_Flow_Cylinder)(cylinder_center=[.75, .5], cylinder_radius=.1, diameter=1., rho=1., viscosity=1.,
                                     pressure_grad=-10., pipe_length=3., N=125, verbose=False)
    assert c.L == pytest.approx(0.1) and c.T == pytest.approx(0.08) and c.Re == pytest.approx(1.5625)
    assert c.omega == pytest.approx(0.413223140496, rel=1e-11) and (c.nx, c.ny) == (3751, 1251)
    assert c.inlet_rho == pytest.approx(1.00368738304, rel=1e-11)
    assert c.obstacle_mask.dtype == bool and c.obstacle_mask.sum() == len(c.obstacle_pixels[0]) > 40000
    kw = dict(diameter=0.8, rho=1.3, viscosity=0.21, pressure_grad=-2.5, pipe_length=1.7, N=41, time_prefactor=0.7)
    s = dry_c(lb.Pipe_Flow)(verbose=False, **kw)
    p = oracle.cython_pipe_parameters(**kw)
    for k in ("L", "T", "Re", "delta_x", "delta_t", "lb_viscosity", "omega", "lx", "ly", "nx", "ny", "inlet_rho", "outlet_rho"):
        assert getattr(s, k) == p[k], k
