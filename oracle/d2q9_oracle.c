/*
 * d2q9_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * Plain-C, single-threaded CPU restatement of the two reference semantics of the
 * D2Q9 BGK collide-and-stream step of latticeboltzmann/2d-lb:
 *
 *   o2_*  "OpenCL path"  : kernels of LB_D2Q9/D2Q9.cl driven in the order of
 *                          LB_D2Q9/dimensionless/opencl_dim.py:372-387
 *   o1_*  "Cython path"  : LB_D2Q9/dimensionless/cython_dim.pyx:160-359, 398-513
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object (oracle/_build/libd2q9_oracle.so).  The product
 * (2d-lb_amd/) never links, imports or calls it.
 *
 * Parity pin: see oracle/README.md.  Every o2_/o1_ routine is checked against
 * fixtures under tests/golden/ that were produced by executing the reference's
 * own sources in the build container (oracle/make_golden.py), and against the
 * reference's analytic Poiseuille known-answer test and printed constants.
 *
 * Build: gcc -O2 -std=gnu99 -ffp-contract=off  (no FMA contraction, so the
 * float/double operation order written here is the order executed).
 *
 * Build-defined extensions (no reference counterpart; marked [BD]): periodic
 * streaming and the four-wall lid-driven cavity closure.  They reuse the
 * reference's cell arithmetic and are the oracle for BASELINE configs 2-4.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

/* lattice set, opencl_dim.py:22-26 / cython_dim.pyx:16-19 */
static const int   CX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
static const int   CY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};

enum { BC_PIPE = 0, BC_PERIODIC = 1, BC_CAVITY = 2, BC_VELOCITY_INLET = 3 };

/* ======================================================================= */
/*  O2 : OpenCL-path semantics.  idx(k,x,y) = k*nx*ny + y*nx + x            */
/* ======================================================================= */

typedef struct {
    int32_t nx, ny, bc_mode, _pad;
    float omega, rho_in, rho_out, lid_u, rho0;
    float u_w, u_e;                       /* BC_VELOCITY_INLET: imposed inlet / outlet speed */
    float cs2, two_cs2, two_cs4;          /* float32 casts made by the host, opencl_dim.py:305 */
    float *f, *fs, *feq, *rho, *u, *v;
    const int32_t *mask;                  /* NULL when there is no obstacle */
    int32_t d2q9i, _pad2;                 /* 1: the kernels of LB_D2Q9/D2Q9i.cl ("incompressible" fork) */
} o2_state;

#define P(k) ((size_t)(k) * plane)

/* D2Q9.cl:139-171 `move`.  Push every population one link; a target outside
 * the box is dropped (fs keeps whatever it held).  [BD] with wrap_x / wrap_y
 * the target index wraps instead (porous_media/single_component.cl:338-375 is
 * the reference's only periodic precedent). */
void o2_stream(const float *f, float *fs, int nx, int ny, int wrap_x, int wrap_y)
{
    const size_t plane = (size_t)nx * ny;
    /* (every (k, x, y) writes its own target: the loops are independent.  The omp pragmas of the o2_* routines only take
     *  effect in the -fopenmp build, which tests/test_gpu_fullsize.py uses to hold the 8192^2 run to this oracle in seconds;
     *  each cell's arithmetic is untouched, so both builds give the same bits: tests/test_oracle_golden.py.) */
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < 9; ++k)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                int tx = x + CX[k], ty = y + CY[k];
                if (wrap_x) tx = (tx + nx) % nx;
                if (wrap_y) ty = (ty + ny) % ny;
                if (tx < 0 || tx >= nx || ty < 0 || ty >= ny) continue;
                fs[P(k) + (size_t)ty * nx + tx] = f[P(k) + (size_t)y * nx + x];
            }
}

/* D2Q9.cl:123-137 `copy_buffer` */
void o2_copy(const float *src, float *dst, int nx, int ny)
{
    const size_t plane = (size_t)nx * ny;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < 9; ++k) memcpy(dst + k * plane, src + k * plane, sizeof(float) * plane);
}

/* D2Q9.cl:173-261 `move_bcs`: Zou-He pressure inlet (x=0) / outlet (x=nx-1),
 * no-slip north/south rows, four corner closures.  Literals such as (2./3.)
 * are double in the reference source, so those products are formed in double
 * and rounded once on the store; that is reproduced here. */
void o2_bc_pipe(float *f, float rho_in, float rho_out, int nx, int ny)
{
    const size_t plane = (size_t)nx * ny;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            const int on_w = (x == 0), on_e = (x == nx - 1);
            const int on_s = (y == 0), on_n = (y == ny - 1);
            if (!(on_w || on_e || on_s || on_n)) continue;
            float *c = f + (size_t)y * nx + x;
            const float f0 = c[P(0)], f1 = c[P(1)], f2 = c[P(2)], f3 = c[P(3)], f4 = c[P(4)],
                        f5 = c[P(5)], f6 = c[P(6)], f7 = c[P(7)], f8 = c[P(8)];
            /* the reference tests the nine regions with independent ifs; for nx,ny >= 2
             * exactly one of them holds on any edge cell */
            if (on_w && !on_s && !on_n) {                                   /* :198-203 */
                float uu = -((f0 + f2 + 2 * f3 + f4 + 2 * f6 + 2 * f7 - rho_in) / rho_in);
                c[P(1)] = (float)(f3 + (2. / 3.) * rho_in * uu);
                c[P(5)] = (float)(-.5 * f2 + .5 * f4 + f7 + (1. / 6.) * uu * rho_in);
                c[P(8)] = (float)(.5 * f2 - .5 * f4 + f6 + (1. / 6.) * uu * rho_in);
            }
            if (on_e && !on_s && !on_n) {                                   /* :205-210 */
                float uu = -1 + (f0 + 2 * f1 + f2 + f4 + 2 * f5 + 2 * f8) / rho_out;
                c[P(3)] = (float)(f1 - (2. / 3.) * rho_out * uu);
                c[P(6)] = (float)(-.5 * f2 + .5 * f4 + f8 - (1. / 6.) * uu * rho_out);
                c[P(7)] = (float)(.5 * f2 - .5 * f4 + f5 - (1. / 6.) * uu * rho_out);
            }
            if (on_n && !on_w && !on_e) {                                   /* :213-217 */
                c[P(4)] = f2;
                c[P(8)] = (float)(.5 * (-f1 + f3 + 2 * f6));
                c[P(7)] = (float)(.5 * (f1 - f3 + 2 * f5));
            }
            if (on_s && !on_w && !on_e) {                                   /* :219-223 */
                c[P(2)] = f4;
                c[P(6)] = (float)(.5 * (f1 - f3 + 2 * f8));
                c[P(5)] = (float)(.5 * (-f1 + f3 + 2 * f7));
            }
            if (on_w && on_s) {                                             /* :228-234 */
                float t = (float)(.5 * (-f0 - 2 * f3 - 2 * f4 - 2 * f7 + rho_in));
                c[P(1)] = f3; c[P(2)] = f4; c[P(5)] = f7; c[P(6)] = t; c[P(8)] = t;
            }
            if (on_w && on_n) {                                             /* :236-242 */
                float t = (float)(.5 * (-f0 - 2 * f2 - 2 * f3 - 2 * f6 + rho_in));
                c[P(1)] = f3; c[P(4)] = f2; c[P(8)] = f6; c[P(5)] = t; c[P(7)] = t;
            }
            if (on_e && on_s) {                                             /* :245-251 */
                float t = (float)(.5 * (-f0 - 2 * f1 - 2 * f4 - 2 * f8 + rho_out));
                c[P(3)] = f1; c[P(2)] = f4; c[P(6)] = f8; c[P(5)] = t; c[P(7)] = t;
            }
            if (on_e && on_n) {                                             /* :253-259 */
                float t = (float)(.5 * (-f0 - 2 * f1 - 2 * f2 - 2 * f5 + rho_out));
                c[P(3)] = f1; c[P(4)] = f2; c[P(7)] = f5; c[P(6)] = t; c[P(8)] = t;
            }
        }
}

/* [BD] Lid-driven cavity closure.  The reference's north/south rule
 * (D2Q9.cl:213-223) is the Zou-He velocity condition with zero wall velocity;
 * rotating it gives the west/east rule, and a tangential wall speed U adds
 * -/+ (1/2) rho_w U to the two diagonal unknowns (the same structure as the
 * velocity-inlet kernel D2Q9.cl:290-303).  Corners reuse the reference corner
 * closure (D2Q9.cl:228-259) with the rest density rho0 in place of the
 * prescribed inlet/outlet density. */
void o2_bc_cavity(float *f, float lid_u, float rho0, int nx, int ny)
{
    const size_t plane = (size_t)nx * ny;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            const int on_w = (x == 0), on_e = (x == nx - 1);
            const int on_s = (y == 0), on_n = (y == ny - 1);
            if (!(on_w || on_e || on_s || on_n)) continue;
            float *c = f + (size_t)y * nx + x;
            const float f0 = c[P(0)], f1 = c[P(1)], f2 = c[P(2)], f3 = c[P(3)], f4 = c[P(4)],
                        f5 = c[P(5)], f6 = c[P(6)], f7 = c[P(7)], f8 = c[P(8)];
            if (on_n && !on_w && !on_e) {
                float rw = f0 + f1 + f3 + 2.f * (f2 + f5 + f6);
                c[P(4)] = f2;
                c[P(7)] = 0.5f * (f1 - f3 + 2.f * f5) - 0.5f * rw * lid_u;
                c[P(8)] = 0.5f * (-f1 + f3 + 2.f * f6) + 0.5f * rw * lid_u;
            }
            if (on_s && !on_w && !on_e) {
                c[P(2)] = f4;
                c[P(6)] = 0.5f * (f1 - f3 + 2.f * f8);
                c[P(5)] = 0.5f * (-f1 + f3 + 2.f * f7);
            }
            if (on_w && !on_s && !on_n) {
                c[P(1)] = f3;
                c[P(5)] = 0.5f * (-f2 + f4 + 2.f * f7);
                c[P(8)] = 0.5f * (f2 - f4 + 2.f * f6);
            }
            if (on_e && !on_s && !on_n) {
                c[P(3)] = f1;
                c[P(6)] = 0.5f * (-f2 + f4 + 2.f * f8);
                c[P(7)] = 0.5f * (f2 - f4 + 2.f * f5);
            }
            if (on_w && on_s) {
                float t = 0.5f * (-f0 - 2 * f3 - 2 * f4 - 2 * f7 + rho0);
                c[P(1)] = f3; c[P(2)] = f4; c[P(5)] = f7; c[P(6)] = t; c[P(8)] = t;
            }
            if (on_w && on_n) {
                float t = 0.5f * (-f0 - 2 * f2 - 2 * f3 - 2 * f6 + rho0);
                c[P(1)] = f3; c[P(4)] = f2; c[P(8)] = f6; c[P(5)] = t; c[P(7)] = t;
            }
            if (on_e && on_s) {
                float t = 0.5f * (-f0 - 2 * f1 - 2 * f4 - 2 * f8 + rho0);
                c[P(3)] = f1; c[P(2)] = f4; c[P(6)] = f8; c[P(5)] = t; c[P(7)] = t;
            }
            if (on_e && on_n) {
                float t = 0.5f * (-f0 - 2 * f1 - 2 * f2 - 2 * f5 + rho0);
                c[P(3)] = f1; c[P(4)] = f2; c[P(7)] = f5; c[P(6)] = t; c[P(8)] = t;
            }
        }
}

/* D2Q9.cl:263-321 `move_bcs_PeriodicBC_VelocityInlet` (only OLD/opencl.py:290-296 launches it): imposed
 * x-velocity u_w at x=0 and u_e at x=nx-1 for 1<=y<=ny-2 (Zou-He, double literals), and on the whole
 * north / south rows a copy of the three links that streaming could not deliver from the same x of
 * the opposite wall row.  Each work-item first reads its own nine links; the copies read the opposite
 * row from memory, planes that no other work-item of the launch writes, so the order is immaterial. */
void o2_bc_velocity_inlet(float *f, float u_w, float u_e, int nx, int ny)
{
    const size_t plane = (size_t)nx * ny;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            float *c = f + (size_t)y * nx + x;
            const float f0 = c[P(0)], f1 = c[P(1)], f2 = c[P(2)], f3 = c[P(3)], f4 = c[P(4)],
                        f5 = c[P(5)], f6 = c[P(6)], f7 = c[P(7)], f8 = c[P(8)];
            if (x == 0 && y >= 1 && y < ny - 1) {                                 /* :291-296 */
                float rho_w = (float)((1. / (1. - u_w)) * (f0 + f2 + f4 + 2 * (f3 + f6 + f7)));
                c[P(1)] = (float)(f3 + (2. / 3.) * rho_w * u_w);
                c[P(5)] = (float)(f7 - (1. / 2.) * (f2 - f4) + (1. / 6.) * rho_w * u_w);
                c[P(8)] = (float)(f6 + (1. / 2.) * (f2 - f4) + (1. / 6.) * rho_w * u_w);
            }
            if (x == nx - 1 && y >= 1 && y < ny - 1) {                            /* :298-303 */
                float rho_e = (float)((1. / (1. + u_e)) * (f0 + f2 + f4 + 2. * (f1 + f5 + f8)));
                c[P(3)] = (float)(f1 - (2. / 3.) * rho_e * u_e);
                c[P(6)] = (float)(f5 + (1. / 2.) * (f2 - f4) - (1. / 6.) * rho_e * u_e);
                c[P(7)] = (float)(f8 - (1. / 2.) * (f2 - f4) - (1. / 6.) * rho_e * u_e);
            }
            if (y == ny - 1) {                                                    /* :306-311 */
                c[P(4)] = f[P(4) + x]; c[P(8)] = f[P(8) + x]; c[P(7)] = f[P(7) + x];
            }
            if (y == 0) {                                                         /* :314-318 */
                const size_t top = (size_t)(ny - 1) * nx + x;
                c[P(2)] = f[P(2) + top]; c[P(6)] = f[P(6) + top]; c[P(5)] = f[P(5) + top];
            }
        }
}

/* D2Q9.cl:323-374 `update_hydro_PeriodicBC_VelocityInlet`: rho everywhere; u,v only for 0<x<nx-1; at
 * the inlet / outlet (1<=y<=ny-2) rho from the Zou-He closure and u = u_w / u_e; v there and u,v on
 * the four corner cells keep whatever they held. */
void o2_moments_velocity_inlet(const float *f, float *rho, float *u, float *v, float u_w, float u_e, int nx, int ny)
{
    const size_t plane = (size_t)nx * ny;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            const size_t i = (size_t)y * nx + x;
            const float f0 = f[P(0) + i], f1 = f[P(1) + i], f2 = f[P(2) + i], f3 = f[P(3) + i],
                        f4 = f[P(4) + i], f5 = f[P(5) + i], f6 = f[P(6) + i], f7 = f[P(7) + i],
                        f8 = f[P(8) + i];
            float r = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + f8;
            rho[i] = r;
            float inv = (float)(1. / r);
            if (x != 0 && x != nx - 1) {
                u[i] = (f1 - f3 + f5 - f6 - f7 + f8) * inv;
                v[i] = (f5 + f2 + f6 - f7 - f4 - f8) * inv;
            }
            if (x == 0 && y != 0 && y < ny - 1) {
                rho[i] = (float)((1. / (1. - u_w)) * (f0 + f2 + f4 + 2. * (f3 + f6 + f7)));
                u[i] = u_w;
            }
            if (x == nx - 1 && y != 0 && y < ny - 1) {
                rho[i] = (float)((1. / (1. + u_e)) * (f0 + f2 + f4 + 2. * (f1 + f5 + f8)));
                u[i] = u_e;
            }
        }
}

/* D2Q9.cl:398-433 `bounceback_in_obstacle`: on mask==1 exchange opposite links. */
void o2_bounceback(const int32_t *mask, float *f, int nx, int ny)
{
    const size_t plane = (size_t)nx * ny;
    static const int A[4] = {1, 2, 5, 6}, B[4] = {3, 4, 7, 8};
    for (size_t i = 0; i < plane; ++i) {
        if (mask[i] != 1) continue;
        for (int p = 0; p < 4; ++p) {
            float a = f[P(A[p]) + i], b = f[P(B[p]) + i];
            f[P(A[p]) + i] = b;
            f[P(B[p]) + i] = a;
        }
    }
}

/* D2Q9.cl:377-396 `set_zero_velocity_in_obstacle` (init only, opencl_dim.py:506-508) */
void o2_zero_velocity(const int32_t *mask, float *u, float *v, int nx, int ny)
{
    for (size_t i = 0; i < (size_t)nx * ny; ++i)
        if (mask[i] == 1) { u[i] = 0.f; v[i] = 0.f; }
}

/* D2Q9.cl:67-100 `update_hydro`: rho is the left-to-right float sum, the
 * reciprocal is a double divide (`1./rho`) rounded to float. */
void o2_moments(const float *f, float *rho, float *u, float *v, int nx, int ny)
{
    const size_t plane = (size_t)nx * ny;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < plane; ++i) {
        const float f0 = f[P(0) + i], f1 = f[P(1) + i], f2 = f[P(2) + i], f3 = f[P(3) + i],
                    f4 = f[P(4) + i], f5 = f[P(5) + i], f6 = f[P(6) + i], f7 = f[P(7) + i],
                    f8 = f[P(8) + i];
        float r = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + f8;
        float inv = (float)(1. / r);
        rho[i] = r;
        u[i] = (f1 - f3 + f5 - f6 - f7 + f8) * inv;
        v[i] = (f5 + f2 + f6 - f7 - f4 - f8) * inv;
    }
}

/* D2Q9.cl:2-64 `update_feq`: w*rho*(1 + cu/cs2 + cu^2/two_cs4 - usq/two_cs2),
 * all float, divisions by the float32-cast constants, weights float32
 * (opencl_dim.py:22-23). */
void o2_feq(float *feq, const float *rho, const float *u, const float *v,
            float cs2, float two_cs2, float two_cs4, int nx, int ny)
{
    static const float W[9] = {(float)(4. / 9.), (float)(1. / 9.), (float)(1. / 9.),
                               (float)(1. / 9.), (float)(1. / 9.), (float)(1. / 36.),
                               (float)(1. / 36.), (float)(1. / 36.), (float)(1. / 36.)};
    const size_t plane = (size_t)nx * ny;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < 9; ++k)
        for (size_t i = 0; i < plane; ++i) {
            float cu = CX[k] * u[i] + CY[k] * v[i];
            float usq = u[i] * u[i] + v[i] * v[i];
            float inner = 1.f + cu / cs2 + cu * cu / two_cs4 - usq / two_cs2;
            feq[P(k) + i] = W[k] * rho[i] * inner;
        }
}

/* D2Q9.cl:102-121 `collide_particles` */
void o2_collide(float *f, const float *feq, float omega, int nx, int ny)
{
    const size_t n = 9u * (size_t)nx * ny;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
        f[i] = f[i] * (1 - omega) + omega * feq[i];
}

/* ---- LB_D2Q9/D2Q9i.cl: the "incompressible" fork.  Identical to D2Q9.cl except for the three routines
 * below (and an unused local in the wall rules); host: dimensionless/opencl_dim_D2Q9i.py. ---------- */

/* D2Q9i.cl:190-205: inlet / outlet of `move_bcs`; walls and corners as in D2Q9.cl */
void o2i_bc_pipe(float *f, float rho_in, float rho_out, int nx, int ny)
{
    const size_t plane = (size_t)nx * ny;
    /* the inlet / outlet columns get the fork's formulas ... */
    for (int y = 1; y < ny - 1; ++y) {
        float *c = f + (size_t)y * nx;
        {
            const float f0 = c[P(0)], f2 = c[P(2)], f3 = c[P(3)], f4 = c[P(4)], f6 = c[P(6)], f7 = c[P(7)];
            float uu = -f0 - f2 - 2 * f3 - f4 - 2 * f6 - 2 * f7 + rho_in;
            c[P(1)] = (float)((1. / 3.) * (3 * f3 + 2 * uu));
            c[P(5)] = (float)((1. / 6.) * (-3 * f2 + 3 * f4 + 6 * f7 + uu));
            c[P(8)] = (float)((1. / 6.) * (3 * f2 - 3 * f4 + 6 * f6 + uu));
        }
        c += nx - 1;
        {
            const float f0 = c[P(0)], f1 = c[P(1)], f2 = c[P(2)], f4 = c[P(4)], f5 = c[P(5)], f8 = c[P(8)];
            float uu = f0 + 2 * f1 + f2 + f4 + 2 * f5 + 2 * f8 - rho_out;
            c[P(3)] = (float)((1. / 3.) * (3 * f1 - 2 * uu));
            c[P(6)] = (float)((1. / 6.) * (-3 * f2 + 3 * f4 + 6 * f8 - uu));
            c[P(7)] = (float)((1. / 6.) * (3 * f2 - 3 * f4 + 6 * f5 - uu));
        }
    }
    /* ... walls and corners the common ones: run the D2Q9 rule on rows 0 and ny-1 only */
    for (int y = 0; y < ny; y += (ny - 1)) {
        float *row = f + (size_t)y * nx;
        /* build a 1-row view by calling the shared routine on a temporary with the same edge logic */
        for (int x = 0; x < nx; ++x) {
            float *c = row + x;
            const int on_w = (x == 0), on_e = (x == nx - 1), on_s = (y == 0), on_n = (y == ny - 1);
            const float f0 = c[P(0)], f1 = c[P(1)], f2 = c[P(2)], f3 = c[P(3)], f4 = c[P(4)],
                        f5 = c[P(5)], f6 = c[P(6)], f7 = c[P(7)], f8 = c[P(8)];
            if (on_n && !on_w && !on_e) { c[P(4)] = f2; c[P(8)] = (float)(.5 * (-f1 + f3 + 2 * f6)); c[P(7)] = (float)(.5 * (f1 - f3 + 2 * f5)); }
            if (on_s && !on_w && !on_e) { c[P(2)] = f4; c[P(6)] = (float)(.5 * (f1 - f3 + 2 * f8)); c[P(5)] = (float)(.5 * (-f1 + f3 + 2 * f7)); }
            if (on_w && on_s) { float t = (float)(.5 * (-f0 - 2 * f3 - 2 * f4 - 2 * f7 + rho_in));
                                c[P(1)] = f3; c[P(2)] = f4; c[P(5)] = f7; c[P(6)] = t; c[P(8)] = t; }
            if (on_w && on_n) { float t = (float)(.5 * (-f0 - 2 * f2 - 2 * f3 - 2 * f6 + rho_in));
                                c[P(1)] = f3; c[P(4)] = f2; c[P(8)] = f6; c[P(5)] = t; c[P(7)] = t; }
            if (on_e && on_s) { float t = (float)(.5 * (-f0 - 2 * f1 - 2 * f4 - 2 * f8 + rho_out));
                                c[P(3)] = f1; c[P(2)] = f4; c[P(6)] = f8; c[P(5)] = t; c[P(7)] = t; }
            if (on_e && on_n) { float t = (float)(.5 * (-f0 - 2 * f1 - 2 * f2 - 2 * f5 + rho_out));
                                c[P(3)] = f1; c[P(4)] = f2; c[P(7)] = f5; c[P(6)] = t; c[P(8)] = t; }
        }
        if (ny == 1) break;
    }
}

/* D2Q9i.cl:88-94 `update_hydro`: momentum, not velocity (no division by rho) */
void o2i_moments(const float *f, float *rho, float *u, float *v, int nx, int ny)
{
    const size_t plane = (size_t)nx * ny;
    for (size_t i = 0; i < plane; ++i) {
        const float f0 = f[P(0) + i], f1 = f[P(1) + i], f2 = f[P(2) + i], f3 = f[P(3) + i],
                    f4 = f[P(4) + i], f5 = f[P(5) + i], f6 = f[P(6) + i], f7 = f[P(7) + i],
                    f8 = f[P(8) + i];
        rho[i] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + f8;
        u[i] = (f1 + f5 + f8 - f6 - f3 - f7);
        v[i] = (f6 + f2 + f5 - f7 - f4 - f8);
    }
}

/* D2Q9i.cl:55-62 `update_feq`: inner = rho + 3 cu + 4.5 cu^2 - 1.5 usq formed in double (the literals
 * 3., 9./2., 3./2. are double), rounded to float, then w*rho*inner in float */
void o2i_feq(float *feq, const float *rho, const float *u, const float *v, int nx, int ny)
{
    static const float W[9] = {(float)(4. / 9.), (float)(1. / 9.), (float)(1. / 9.),
                               (float)(1. / 9.), (float)(1. / 9.), (float)(1. / 36.),
                               (float)(1. / 36.), (float)(1. / 36.), (float)(1. / 36.)};
    const size_t plane = (size_t)nx * ny;
    for (int k = 0; k < 9; ++k)
        for (size_t i = 0; i < plane; ++i) {
            float cu = CX[k] * u[i] + CY[k] * v[i];
            float usq = u[i] * u[i] + v[i] * v[i];
            float inner = (float)(rho[i] + 3. * cu + (9. / 2.) * (cu * cu) - (3. / 2.) * usq);
            feq[P(k) + i] = W[k] * rho[i] * inner;
        }
}

/* Individual phases in the order of opencl_dim.py:380-387 (Pipe_Flow.run) with
 * the obstacle hook of Pipe_Flow_Cylinder.move_bcs (:510-518). */
void o2_phase_move(o2_state *s)
{
    const int wrap = (s->bc_mode == BC_PERIODIC);
    o2_stream(s->f, s->fs, s->nx, s->ny, wrap, wrap);
    o2_copy(s->fs, s->f, s->nx, s->ny);
}

void o2_phase_bcs(o2_state *s)
{
    if (s->bc_mode == BC_PIPE && s->d2q9i) o2i_bc_pipe(s->f, s->rho_in, s->rho_out, s->nx, s->ny);
    else if (s->bc_mode == BC_PIPE) o2_bc_pipe(s->f, s->rho_in, s->rho_out, s->nx, s->ny);
    if (s->bc_mode == BC_CAVITY) o2_bc_cavity(s->f, s->lid_u, s->rho0, s->nx, s->ny);
    if (s->bc_mode == BC_VELOCITY_INLET) o2_bc_velocity_inlet(s->f, s->u_w, s->u_e, s->nx, s->ny);
    if (s->mask) o2_bounceback(s->mask, s->f, s->nx, s->ny);
}

void o2_run(o2_state *s, int n)
{
    for (int it = 0; it < n; ++it) {
        o2_phase_move(s);
        o2_phase_bcs(s);
        if (s->bc_mode == BC_VELOCITY_INLET)
            o2_moments_velocity_inlet(s->f, s->rho, s->u, s->v, s->u_w, s->u_e, s->nx, s->ny);
        else if (s->d2q9i)
            o2i_moments(s->f, s->rho, s->u, s->v, s->nx, s->ny);
        else
            o2_moments(s->f, s->rho, s->u, s->v, s->nx, s->ny);
        if (s->d2q9i) {
            /* opencl_dim_D2Q9i.py:494-503: the cylinder class re-zeroes u,v in the obstacle every step */
            if (s->mask) o2_zero_velocity(s->mask, s->u, s->v, s->nx, s->ny);
            o2i_feq(s->feq, s->rho, s->u, s->v, s->nx, s->ny);
        } else
            o2_feq(s->feq, s->rho, s->u, s->v, s->cs2, s->two_cs2, s->two_cs4, s->nx, s->ny);
        o2_collide(s->f, s->feq, s->omega, s->nx, s->ny);
    }
}
#undef P

/* ======================================================================= */
/*  O1 : Cython-path semantics.  idx(k,i,j) = k*nx*ny + i*ny + j  (j = y)   */
/*  f, feq, rho float32; u, v float64 containers (cython_dim.pyx:156-157).  */
/* ======================================================================= */

typedef struct {
    int32_t nx, ny;      /* nx = lx+1, ny = ly+1 */
    int32_t numpy2;      /* 1: NumPy>=2 (NEP 50) promotions = how the fixtures were made here;
                            0: NumPy 1.x value-based casting = what the authors ran */
    int32_t _pad;
    double omega, rho_in, rho_out;       /* np.float64 scalars in the reference */
    float *f, *feq, *rho;
    double *u, *v;
    const uint8_t *mask; /* (nx,ny) C order, NULL when no obstacle */
} o1_state;

#define F(k, i, j) f[(size_t)(k) * plane + (size_t)(i) * ny + (j)]

/* cython_dim.pyx:204-269 `move_bcs` (+ :468-513 obstacle swap).  Inlet/outlet
 * use the *stored* u of the previous update_hydro. */
void o1_move_bcs(o1_state *s)
{
    const int nx = s->nx, ny = s->ny, lx = nx - 1, ly = ny - 1;
    const size_t plane = (size_t)nx * ny;
    float *f = s->f;
    const double rin = s->rho_in, rout = s->rho_out;
    /* numpy slice arithmetic :214-224: the f32 terms are summed in f32, the
     * u-term is f64, the final add is f64, the store rounds to f32 */
    for (int j = 1; j < ly; ++j) {
        const double u0 = s->u[(size_t)0 * ny + j];
        const float f2 = F(2, 0, j), f3 = F(3, 0, j), f4 = F(4, 0, j), f6 = F(6, 0, j), f7 = F(7, 0, j);
        F(1, 0, j) = (float)((double)f3 + ((2. / 3.) * rin) * u0);
        F(5, 0, j) = (float)((double)((-.5f * f2 + .5f * f4) + f7) + ((1. / 6.) * u0) * rin);
        F(8, 0, j) = (float)((double)((.5f * f2 - .5f * f4) + f6) + ((1. / 6.) * u0) * rin);
    }
    for (int j = 1; j < ly; ++j) {
        const double ul = s->u[(size_t)lx * ny + j];
        const float f1 = F(1, lx, j), f2 = F(2, lx, j), f4 = F(4, lx, j), f5 = F(5, lx, j), f8 = F(8, lx, j);
        F(3, lx, j) = (float)((double)f1 - ((2. / 3.) * rout) * ul);
        F(6, lx, j) = (float)((double)((-.5f * f2 + .5f * f4) + f8) - ((1. / 6.) * ul) * rout);
        F(7, lx, j) = (float)((double)((.5f * f2 - .5f * f4) + f5) - ((1. / 6.) * ul) * rout);
    }
    const float rinf = (float)rin, routf = (float)rout;   /* `cdef float`, :227-228 */
    for (int i = 1; i < lx; ++i) {                        /* plain bounce-back walls :230-240 */
        F(4, i, ly) = F(2, i, ly); F(8, i, ly) = F(6, i, ly); F(7, i, ly) = F(5, i, ly);
    }
    for (int i = 1; i < lx; ++i) {
        F(2, i, 0) = F(4, i, 0); F(6, i, 0) = F(8, i, 0); F(5, i, 0) = F(7, i, 0);
    }
    /* corners :242-269.  Cython writes the integer literal 2 as `2.0` next to a C float,
     * so the whole bracket is evaluated in double and rounded once on the store. */
    F(1, 0, 0) = F(3, 0, 0); F(2, 0, 0) = F(4, 0, 0); F(5, 0, 0) = F(7, 0, 0);
    { float t = (float)(.5 * (-F(0, 0, 0) - 2.0 * F(3, 0, 0) - 2.0 * F(4, 0, 0) - 2.0 * F(7, 0, 0) + rinf));
      F(6, 0, 0) = t; F(8, 0, 0) = t; }
    F(1, 0, ly) = F(3, 0, ly); F(4, 0, ly) = F(2, 0, ly);
    { float t = (float)(.5 * (-F(0, 0, ly) - 2.0 * F(2, 0, ly) - 2.0 * F(3, 0, ly) - 2.0 * F(6, 0, ly) + rinf));
      F(5, 0, ly) = t; F(7, 0, ly) = t; }
    F(8, 0, ly) = F(6, 0, ly);
    F(3, lx, 0) = F(1, lx, 0); F(2, lx, 0) = F(4, lx, 0); F(6, lx, 0) = F(8, lx, 0);
    { float t = (float)(.5 * (-F(0, lx, 0) - 2.0 * F(1, lx, 0) - 2.0 * F(4, lx, 0) - 2.0 * F(8, lx, 0) + routf));
      F(5, lx, 0) = t; F(7, lx, 0) = t; }
    F(3, lx, ly) = F(1, lx, ly); F(4, lx, ly) = F(2, lx, ly);
    { float t = (float)(.5 * (-F(0, lx, ly) - 2.0 * F(1, lx, ly) - 2.0 * F(2, lx, ly) - 2.0 * F(5, lx, ly) + routf));
      F(6, lx, ly) = t; F(7, lx, ly) = F(5, lx, ly); F(8, lx, ly) = t; }

    if (s->mask) {                                        /* :468-513 */
        static const int A[4] = {1, 2, 5, 6}, B[4] = {3, 4, 7, 8};
        for (size_t c = 0; c < plane; ++c) {
            if (!s->mask[c]) continue;
            for (int p = 0; p < 4; ++p) {
                float a = f[(size_t)A[p] * plane + c], b = f[(size_t)B[p] * plane + c];
                f[(size_t)A[p] * plane + c] = b;
                f[(size_t)B[p] * plane + c] = a;
            }
        }
    }
}

/* cython_dim.pyx:271-299 `move`: in-place pull with the reference's loop order
 * and index ranges (some tangential links never move on one edge row/column). */
void o1_move(o1_state *s)
{
    const int nx = s->nx, ny = s->ny, lx = nx - 1, ly = ny - 1;
    const size_t plane = (size_t)nx * ny;
    float *f = s->f;
    /* four loop nests over disjoint link pairs: independent of each other, sequential inside */
#pragma omp parallel sections
    {
#pragma omp section
        for (int j = ly; j > 0; --j)
            for (int i = 0; i < lx; ++i) { F(2, i, j) = F(2, i, j - 1); F(6, i, j) = F(6, i + 1, j - 1); }
#pragma omp section
        for (int j = ly; j > 0; --j)
            for (int i = lx; i > 0; --i) { F(1, i, j) = F(1, i - 1, j); F(5, i, j) = F(5, i - 1, j - 1); }
#pragma omp section
        for (int j = 0; j < ly; ++j)
            for (int i = lx; i > 0; --i) { F(4, i, j) = F(4, i, j + 1); F(8, i, j) = F(8, i - 1, j + 1); }
#pragma omp section
        for (int j = 0; j < ly; ++j)
            for (int i = 0; i < lx; ++i) { F(3, i, j) = F(3, i + 1, j); F(7, i, j) = F(7, i + 1, j + 1); }
    }
}

/* cython_dim.pyx:302-333 `update_hydro` (+ :459-466 obstacle zeroing) */
void o1_update_hydro(o1_state *s)
{
    /* the omp pragmas of the o1_* phases only take effect in the -fopenmp build used for the
     * "all host cores" CPU baseline; every cell is independent, so results do not change */
    const int nx = s->nx, ny = s->ny, lx = nx - 1, ly = ny - 1;
    const size_t plane = (size_t)nx * ny;
    const float *f = s->f;
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < plane; ++c) {
        /* np.sum(f, axis=0): plane-by-plane float32 accumulation */
        float r = f[c];
        for (int k = 1; k < 9; ++k) r += f[(size_t)k * plane + c];
        s->rho[c] = r;
        float inv = 1.f / r;
        const float f1 = f[plane + c], f2 = f[2 * plane + c], f3 = f[3 * plane + c], f4 = f[4 * plane + c],
                    f5 = f[5 * plane + c], f6 = f[6 * plane + c], f7 = f[7 * plane + c], f8 = f[8 * plane + c];
        s->u[c] = (double)((f1 - f3 + f5 - f6 - f7 + f8) * inv);
        s->v[c] = (double)((f5 + f2 + f6 - f7 - f4 - f8) * inv);
    }
    for (int i = 0; i < nx; ++i) {
        s->u[(size_t)i * ny] = 0; s->u[(size_t)i * ny + ly] = 0;
        s->v[(size_t)i * ny] = 0; s->v[(size_t)i * ny + ly] = 0;
    }
    for (int j = 0; j < ny; ++j) {
        s->rho[j] = (float)s->rho_in;
        s->rho[(size_t)lx * ny + j] = (float)s->rho_out;
        float a = (F(0, 0, j) + F(2, 0, j) + F(4, 0, j)) + 2 * (F(3, 0, j) + F(6, 0, j) + F(7, 0, j));
        float b = (F(0, lx, j) + F(2, lx, j) + F(4, lx, j)) + 2 * (F(1, lx, j) + F(5, lx, j) + F(8, lx, j));
        if (s->numpy2) {
            s->u[j] = 1 - (double)a / s->rho_in;
            s->u[(size_t)lx * ny + j] = -1 + (double)b / s->rho_out;
        } else {
            s->u[j] = (double)(1 - a / (float)s->rho_in);
            s->u[(size_t)lx * ny + j] = (double)(-1 + b / (float)s->rho_out);
        }
    }
    if (s->mask)
        for (size_t c = 0; c < plane; ++c)
            if (s->mask[c]) { s->u[c] = 0; s->v[c] = 0; }
}

/* cython_dim.pyx:160-189 `update_feq` (Succi's expansion; float64 temporaries
 * because u,v are float64; `w*rho` is float32 because w is a Python float). */
void o1_update_feq(o1_state *s)
{
    const int nx = s->nx, ny = s->ny;
    const size_t plane = (size_t)nx * ny;
    const double cs = 1.0 / __builtin_sqrt(3.0);
    const double cs2 = cs * cs, cs22 = 2 * cs2, cssq = 2.0 / 9.0;
    const float w0 = (float)(4. / 9.), w1 = (float)(1. / 9.), w2 = (float)(1. / 36.);
    float *feq = s->feq;
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < plane; ++c) {
        const double u = s->u[c], v = s->v[c];
        const float rho = s->rho[c];
        const double ul = u / cs2, vl = v / cs2, uv = ul * vl;
        const double usq = u * u, vsq = v * v;
        const double sumsq = (usq + vsq) / cs22;
        const double sumsq2 = sumsq * (1. - cs2) / cs2;
        const double u2 = usq / cssq, v2 = vsq / cssq;
        const double r0 = (double)(w0 * rho), r1 = (double)(w1 * rho), r2 = (double)(w2 * rho);
        feq[c]             = (float)(r0 * (1. - sumsq));
        feq[plane + c]     = (float)(r1 * (1. - sumsq + u2 + ul));
        feq[2 * plane + c] = (float)(r1 * (1. - sumsq + v2 + vl));
        feq[3 * plane + c] = (float)(r1 * (1. - sumsq + u2 - ul));
        feq[4 * plane + c] = (float)(r1 * (1. - sumsq + v2 - vl));
        feq[5 * plane + c] = (float)(r2 * (1. + sumsq2 + ul + vl + uv));
        feq[6 * plane + c] = (float)(r2 * (1. + sumsq2 - ul + vl - uv));
        feq[7 * plane + c] = (float)(r2 * (1. + sumsq2 - ul - vl + uv));
        feq[8 * plane + c] = (float)(r2 * (1. + sumsq2 + ul - vl - uv));
    }
}

/* cython_dim.pyx:336-344 `collide_particles` */
void o1_collide(o1_state *s)
{
    const size_t n = 9u * (size_t)s->nx * s->ny;
    float *f = s->f;
    const float *feq = s->feq;
    if (s->numpy2) {
        const double om = s->omega, om1 = 1. - s->omega;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) f[i] = (float)((double)f[i] * om1 + om * (double)feq[i]);
    } else {
        const float om = (float)s->omega, om1 = (float)(1. - s->omega);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) f[i] = f[i] * om1 + om * feq[i];
    }
}

/* cython_dim.pyx:346-359 `run` */
void o1_run(o1_state *s, int n)
{
    for (int it = 0; it < n; ++it) {
        o1_move_bcs(s);
        o1_move(s);
        o1_update_hydro(s);
        o1_update_feq(s);
        o1_collide(s);
    }
}
#undef F
