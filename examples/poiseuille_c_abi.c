/* The C ABI without Python: a 2D Poiseuille pipe through include/lb_hip.h only.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/poiseuille_c_abi.c -L2d-lb_amd/LB_D2Q9 -llbhip \
 *       -Wl,-rpath,$PWD/2d-lb_amd/LB_D2Q9 -lm -o /tmp/poiseuille_c_abi && /tmp/poiseuille_c_abi
 *
 * What the reference's host does through pyopencl (opencl_dim.py:258-327 init_hydro / update_feq / init_pop, :372-387 run,
 * :390-415 get_fields), written against the entry points that replace those calls: density ramp from the inlet to the outlet,
 * fluid at rest, f = feq, n steps, read rho / u / v back.  Checked against the steady plane-Poiseuille profile
 *   u(y) = G / (2 nu) y (D - y),   G = cs^2 (rho_in - rho_out) / (nx - 1),  D = ny - 1,  nu = (1/omega - 1/2) / 3.
 * Exit status 0 = within 2 % of the parabola's peak (the lattice solution carries its own compressibility error). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "lb_hip.h"

#define TRY(call)                                                                    \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != LB_OK) {                                                          \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, lb_last_error());          \
            return 2;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv)
{
    const int nx = 128, ny = 33, steps = argc > 1 ? atoi(argv[1]) : 6000;
    const float omega = 1.0f, rho_in = 1.002f, rho_out = 1.0f;
    if (lb_abi_version() != LB_ABI_VERSION) {
        fprintf(stderr, "header ABI %d, library ABI %d\n", LB_ABI_VERSION, lb_abi_version());
        return 2;
    }
    if (lb_device_count() < 1) {
        fprintf(stderr, "no GPU: %s\n", lb_last_error());
        return 3;
    }
    lb_params p = {0};
    p.nx = nx; p.ny = ny; p.y0 = 0; p.local_ny = ny;
    p.bc_mode = LB_BC_PIPE; p.device = 0;
    p.omega = omega; p.inlet_rho = rho_in; p.outlet_rho = rho_out; p.rho0 = 1.0f;
    p.semantics = LB_SEM_OPENCL;
    lb_sim *sim = NULL;
    TRY(lb_create(&p, &sim));

    const size_t n = (size_t)nx * ny;
    float *rho = malloc(n * sizeof(float)), *u = calloc(n, sizeof(float)), *v = calloc(n, sizeof(float));
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x)            /* device order: [y][x]; the ramp of opencl_dim.py:266-283 */
            rho[(size_t)y * nx + x] = rho_in - x * (rho_in - rho_out) / nx;
    TRY(lb_set_macro(sim, rho, u, v));
    TRY(lb_update_feq(sim));
    TRY(lb_init_pop(sim));
    TRY(lb_run(sim, steps));
    TRY(lb_sync(sim));
    TRY(lb_get_macro(sim, rho, u, v));

    const double nu = (1.0 / omega - 0.5) / 3.0, D = ny - 1, G = (1.0 / 3.0) * (rho_in - rho_out) / (nx - 1);
    const int xm = nx / 2;
    double worst = 0.0, peak = G / (2 * nu) * (D / 2) * (D / 2);
    for (int y = 0; y < ny; ++y) {
        const double want = G / (2 * nu) * y * (D - y), got = u[(size_t)y * nx + xm];
        if (fabs(got - want) > worst) worst = fabs(got - want);
    }
    char kernel[160];
    TRY(lb_hot_kernel(sim, kernel, (int)sizeof kernel));
    printf("%d x %d pipe, %d steps on %s\n", nx, ny, steps, kernel);
    printf("u(centre) = %.6e, parabola peak %.6e, max |u - parabola| at x = %d: %.3e (%.2f %% of the peak)\n",
           u[(size_t)(ny / 2) * nx + xm], peak, xm, worst, 100.0 * worst / peak);
    TRY(lb_destroy(sim));
    free(rho); free(u); free(v);
    return worst <= 0.02 * peak ? 0 : 1;
}
