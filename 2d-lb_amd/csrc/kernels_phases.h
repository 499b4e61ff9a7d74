// kernels_phases.h -- one kernel per reference kernel (un-fused, API / test parity) and the halo
// pack / unpack kernels of the row-slab decomposition.  Included by lb_hip.cpp only.
#pragma once
#include "d2q9_cell.h"

namespace {

// ---- un-fused kernels: the reference's phases one by one (API / test parity) ----------------
struct PhaseArgs {
    float *f, *fs, *feq;   // plane 0, row 0, x 0
    float *rho, *u, *v;
    const uint8_t *mask;
    long long plane;
    int pitch, nx, ny, bc;
    float omega, rho_in, rho_out, lid_u, rho0;
};

__device__ __constant__ int d_cx[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
__device__ __constant__ int d_cy[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
__device__ __constant__ float d_w[9] = {4.f / 9.f,  1.f / 9.f,  1.f / 9.f,  1.f / 9.f, 1.f / 9.f,
                                        1.f / 36.f, 1.f / 36.f, 1.f / 36.f, 1.f / 36.f};

// D2Q9.cl:139-171 in pull form: a cell whose upstream neighbour is outside the box keeps the
// stale content of f_streamed, exactly like the reference's dropped push.
__global__ void k_move(const PhaseArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= a.nx) return;
    int sx = x - d_cx[k], sy = y - d_cy[k];
    if (a.bc == LB_BC_PERIODIC) {
        sx = (sx + a.nx) % a.nx;
        sy = (sy + a.ny) % a.ny;
    }
    if (sx < 0 || sx >= a.nx || sy < 0 || sy >= a.ny) return;
    a.fs[k * a.plane + (long long)y * a.pitch + x] = a.f[k * a.plane + (long long)sy * a.pitch + sx];
}

__global__ void k_bcs(const PhaseArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x, S = a.plane;
    float *f = a.f + o;
    Cell c = {f[0], f[S], f[2 * S], f[3 * S], f[4 * S], f[5 * S], f[6 * S], f[7 * S], f[8 * S]};
    if (x == 0 || x == a.nx - 1 || y == 0 || y == a.ny - 1) {
        if (a.bc == LB_BC_PIPE) bc_pipe_cell(c, x, y, a.nx, a.ny, a.rho_in, a.rho_out);
        if (a.bc == LB_BC_CAVITY) bc_cavity_cell(c, x, y, a.nx, a.ny, a.lid_u, a.rho0);
    }
    bounce_cell(c, a.mask && a.mask[o]);
    f[S] = c.f1; f[2 * S] = c.f2; f[3 * S] = c.f3; f[4 * S] = c.f4;
    f[5 * S] = c.f5; f[6 * S] = c.f6; f[7 * S] = c.f7; f[8 * S] = c.f8;
}

__global__ void k_hydro(const PhaseArgs a)   // D2Q9.cl:67-100
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = a.f[k * a.plane + o];
    const float rho = f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7] + f[8];
    const float inv = 1.0f / rho;
    a.rho[o] = rho;
    a.u[o] = (f[1] - f[3] + f[5] - f[6] - f[7] + f[8]) * inv;
    a.v[o] = (f[5] + f[2] + f[6] - f[7] - f[4] - f[8]) * inv;
}

__global__ void k_feq(const PhaseArgs a)     // D2Q9.cl:2-64
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x;
    const float rho = a.rho[o], ux = a.u[o], uy = a.v[o];
    const float usq = ux * ux + uy * uy;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float cu = d_cx[k] * ux + d_cy[k] * uy;
        a.feq[k * a.plane + o] = d_w[k] * rho * (1.f + 3.f * cu + 4.5f * cu * cu - 1.5f * usq);
    }
}

__global__ void k_collide(const PhaseArgs a) // D2Q9.cl:102-121
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= a.nx) return;
    const long long o = k * a.plane + (long long)y * a.pitch + x;
    a.f[o] = a.f[o] * (1.f - a.omega) + a.omega * a.feq[o];
}

__global__ void k_zero_vel(const PhaseArgs a) // D2Q9.cl:377-396
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.nx) return;
    const long long o = (long long)y * a.pitch + x;
    if (a.mask[o]) { a.u[o] = 0.f; a.v[o] = 0.f; }
}

// Halo pack / unpack: the 3-deep halo of one edge is 18 row segments scattered over the planes
// (HaloSeg tables on the host side).  One tiny kernel gathers both edges into two contiguous buffers
// (so that an exchange is one send + one receive per neighbour instead of eighteen), one scatters the
// received buffers into the ghost rows.  Table t: 0 north-out, 1 south-out, 2 south-in, 3 north-in;
// rows of the north tables count from row H.
__device__ __constant__ int d_halo_k[4][18] = {
    {2, 5, 6, 0, 1, 3, 2, 5, 6, 0, 1, 2, 3, 4, 5, 6, 7, 8}, {0, 1, 2, 3, 4, 5, 6, 7, 8, 0, 1, 3, 4, 7, 8, 4, 7, 8},
    {2, 5, 6, 0, 1, 3, 2, 5, 6, 0, 1, 2, 3, 4, 5, 6, 7, 8}, {0, 1, 2, 3, 4, 5, 6, 7, 8, 0, 1, 3, 4, 7, 8, 4, 7, 8}};
__device__ __constant__ int d_halo_row[4][18] = {
    {-3, -3, -3, -2, -2, -2, -2, -2, -2, -1, -1, -1, -1, -1, -1, -1, -1, -1},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2},
    {-3, -3, -3, -2, -2, -2, -2, -2, -2, -1, -1, -1, -1, -1, -1, -1, -1, -1},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2}};

__global__ void k_halo_pack(const float *origin, long long plane, int pitch, int h, int nx, float *buf_n, float *buf_s)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, seg = blockIdx.y, north = (blockIdx.z == 0);
    if (x >= nx) return;
    float *buf = north ? buf_n : buf_s;
    if (!buf) return;
    const int t = north ? 0 : 1;
    const long long row = (north ? h : 0) + d_halo_row[t][seg];
    buf[(long long)seg * nx + x] = origin[d_halo_k[t][seg] * plane + row * pitch + x];
}

__global__ void k_halo_unpack(float *origin, long long plane, int pitch, int h, int nx, const float *buf_s, const float *buf_n)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, seg = blockIdx.y, north = (blockIdx.z == 0);
    if (x >= nx) return;
    const float *buf = north ? buf_n : buf_s;
    if (!buf) return;
    const int t = north ? 3 : 2;
    const long long row = (north ? h : 0) + d_halo_row[t][seg];
    origin[d_halo_k[t][seg] * plane + row * pitch + x] = buf[(long long)seg * nx + x];
}

}  // namespace
