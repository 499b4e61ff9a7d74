// resident_handshake_probe.hip -- what a neighbour handshake between persistent workgroups costs on MI355X (round 5).
//   hipcc --offload-arch=gfx950 -O3 tools/resident_handshake_probe.hip -o /tmp/hs_probe && /tmp/hs_probe
// A lattice-resident kernel for 1024^2 (BASELINE config 2) would keep a 64 x 64 tile per CU in registers for a whole run() and
// exchange only the tile skirts through L2, meeting its four neighbours at device-side sequence flags every K steps.  This probe
// runs that skeleton without the physics: 256 (or 512) persistent workgroups in a 16 x 16 (32 x 16) torus; per round every
// workgroup stores `bytes` of skirt to a global buffer, publishes its round number (release, agent scope), waits until its four
// neighbours have published theirs (acquire) and reads their skirts.  Prints microseconds per round = the floor under K time steps.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void k_rounds(unsigned *flags, float *skirts, int gx, int gy, int rounds, int floats_per_edge,
                                                unsigned long long *ticks)
{
    const int wg = blockIdx.x, x = wg % gx, y = wg / gx;
    const int nb[4] = {((x + 1) % gx) + y * gx, ((x + gx - 1) % gx) + y * gx, x + ((y + 1) % gy) * gx, x + ((y + gy - 1) % gy) * gx};
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds; ++r) {
        // my skirt of this round (double-buffered by the round's parity: a neighbour may still be reading the previous one)
        float *mine = skirts + ((size_t)(r & 1) * gridDim.x + wg) * 4 * floats_per_edge;
        for (int i = threadIdx.x; i < 4 * floats_per_edge; i += blockDim.x) mine[i] = (float)(r + i) + acc * 1e-9f;
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&flags[wg * 32], (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < 4) {
            while (__hip_atomic_load(&flags[nb[threadIdx.x] * 32], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r)
                __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        for (int e = 0; e < 4; ++e) {
            const float *theirs = skirts + ((size_t)(r & 1) * gridDim.x + nb[e]) * 4 * floats_per_edge + e * floats_per_edge;
            for (int i = threadIdx.x; i < floats_per_edge; i += blockDim.x) acc += __builtin_nontemporal_load(theirs + i);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) ticks[wg] = t1 - t0;
    if (acc == 12345.678f) flags[0] = 0;        // (keeps acc alive)
}

int main()
{
    int dev_cus = 256;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) == hipSuccess) dev_cus = p.multiProcessorCount;
    for (int wgs : {256, 512}) {
        const int gx = wgs == 256 ? 16 : 32, gy = 16;
        for (int floats_per_edge : {0, 3 * 64, 3 * 64 * 4}) {         // no payload; 3 links x 64 cells; x 4 rows (K = 4)
            unsigned *flags; float *skirts; unsigned long long *ticks;
            hipMalloc(&flags, wgs * 32 * sizeof(unsigned)); hipMemset(flags, 0, wgs * 32 * sizeof(unsigned));
            hipMalloc(&skirts, (size_t)2 * wgs * 4 * (floats_per_edge + 1) * sizeof(float));
            hipMalloc(&ticks, wgs * sizeof(unsigned long long));
            const int rounds = 2000;
            void *args[] = {&flags, &skirts, (void *)&gx, (void *)&gy, (void *)&rounds, (void *)&floats_per_edge, &ticks};
            // (a cooperative launch guarantees co-residency: a plain one of <= CUs x 2 workgroups of 256 threads is co-resident as well)
            hipError_t e = hipLaunchCooperativeKernel((const void *)k_rounds, dim3(wgs), dim3(256), args, 0, 0);
            if (e != hipSuccess) { printf("cooperative launch refused (%s)\n", hipGetErrorString(e)); return 1; }
            e = hipDeviceSynchronize();
            unsigned long long *h = (unsigned long long *)malloc(wgs * sizeof(unsigned long long));
            hipMemcpy(h, ticks, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            unsigned long long mx = 0;
            for (int i = 0; i < wgs; ++i) mx = h[i] > mx ? h[i] : mx;
            printf("%3d workgroups (%d CUs), %5d B per edge: %s, %.2f us per round\n", wgs, dev_cus, floats_per_edge * 4,
                   hipGetErrorString(e), mx / 100.0 / rounds);
            free(h); hipFree(flags); hipFree(skirts); hipFree(ticks);
        }
    }
    return 0;
}
