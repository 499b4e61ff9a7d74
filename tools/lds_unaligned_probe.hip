// lds_unaligned_probe.hip -- does gfx950 serve a ds_read_b128 whose address is only dword-aligned?  (round 5: the marching kernels'
// stage windows are rows of cells in LDS; a shifted read would replace three moves and a DPP per shifted link.)
//   hipcc --offload-arch=gfx950 -O3 tools/lds_unaligned_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4a __attribute__((ext_vector_type(4)));
__global__ void k(float *out, int shift_bytes)
{
    __shared__ float lds[64 * 4 + 64];
    for (int i = threadIdx.x; i < 64 * 4 + 64; i += 64) lds[i] = (float)i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(&lds[16]) + threadIdx.x * 16 + shift_bytes;     // (LDS addresses are 32-bit offsets)
    f4a v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main()
{
    float *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int shift : {0, 4, -4, 8, 12}) {
        hipMemset(d, 0, sizeof(h));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, shift);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += (h[i] != (float)(16 + i + shift / 4));
        printf("shift %+d bytes: %s, %d of 256 elements wrong (lane 1 got %g %g %g %g, expected %d..)\n", shift, hipGetErrorString(e), bad,
               h[4], h[5], h[6], h[7], 20 + shift / 4);
    }
    return 0;
}
