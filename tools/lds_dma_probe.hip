// lds_dma_probe.hip -- what `buffer_load_dwordx4 ... lds` / `global_load_lds_dword` do on gfx950 (round 6, for k_deep2's gather ahead):
// where a lane's 16 bytes land (M0 base, + lane * 16 ?), whether the instruction offset moves the LDS address too, whether an M0 base
// beyond 64 KB works, whether exec-masked lanes stay untouched, and that `s_waitcnt vmcnt(0)` is the wait.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4a __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u4v rsrc_of(const float *p)
{
    const unsigned long long v = (unsigned long long)p;
    return u4v{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)), 0xffffffffu, 0x00020000u};
}

// mode 0: x4 load, no instruction offset, M0 = base;  1: instruction offset 4 (does LDS move?);  2: M0 base at 96 KB;
// 3: dword load, only lanes 5 and 60 active;  4: global_load_lds_dwordx4
__global__ __launch_bounds__(64) void k(const float *src, float *out, int mode)
{
    extern __shared__ f4a lds[];                        // 112 KB
    const int lane = threadIdx.x;
    for (int i = lane; i < 112 * 64; i += 64) lds[i] = f4a{-1.f, -1.f, -1.f, -1.f};
    __syncthreads();
    const u4v r = rsrc_of(src);
    const unsigned base = mode == 2 ? 96u * 1024u : 2048u;     // bytes into this workgroup's LDS
    const int vo = lane * 16;
    if (mode == 0 || mode == 2)
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(vo), "s"(r), "s"(base) : "memory", "m0");
    else if (mode == 1)
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:4 lds" :: "v"(vo), "s"(r), "s"(base) : "memory", "m0");
    else if (mode == 3) {
        if (lane == 5 || lane == 60)
            asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" :: "v"(src + 1000 + lane), "s"(base) : "memory", "m0");
    } else
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src + lane * 4), "s"(base) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // dump 4 KB around the base (and the first 4 KB of LDS for mode 2)
    const f4a *p = reinterpret_cast<const f4a *>(reinterpret_cast<const char *>(lds) + base - 1024);
    for (int i = lane; i < 256; i += 64) reinterpret_cast<f4a *>(out)[i] = p[i];
    for (int i = lane; i < 256; i += 64) reinterpret_cast<f4a *>(out)[256 + i] = lds[i];
}

int main()
{
    float *src, *out;
    hipMalloc(&src, 1 << 20); hipMalloc(&out, 8192);
    float *h = (float *)malloc(1 << 20), *o = (float *)malloc(8192);
    for (int i = 0; i < (1 << 18); ++i) h[i] = (float)i;
    hipMemcpy(src, h, 1 << 20, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 112 * 1024, 0, src, out, mode);
        if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d: launch failed: %s\n", mode, hipGetErrorString(hipGetLastError())); continue; }
        hipMemcpy(o, out, 8192, hipMemcpyDeviceToHost);
        // where did data land?  report every float != -1 in the window [base - 1 KB, base + 3 KB) as (byte offset from base: value)
        printf("mode %d:", mode);
        int n = 0, first = -1, last = -1;
        for (int i = 0; i < 1024; ++i)
            if (o[i] != -1.f) { if (first < 0) first = i; last = i; ++n; }
        printf(" %d floats written, bytes [%d, %d] relative to the M0 base;", n, first < 0 ? 0 : first * 4 - 1024, last < 0 ? 0 : last * 4 - 1024 + 3);
        if (first >= 0) {
            printf(" first 8:");
            for (int i = first; i < first + 8 && i < 1024; ++i) printf(" %g", o[i]);
            printf(" ... lane 1's:");
            for (int i = 256 + 4; i < 256 + 8; ++i) printf(" %g", o[i]);
        }
        int low = 0;
        for (int i = 1024; i < 2048; ++i) low += o[i] != -1.f;
        printf("; first 4 KB of LDS: %d floats written\n", low);
    }
    return 0;
}
