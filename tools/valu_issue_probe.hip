// valu_issue_probe.hip -- how fast ONE wave per SIMD issues vector-ALU instructions on gfx950, against two (round 5).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
// Per (waves per SIMD, independent chains): cycles of the shader clock (s_memtime) per v_pk_fma_f32 and per v_fma_f32 as seen by one
// wave, and the shader clock itself (s_memtime against the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int CHAINS, bool PK>
__global__ __launch_bounds__(64) void k_chain(float *out, unsigned long long *clk, int iters, int pad_lds)
{
    extern __shared__ float lds[];
    f2 x[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = f2{1.f + threadIdx.x * 1e-3f + c, 2.f};
    const f2 a = f2{0.999f, 1.001f}, b = f2{1e-3f, -1e-3f};
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 64 / CHAINS; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c].x) : "v"(a.x), "v"(b.x));
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += x[c].x + x[c].y;
    if (pad_lds < 0) lds[threadIdx.x] = s;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int CHAINS, bool PK>
void run(int waves_per_simd, float *out, unsigned long long *clk, unsigned long long *h)
{
    const int iters = 20000, n_wg = 256 * 4 * waves_per_simd;
    // (LDS padding so that exactly waves_per_simd workgroups of one wave fit a SIMD's share: 160 KB / (4 * wps))
    const int lds = waves_per_simd == 1 ? 40 * 1024 : 20 * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain<CHAINS, PK>), dim3(n_wg), dim3(64), lds, 0, out, clk, 100, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_chain<CHAINS, PK>), dim3(n_wg), dim3(64), lds, 0, out, clk, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * n_wg, hipMemcpyDeviceToHost);
    double ct = 0, cr = 0;
    for (int i = 0; i < n_wg; ++i) { ct += h[2 * i]; cr += h[2 * i + 1]; }
    ct /= n_wg; cr /= n_wg;
    const double n_inst = (double)iters * 64;
    printf("%s chains=%d waves/SIMD=%d: %.2f s_memtime ticks per instruction per wave; launch %.3f ms; s_memtime %.1f MHz (vs the 100 MHz clock)\n",
           PK ? "v_pk_fma_f32" : "v_fma_f32   ", CHAINS, waves_per_simd, ct / n_inst, ms, ct / cr * 100.0);
}

int main()
{
    float *out; unsigned long long *clk;
    hipMalloc(&out, 4 * 64 * 4096); hipMalloc(&clk, 16 * 4096);
    unsigned long long *h = (unsigned long long *)malloc(16 * 4096);
    for (int w = 1; w <= 2; ++w) {
        run<1, true>(w, out, clk, h); run<2, true>(w, out, clk, h); run<4, true>(w, out, clk, h); run<8, true>(w, out, clk, h);
        run<1, false>(w, out, clk, h); run<2, false>(w, out, clk, h); run<8, false>(w, out, clk, h);
    }
    return 0;
}
