// Streaming-copy yardstick for the HBM roofline (tools/copy_bench.hip; build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/copy_bench.hip -o /tmp/copy_bench && /tmp/copy_bench
// Copies 2.4 GB (one 8192^2 lattice) with 16-byte-per-lane accesses in several shapes and prints GB/s (read + written).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// grid-stride, U independent 16-byte loads per lane in flight, then U stores
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_u(const f4 *__restrict__ src, f4 *__restrict__ dst, long long n4)
{
    const long long tile = (long long)U * blockDim.x;
    for (long long base = (long long)blockIdx.x * tile; base < n4; base += (long long)gridDim.x * tile) {
        f4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const long long i = base + (long long)j * blockDim.x + threadIdx.x;
            if (i < n4) v[j] = NTL ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const long long i = base + (long long)j * blockDim.x + threadIdx.x;
            if (i < n4) { if (NTS) __builtin_nontemporal_store(v[j], dst + i); else dst[i] = v[j]; }
        }
    }
}

// 8 bytes per lane, one float2 per thread, no loop (would narrower marching strips -- 2 cells per lane -- still stream?)
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void copy_f2(const f2 *__restrict__ src, f2 *__restrict__ dst, long long n2)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n2) __builtin_nontemporal_store(src[i], dst + i);
}
// 9 planes x 512 B per wave (f2 per lane), rows dealt to blocks: the narrow strip's shape
__global__ __launch_bounds__(256) void copy_planes_f2(const f2 *__restrict__ src, f2 *__restrict__ dst, long long plane2, int row2, int rows)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= row2) return;
    for (int y = blockIdx.y; y < rows; y += gridDim.y) {
        f2 v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = src[k * plane2 + (long long)y * row2 + l];
#pragma unroll
        for (int k = 0; k < 9; ++k) __builtin_nontemporal_store(v[k], dst + k * plane2 + (long long)y * row2 + l);
    }
}

// the lattice step's shape without its arithmetic: a wave copies 1 KiB from each of 9 planes of one row, rows dealt to blocks
template <bool NTS>
__global__ __launch_bounds__(256) void copy_planes(const f4 *__restrict__ src, f4 *__restrict__ dst, long long plane4, int row4, int rows)
{
    const int lane4 = blockIdx.x * blockDim.x + threadIdx.x;     // float4 index inside the row
    if (lane4 >= row4) return;
    for (int y = blockIdx.y; y < rows; y += gridDim.y) {
        f4 v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = src[k * plane4 + (long long)y * row4 + lane4];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            f4 *p = dst + k * plane4 + (long long)y * row4 + lane4;
            if (NTS) __builtin_nontemporal_store(v[k], p); else *p = v[k];
        }
    }
}

// The marching kernels' access pattern without arithmetic: wave (strip s, segment g) walks its rows, 9 x 1 KiB loads then
// 9 x 1 KiB stores per row.  LAYOUT 0: planes are [row][8192] (a wave's consecutive rows lie 32 KiB apart, x-neighbour strips
// 1 KiB apart) -- the engine's layout;  LAYOUT 1: strip-major planes [strip][row][256] (a wave's rows are contiguous).
template <int LAYOUT>
__global__ __launch_bounds__(128) void march_copy(const f4 *__restrict__ src, f4 *__restrict__ dst, long long plane4, int n,
                                                  int strips, int seg_rows)
{
    const int item = blockIdx.x * 2 + threadIdx.y, lane = threadIdx.x;
    const int s = item % strips, g = item / strips;
    const int y0 = g * seg_rows, y1 = min(y0 + seg_rows, n);
    const long long row4 = n / 4;
    for (int y = y0; y < y1; ++y) {
        f4 v[9];
        // LAYOUT 2: [row][plane][8192] (the nine planes of a row interleaved: 18 fronts per segment become 2);
        // 3: [row][strip][plane][256] (a wave's nine 1 KiB accesses are one 9 KiB block); 4: [strip][row][plane][256]
        // (a wave's whole march is one contiguous stream)
        long long o, ks = plane4;
        if (LAYOUT == 0) o = (long long)y * row4 + s * 64 + lane;
        else if (LAYOUT == 1) o = ((long long)s * n + y) * 64 + lane;
        else if (LAYOUT == 2) { o = (long long)y * 9 * row4 + s * 64 + lane; ks = row4; }
        else if (LAYOUT == 3) { o = ((long long)y * strips + s) * 9 * 64 + lane; ks = 64; }
        else { o = ((long long)s * n + y) * 9 * 64 + lane; ks = 64; }
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = src[k * ks + o];
#pragma unroll
        for (int k = 0; k < 9; ++k) __builtin_nontemporal_store(v[k], dst + k * ks + o);
    }
}

// the same march with 128-cell strips (8 bytes per lane: what a two-cells-per-lane marching kernel would issue)
__global__ __launch_bounds__(256) void march_copy_f2(const f2 *__restrict__ src, f2 *__restrict__ dst, long long plane2, int n,
                                                     int strips, int seg_rows)
{
    const int item = blockIdx.x * 4 + threadIdx.y, lane = threadIdx.x;
    const int s = item % strips, g = item / strips;
    const int y0 = g * seg_rows, y1 = min(y0 + seg_rows, n);
    const long long row2 = n / 2;
    for (int y = y0; y < y1; ++y) {
        f2 v[9];
        const long long o = (long long)y * row2 + s * 64 + lane;
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = src[k * plane2 + o];
#pragma unroll
        for (int k = 0; k < 9; ++k) __builtin_nontemporal_store(v[k], dst + k * plane2 + o);
    }
}

// k_step's launch shape without its arithmetic: blocks of (64 * WX) x RY threads, a wave = 256 cells of one row, nine planes;
// DELAY dependent FMAs per loaded value between the loads and the stores (the collision's place)
template <int DELAY>
__global__ __launch_bounds__(256) void copy_step_shape(const f4 *__restrict__ src, f4 *__restrict__ dst, long long plane4, int row4, int rows)
{
    const int lane4 = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (lane4 >= row4 || y >= rows) return;
    f4 v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = src[k * plane4 + (long long)y * row4 + lane4];
    if (DELAY) {
        f4 acc = v[0];
        for (int i = 0; i < DELAY; ++i)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = acc * 1.0000001f + v[k];
        if (acc.x == 12345.678f) v[0] = acc;           // (never; keeps the loop)
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) __builtin_nontemporal_store(v[k], dst + k * plane4 + (long long)y * row4 + lane4);
}

// march_copy with other item -> (strip, segment) mappings.  MAP 0: a workgroup's two waves take x-adjacent strips of one segment
// (k_step4's); 1: the same strip, two vertically adjacent segments; 2: column-major (consecutive workgroups walk down one strip's
// segments); 3: as 0 with the 8 x 8 transposition of workgroup ids that keeps 8 consecutive items on one XCD (xcd_item)
template <int MAP>
__global__ __launch_bounds__(128) void march_copy_map(const f4 *__restrict__ src, f4 *__restrict__ dst, long long plane4, int n,
                                                      int strips, int segs, int seg_rows)
{
    int wg = blockIdx.x;
    if (MAP == 3) { const int blk = wg & ~63, i = wg & 63; if (blk + 64 <= (int)gridDim.x) wg = blk + (i & 7) * 8 + (i >> 3); }
    const int wy = threadIdx.y, lane = threadIdx.x;
    int s, g;
    if (MAP == 1) { s = wg % strips; g = 2 * (wg / strips) + wy; }
    else if (MAP == 2) { const int item = wg * 2 + wy; g = item % segs; s = item / segs; }
    else { const int item = wg * 2 + wy; s = item % strips; g = item / strips; }
    if (s >= strips || g >= segs) return;
    const int y0 = g * seg_rows, y1 = min(y0 + seg_rows, n);
    const long long row4 = n / 4;
    for (int y = y0; y < y1; ++y) {
        f4 v[9];
        const long long o = (long long)y * row4 + s * 64 + lane;
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = src[k * plane4 + o];
#pragma unroll
        for (int k = 0; k < 9; ++k) __builtin_nontemporal_store(v[k], dst + k * plane4 + o);
    }
}

template <typename F>
static double time_ms(F launch, int iters)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t / iters);
    }
    std::sort(ms.begin(), ms.end());
    return ms[2];
}

int main()
{
    const int n = 8192;
    const long long plane4 = (long long)n * n / 4, n4 = 9 * plane4;
    const size_t bytes = (size_t)n4 * 16;
    f4 *src, *dst;
    const size_t slack = 8u << 20;              // room for the destination-offset experiment below
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes + slack));
    CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes + slack));
    if (getenv("COPY_BENCH_OFFSETS")) {
        // does the distance between the lattice read and the lattice written matter (HBM channel / bank conflicts between
        // the read front and the write front)?  march_copy in the interleaved-row layout, destination shifted
        const int strips = n / 256, segs = 256 * 8 / strips, seg_rows = (n + segs - 1) / segs, items = strips * segs;
        printf("src %p dst %p (dst - src = %lld B)\n", (void *)src, (void *)dst, (long long)((char *)dst - (char *)src));
        for (size_t off : {(size_t)0, (size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)147456,
                           (size_t)262144, (size_t)1048576, (size_t)2097152 + 147456}) {
            f4 *d = dst + off / 16;
            double ms = time_ms([&] { hipLaunchKernelGGL(march_copy<2>, dim3(items / 2), dim3(64, 2), 0, 0, src, d, plane4, n, strips, seg_rows); }, 10);
            printf("march_copy [row][plane][x], destination + %8zu B: %7.1f us  %7.1f GB/s\n", off, ms * 1e3, 2.0 * bytes / 1e9 / (ms * 1e-3));
        }
        return 0;
    }
    const double gb = 2.0 * bytes / 1e9;
#define RUN(name, ...) do { double ms = time_ms([&] { __VA_ARGS__; }, 10); printf("%-58s %7.1f us  %7.1f GB/s\n", name, ms * 1e3, gb / (ms * 1e-3)); } while (0)
    for (int blocks : {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        char nm[128];
        snprintf(nm, sizeof nm, "copy_u<8,plain,plain>   grid %5d", blocks);
        RUN(nm, hipLaunchKernelGGL((copy_u<8, false, false>), dim3(blocks), dim3(256), 0, 0, src, dst, n4));
        snprintf(nm, sizeof nm, "copy_u<8,plain,nt>      grid %5d", blocks);
        RUN(nm, hipLaunchKernelGGL((copy_u<8, false, true>), dim3(blocks), dim3(256), 0, 0, src, dst, n4));
        snprintf(nm, sizeof nm, "copy_u<8,nt,nt>         grid %5d", blocks);
        RUN(nm, hipLaunchKernelGGL((copy_u<8, true, true>), dim3(blocks), dim3(256), 0, 0, src, dst, n4));
        snprintf(nm, sizeof nm, "copy_u<4,plain,nt>      grid %5d", blocks);
        RUN(nm, hipLaunchKernelGGL((copy_u<4, false, true>), dim3(blocks), dim3(256), 0, 0, src, dst, n4));
        snprintf(nm, sizeof nm, "copy_u<16,plain,nt>     grid %5d", blocks);
        RUN(nm, hipLaunchKernelGGL((copy_u<16, false, true>), dim3(blocks), dim3(256), 0, 0, src, dst, n4));
        snprintf(nm, sizeof nm, "copy_u<2,plain,nt>      grid %5d", blocks);
        RUN(nm, hipLaunchKernelGGL((copy_u<2, false, true>), dim3(blocks), dim3(256), 0, 0, src, dst, n4));
    }
    // one block per (256 float4 of a row, row): no grid-stride loop at all
    RUN("copy_u<1,plain,nt>      grid = n4/256 (no loop)", hipLaunchKernelGGL((copy_u<1, false, true>), dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, src, dst, n4));
    RUN("copy_u<4,plain,nt>      grid = n4/1024 (no loop)", hipLaunchKernelGGL((copy_u<4, false, true>), dim3((unsigned)(n4 / 1024)), dim3(256), 0, 0, src, dst, n4));
    for (int gy : {256, 1024, 8192}) {
        char nm[128];
        snprintf(nm, sizeof nm, "copy_planes<plain> (9 planes x 1 KiB per wave)  grid.y %4d", gy);
        RUN(nm, hipLaunchKernelGGL((copy_planes<false>), dim3(n / 4 / 256, gy), dim3(256), 0, 0, src, dst, plane4, n / 4, n));
        snprintf(nm, sizeof nm, "copy_planes<nt>    (9 planes x 1 KiB per wave)  grid.y %4d", gy);
        RUN(nm, hipLaunchKernelGGL((copy_planes<true>), dim3(n / 4 / 256, gy), dim3(256), 0, 0, src, dst, plane4, n / 4, n));
    }
    RUN("copy_f2 (8 B per lane, one per thread, no loop)", hipLaunchKernelGGL(copy_f2, dim3((unsigned)(n4 * 2 / 256)), dim3(256), 0, 0, (const f2 *)src, (f2 *)dst, n4 * 2));
    RUN("copy_planes_f2 (9 planes x 512 B per wave) grid.y 8192", hipLaunchKernelGGL(copy_planes_f2, dim3(n / 2 / 256, 8192), dim3(256), 0, 0, (const f2 *)src, (f2 *)dst, plane4 * 2, n / 2, n));
    for (int wpc : {4, 8, 16, 32}) {
        const int strips = n / 256, segs = 256 * wpc / strips, seg_rows = (n + segs - 1) / segs, items = strips * segs;
        char nm[128];
        snprintf(nm, sizeof nm, "march_copy row-major planes,   %2d waves/CU, %3d-row segments", wpc, seg_rows);
        RUN(nm, hipLaunchKernelGGL(march_copy<0>, dim3(items / 2), dim3(64, 2), 0, 0, src, dst, plane4, n, strips, seg_rows));
        snprintf(nm, sizeof nm, "march_copy strip-major planes, %2d waves/CU, %3d-row segments", wpc, seg_rows);
        RUN(nm, hipLaunchKernelGGL(march_copy<1>, dim3(items / 2), dim3(64, 2), 0, 0, src, dst, plane4, n, strips, seg_rows));
        snprintf(nm, sizeof nm, "march_copy [row][plane][x],         %2d waves/CU, %3d-row segments", wpc, seg_rows);
        RUN(nm, hipLaunchKernelGGL(march_copy<2>, dim3(items / 2), dim3(64, 2), 0, 0, src, dst, plane4, n, strips, seg_rows));
        snprintf(nm, sizeof nm, "march_copy [row][strip][plane][256], %2d waves/CU, %3d-row segments", wpc, seg_rows);
        RUN(nm, hipLaunchKernelGGL(march_copy<3>, dim3(items / 2), dim3(64, 2), 0, 0, src, dst, plane4, n, strips, seg_rows));
        snprintf(nm, sizeof nm, "march_copy [strip][row][plane][256], %2d waves/CU, %3d-row segments", wpc, seg_rows);
        RUN(nm, hipLaunchKernelGGL(march_copy<4>, dim3(items / 2), dim3(64, 2), 0, 0, src, dst, plane4, n, strips, seg_rows));
    }
    for (int wpc : {8, 12, 16, 24}) {
        const int strips = n / 128, segs = 256 * wpc / strips, seg_rows = (n + segs - 1) / segs, items = strips * segs;
        char nm[128];
        snprintf(nm, sizeof nm, "march_copy_f2 (128-cell strips), %2d waves/CU, %3d-row segments", wpc, seg_rows);
        RUN(nm, hipLaunchKernelGGL(march_copy_f2, dim3(items / 4), dim3(64, 4), 0, 0, (const f2 *)src, (f2 *)dst, plane4 * 2, n, strips, seg_rows));
    }
    for (int shape = 0; shape < 3; ++shape) {
        const dim3 blk = shape == 0 ? dim3(256, 1) : (shape == 1 ? dim3(128, 2) : dim3(64, 4));
        const dim3 grd(n / 4 / blk.x, n / blk.y);
        char nm[128];
        snprintf(nm, sizeof nm, "copy_step_shape block %3d x %d, no delay", blk.x, blk.y);
        RUN(nm, hipLaunchKernelGGL(copy_step_shape<0>, grd, blk, 0, 0, src, dst, plane4, n / 4, n));
        snprintf(nm, sizeof nm, "copy_step_shape block %3d x %d, 4 x 9 FMAs", blk.x, blk.y);
        RUN(nm, hipLaunchKernelGGL(copy_step_shape<4>, grd, blk, 0, 0, src, dst, plane4, n / 4, n));
        snprintf(nm, sizeof nm, "copy_step_shape block %3d x %d, 16 x 9 FMAs", blk.x, blk.y);
        RUN(nm, hipLaunchKernelGGL(copy_step_shape<16>, grd, blk, 0, 0, src, dst, plane4, n / 4, n));
    }
    {
        const int strips = n / 256, segs = 64, seg_rows = n / segs, items = strips * segs;
        RUN("march_copy_map 0: x-adjacent strips per workgroup (8 waves/CU)", hipLaunchKernelGGL(march_copy_map<0>, dim3(items / 2), dim3(64, 2), 0, 0, src, dst, plane4, n, strips, segs, seg_rows));
        RUN("march_copy_map 1: one strip, two segments per workgroup", hipLaunchKernelGGL(march_copy_map<1>, dim3(items / 2), dim3(64, 2), 0, 0, src, dst, plane4, n, strips, segs, seg_rows));
        RUN("march_copy_map 2: column-major items", hipLaunchKernelGGL(march_copy_map<2>, dim3(items / 2), dim3(64, 2), 0, 0, src, dst, plane4, n, strips, segs, seg_rows));
        RUN("march_copy_map 3: x-adjacent + XCD transposition", hipLaunchKernelGGL(march_copy_map<3>, dim3(items / 2), dim3(64, 2), 0, 0, src, dst, plane4, n, strips, segs, seg_rows));
    }
    // the same march with the residency of k_step4 (36 KiB of LDS per two-wave workgroup = 8 waves per CU) and MORE items
    // than wave slots: short-lived marching waves dispatched in row-major order = a band sweeping the grid
    for (int seg_rows : {128, 64, 32, 16, 8}) {
        const int strips = n / 256, segs = n / seg_rows, items = strips * segs;
        char nm[128];
        snprintf(nm, sizeof nm, "march_copy, 8 waves/CU resident, %5d items of %3d rows", items, seg_rows);
        RUN(nm, hipLaunchKernelGGL(march_copy<0>, dim3(items / 2), dim3(64, 2), 36 * 1024, 0, src, dst, plane4, n, strips, seg_rows));
    }
    RUN("hipMemcpyDtoD", CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)));
    return 0;
}
