// queue_order_repro.hip -- does HIP order kernels across streams of different priority by events alone when several
// processes share the GPU?  Stand-alone (no liblbhip): the question behind lb_run_group's rare mismatches under
// contention, reduced to its pattern.
//
// NB buffers, NS streams (argument 5, default 12: more than the runtime's hardware queues, as lb_run_group's three streams
// per member are; the first NS / 3 created with the device's highest priority when `prio` = 1).  Every operation
// picks a stream, a destination buffer and two source buffers at random, makes its stream wait for the event of the LAST
// ACCESS of each of the three (read or write: every buffer sees a total order of accesses, enforced by events only),
// launches one kernel -- it checks that both sources hold the value their last writer stored and then overwrites the
// destination -- and records the three events again.  Kernel sizes mimic the library's mix (halo pack: a few
// workgroups; edge bands: 64 waves; interior: the whole chip).  A source that does not hold the expected value = a kernel
// ran before one it was ordered behind.  Prints the number of such elements; exit status 1 if any.
//
//   hipcc --offload-arch=gfx950 -O2 tools/queue_order_repro.hip -o tools/_build/queue_order_repro
//   for i in 1 2 3 4 5; do tools/_build/queue_order_repro 20000 $i 1 & done; wait       # five processes, priorities on
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_op(int *dst, int val, const int *s0, int e0, const int *s1, int e1, int n, unsigned *bad)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (s0[i] != e0 || s1[i] != e1) atomicAdd(bad, 1u);
        dst[i] = val;
    }
}

int main(int argc, char **argv)
{
    const int ops = argc > 1 ? atoi(argv[1]) : 20000, seed = argc > 2 ? atoi(argv[2]) : 1, prio = argc > 3 ? atoi(argv[3]) : 1;
    const int sync_every = argc > 4 ? atoi(argv[4]) : 0;          // > 0: hipDeviceSynchronize every so many operations
    constexpr int NB = 8, NSMAX = 16, N = 1 << 18;
    const int NS = argc > 5 ? atoi(argv[5]) : 12, NHI = NS / 3;       // streams (lb_run_group: three per member, one of them high priority)
    int lo = 0, hi = 0, *buf[NB], val[NB] = {0};
    unsigned *bad, host_bad = 0;
    hipStream_t st[NSMAX];
    if (NS < 1 || NS > NSMAX) return 2;
    hipEvent_t ev[NB];
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    for (int s = 0; s < NS; ++s) CK(hipStreamCreateWithPriority(&st[s], hipStreamNonBlocking, (prio && s < NHI) ? hi : lo));
    CK(hipMalloc(&bad, sizeof(unsigned)));
    CK(hipMemset(bad, 0, sizeof(unsigned)));
    for (int b = 0; b < NB; ++b) {
        CK(hipMalloc(&buf[b], sizeof(int) * N));
        CK(hipMemset(buf[b], 0, sizeof(int) * N));
        CK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
    }
    CK(hipDeviceSynchronize());
    for (int b = 0; b < NB; ++b) CK(hipEventRecord(ev[b], st[0]));
    srand(seed);
    for (int t = 1; t <= ops; ++t) {
        const int s = rand() % NS, d = rand() % NB, a = (d + 1 + rand() % (NB - 1)) % NB;
        int b = (d + 1 + rand() % (NB - 1)) % NB;
        if (b == a) b = (a + 1) % NB == d ? (a + 2) % NB : (a + 1) % NB;
        static const int grids[4] = {8, 16, 64, 2048};                 // (grid-stride: every launch covers the whole buffer)
        const int g = grids[rand() % 4];
        for (int x : {d, a, b}) CK(hipStreamWaitEvent(st[s], ev[x], 0));
        hipLaunchKernelGGL(k_op, dim3(g), dim3(256), 0, st[s], buf[d], t, buf[a], val[a], buf[b], val[b], N, bad);
        CK(hipGetLastError());
        for (int x : {d, a, b}) CK(hipEventRecord(ev[x], st[s]));
        val[d] = t;
        if (sync_every && t % sync_every == 0) CK(hipDeviceSynchronize());
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&host_bad, bad, sizeof(unsigned), hipMemcpyDeviceToHost));
    printf("queue_order_repro: ops %d seed %d streams %d (%d high priority: %s) sync_every %d -> %u out-of-order element reads\n",
           ops, seed, NS, NHI, prio ? "on" : "off", sync_every, host_bad);
    return host_bad ? 1 : 0;
}
