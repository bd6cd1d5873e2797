// membench5.hip -- round-2 experiment (not product code): does the fused kernel's stream pattern run faster when the
// prefetch is never drained?  Same traffic as step_k_kernel<genset+battery> (per grid and step: actions 3 x 8 B, two
// series values, reward + soc 8 B each, done 1 B = 57 B), every launch on FRESH memory (no Infinity-Cache reuse).
//   A  register ring with conditional refill (what round 1 shipped: hipcc drains with vmcnt(0) at the loop latch)
//   B  register ring, unconditional clamped refill (counted vmcnt waits), U = 4 / 8
//   C  wave-private LDS ring filled by LDS-DMA (global_load_lds_dwordx4, 16 B per lane, no VGPR destination),
//      explicit counted vmcnt, R = 4 / 8 steps in flight
//   2s the same kernels as two half-size launches on two streams
// build: hipcc --offload-arch=gfx950 -O3 tools/membench5.hip -o gpurun_out/membench5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct Streams {
    const double *act, *ts1, *ts2;
    double *o1, *o2;
    uint8_t *ob;
};

__device__ __forceinline__ double fake_step(double a0, double a1, double a2, double l, double p, double &c)
{
    const double r = a0 + a1 * a2 + l - p;
    c += r;
    return r;
}

// ---- A: conditional refill ------------------------------------------------------------------------------
template <int U, bool BYTE>
__global__ __launch_bounds__(256) void walk_a(Streams s, long N, long n0, long n1, int K, long row0)
{
    const long i = n0 + (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n1) return;
    double ra[U][3], rl[U], rp[U];
    auto ld = [&](int u, long k) {
        const long off = (row0 + k) * N + i;
        ra[u][0] = s.act[off * 3]; ra[u][1] = s.act[off * 3 + 1]; ra[u][2] = s.act[off * 3 + 2];
        rl[u] = s.ts1[off]; rp[u] = s.ts2[off];
    };
#pragma unroll
    for (int u = 0; u < U; u++) ld(u, u);
    double c = 0.0;
    for (int k0 = 0; k0 < K; k0 += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = k0 + u;
            const double r = fake_step(ra[u][0], ra[u][1], ra[u][2], rl[u], rp[u], c);
            if (k + U < K) ld(u, k + U);
            const long off = (row0 + k) * N + i;
            s.o1[off] = r; s.o2[off] = c;
            if (BYTE) s.ob[off] = (uint8_t)k;
        }
    }
}

// ---- B: unconditional clamped refill -----------------------------------------------------------------------
template <int U, bool BYTE>
__global__ __launch_bounds__(256) void walk_b(Streams s, long N, long n0, long n1, int K, long row0)
{
    const long i = n0 + (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n1) return;
    double ra[U][3], rl[U], rp[U];
    const double *pa = s.act + ((long)row0 * N + i) * 3;
    const double *p1 = s.ts1 + (long)row0 * N + i;
    const double *p2 = s.ts2 + (long)row0 * N + i;
    double *q1 = s.o1 + (long)row0 * N + i, *q2 = s.o2 + (long)row0 * N + i;
    uint8_t *qb = s.ob + (long)row0 * N + i;
#pragma unroll
    for (int u = 0; u < U; u++) {
        ra[u][0] = pa[(long)u * N * 3]; ra[u][1] = pa[(long)u * N * 3 + 1]; ra[u][2] = pa[(long)u * N * 3 + 2];
        rl[u] = p1[(long)u * N]; rp[u] = p2[(long)u * N];
    }
    double c = 0.0;
    for (int k0 = 0; k0 < K; k0 += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = k0 + u;
            const double r = fake_step(ra[u][0], ra[u][1], ra[u][2], rl[u], rp[u], c);
            const long kn = (k + U < K) ? (long)(k + U) : (long)(K - 1);        // clamped: always a legal row
            ra[u][0] = pa[kn * N * 3]; ra[u][1] = pa[kn * N * 3 + 1]; ra[u][2] = pa[kn * N * 3 + 2];
            rl[u] = p1[kn * N]; rp[u] = p2[kn * N];
            q1[(long)k * N] = r; q2[(long)k * N] = c;
            if (BYTE) qb[(long)k * N] = (uint8_t)k;
            __builtin_amdgcn_sched_barrier(0);          // keep the steps apart: each waits for ITS slot only (counted vmcnt)
        }
    }
}

// ---- C: wave-private LDS ring, LDS-DMA fills ---------------------------------------------------------------
// slot of one step and one wave: [actions 64 x 24 B = 1536][ts1 512][ts2 512] = 2560 B, filled by three
// global_load_lds_dwordx4 (lane -> 16 B): #1 actions bytes 0..1023, #2 lanes 0-31 actions bytes 1024..1535 and lanes
// 32-63 ts1, #3 lanes 0-31 ts2 (lanes 32-63 re-read ts2's first bytes into a pad).
constexpr int SLOT = 2560 + 512;       // + pad for the idle half of fill #3

__device__ __forceinline__ void glds16(const void *g, __attribute__((address_space(3))) void *l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, l, 16, 0, 0);
}

template <int R, bool BYTE>
__global__ __launch_bounds__(256) void walk_c(Streams s, long N, long n0, long n1, int K, long row0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long w0 = n0 + (long)blockIdx.x * 256 + wave * 64;     // first grid of the wave
    if (w0 >= n1) return;
    const long i = w0 + lane;
    const bool live = i < n1;
    unsigned char *ring = lds_raw + (size_t)wave * R * SLOT;
    // per-lane source pointers of the three fills at row 0 (bytes); a row is N*24 / N*8 bytes further
    const char *a_base = (const char *)(s.act + ((long)row0 * N + w0) * 3);
    const char *g1 = a_base + lane * 16;
    const char *g2 = lane < 32 ? a_base + 1024 + lane * 16 : (const char *)(s.ts1 + (long)row0 * N + w0) + (lane - 32) * 16;
    const char *g3 = (const char *)(s.ts2 + (long)row0 * N + w0) + (lane & 31) * 16;
    const long stride_a = N * 24, stride_t = N * 8;
    const long st2 = lane < 32 ? stride_a : stride_t;
    auto fill = [&](int slot, long k) {
        auto *dst = (__attribute__((address_space(3))) unsigned char *)(ring + slot * SLOT);
        glds16(g1 + k * stride_a, dst);
        glds16(g2 + k * st2, dst + 1024);
        glds16(g3 + k * stride_t, dst + 2048);
    };
#pragma unroll
    for (int u = 0; u < R; u++) fill(u, u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // prologue: drained once (its counts differ)
    double *q1 = s.o1 + (long)row0 * N + i, *q2 = s.o2 + (long)row0 * N + i;
    uint8_t *qb = s.ob + (long)row0 * N + i;
    double c = 0.0;
    constexpr int NST = 2 + (BYTE ? 1 : 0);
    for (int k0 = 0; k0 < K; k0 += R) {
#pragma unroll
        for (int u = 0; u < R; u++) {
            const int k = k0 + u;
            // issued after this slot's fill: the stores of that step, then R-1 steps of (3 fills + NST stores)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 1) * (3 + NST) + NST) : "memory");
            const unsigned char *sl = ring + u * SLOT;
            const double a0 = *(const double *)(sl + lane * 24);
            const double a1 = *(const double *)(sl + lane * 24 + 8);
            const double a2 = *(const double *)(sl + lane * 24 + 16);
            const double l = *(const double *)(sl + 1536 + lane * 8);
            const double p = *(const double *)(sl + 2048 + lane * 8);
            const double r = fake_step(a0, a1, a2, l, p, c);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slot has been read: it may be refilled
            const long kn = (k + R < K) ? (long)(k + R) : (long)(K - 1);
            fill(u, kn);
            if (live) {
                q1[(long)k * N] = r; q2[(long)k * N] = c;
                if (BYTE) qb[(long)k * N] = (uint8_t)k;
            } else {                                                     // keep the vmcnt arithmetic wave-uniform
                asm volatile("" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

typedef void (*kern_t)(Streams, long, long, long, int, long);

static void run(const char *name, kern_t kern, size_t lds, long N, int K, int reps, int nstreams, int bytes_per)
{
    const long rows = (long)K * reps;
    Streams s;
    hipMalloc((void **)&s.act, rows * N * 24); hipMalloc((void **)&s.ts1, rows * N * 8); hipMalloc((void **)&s.ts2, rows * N * 8);
    hipMalloc((void **)&s.o1, rows * N * 8); hipMalloc((void **)&s.o2, rows * N * 8); hipMalloc((void **)&s.ob, rows * N);
    hipMemset((void *)s.act, 0, rows * N * 24); hipMemset((void *)s.ts1, 0, rows * N * 8); hipMemset((void *)s.ts2, 0, rows * N * 8);
    hipStream_t st[4];
    for (int j = 0; j < nstreams; j++) hipStreamCreateWithFlags(&st[j], hipStreamNonBlocking);
    if (lds > 64 * 1024) hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0[4], e1[4];
    for (int j = 0; j < nstreams; j++) { hipEventCreate(&e0[j]); hipEventCreate(&e1[j]); }
    const long per = ((N / nstreams) + 255) / 256 * 256;
    double best = 0;
    for (int pass = 0; pass < 3; pass++) {
        hipDeviceSynchronize();
        for (int j = 0; j < nstreams; j++) hipEventRecord(e0[j], st[j]);
        for (int r = 0; r < reps; r++)
            for (int j = 0; j < nstreams; j++) {
                const long n0 = j * per, n1 = (j == nstreams - 1) ? N : (j + 1) * per;
                const int blocks = (int)((n1 - n0 + 255) / 256);
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st[j], s, N, n0, n1, K, (long)r * K);
            }
        float worst = 0;
        for (int j = 0; j < nstreams; j++) hipEventRecord(e1[j], st[j]);
        for (int j = 0; j < nstreams; j++) {
            hipEventSynchronize(e1[j]);
            float ms; hipEventElapsedTime(&ms, e0[j], e1[j]);
            if (ms > worst) worst = ms;
        }
        const double gbs = (double)N * K * bytes_per * reps / (worst * 1e-3) / 1e9;
        if (pass > 0 && gbs > best) best = gbs;
        if (pass == 2)
            printf("%-58s N=%7ld K=%d x%d stream(s): %7.1f GB/s  %6.1f us/round  (%s)\n", name, N, K, nstreams, best,
                   (double)N * K * bytes_per / best / 1e3, hipGetErrorString(hipGetLastError()));
    }
    hipFree((void *)s.act); hipFree((void *)s.ts1); hipFree((void *)s.ts2); hipFree(s.o1); hipFree(s.o2); hipFree(s.ob);
    for (int j = 0; j < nstreams; j++) hipStreamDestroy(st[j]);
}

int main(int argc, char **argv)
{
    const long sizes[] = {100000L, 131072L, 1000000L};
    for (long n : sizes) {
        const int reps = n > 500000 ? 6 : 40;
        for (int ns = 1; ns <= 2; ns++) {
            run("A ring 4, conditional refill, +byte", walk_a<4, true>, 0, n, 64, reps, ns, 57);
            run("A ring 4, conditional refill, no byte", walk_a<4, false>, 0, n, 64, reps, ns, 56);
            run("B ring 4, clamped refill, +byte", walk_b<4, true>, 0, n, 64, reps, ns, 57);
            run("B ring 8, clamped refill, +byte", walk_b<8, true>, 0, n, 64, reps, ns, 57);
            run("B ring 8, clamped refill, no byte", walk_b<8, false>, 0, n, 64, reps, ns, 56);
            run("C LDS-DMA ring 4, +byte", walk_c<4, true>, 4 * 4 * SLOT, n, 64, reps, ns, 57);
            run("C LDS-DMA ring 8, +byte", walk_c<8, true>, 4 * 8 * SLOT, n, 64, reps, ns, 57);
            run("C LDS-DMA ring 8, no byte", walk_c<8, false>, 4 * 8 * SLOT, n, 64, reps, ns, 56);
        }
    }
    return 0;
}
