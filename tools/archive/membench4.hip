// membench4.hip -- time-blocked streams WITH lane-contiguous loads redistributed through LDS   (experiment)
// Layout: actions [K/B, N, B, 3], load / pv [K/B, N, B], reward / soc [K/B, N, B], done [K/B, N, B] u8, B = 4.
// A wave owns 64 grids: per block of B steps it pulls 6 KB + 2 KB + 2 KB of CONSECUTIVE bytes with 16-B loads
// (lane l takes bytes [16 l + 1024 j)), parks them in its private LDS region, and every lane reads back its own
// grid's B steps; results go the same way back.  Next block's loads are in flight (registers) during the compute.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int B = 4;
constexpr int NA = 3 * B * 8 * 64 / 1024;     // 1-KiB wave-loads per block for the actions: 6
constexpr int NT = B * 8 * 64 / 1024;         // for one series / one fp64 output: 2
constexpr int LDS_WAVE = (NA + 2 * NT) * 1024 + 2 * NT * 1024;   // inputs + outputs, bytes

__global__ __launch_bounds__(256) void walk(const double2 *__restrict__ act, const double2 *__restrict__ ts1,
                                            const double2 *__restrict__ ts2, double2 *__restrict__ o1, double2 *__restrict__ o2,
                                            unsigned *__restrict__ ob, long N, int KB, long blk0)
{
    extern __shared__ double2 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long g0 = ((long)blockIdx.x * 4 + wave) * 64;
    if (g0 >= N) return;                                   // N is a multiple of 64 in this experiment
    double2 *L = lds + wave * (LDS_WAVE / 16);
    double2 *La = L, *Ll = L + NA * 64, *Lp = Ll + NT * 64, *Lo1 = Lp + NT * 64, *Lo2 = Lo1 + NT * 64;
    double2 ra[NA], rl[NT], rp[NT];
    auto ld = [&](long kb) {
        const long b16 = (blk0 + kb) * N + g0;             // first grid of the wave in this block, in grids
#pragma unroll
        for (int j = 0; j < NA; j++) ra[j] = act[b16 * (3 * B / 2) + j * 64 + lane];
#pragma unroll
        for (int j = 0; j < NT; j++) { rl[j] = ts1[b16 * (B / 2) + j * 64 + lane]; rp[j] = ts2[b16 * (B / 2) + j * 64 + lane]; }
    };
    ld(0);
    double c = 0.0;
    for (int kb = 0; kb < KB; kb++) {
        // registers (linear order) -> LDS
#pragma unroll
        for (int j = 0; j < NA; j++) La[j * 64 + lane] = ra[j];
#pragma unroll
        for (int j = 0; j < NT; j++) { Ll[j * 64 + lane] = rl[j]; Lp[j * 64 + lane] = rp[j]; }
        if (kb + 1 < KB) ld(kb + 1);                        // next block in flight during the compute
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        // every lane reads its own grid's B steps
        double2 a[3 * B / 2], l[B / 2], p[B / 2], r[B / 2], q[B / 2];
#pragma unroll
        for (int j = 0; j < 3 * B / 2; j++) a[j] = La[lane * (3 * B / 2) + j];
#pragma unroll
        for (int j = 0; j < B / 2; j++) { l[j] = Ll[lane * (B / 2) + j]; p[j] = Lp[lane * (B / 2) + j]; }
#pragma unroll
        for (int j = 0; j < B / 2; j++) {
            r[j].x = a[3 * j].x + a[3 * j].y * a[3 * j + 1].x + l[j].x - p[j].x; c += r[j].x; q[j].x = c;
            r[j].y = a[3 * j + 1].y + a[3 * j + 2].x * a[3 * j + 2].y + l[j].y - p[j].y; c += r[j].y; q[j].y = c;
        }
#pragma unroll
        for (int j = 0; j < B / 2; j++) { Lo1[lane * (B / 2) + j] = r[j]; Lo2[lane * (B / 2) + j] = q[j]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        const long b16 = (blk0 + kb) * N + g0;
#pragma unroll
        for (int j = 0; j < NT; j++) { o1[b16 * (B / 2) + j * 64 + lane] = Lo1[j * 64 + lane]; o2[b16 * (B / 2) + j * 64 + lane] = Lo2[j * 64 + lane]; }
        ob[b16 + lane] = (unsigned)kb;                      // B = 4 done bytes per grid
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
}

int main()
{
    for (long N : {100032L, 131072L, 1000000L}) {
        const int K = 64, KB = K / B, reps = N > 500000 ? 6 : 40;
        const long blks = (long)KB * reps;
        double2 *act, *ts1, *ts2, *o1, *o2; unsigned *ob;
        hipMalloc(&act, blks * N * 24 * B); hipMalloc(&ts1, blks * N * 8 * B); hipMalloc(&ts2, blks * N * 8 * B);
        hipMalloc(&o1, blks * N * 8 * B); hipMalloc(&o2, blks * N * 8 * B); hipMalloc(&ob, blks * N * B);
        hipMemset(act, 0, blks * N * 24 * B); hipMemset(ts1, 0, blks * N * 8 * B); hipMemset(ts2, 0, blks * N * 8 * B);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int blocks = (N / 64 + 3) / 4;
        for (int pass = 0; pass < 2; pass++) {
            hipEventRecord(e0);
            for (int r = 0; r < reps; r++) walk<<<blocks, 256, 4 * LDS_WAVE>>>(act, ts1, ts2, o1, o2, ob, N, KB, (long)r * KB);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (pass == 1)
                printf("D: time-blocked by %d, lane-contiguous loads via LDS   N=%7ld K=%d: %7.1f GB/s  %6.1f us/launch  (%s)\n", B, N, K,
                       (double)N * K * 57.0 * reps / (ms * 1e-3) / 1e9, ms * 1e3 / reps, hipGetErrorString(hipGetLastError()));
        }
        hipFree(act); hipFree(ts1); hipFree(ts2); hipFree(o1); hipFree(o2); hipFree(ob);
    }
    return 0;
}
