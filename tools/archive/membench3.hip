// membench3.hip -- would a TIME-BLOCKED layout lift the fused kernel's HBM throughput?   (experiment)
//   C: actions [K/8, N, 8, 3], load / pv [K/8, N, 8], outputs reward / soc [K/8, N, 8], done [K/8, N, 8] u8:
//      every lane streams 8 consecutive steps of its grid with 16-byte loads; a wave covers contiguous 12 KB / 4 KB.
// Fresh memory every launch (HBM only), same bytes per step (57) as membench2's layouts A / B.
#include <hip/hip_runtime.h>
#include <cstdio>


template <int B>
__global__ __launch_bounds__(256) void walk(const double2 *__restrict__ act, const double2 *__restrict__ ts1,
                                            const double2 *__restrict__ ts2, double2 *__restrict__ o1, double2 *__restrict__ o2,
                                            uint2 *__restrict__ ob, long N, int KB, long blk0)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    constexpr int NA = 3 * B / 2, NT = B / 2;
    double2 a[2][NA], l[2][NT], p[2][NT];
    auto ld = [&](int s, long kb) {
        const long base = (blk0 + kb) * N + i;
#pragma unroll
        for (int j = 0; j < NA; j++) a[s][j] = act[base * NA + j];
#pragma unroll
        for (int j = 0; j < NT; j++) { l[s][j] = ts1[base * NT + j]; p[s][j] = ts2[base * NT + j]; }
    };
    ld(0, 0);
    double c = 0.0;
    for (int kb = 0; kb < KB; kb++) {
        const int s = kb & 1;
        if (kb + 1 < KB) ld(s ^ 1, kb + 1);
        const long base = (blk0 + kb) * N + i;
        double2 r[NT], q[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) {
            r[j].x = a[s][3 * j].x + a[s][3 * j].y * a[s][3 * j + 1].x + l[s][j].x - p[s][j].x; c += r[j].x; q[j].x = c;
            r[j].y = a[s][3 * j + 1].y + a[s][3 * j + 2].x * a[s][3 * j + 2].y + l[s][j].y - p[s][j].y; c += r[j].y; q[j].y = c;
        }
#pragma unroll
        for (int j = 0; j < NT; j++) { o1[base * NT + j] = r[j]; o2[base * NT + j] = q[j]; }
        if (B == 8) ob[base] = make_uint2((unsigned)kb, 0u);
        else if (B == 4) ((unsigned *)ob)[base] = (unsigned)kb;
        else ((unsigned short *)ob)[base] = (unsigned short)kb;
    }
}

template <int B>
void run()
{
    for (long N : {100000L, 131072L, 1000000L}) {
        const int K = 64, KB = K / B, reps = N > 500000 ? 6 : 40;
        const long blks = (long)KB * reps;
        double2 *act, *ts1, *ts2, *o1, *o2; uint2 *ob;
        hipMalloc(&act, blks * N * 24 * B); hipMalloc(&ts1, blks * N * 8 * B); hipMalloc(&ts2, blks * N * 8 * B);
        hipMalloc(&o1, blks * N * 8 * B); hipMalloc(&o2, blks * N * 8 * B); hipMalloc(&ob, blks * N * B);
        hipMemset(act, 0, blks * N * 24 * B); hipMemset(ts1, 0, blks * N * 8 * B); hipMemset(ts2, 0, blks * N * 8 * B);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int blocks = (N + 255) / 256;
        for (int pass = 0; pass < 2; pass++) {
            hipEventRecord(e0);
            for (int r = 0; r < reps; r++) walk<B><<<blocks, 256>>>(act, ts1, ts2, o1, o2, ob, N, KB, (long)r * KB);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (pass == 1)
                printf("C: time-blocked by %d steps, 16-B loads   N=%7ld K=%d: %7.1f GB/s  %6.1f us/launch\n", B, N, K,
                       (double)N * K * 57.0 * reps / (ms * 1e-3) / 1e9, ms * 1e3 / reps);
        }
        hipFree(act); hipFree(ts1); hipFree(ts2); hipFree(o1); hipFree(o2); hipFree(ob);
    }
}

int main()
{
    run<2>(); run<4>(); run<8>();
    return 0;
}
