// membench2.hip -- HBM-only (no Infinity-Cache reuse) comparison of two stream layouts for the fused step kernel:
//   A (shipped): actions [K,N,3] (3 x 8-B loads, stride 24 B) + load[K,N] + pv[K,N] -> reward[K,N], soc[K,N], done[K,N] u8
//   B (proposed): actions [K,N,3] + loadpv[K,N,2] (one 16-B load) -> {reward,soc}[K,N,2] (one 16-B store), done u8
// Every launch walks a fresh 64-row window of arrays that are REPS launches long.  (experiment, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <bool PAIR, int U>
__global__ __launch_bounds__(256) void walk(const double *__restrict__ act, const double *__restrict__ ts1,
                                            const double *__restrict__ ts2, double *__restrict__ o1, double *__restrict__ o2,
                                            uint8_t *__restrict__ ob, long N, int K, long row0)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    double ra[U][3], rl[U], rp[U];
    auto ld = [&](int u, long k) {
        const long off = (row0 + k) * N + i;
        ra[u][0] = act[off * 3]; ra[u][1] = act[off * 3 + 1]; ra[u][2] = act[off * 3 + 2];
        if (PAIR) { const double2 v = ((const double2 *)ts1)[off]; rl[u] = v.x; rp[u] = v.y; }
        else { rl[u] = ts1[off]; rp[u] = ts2[off]; }
    };
#pragma unroll
    for (int u = 0; u < U; u++) ld(u, u);
    double c = 0.0;
    for (int k0 = 0; k0 < K; k0 += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = k0 + u;
            const double r = ra[u][0] + ra[u][1] * ra[u][2] + rl[u] - rp[u];
            c += r;
            if (k + U < K) ld(u, k + U);
            const long off = (row0 + k) * N + i;
            if (PAIR) { ((double2 *)o1)[off] = make_double2(r, c); }
            else { o1[off] = r; o2[off] = c; }
            ob[off] = (uint8_t)k;
        }
    }
}

template <bool PAIR>
void run(long N, int K, int reps, const char *name)
{
    const long rows = (long)K * reps;
    double *act, *ts1, *ts2, *o1, *o2; uint8_t *ob;
    hipMalloc(&act, rows * N * 24); hipMalloc(&ts1, rows * N * (PAIR ? 16 : 8)); hipMalloc(&ts2, rows * N * 8);
    hipMalloc(&o1, rows * N * (PAIR ? 16 : 8)); hipMalloc(&o2, rows * N * 8); hipMalloc(&ob, rows * N);
    hipMemset(act, 0, rows * N * 24); hipMemset(ts1, 0, rows * N * (PAIR ? 16 : 8)); hipMemset(ts2, 0, rows * N * 8);
    const int blocks = (N + 255) / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; pass++) {
        hipEventRecord(e0);
        for (int r = 0; r < reps; r++) walk<PAIR, 4><<<blocks, 256>>>(act, ts1, ts2, o1, o2, ob, N, K, (long)r * K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (pass == 1)
            printf("%-34s N=%7ld K=%d: %7.1f GB/s  %6.1f us/launch\n", name, N, K, (double)N * K * 57.0 * reps / (ms * 1e-3) / 1e9,
                   ms * 1e3 / reps);
    }
    hipFree(act); hipFree(ts1); hipFree(ts2); hipFree(o1); hipFree(o2); hipFree(ob);
}

int main()
{
    for (long n : {100000L, 131072L, 1000000L}) {
        const int reps = n > 500000 ? 6 : 40;
        run<false>(n, 64, reps, "A: 3+1+1 loads, 2 stores + byte");
        run<true>(n, 64, reps, "B: 3+1(16B) loads, 1(16B) store + byte");
    }
    return 0;
}
