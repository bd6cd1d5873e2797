// membench6.hip -- round-2 experiment (not product code): the fused kernel's traffic (per grid and step: actions 3 x 8 B, two
// series values, reward + soc 8 B each, done 1 B = 57 B) with two memory layouts, every launch on FRESH memory.
//   T  time-major rows (what ships): stream[k * N + i]; a wave reads 512 contiguous bytes of a row, the next step's 512 bytes
//      are a whole row (N * 8 B = 800 KB) further on
//   W  wave-blocked: stream[(wave * K_total + k) * 64 + lane]: a wave's 512-byte pieces of consecutive steps are contiguous
//      (32 KB per stream and wave for a 64-step launch); actions [(wave * K_total + k) * 64 + lane] * 3
// build: hipcc --offload-arch=gfx950 -O3 tools/membench6.hip -o tools/bin/membench6
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

template <int U, bool WAVE_BLOCKED>
__global__ __launch_bounds__(256) void walk(Streams s, long N, long n1, int K, long row0, long rows_total)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n1) return;
    const long wave = i >> 6, lane = i & 63;
    auto index = [&](long k) -> long {
        return WAVE_BLOCKED ? (wave * rows_total + row0 + k) * 64 + lane : (row0 + k) * N + i;
    };
    double ra[U][3], rl[U], rp[U];
    auto ld = [&](int u, long k) {
        const long off = index(k);
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
            const long off = index(k);
            s.o1[off] = r; s.o2[off] = c;
            s.ob[off] = (uint8_t)k;
        }
    }
}

template <bool WB>
static double run(const Streams &s, long N, int K, long rows_total, int launches, int streams)
{
    hipStream_t st[2];
    for (int j = 0; j < 2; j++) hipStreamCreate(&st[j]);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long half = ((N / streams) + 255) / 256 * 256;
    auto launch_all = [&](int first, int count) {
        for (int l = first; l < first + count; l++) {
            const long row0 = (long)(l % (rows_total / K)) * K;
            if (streams == 1) walk<4, WB><<<(unsigned)((N + 255) / 256), 256, 0, st[0]>>>(s, N, N, K, row0, rows_total);
            else {                                             // two halves of the grids on two streams, never joined
                // second half = grids [half, N): shift the pointers so that the kernel's i starts at 0
                Streams s2 = s;
                const long sh = WB ? (half / 64) * rows_total * 64 : half;
                s2.act += sh * 3; s2.ts1 += sh; s2.ts2 += sh; s2.o1 += sh; s2.o2 += sh; s2.ob += sh;
                walk<4, WB><<<(unsigned)(half / 256), 256, 0, st[0]>>>(s, N, half, K, row0, rows_total);
                walk<4, WB><<<(unsigned)((N - half + 255) / 256), 256, 0, st[1]>>>(s2, N, N - half, K, row0, rows_total);
            }
        }
    };
    launch_all(0, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, st[0]);
    launch_all(8, launches);
    hipEventRecord(e1, st[0]);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)N * K * 57.0 * launches / (ms * 1e-3) / 1e9;
}

int main(int argc, char **argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 100000;
    const int K = 64;
    const long rows_total = (N > 400000 ? 8L : 64L) * K;      // 64 (8 for very large N) launches' worth of fresh rows, then wrap
    const long Npad = (N + 63) / 64 * 64;
    const size_t elems = (size_t)Npad * rows_total;
    Streams s;
    double *act, *ts1, *ts2, *o1, *o2; uint8_t *ob;
    if (hipMalloc(&act, elems * 24) != hipSuccess || hipMalloc(&ts1, elems * 8) != hipSuccess || hipMalloc(&ts2, elems * 8) != hipSuccess ||
        hipMalloc(&o1, elems * 8) != hipSuccess || hipMalloc(&o2, elems * 8) != hipSuccess || hipMalloc(&ob, elems) != hipSuccess) {
        fprintf(stderr, "allocation of %.1f GB failed\n", elems * 57.0 / 1e9);
        return 1;
    }
    hipMemset(act, 0, elems * 24); hipMemset(ts1, 0, elems * 8); hipMemset(ts2, 0, elems * 8);
    s.act = act; s.ts1 = ts1; s.ts2 = ts2; s.o1 = o1; s.o2 = o2; s.ob = ob;
    for (int rep = 0; rep < 2; rep++) {
        printf("N = %ld, K = %d, fresh rows every launch (%.1f GB footprint)\n", N, K, elems * 57.0 / 1e9);
        printf("  time-major rows,  one stream : %7.1f GB/s\n", run<false>(s, N, K, rows_total, 48, 1));
        printf("  time-major rows,  two streams: %7.1f GB/s\n", run<false>(s, N, K, rows_total, 48, 2));
        printf("  wave-blocked,     one stream : %7.1f GB/s\n", run<true>(s, N, K, rows_total, 48, 1));
        printf("  wave-blocked,     two streams: %7.1f GB/s\n", run<true>(s, N, K, rows_total, 48, 2));
    }
    return 0;
}
