// membench.hip -- what does MI355X HBM deliver for THIS kernel's access pattern?  (experiment, not product code)
// Each lane walks down K rows of NS read streams [K, N] and writes NW streams [K, N] (8 or 16 bytes per lane and
// row), optionally plus a 1-byte-per-element stream -- the shape of step_k_kernel's traffic.
// build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o gpurun_out/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename V, int NS, int NW, bool BYTE, int U>
__global__ __launch_bounds__(256) void walk(const V *__restrict__ in, V *__restrict__ out, uint8_t *__restrict__ bout,
                                            long N, int K)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    V ring[U][NS];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int s = 0; s < NS; s++) ring[u][s] = in[((long)s * K + u) * N + i];
    for (int k0 = 0; k0 < K; k0 += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = k0 + u;
            V acc = ring[u][0];
#pragma unroll
            for (int s = 1; s < NS; s++) acc += ring[u][s];
            if (k + U < K) {
#pragma unroll
                for (int s = 0; s < NS; s++) ring[u][s] = in[((long)s * K + k + U) * N + i];
            }
#pragma unroll
            for (int w = 0; w < NW; w++) out[((long)w * K + k) * N + i] = acc;
            if (BYTE) bout[(long)k * N * (sizeof(V) / 8) + i * (sizeof(V) / 8)] = (uint8_t)k;
        }
    }
}

template <typename V, int NS, int NW, bool BYTE, int U>
double run(long Ngrids, int K, const char *name)
{
    const long N = Ngrids / (sizeof(V) / 8);       // lanes
    V *in, *out; uint8_t *b;
    hipMalloc(&in, sizeof(V) * N * K * NS);
    hipMalloc(&out, sizeof(V) * N * K * (NW ? NW : 1));
    hipMalloc(&b, Ngrids * K);
    hipMemset(in, 0, sizeof(V) * N * K * NS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = (N + 255) / 256;
    walk<V, NS, NW, BYTE, U><<<blocks, 256>>>(in, out, b, N, K);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; r++) walk<V, NS, NW, BYTE, U><<<blocks, 256>>>(in, out, b, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)Ngrids * K * (8.0 * (NS + NW) + (BYTE ? 1 : 0));
    const double gbs = bytes * reps / (ms * 1e-3) / 1e9;
    printf("%-44s grids=%8ld K=%3d  %8.1f GB/s  (%.1f us/launch)\n", name, Ngrids, K, gbs, ms * 1e3 / reps);
    hipFree(in); hipFree(out); hipFree(b);
    return gbs;
}

int main()
{
    for (long n : {100000L, 1000000L}) {
        const int K = 64;
        run<double, 5, 0, false, 4>(n, K, "read 5 streams, 8 B/lane");
        run<double2, 5, 0, false, 4>(n, K, "read 5 streams, 16 B/lane");
        run<double, 5, 2, false, 4>(n, K, "read 5 + write 2, 8 B/lane");
        run<double2, 5, 2, false, 4>(n, K, "read 5 + write 2, 16 B/lane");
        run<double, 5, 2, true, 4>(n, K, "read 5 + write 2 + byte, 8 B/lane");
        run<double2, 5, 2, true, 4>(n, K, "read 5 + write 2 + byte(s), 16 B/lane");
        run<double, 5, 2, true, 8>(n, K, "read 5 + write 2 + byte, 8 B/lane, ring 8");
        run<double, 1, 1, false, 4>(n, K, "copy 1:1, 8 B/lane");
        run<double2, 1, 1, false, 4>(n, K, "copy 1:1, 16 B/lane");
    }
    return 0;
}
