// fillbench.hip -- experiment: what does MI355X sustain for PURE WRITES (the observation-row traffic of config 5 is 85 %
// writes)?  Grid-stride fill of a 4 GB buffer: 8 / 16 B per lane, plain / non-temporal stores, and a 1:1 copy for scale.
// build: hipcc --offload-arch=gfx950 -O3 tools/fillbench.hip -o tools/bin/fillbench
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename V, bool NT>
__global__ __launch_bounds__(256) void fill(V *__restrict__ out, long n, double x)
{
    V v;
    for (int j = 0; j < (int)(sizeof(V) / 8); j++) ((double *)&v)[j] = x + j;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

template <typename V>
__global__ __launch_bounds__(256) void copy(const V *__restrict__ in, V *__restrict__ out, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = in[i];
}

typedef double v2 __attribute__((ext_vector_type(2)));

template <typename F>
static void timeit(const char *name, double bytes, F launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) launch();
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; r++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f GB/s\n", name, bytes * reps / (ms * 1e-3) / 1e9);
}

int main()
{
    const long bytes = 4L << 30;
    double *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes);
    for (int blocks : {2048, 8192, 65536}) {
        printf("-- %d workgroups\n", blocks);
        timeit("fill  8 B/lane plain", bytes, [&] { fill<double, false><<<blocks, 256>>>(b, bytes / 8, 1.0); });
        timeit("fill  8 B/lane non-temporal", bytes, [&] { fill<double, true><<<blocks, 256>>>(b, bytes / 8, 1.0); });
        timeit("fill 16 B/lane plain", bytes, [&] { fill<v2, false><<<blocks, 256>>>((v2 *)b, bytes / 16, 1.0); });
        timeit("fill 16 B/lane non-temporal", bytes, [&] { fill<v2, true><<<blocks, 256>>>((v2 *)b, bytes / 16, 1.0); });
        timeit("copy 16 B/lane (read + write bytes)", 2.0 * bytes, [&] { copy<v2><<<blocks, 256>>>((const v2 *)a, (v2 *)b, bytes / 16); });
    }
    return 0;
}
