// launchbench.hip -- what does a back-to-back kernel launch cost on this GPU when the kernel does (almost) nothing?
// The single-step cadence of the engine (4.9 us per env-step of 100 000 grids) is measured against this floor.
// build: hipcc --offload-arch=gfx950 -O3 tools/launchbench.hip -o tools/bin/launchbench
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void empty_kernel() {}
__global__ void touch_kernel(const double *in, double *out, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0;                       // one load -> one store per lane: a single memory round trip
}
struct Big { double pad[55]; };                             // ~440 B of kernel arguments, as the engine's KArgs
__global__ void args_kernel(Big b, double *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && b.pad[7] == 42.0) out[0] = 1.0; }
__global__ void ptr_kernel(const Big *b, double *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && b->pad[7] == 42.0) out[0] = 1.0; }
struct Mid { double pad[15]; };                             // 120 B
__global__ void mid_kernel(Mid b, double *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && b.pad[7] == 42.0) out[0] = 1.0; }
struct Mid2 { double pad[31]; };                            // 248 B
__global__ void mid2_kernel(Mid2 b, double *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && b.pad[7] == 42.0) out[0] = 1.0; }

int main()
{
    const long n = 100000;
    double *a, *b;
    (void)hipMalloc(&a, n * 8); (void)hipMalloc(&b, n * 8);
    (void)hipMemset(a, 0, n * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    Big big = {};
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 200; i++) launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 4000; i++) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %6.2f us per launch\n", name, ms * 1e3 / 4000);
    };
    run("empty kernel, 1 workgroup", [&] { empty_kernel<<<1, 256>>>(); });
    run("empty kernel, 391 workgroups", [&] { empty_kernel<<<391, 256>>>(); });
    run("empty kernel, 391 workgroups, 440 B of arguments", [&] { args_kernel<<<391, 256>>>(big, b); });
    Big *dbig; (void)hipMalloc(&dbig, sizeof(Big)); (void)hipMemcpy(dbig, &big, sizeof(Big), hipMemcpyHostToDevice);
    Mid mid = {}; Mid2 mid2 = {};
    run("empty kernel, 391 workgroups, 120 B of arguments", [&] { mid_kernel<<<391, 256>>>(mid, b); });
    run("empty kernel, 391 workgroups, 248 B of arguments", [&] { mid2_kernel<<<391, 256>>>(mid2, b); });
    run("empty kernel, 391 workgroups, pointer to 440 B in device memory", [&] { ptr_kernel<<<391, 256>>>(dbig, b); });
    run("one load + one store per lane, 391 workgroups (100 000)", [&] { touch_kernel<<<391, 256>>>(a, b, n); });
    return 0;
}
