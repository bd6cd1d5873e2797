// launchbench2.hip -- round 6: what do TWO host threads issuing to TWO streams cost each other inside the HIP runtime?
// The Gym single step as two dependent launch chains (mgx_set_launch_threads) is host-bound: each thread needs 6 us per launch
// where one thread alone needs 3.7.  This separates the runtime's share: the same 440-byte-argument kernel, three launch APIs
// (hipLaunchKernelGGL, hipModuleLaunchKernel via hipGetFuncBySymbol, hipExtModuleLaunchKernel), 1 thread x 1 stream, 1 thread x 2
// streams alternating, 2 threads x 1 stream each.  Host time per launch (issue only) and GPU cadence.
// build: hipcc --offload-arch=gfx950 -O3 -pthread tools/launchbench2.hip -o tools/bin/launchbench2
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

struct Big { double pad[55]; };
__global__ void args_kernel(Big b, double *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && b.pad[7] == 42.0) out[0] = 1.0; }
__global__ void touch_kernel(Big bb, const double *in, double *out, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] + bb.pad[3];
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const long n = 50000;
    const int NL = 20000;
    double *a, *b[2];
    (void)hipMalloc(&a, n * 8); (void)hipMalloc(&b[0], n * 8); (void)hipMalloc(&b[1], n * 8);
    (void)hipMemset(a, 0, n * 8);
    hipStream_t st[2];
    for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    Big big = {};
    hipFunction_t fn = nullptr;
    hipError_t ef = hipGetFuncBySymbol(&fn, (const void *)touch_kernel);
    printf("hipGetFuncBySymbol: %s\n", hipGetErrorString(ef));
    const unsigned blocks = (unsigned)((n + 255) / 256);

    auto launch_ggl = [&](int j) { touch_kernel<<<blocks, 256, 0, st[j]>>>(big, a, b[j], n); };
    auto launch_mod = [&](int j) {
        struct { Big bb; const double *in; double *out; long n; } args{big, a, b[j], n};
        size_t sz = sizeof(args);
        void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        (void)hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, st[j], nullptr, cfg);
    };
    auto launch_ext = [&](int j) {
        struct { Big bb; const double *in; double *out; long n; } args{big, a, b[j], n};
        size_t sz = sizeof(args);
        void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        (void)hipExtModuleLaunchKernel(fn, blocks * 256, 1, 1, 256, 1, 1, 0, st[j], nullptr, cfg, nullptr, nullptr, 0);
    };

    auto bench = [&](const char *name, auto launch) {
        // (a) one thread, one stream
        for (int i = 0; i < 500; i++) launch(0);
        (void)hipDeviceSynchronize();
        double t0 = now_us();
        for (int i = 0; i < NL; i++) launch(0);
        double t1 = now_us();
        (void)hipDeviceSynchronize();
        double t2 = now_us();
        printf("%-28s 1 thread  x 1 stream : host %5.2f us per launch, wall %5.2f us per launch\n", name, (t1 - t0) / NL, (t2 - t0) / NL);
        // (b) one thread, two streams alternating
        t0 = now_us();
        for (int i = 0; i < NL; i++) { launch(0); launch(1); }
        t1 = now_us();
        (void)hipDeviceSynchronize();
        t2 = now_us();
        printf("%-28s 1 thread  x 2 streams: host %5.2f us per launch, wall %5.2f us per PAIR\n", name, (t1 - t0) / (2 * NL), (t2 - t0) / NL);
        // (c) two threads, one stream each
        std::atomic<int> go{0};
        double host[2];
        std::vector<std::thread> th;
        for (int j = 0; j < 2; j++)
            th.emplace_back([&, j] {
                while (!go.load()) {}
                const double s0 = now_us();
                for (int i = 0; i < NL; i++) launch(j);
                host[j] = (now_us() - s0) / NL;
            });
        t0 = now_us();
        go.store(1);
        for (auto &t : th) t.join();
        (void)hipDeviceSynchronize();
        t2 = now_us();
        printf("%-28s 2 threads x 1 stream : host %5.2f / %5.2f us per launch, wall %5.2f us per PAIR\n", name, host[0], host[1], (t2 - t0) / NL);
    };
    bench("hipLaunchKernelGGL", launch_ggl);
    if (ef == hipSuccess) {
        bench("hipModuleLaunchKernel", launch_mod);
        bench("hipExtModuleLaunchKernel", launch_ext);
    }
    return 0;
}
