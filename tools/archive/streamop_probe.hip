// Probe (round 4): do HIP stream memory operations work on this stack, and what do they cost?
//   hipStreamWriteValue32 on device memory that a RESIDENT kernel polls; hipStreamWaitValue32 on signal memory that the kernel writes.
// Everything is bounded by wall-clock timeouts inside the kernels: nothing here can hang the device.
// build: hipcc --offload-arch=gfx950 -O2 tools/streamop_probe.hip -o tools/bin/streamop_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// one resident wave: for k = 0 .. n-1: wait until *seq > k (or timeout), stamp the wall clock, write done[0] = k + 1
__global__ void server(volatile uint32_t *seq, uint32_t *done, int n, int64_t timeout_ticks, int64_t *stamps, uint32_t *status)
{
    if (threadIdx.x != 0) return;
    const int64_t t0 = wall_clock64();
    for (int k = 0; k < n; k++) {
        while (__hip_atomic_load((uint32_t *)seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) <= (uint32_t)k) {
            if (wall_clock64() - t0 > timeout_ticks) { *status = 2; __hip_atomic_store(done, 0xffffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); return; }
            __builtin_amdgcn_s_sleep(2);
        }
        stamps[k] = wall_clock64();
        __hip_atomic_store(done, (uint32_t)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *status = 1;
}

__global__ void touch(uint32_t *p, int64_t *stamp) { if (threadIdx.x == 0) { *stamp = wall_clock64(); p[0] += 1; } }

int main()
{
    int can_wait = -1;
    CK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can_wait);
    uint32_t *seq, *done, *status, *scratch;
    int64_t *stamps, *tstamp;
    const int n = 2000;
    CK(hipMalloc(&seq, 64));
    hipError_t es = hipExtMallocWithFlags((void **)&done, 8, hipMallocSignalMemory);      // (a HSA signal: exactly 8 bytes)
    (void)hipGetLastError();
    printf("hipExtMallocWithFlags(hipMallocSignalMemory) -> %s\n", hipGetErrorString(es));
    if (es != hipSuccess) CK(hipMalloc(&done, 64));
    CK(hipMalloc(&status, 64)); CK(hipMalloc(&scratch, 64)); CK(hipMalloc(&stamps, n * 8)); CK(hipMalloc(&tstamp, n * 8));
    CK(hipMemset(seq, 0, 64)); CK(hipMemset(done, 0, 8)); CK(hipMemset(status, 0, 64)); CK(hipMemset(scratch, 0, 64));
    hipStream_t ss, sp;
    CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    CK(hipDeviceSynchronize());
    server<<<1, 64, 0, ss>>>(seq, done, n, (int64_t)3 * 100000000, stamps, status);       // 3 s at 100 MHz
    CK(hipGetLastError());
    // (A) host posts n steps with stream write ops, waits for each through a stream wait op, then a tiny kernel: the round trip
    auto t0 = std::chrono::steady_clock::now();
    hipError_t ew = hipSuccess, ewt = hipSuccess;
    for (int k = 0; k < n && ew == hipSuccess && ewt == hipSuccess; k++) {
        ew = hipStreamWriteValue32(sp, seq, (uint32_t)(k + 1), 0);
        if (ew != hipSuccess) break;
        ewt = hipStreamWaitValue32(sp, done, (uint32_t)(k + 1), hipStreamWaitValueGte, 0xffffffffu);
        if (ewt != hipSuccess) break;
        touch<<<1, 64, 0, sp>>>(scratch, tstamp + k);
    }
    auto t1 = std::chrono::steady_clock::now();
    printf("hipStreamWriteValue32 -> %s, hipStreamWaitValue32 -> %s; host issue %.2f us per (write, wait, kernel)\n", hipGetErrorString(ew),
           hipGetErrorString(ewt), std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
    if (ew != hipSuccess || ewt != hipSuccess) {           // release the server whatever happened
        uint32_t big = 0x7fffffffu;
        CK(hipMemcpyAsync(seq, &big, 4, hipMemcpyHostToDevice, sp));
    }
    CK(hipStreamSynchronize(sp));
    auto t2 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(ss));
    uint32_t st = 0, sc = 0;
    CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&sc, scratch, 4, hipMemcpyDeviceToHost));
    std::vector<int64_t> a(n), b(n);
    CK(hipMemcpy(a.data(), stamps, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), tstamp, n * 8, hipMemcpyDeviceToHost));
    printf("server status %u (1 = served all), touch kernels ran %u; whole loop %.2f us per step (wall)\n", st, sc,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
    if (st == 1 && sc == (uint32_t)n) {
        double cad = 0, lag = 0;
        for (int k = n / 2; k < n - 1; k++) { cad += (a[k + 1] - a[k]) * 0.01; lag += (b[k] - a[k]) * 0.01; }
        printf("device clocks: step-to-step cadence %.2f us, server stamp -> dependent kernel start %.2f us\n", cad / (n / 2 - 1), lag / (n / 2 - 1));
    }
    return 0;
}
