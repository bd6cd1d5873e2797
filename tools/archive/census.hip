// census.hip -- where does the dispatcher put the workgroups of a 391 x 256-thread launch?  (experiment)
// Each workgroup records its XCC / SE / CU id (HW_REG_HW_ID / XCC_ID) and spins ~20 us so that all are co-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <cstdlib>

__global__ void census(unsigned *out, int spin, int vgpr_hog)
{
    extern __shared__ double lds[];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = xcc; }
    if (vgpr_hog == 12345) lds[threadIdx.x] = 1.0;
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 391;
    const int threads = argc > 2 ? atoi(argv[2]) : 256;
    const size_t lds = argc > 3 ? atoi(argv[3]) : 0;
    unsigned *d; hipMalloc(&d, blocks * 8);
    if (lds > 65536) hipFuncSetAttribute((const void *)census, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    census<<<blocks, threads, lds>>>(d, 40000, 0);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 2);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu, per_xcc;
    for (int b = 0; b < blocks; b++) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        per_xcc[xcc]++;
    }
    std::map<int, int> hist;
    for (auto &kv : per_cu) hist[kv.second]++;
    printf("blocks=%d threads=%d lds=%zu: distinct CUs used = %zu;  blocks-per-CU histogram:", blocks, threads, lds, per_cu.size());
    for (auto &kv : hist) printf("  %d blocks: %d CUs", kv.first, kv.second);
    printf("\n  per XCC:");
    for (auto &kv : per_xcc) printf(" %d", kv.second);
    printf("\n");
    return 0;
}
