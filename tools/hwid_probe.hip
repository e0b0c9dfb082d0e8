// Where do the workgroups of a persistent one-wave-per-workgroup grid land?  Prints, for a grid of CUs x 16 workgroups with the render
// kernel's LDS footprint, the (XCC, SE, CU) of every workgroup as read from HW_REG_HW_ID / HW_REG_XCC_ID.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/hwid_probe tools/hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>

__global__ void __launch_bounds__(64) probe(uint32_t* out, int spin)
{
    extern __shared__ uint32_t lds[];
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); // HW_REG_XCC_ID
    lds[threadIdx.x] = hw;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while ((long long)(__builtin_readcyclecounter() - t0) < (long long)spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = lds[0];
        out[2 * blockIdx.x + 1] = xcc;
    }
}

int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int grid = pr.multiProcessorCount * 16;
    uint32_t* d;
    hipMalloc(&d, grid * 8);
    hipMemset(d, 0, grid * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 10040);
    probe<<<grid, 64, 10040>>>(d, 200000);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<uint32_t> h(grid * 2);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> per_cu;
    printf("CUs %d grid %d\nfirst 80 workgroups: block xcc se sh cu simd wave (raw hw_id)\n", pr.multiProcessorCount, grid);
    for (int b = 0; b < grid; ++b) {
        const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, simd = (hw >> 4) & 3, wave = hw & 0xf;
        if (b < 80) printf("%4d %u %u %u %2u %u %2u (%08x)\n", b, xcc, se, sh, cu, simd, wave, hw);
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    printf("distinct (xcc,se,sh,cu): %zu\n", per_cu.size());
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    for (auto& kv : hist) printf("  %d CUs hold %d workgroups\n", kv.second, kv.first);
    for (uint32_t x = 0; x < 8; ++x) {
        printf("xcc %u:", x);
        for (auto& kv : per_cu) if ((kv.first >> 12) == x) printf(" %u.%u.%u=%d", (kv.first >> 8) & 0xf, (kv.first >> 4) & 0xf, kv.first & 0xf, kv.second);
        printf("\n");
    }
    return 0;
}
