// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE for THIS kernel's access pattern (MI355X_MICROARCH.md: "other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern"): every lane reads one 64-byte record
// (4 x global_load_dwordx4) at a pseudo-random record index, like a BVH-node fetch.  Known bytes = threads * iters * 64.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib tools/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib <table MiB> <iters>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void gather64(const float4* __restrict__ table, unsigned n_rec, int iters, float* out, int active_lanes)
{
    if ((int)(threadIdx.x & 63) >= active_lanes) return; // lane-occupancy experiment: cost of a vector-memory instruction vs active lanes
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned s = tid * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        unsigned r = (unsigned)(((unsigned long long)s * n_rec) >> 32);
        const float4* p = table + (size_t)r * 4;
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
        s ^= __float_as_uint(acc) & 1u; // dependent chain like a traversal
    }
    out[tid] = acc;
}

int main(int argc, char** argv)
{
    long mib = argc > 1 ? atol(argv[1]) : 1024; // negative: KiB
    int iters = argc > 2 ? atoi(argv[2]) : 256;
    int active = argc > 3 ? atoi(argv[3]) : 64;
    size_t bytes = mib < 0 ? (size_t)(-mib) << 10 : (size_t)mib << 20;
    unsigned n_rec = (unsigned)(bytes / 64);
    float4* table;
    float* out;
    int grid = 4096, block = 256;
    hipMalloc(&table, bytes);
    hipMemset(table, 0, bytes);
    hipMalloc(&out, (size_t)grid * block * 4);
    hipLaunchKernelGGL(gather64, dim3(grid), dim3(block), 0, 0, table, n_rec, iters, out, active);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(gather64, dim3(grid), dim3(block), 0, 0, table, n_rec, iters, out, active);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double known = (double)grid * block * iters * 64.0 * active / 64.0;
    printf("{\"active_lanes\": %d, \"table_MiB\": %ld, \"iters\": %d, \"known_bytes_per_launch\": %.0f, \"ms\": %.3f, \"GBps\": %.1f}\n", active, mib, iters, known, ms, known / ms / 1e6);
    return 0;
}
