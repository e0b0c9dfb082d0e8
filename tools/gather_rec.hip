// gather_rec.hip -- gather-rate ceiling for per-lane records of REC bytes (64/128/192/256) at random record indices, dependent chain.
//   hipcc --offload-arch=gfx950 -O3 -DREC=128 -o gather_rec tools/gather_rec.hip ; ./gather_rec <table KiB> <iters>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifndef REC
#define REC 64
#endif
__global__ void gather(const float4* __restrict__ table, unsigned n_rec, int iters, float* out)
{
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned s = tid * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        unsigned r = (unsigned)(((unsigned long long)s * n_rec) >> 32);
        const float4* p = table + (size_t)r * (REC / 16);
        float4 v[REC / 16];
#pragma unroll
        for (int k = 0; k < REC / 16; ++k) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < REC / 16; ++k) acc += v[k].x + v[k].w;
        s ^= __float_as_uint(acc) & 1u;
    }
    out[tid] = acc;
}
int main(int argc, char** argv)
{
    long kib = argc > 1 ? atol(argv[1]) : 65536;
    int iters = argc > 2 ? atoi(argv[2]) : 256;
    size_t bytes = (size_t)kib << 10;
    unsigned n_rec = (unsigned)(bytes / REC);
    float4* table; float* out;
    int grid = 4096, block = 256;
    (void)hipMalloc(&table, bytes); (void)hipMemset(table, 0, bytes); (void)hipMalloc(&out, (size_t)grid * block * 4);
    hipLaunchKernelGGL(gather, dim3(grid), dim3(block), 0, 0, table, n_rec, iters, out);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(gather, dim3(grid), dim3(block), 0, 0, table, n_rec, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    double recs = (double)grid * block * iters;
    printf("{\"rec_bytes\": %d, \"table_KiB\": %ld, \"ms\": %.3f, \"Grec_per_s\": %.1f, \"GBps\": %.1f}\n", REC, kib, ms, recs / ms / 1e6, recs * REC / ms / 1e6);
    return 0;
}
