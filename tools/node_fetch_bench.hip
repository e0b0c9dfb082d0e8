// node_fetch_bench.hip -- how should a wave fetch one 128-byte BVH record per lane?  (round 3; MI355X)
//
// The render kernel's node step issues 7 x global_load_dwordx4 per lane, all to the lane's own 128-byte record: 7 vL1D accesses
// per lane and step.  This bench measures dependent random record walks (next index comes out of the record) in four forms:
//   mode 0  per-lane: 7 x global_load_dwordx4 from the lane's record (the kernel's form)
//   mode 1  cooperative LDS-DMA: round r, lane i fetches 16 bytes of the record of lane 8r + i/8 (8 lanes = one whole 128-byte
//           line per record) straight into LDS (global_load_lds_dwordx4), then every lane reads its record with 7 x ds_read_b128;
//           the chunk order inside a record is rotated by (owner lane >> 1) so that the b128 reads are bank-conflict free
//   mode 2  as mode 1 with register staging (global_load_dwordx4 + ds_write_b128)
//   mode 3  groups of 8 lanes walk ONE record chain each (8 chains per wave): one global_load_dwordx4 per step
// usage: node_fetch_bench <mode> <table KiB> <iters> <active lanes> <lds pad KiB per wave (occupancy knob)>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

__device__ __forceinline__ f32x4 ldg4(const void* base, size_t off) { return *(const f32x4 AS1*)((const char AS1*)base + off); }

// some arithmetic on the record (stands for the slab tests) and the next index out of it
__device__ __forceinline__ uint32_t consume(const f32x4* c, uint32_t it, float& acc)
{
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += c[k].x * c[k].y + c[k].z * c[k].w;
    acc += s;
    const uint32_t nx[4] = {__float_as_uint(c[6].x), __float_as_uint(c[6].y), __float_as_uint(c[6].z), __float_as_uint(c[6].w)};
    return nx[(it + (__float_as_uint(acc) & 1u)) & 3u];
}

template <int MODE>
__global__ void __launch_bounds__(64, 4) walk(const char* __restrict__ table, uint32_t n_rec, int iters, int active, float* out)
{
    extern __shared__ __attribute__((aligned(128))) uint32_t lds[];
    const int lane = threadIdx.x;
    uint32_t cur = (uint32_t)(((unsigned long long)((blockIdx.x * 64u + lane) * 2654435761u + 12345u) * n_rec) >> 32);
    float acc = 0.0f;
    const bool on = lane < active;
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            if (on) {
                f32x4 c[7];
                const size_t b = (size_t)cur * 128;
#pragma unroll
                for (int k = 0; k < 7; ++k) c[k] = ldg4(table, b + 16 * k);
                cur = consume(c, it, acc);
            }
        }
    } else if (MODE == 1 || MODE == 2) {
        const int n_rounds = (active + 7) / 8;
        for (int it = 0; it < iters; ++it) {
            uint32_t refs[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) refs[r] = (uint32_t)__builtin_amdgcn_ds_bpermute((8 * r + (lane >> 3)) * 4, (int)cur);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (r >= n_rounds) break;
                const int owner = 8 * r + (lane >> 3);
                const int chunk = ((lane & 7) + (owner >> 1)) & 7;
                const char AS1* src = (const char AS1*)table + (size_t)refs[r] * 128 + chunk * 16;
                if (MODE == 1) {
                    if (owner < active) __builtin_amdgcn_global_load_lds((const void AS1*)src, (void AS3*)(lds + r * 256), 16, 0, 0);
                } else {
                    if (owner < active) {
                        const f32x4 v = *(const f32x4 AS1*)src;
                        *(f32x4*)(lds + r * 256 + lane * 4) = v;
                    }
                }
            }
            if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (on) {
                f32x4 c[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) c[k] = *(const f32x4*)(lds + lane * 32 + (((k - (lane >> 1)) & 7) * 4));
                cur = consume(c, it, acc);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
    } else {
        // 8 lanes per chain; lane k of a group holds chunk k of the group's record
        const int g = lane >> 3, sub = lane & 7;
        cur = (uint32_t)(((unsigned long long)((blockIdx.x * 8u + g) * 2654435761u + 12345u) * n_rec) >> 32);
        const bool gon = g * 8 < active;
        for (int it = 0; it < iters; ++it) {
            if (gon) {
                const f32x4 c = ldg4(table, (size_t)cur * 128 + sub * 16);
                float s = c.x * c.y + c.z * c.w;
                // group reduction (stands for the nearest-child search): xor 1, xor 2, mirror within 8
                s += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0xB1, 0xf, 0xf, true));
                s += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x4E, 0xf, 0xf, true));
                s += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x141, 0xf, 0xf, true));
                acc += s;
                // next index: word (it & 3) of chunk 6, broadcast to the group
                const uint32_t nx[4] = {__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), __float_as_uint(c.w)};
                const uint32_t mine = nx[(it + (__float_as_uint(acc) & 1u)) & 3u];
                cur = (uint32_t)__builtin_amdgcn_ds_bpermute((g * 8 + 6) * 4, (int)mine);
            }
        }
    }
    out[blockIdx.x * 64 + lane] = acc + (float)cur;
}

int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const long kib = argc > 2 ? atol(argv[2]) : 16384;
    const int iters = argc > 3 ? atoi(argv[3]) : 512;
    const int active = argc > 4 ? atoi(argv[4]) : 64;
    const int pad_kib = argc > 5 ? atoi(argv[5]) : 9;
    const size_t bytes = (size_t)kib << 10;
    const uint32_t n_rec = (uint32_t)(bytes / 128);
    // records: random floats in chunks 0..5, four random next indices in chunk 6
    uint32_t* h = (uint32_t*)malloc(bytes);
    uint64_t s = 0x5EEDull;
    for (uint32_t r = 0; r < n_rec; ++r) {
        for (int w = 0; w < 32; ++w) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const uint32_t x = (uint32_t)(s >> 33);
            if (w >= 24 && w < 28) h[r * 32 + w] = (uint32_t)(((unsigned long long)x * n_rec) >> 31) % n_rec;
            else { const float f = (float)(x & 0xffff) * (1.0f / 65536.0f); h[r * 32 + w] = *(const uint32_t*)&f; }
        }
    }
    char* table; float* out;
    int dev_cus = 256;
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0); dev_cus = prop.multiProcessorCount;
    const size_t lds_bytes = (size_t)pad_kib << 10;
    const int waves_per_cu = (int)(160 * 1024 / lds_bytes) < 16 ? (int)(160 * 1024 / lds_bytes) : 16;
    const int grid = dev_cus * waves_per_cu;
    (void)hipMalloc(&table, bytes); (void)hipMemcpy(table, h, bytes, hipMemcpyHostToDevice); (void)hipMalloc(&out, (size_t)grid * 64 * 4);
    void (*fn)(const char*, uint32_t, int, int, float*) = mode == 0 ? walk<0> : mode == 1 ? walk<1> : mode == 2 ? walk<2> : walk<3>;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds_bytes, 0, (const char*)table, n_rec, iters, active, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
    const double recs = (double)grid * (mode == 3 ? (active + 7) / 8 : active) * iters;
    printf("{\"mode\": %d, \"table_KiB\": %ld, \"active\": %d, \"waves_per_cu\": %d, \"ms\": %.3f, \"Grec_per_s\": %.2f, \"us_per_step\": %.3f}\n", mode, kib, active, waves_per_cu, best,
           recs / best / 1e6, best * 1e3 / iters);
    return 0;
}
