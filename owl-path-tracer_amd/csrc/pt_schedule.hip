// pt_schedule.hip -- the small kernels around the render launch: the cost-ordered pixel queue (counting sort of the cost pre-pass's
// classes), the tier plan of the whole-pixel schedule, and helpers (RGBA8 pack of a reduced frame, triangle-id packing, parameter block store).
#include <hip/hip_runtime.h>

#include "pt_device.h"
#include "pt_launch.h"
#include "pt_tiers.h"
#include "pt_types.h"

using namespace ptd;

// ---- cost-ordered pixel queue ------------------------------------------------------------------------------------------
// A pixel's samples are sequential (one RNG stream, device.cu:226-243), so a frame cannot end before its most expensive pixel
// does.  The host runs a short pre-pass that records rays per pixel (cost_out), then these three kernels build a queue with
// the expensive pixels first (stable counting sort over PT_SORT_BUCKETS cost classes, so neighbours stay neighbours).
#define PT_SORT_BLOCK 256
#define PT_SORT_ITEMS 16

// Cost of a pixel for the ordering and the tier plan: the class the pre-pass recorded for it (pt_cost_class), de-noised.  A few
// samples are a noisy estimate, and a pixel that is taken for cheaper than it is spends the frame in a wave that is too dense for it
// and ends long after everything else.  Expensive regions are spatially coherent, so the estimate is the mean class (= geometric mean
// of the durations) over those pixels of the (2R+1)^2 neighbourhood that look like the same surface - within +-PT_COST_BAND classes
// (+-54 %) of the pixel's own - and never less than the pixel's own.  (The neighbourhood MAXIMUM - rounds 1 and 2 - is safe but puts
// three times as many pixels into the expensive classes as belong there; the tier plan then runs out of waves.)  Pixels of other ranks
// and outside the image hold 0 and take no part.
#define PT_COST_BAND 10
__device__ __forceinline__ uint32_t pixel_cost(const uint8_t* __restrict__ img, uint32_t pid, int W, int H, int R)
{
    const int x = (int)(pid % (uint32_t)W), y = (int)(pid / (uint32_t)W);
    const int own = (int)img[pid];
    int sum = 0, cnt = 0;
    for (int dy = -R; dy <= R; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -R; dx <= R; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const int v = (int)img[(size_t)yy * W + xx];
            if (v != 0 && v >= own - PT_COST_BAND && v <= own + PT_COST_BAND) { sum += v; ++cnt; }
        }
    }
    const int mean = cnt ? (sum + cnt - 1) / cnt : own;
    return (uint32_t)(mean > own ? mean : own);
}

// Cost classes of the queue: PT_SORT_BUCKETS buckets of four pre-pass classes each (19 % wide), from class PT_COST_TOP (20 ms for the
// pre-pass's samples of one pixel) down; everything below 80 us shares the last bucket.  0 = most expensive.
__device__ __forceinline__ int cost_bucket(uint32_t cls)
{
    const int b = (PT_COST_TOP - (int)cls) / 4;
    return b < 0 ? 0 : (b > PT_SORT_BUCKETS - 1 ? PT_SORT_BUCKETS - 1 : b);
}

__global__ void __launch_bounds__(PT_SORT_BLOCK) pt_sort_hist_kernel(const uint8_t* __restrict__ img, const uint32_t* __restrict__ in, int W, int H, int R, uint32_t n,
                                                                    uint32_t c0, uint32_t* __restrict__ block_hist, uint8_t* __restrict__ bucket)
{
    __shared__ uint32_t h[PT_SORT_BUCKETS];
    if (threadIdx.x < PT_SORT_BUCKETS) h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t first = (blockIdx.x * PT_SORT_BLOCK + threadIdx.x) * PT_SORT_ITEMS;
    for (uint32_t i = first; i < first + PT_SORT_ITEMS && i < n; ++i) {
        const int b = cost_bucket(pixel_cost(img, in[i], W, H, R));
        bucket[i] = (uint8_t)b; // per queue entry, for the scatter pass
        atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < PT_SORT_BUCKETS) block_hist[threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of block_hist in (bucket, block) order; one workgroup
__global__ void __launch_bounds__(PT_SORT_BLOCK) pt_sort_scan_kernel(uint32_t* __restrict__ block_hist, uint32_t n_entries)
{
    __shared__ uint32_t part[PT_SORT_BLOCK];
    const uint32_t per = (n_entries + PT_SORT_BLOCK - 1) / PT_SORT_BLOCK;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n_entries ? lo + per : n_entries;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += block_hist[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < PT_SORT_BLOCK; ++t) { const uint32_t v = part[t]; part[t] = run; run += v; }
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = block_hist[i]; block_hist[i] = run; run += v; }
}

__global__ void __launch_bounds__(PT_SORT_BLOCK) pt_sort_scatter_kernel(const uint8_t* __restrict__ bucket, const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                                       uint32_t n, const uint32_t* __restrict__ block_off)
{
    __shared__ uint32_t cnt[PT_SORT_BUCKETS][PT_SORT_BLOCK];
    const uint32_t t = threadIdx.x;
    for (int b = 0; b < PT_SORT_BUCKETS; ++b) cnt[b][t] = 0u;
    const uint32_t first = (blockIdx.x * PT_SORT_BLOCK + t) * PT_SORT_ITEMS;
    for (uint32_t i = first; i < first + PT_SORT_ITEMS && i < n; ++i) cnt[bucket[i]][t] += 1u;
    __syncthreads();
    if (t < PT_SORT_BUCKETS) { // queue position of (bucket t, thread j) of this block
        uint32_t run = block_off[t * gridDim.x + blockIdx.x];
        for (int j = 0; j < PT_SORT_BLOCK; ++j) { const uint32_t v = cnt[t][j]; cnt[t][j] = run; run += v; }
    }
    __syncthreads();
    for (uint32_t i = first; i < first + PT_SORT_ITEMS && i < n; ++i) out[cnt[bucket[i]][t]++] = in[i];
}

// ---- tier plan of the whole-pixel schedule (take_ticket): pt_tiers.h ----------------------------------------------------------
// One thread.  block_off: the scanned histogram of the sort (bucket b starts at queue entry block_off[b * nb]).
__global__ void pt_plan_tiers_kernel(const uint32_t* __restrict__ block_off, uint32_t nb, uint32_t n, int capacity, int ns, int force, uint32_t* __restrict__ tiers)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t start[PT_SORT_BUCKETS + 1];
    for (int b = 0; b < PT_SORT_BUCKETS; ++b) start[b] = block_off[(size_t)b * nb];
    start[PT_SORT_BUCKETS] = n;
    pt_plan_tiers(start, capacity, ns, force, tiers);
}

extern "C" hipError_t pt_launch_plan_tiers(const uint32_t* scratch, uint32_t n, int capacity, int ns, int force, uint32_t* tiers, hipStream_t stream)
{
    const uint32_t per_block = PT_SORT_BLOCK * PT_SORT_ITEMS;
    const uint32_t nb = (n + per_block - 1) / per_block;
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_plan_tiers_kernel, dim3(1), dim3(1), 0, stream, scratch, nb, n, capacity, ns, force, tiers);
    return hipGetLastError();
}

extern "C" size_t pt_sort_scratch_bytes(uint32_t n)
{
    const uint32_t per_block = PT_SORT_BLOCK * PT_SORT_ITEMS;
    return (size_t)((n + per_block - 1) / per_block) * PT_SORT_BUCKETS * sizeof(uint32_t);
}

extern "C" hipError_t pt_launch_sort_pixels(const uint8_t* cost_img, int W, int H, int radius, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t c0,
                                            uint32_t* scratch, uint8_t* bucket, hipStream_t stream)
{
    const uint32_t per_block = PT_SORT_BLOCK * PT_SORT_ITEMS;
    const uint32_t nb = (n + per_block - 1) / per_block;
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_sort_hist_kernel, dim3(nb), dim3(PT_SORT_BLOCK), 0, stream, cost_img, in, W, H, radius, n, c0, scratch, bucket);
    hipLaunchKernelGGL(pt_sort_scan_kernel, dim3(1), dim3(PT_SORT_BLOCK), 0, stream, scratch, nb * PT_SORT_BUCKETS);
    hipLaunchKernelGGL(pt_sort_scatter_kernel, dim3(nb), dim3(PT_SORT_BLOCK), 0, stream, (const uint8_t*)bucket, in, out, n, (const uint32_t*)scratch);
    return hipGetLastError();
}

// RGBA8 image of a float3 framebuffer (device.cu:252-253 applied to a whole frame).  With N ranks only the float3 frame is reduced
// (one collective, the same on every rank whatever buffers its caller passed); the root quantises the reduced frame here - every
// pixel has one non-zero contributor, so this is bit for bit what the owning rank's kernel would have stored.
__global__ void __launch_bounds__(256) pt_pack_rgba8_kernel(const float* __restrict__ rgb, uint32_t* __restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_rgba(V(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
}

extern "C" hipError_t pt_launch_pack_rgba8(const float* rgb, uint32_t* out, long long n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pt_pack_rgba8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rgb, out, n);
    return hipGetLastError();
}

// PtTri::id -> id << 8 | min(material + 1, 255) on the device copy of the triangle records (pt_api.cpp, upload_scene_to_device): the
// host keeps its records unpacked, and packing there would mean a second 48-byte-per-triangle copy on every upload.
__global__ void __launch_bounds__(256) pt_pack_tri_ids_kernel(PtTri* __restrict__ tris, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t id = tris[i].id, m = tris[i].material;
    if (id != 0x7fffffff) tris[i].id = (int32_t)(((uint32_t)id << 8) | (uint32_t)(m + 1 < 255 ? m + 1 : 255));
}

extern "C" hipError_t pt_launch_pack_tri_ids(PtTri* tris, long long n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pt_pack_tri_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tris, n);
    return hipGetLastError();
}

// The parameter block of a wavefront launch, stored by a one-thread kernel: kernel arguments are captured when the launch is
// enqueued, so the host copy may be reused for the next launch at once (a hipMemcpyAsync from pageable memory is only safe while the
// runtime stages it at enqueue time).
__global__ void pt_store_params_kernel(const PtKernelParams p, PtKernelParams* __restrict__ dst) { *dst = p; }

extern "C" hipError_t pt_launch_store_params(const PtKernelParams* p, PtKernelParams* d_dst, hipStream_t stream)
{
    hipLaunchKernelGGL(pt_store_params_kernel, dim3(1), dim3(1), 0, stream, *p, d_dst);
    return hipGetLastError();
}

