// pt_kernel_aux.hip -- the render path's second implementation and its validation hooks (never on the product path):
//   pt_render_kernel   the simple persistent lane-per-pixel megakernel (option kernel=1): kept for A/B measurements and as an independent
//                      implementation in the parity suite (tests/test_gpu_parity.py::test_lane_per_pixel_variant_bitwise);
//   pt_debug_kernel    batched evaluation of the kernels' building blocks for the bit-exact comparisons with the oracle (pt_debug_eval).
#include "pt_launch.h"
#include "pt_trace.h"

namespace {

// Pixel finished for this launch: final average + framebuffer store (device.cu:246-253) or chunk state save.
__device__ __forceinline__ void finish_pixel(const PtKernelParams& P, uint32_t pid, int px, int py, uint32_t rng, v3 color)
{
    if (P.sample_begin + P.sample_count >= P.max_samples) {
        v3 out = color * (1.0f / (float)P.max_samples);                          // device.cu:247
        size_t ofs = (size_t)px + (size_t)P.width * (size_t)(P.height - 1 - py); // device.cu:251
        P.out_rgb[3 * ofs] = out.x;
        P.out_rgb[3 * ofs + 1] = out.y;
        P.out_rgb[3 * ofs + 2] = out.z;
        if (P.out_rgba8) P.out_rgba8[ofs] = make_rgba(out);
    } else {
        P.rng_state[pid] = rng;
        P.accum[3 * (size_t)pid] = color.x;
        P.accum[3 * (size_t)pid + 1] = color.y;
        P.accum[3 * (size_t)pid + 2] = color.z;
    }
}

// Next pixel of the queue: device.cu:224-228 (queue instead of a 2-D launch).  Returns false when exhausted.
__device__ __forceinline__ bool fetch_pixel(const PtKernelParams& P, uint32_t& pid, int& px, int& py, uint32_t& rng, v3& color)
{
    uint32_t q = atomicAdd(P.queue_head, 1u); // hipcc aggregates this into one atomic per wave
    if (q >= P.n_pixels) return false;
    pid = P.pixel_ids[q];
    px = (int)(pid % (uint32_t)P.width);
    py = (int)(pid / (uint32_t)P.width);
    if (P.sample_begin == 0) {
        rng = rng_init((uint32_t)px, (uint32_t)py);
        color = vs(0.0f);
    } else {
        rng = P.rng_state[pid];
        color = V(P.accum[3 * (size_t)pid], P.accum[3 * (size_t)pid + 1], P.accum[3 * (size_t)pid + 2]);
    }
    return true;
}

} // namespace

// =====================================================================================================================
// v1: persistent lane-per-pixel megakernel (option kernel=1)
// =====================================================================================================================

template <bool COUNT>
__global__ void __launch_bounds__(PT_BLOCK) pt_render_kernel(const PtKernelParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* stack = lds + threadIdx.x; // stack[level * PT_BLOCK]

    uint32_t pid = 0;
    int px = 0, py = 0, s = 0;
    v3 color = vs(0.0f);
    bool have_pixel = false, done = false;
    PathState ps;
    ps.rng = 0; ps.org = vs(0.0f); ps.dir = vs(0.0f); ps.throughput = vs(1.0f); ps.depth = 0; ps.lobe = kLobeNone; ps.retries = 0;
    bool new_path = true, retry = false;
    Hit h;
    h.t = kTMax; h.u = h.v = 0.0f; h.slot = -1; h.id = 0x7fffffff;
    Counters cn;

    for (;;) {
        if (!have_pixel && !done) {
            if (fetch_pixel(P, pid, px, py, ps.rng, color)) {
                s = 0;
                have_pixel = true;
                new_path = true;
            } else {
                done = true;
            }
        }
        if (__ballot(!done) == 0ull) break;
        if (done) continue;

        if (new_path) {
            gen_camera_ray(P, px, py, ps);
            new_path = false;
        }
        // owl::traceRay, device.cu:133 (a NaN/Inf retry re-shades the same hit: same ray, same result)
        if (!retry) closest_hit<COUNT>(P, stack, ps.org, ps.dir, h, cn);
        if (COUNT) ++cn.rays;
        retry = false;

        v3 radiance;
        int r = shade_hit<COUNT>(P, P.materials, h.slot, h.u, h.v, ps, radiance, cn);
        if (r == SR_RETRY) {
            retry = true;
        } else if (r == SR_END) {
            color = color + radiance * ps.throughput; // device.cu:217,243
            if (COUNT) ++cn.samples;
            ++s;
            new_path = true;
            if (s == P.sample_count) {
                finish_pixel(P, pid, px, py, ps.rng, color);
                have_pixel = false;
            }
        }
    }
    flush_counters<COUNT>(P, cn);
}

// ---- validation kernels (tests only; see pt_debug_eval in include/mi355pt.h) -----------------------------

enum {
    PT_OP_SIN = 0, PT_OP_COS, PT_OP_TAN, PT_OP_ATAN, PT_OP_ATAN2, PT_OP_ASIN, PT_OP_LOG, PT_OP_EXP, PT_OP_POW, PT_OP_SQRT, PT_OP_DIV,
    PT_OP_SAMPLE_DISNEY = 20, // in: mat[17], wo[3], rng bits, lobe bits (22) -> out: f[3], wi[3], pdf, lobe bits, rng bits (9)
    PT_OP_CLOSEST_HIT = 21,   // in: o[3], d[3] (6) -> out: hit, t, u, v, id bits (5)
    PT_OP_FRAME = 22,         // in: n[3], w[3] (6) -> out: t[3], b[3], local[3], world(local)[3] (12)
    PT_OP_RNG = 23            // in: seed_u bits, seed_v bits (2) -> out: state0 bits, f0, f1, f2, state3 bits (5)
};

__global__ void __launch_bounds__(PT_BLOCK) pt_debug_kernel(const PtKernelParams P, int op, const float* __restrict__ in, int in_stride,
                                                           float* __restrict__ out, int out_stride, long long n)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* stack = lds + threadIdx.x;
    long long i = (long long)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* x = in + i * in_stride;
    float* y = out + i * out_stride;
    switch (op) {
    case PT_OP_SIN: { float s, c; sincos_(x[0], s, c); y[0] = s; break; }
    case PT_OP_COS: { float s, c; sincos_(x[0], s, c); y[0] = c; break; }
    case PT_OP_TAN: y[0] = tan_(x[0]); break;
    case PT_OP_ATAN: y[0] = atan_(x[0]); break;
    case PT_OP_ATAN2: y[0] = atan2_(x[0], x[1]); break;
    case PT_OP_ASIN: y[0] = asin_(x[0]); break;
    case PT_OP_LOG: y[0] = log_(x[0]); break;
    case PT_OP_EXP: y[0] = exp_(x[0]); break;
    case PT_OP_POW: y[0] = pow_(x[0], x[1]); break;
    case PT_OP_SQRT: y[0] = sqrt_(x[0]); break;
    case PT_OP_DIV: y[0] = x[0] / x[1]; break;
    case PT_OP_SAMPLE_DISNEY: {
        Material m = material_load(x);
        v3 wo = V(x[17], x[18], x[19]);
        uint32_t rng = __float_as_uint(x[20]);
        int lobe = __float_as_int(x[21]);
        v3 wi = vs(0.0f);
        float pdf = 0.0f;
        v3 f = sample_disney(m, wo, rng, wi, pdf, lobe);
        y[0] = f.x; y[1] = f.y; y[2] = f.z; y[3] = wi.x; y[4] = wi.y; y[5] = wi.z; y[6] = pdf;
        y[7] = __int_as_float(lobe); y[8] = __uint_as_float(rng);
        break;
    }
    case PT_OP_CLOSEST_HIT: {
        Hit h;
        Counters cn;
        closest_hit<false>(P, stack, V(x[0], x[1], x[2]), V(x[3], x[4], x[5]), h, cn);
        y[0] = h.slot >= 0 ? 1.0f : 0.0f; y[1] = h.t; y[2] = h.u; y[3] = h.v; y[4] = __int_as_float(h.slot >= 0 ? (P.hit_slot_mask != 0xffffffffu ? h.id >> 8 : h.id) : -1);
        break;
    }
    case PT_OP_FRAME: {
        v3 nn = V(x[0], x[1], x[2]), w = V(x[3], x[4], x[5]), t, b;
        onb(nn, t, b);
        v3 l = to_local(t, b, nn, w);
        v3 g = to_world(t, b, nn, l);
        y[0] = t.x; y[1] = t.y; y[2] = t.z; y[3] = b.x; y[4] = b.y; y[5] = b.z;
        y[6] = l.x; y[7] = l.y; y[8] = l.z; y[9] = g.x; y[10] = g.y; y[11] = g.z;
        break;
    }
    case PT_OP_RNG: {
        uint32_t st = rng_init(__float_as_uint(x[0]), __float_as_uint(x[1]));
        y[0] = __uint_as_float(st);
        y[1] = rng_next(st); y[2] = rng_next(st); y[3] = rng_next(st);
        y[4] = __uint_as_float(st);
        break;
    }
    default: break;
    }
}

extern "C" hipError_t pt_launch_debug(const PtKernelParams* p, int op, const float* in, int in_stride, float* out, int out_stride, long long n,
                                      size_t lds_bytes, hipStream_t stream)
{
    int grid = (int)((n + PT_BLOCK - 1) / PT_BLOCK);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(pt_debug_kernel, dim3(grid), dim3(PT_BLOCK), lds_bytes, stream, *p, op, in, in_stride, out, out_stride, n);
    return hipGetLastError();
}


extern "C" hipError_t pt_launch_render_lane(const PtKernelParams* p, int grid, size_t lds_bytes, hipStream_t stream, int count)
{
    if (count) hipLaunchKernelGGL(pt_render_kernel<true>, dim3(grid), dim3(PT_BLOCK), lds_bytes, stream, *p);
    else hipLaunchKernelGGL(pt_render_kernel<false>, dim3(grid), dim3(PT_BLOCK), lds_bytes, stream, *p);
    return hipGetLastError();
}

extern "C" hipError_t pt_lane_kernel_geometry(int count, int stack_entries, int* block, size_t* lds_bytes, int* ns, int* vgprs, int* max_blocks_per_cu)
{
    const void* fn = count ? (const void*)pt_render_kernel<true> : (const void*)pt_render_kernel<false>;
    *block = PT_BLOCK;
    *lds_bytes = (size_t)stack_entries * PT_BLOCK * 4;
    *ns = PT_BLOCK;
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, fn);
    if (e != hipSuccess) return e;
    *vgprs = fa.numRegs;
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, *block, *lds_bytes);
    *max_blocks_per_cu = nb;
    return e;
}

extern "C" int pt_debug_block(void) { return PT_BLOCK; }
