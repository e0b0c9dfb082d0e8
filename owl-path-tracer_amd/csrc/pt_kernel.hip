// pt_kernel.hip -- the render megakernel for gfx950 (MI355X): ray generation, BVH2 traversal with an
// LDS-resident per-lane stack, Moeller-Trumbore intersection, Disney BSDF sampling and the Russian-roulette
// bounce loop of the reference (path_tracer/src/device/device.cu:113-254), one persistent wave64 lane per pixel.
//
// Mapping (DESIGN.md "kernel"):
//   * a lane owns a pixel and walks its samples IN ORDER, because the reference threads one RNG stream per
//     pixel through all samples and bounces (device.cu:226-243) -- splitting samples would change the image;
//   * lanes pull pixels from a global queue (one wave-aggregated atomic per refill), so a wave never waits for
//     its slowest pixel: a finished lane immediately regenerates a path / fetches the next pixel;
//   * every trip of the outer loop traces exactly one ray per live lane (path regeneration), so the traversal
//     loop always runs with as many lanes as the wave has live pixels;
//   * the traversal stack is stack[level][lane] in LDS: bank = lane % 32 whatever the level, conflict-free.
#include "pt_device.h"
#include "pt_types.h"

using namespace ptd;

#define PT_BLOCK 256
#define PT_DONE (-1) // ~0: a leaf reference with count 0 never occurs

namespace {

struct Hit {
    float t, u, v;
    int slot; // leaf-order index of the triangle
    int id;   // global triangle id (tie-break + shading record)
};

__device__ __forceinline__ float fmin_hw(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fmax_hw(float a, float b) { return __builtin_fmaxf(a, b); }

// Slab test.  (bound - o) * inv keeps the error relative (2 roundings), the 1.0000004 factor covers it and
// the host builder pads every box (pt_bvh.cpp), so the test is conservative w.r.t. every hit tri_test can
// report: the closest hit does not depend on BVH topology or traversal order.  NaNs (0 * inf) are ignored
// by min/max exactly as in the oracle; signed zeros cannot change the comparison.
__device__ __forceinline__ bool box_test(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, v3 o, v3 inv,
                                         float tbest, float& tnear)
{
    float t0x = (bminx - o.x) * inv.x, t1x = (bmaxx - o.x) * inv.x;
    float t0y = (bminy - o.y) * inv.y, t1y = (bmaxy - o.y) * inv.y;
    float t0z = (bminz - o.z) * inv.z, t1z = (bmaxz - o.z) * inv.z;
    float tn = fmax_hw(fmax_hw(fmin_hw(t0x, t1x), fmin_hw(t0y, t1y)), fmax_hw(fmin_hw(t0z, t1z), kTMin));
    float tf = fmin_hw(fmin_hw(fmax_hw(t0x, t1x), fmax_hw(t0y, t1y)), fmin_hw(fmax_hw(t0z, t1z), tbest));
    tnear = tn;
    return tn <= tf * 1.0000004f;
}

// Moeller-Trumbore, two-sided, kTMin < t; ties in t go to the lower global id (order independent result).
__device__ __forceinline__ void tri_test(const PtTri* __restrict__ tris, int slot, v3 o, v3 d, Hit& h)
{
    const float4* tp = reinterpret_cast<const float4*>(tris + slot);
    float4 a = tp[0], b = tp[1], c = tp[2];
    v3 p0 = V(a.x, a.y, a.z), p1 = V(a.w, b.x, b.y), p2 = V(b.z, b.w, c.x);
    int id = __float_as_int(c.y);
    v3 e1 = p1 - p0, e2 = p2 - p0;
    v3 pv = cross(d, e2);
    float det = dot(e1, pv);
    float inv = 1.0f / det;
    v3 tv = o - p0;
    float u = dot(tv, pv) * inv;
    v3 qv = cross(tv, e1);
    float v = dot(d, qv) * inv;
    float t = dot(e2, qv) * inv;
    if (u >= 0.0f && v >= 0.0f && u + v <= 1.0f && t > kTMin && (t < h.t || (t == h.t && id < h.id))) {
        h.t = t; h.u = u; h.v = v; h.id = id; h.slot = slot;
    }
}

template <bool COUNT>
__device__ __forceinline__ void closest_hit(const PtKernelParams& P, uint32_t* stack, v3 o, v3 d, Hit& h, uint32_t& n_nodes, uint32_t& n_tris)
{
    h.t = kTMax; h.u = 0.0f; h.v = 0.0f; h.id = 0x7fffffff; h.slot = -1;
    const v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int sp = 0;
    int cur = P.root;
    const PtNode* __restrict__ nodes = P.nodes;
    const PtTri* __restrict__ tris = P.tris;
    for (;;) {
        while (cur >= 0) { // internal nodes: runs until every lane of the wave is at a leaf or finished
            const float4* np = reinterpret_cast<const float4*>(nodes + cur);
            float4 a = np[0], b = np[1], c = np[2];
            int4 ch = reinterpret_cast<const int4*>(np)[3];
            if (COUNT) ++n_nodes;
            float tl, tr;
            bool hl = box_test(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, h.t, tl);
            bool hr = box_test(b.z, b.w, c.x, c.y, c.z, c.w, o, inv, h.t, tr);
            if (hl && hr) {
                bool swap = tr < tl;
                int nearc = swap ? ch.y : ch.x;
                int farc = swap ? ch.x : ch.y;
                stack[sp * PT_BLOCK] = (uint32_t)farc;
                ++sp;
                cur = nearc;
            } else if (hl) {
                cur = ch.x;
            } else if (hr) {
                cur = ch.y;
            } else if (sp > 0) {
                --sp;
                cur = (int)stack[sp * PT_BLOCK];
            } else {
                cur = PT_DONE;
            }
        }
        if (cur == PT_DONE) break;
        uint32_t code = ~(uint32_t)cur;
        int first = (int)(code >> 3), count = (int)(code & 7u);
        for (int i = 0; i < count; ++i) {
            if (COUNT) ++n_tris;
            tri_test(tris, first + i, o, d, h);
        }
        if (sp == 0) break;
        --sp;
        cur = (int)stack[sp * PT_BLOCK];
    }
}

__device__ __forceinline__ v3 interp3(float bw, float bx, float by, v3 a, v3 b, v3 c)
{
    // (1-u-v)*a + u*b + v*c  (device.cu:59,72,86-89), evaluated as an fma chain
    return V(fma_(by, c.x, fma_(bx, b.x, bw * a.x)), fma_(by, c.y, fma_(bx, b.y, bw * a.y)), fma_(by, c.z, fma_(bx, b.z, bw * a.z)));
}

} // namespace

template <bool COUNT>
__global__ void __launch_bounds__(PT_BLOCK) pt_render_kernel(const PtKernelParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* stack = lds + threadIdx.x;                                                    // stack[level * PT_BLOCK]
    float* lmat = reinterpret_cast<float*>(lds + (size_t)P.stack_entries * PT_BLOCK);        // material table copy
    for (int i = threadIdx.x; i < P.n_materials * PT_MAT_STRIDE; i += PT_BLOCK) lmat[i] = P.materials[i];
    __syncthreads();

    const v3 cam_origin = V(P.cam[0], P.cam[1], P.cam[2]);
    const v3 cam_llc = V(P.cam[3], P.cam[4], P.cam[5]);
    const v3 cam_hor = V(P.cam[6], P.cam[7], P.cam[8]);
    const v3 cam_ver = V(P.cam[9], P.cam[10], P.cam[11]);
    const float inv_w = 0.0f; (void)inv_w;

    // per-pixel state
    uint32_t pid = 0, rng = 0;
    int px = 0, py = 0, s = 0;
    v3 color = vs(0.0f);
    bool have_pixel = false, done = false;
    // per-path state
    v3 org = vs(0.0f), dir = vs(0.0f), throughput = vs(1.0f);
    int depth = 0, lobe = kLobeNone, retries = 0;
    bool new_path = true, retry = false;
    Hit h;
    h.t = kTMax; h.u = h.v = 0.0f; h.slot = -1; h.id = 0x7fffffff;

    uint32_t c_rays = 0, c_nodes = 0, c_tris = 0, c_scat = 0, c_env = 0, c_samples = 0, c_retry = 0;

    for (;;) {
        // ---- pixel fetch: device.cu:224-228 per pixel, queue instead of a 2-D launch --------------------
        if (!have_pixel && !done) {
            uint32_t q = atomicAdd(P.queue_head, 1u); // hipcc aggregates this into one atomic per wave
            if (q < P.n_pixels) {
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
                s = 0;
                have_pixel = true;
                new_path = true;
            } else {
                done = true;
            }
        }
        if (__ballot(!done) == 0ull) break;
        if (done) continue;

        // ---- ray generation: device.cu:231-241 ---------------------------------------------------------
        if (new_path) {
            float rx = rng_next(rng);
            float ry = rng_next(rng);
            float su = ((float)px + rx) / (float)P.width;
            float sv = ((float)py + ry) / (float)P.height;
            org = cam_origin;
            dir = normalize(((cam_llc + cam_hor * su) + cam_ver * sv) - cam_origin);
            throughput = vs(1.0f);
            depth = 0;
            lobe = kLobeNone;
            retries = 0;
            new_path = false;
        }

        // ---- owl::traceRay: device.cu:133 (a NaN/Inf retry re-shades the same hit: same ray, same result) ----
        if (!retry) {
            closest_hit<COUNT>(P, stack, org, dir, h, c_nodes, c_tris);
        }
        if (COUNT) ++c_rays;
        retry = false;

        // ---- shade: device.cu:136-214 -------------------------------------------------------------------
        v3 radiance = vs(0.0f);
        bool end_path = false;
        if (h.slot < 0) { // miss: device.cu:136-148
            if (P.env_use_map && P.env_map.width > 0) {
                float tu, tv;
                uv_on_sphere(dir, tu, tv);
                radiance = radiance + tex_nearest(P.env_map.texels, P.env_map.width, P.env_map.height, tu, tv);
                if (COUNT) ++c_env;
            } else if (P.env_use_auto) {
                radiance = radiance + lerp3(vs(1.0f), V(0.5f, 0.7f, 1.0f), 0.5f * (dir.y + 1.0f));
            } else {
                radiance = radiance + V(P.env_color[0], P.env_color[1], P.env_color[2]);
            }
            radiance = radiance * P.env_intensity;
            end_path = true;
        } else {
            const float4* sp4 = reinterpret_cast<const float4*>(P.shade + h.id);
            float4 s0 = sp4[0], s1 = sp4[1], s2 = sp4[2], s3 = sp4[3];
            int mi = __float_as_int(s2.y);
            Material mat = material_default(); // device.cu:150-154
            int tex_slot = -1;
            if (mi >= 0) {
                const float* mp = lmat + mi * PT_MAT_STRIDE;
                mat = material_load(mp);
                tex_slot = __float_as_int(mp[17]);
            }
            if (mat.emission > 0.0f) { // device.cu:157-161: assignment, white, two-sided
                radiance = vs(mat.emission);
                end_path = true;
            } else {
                // attribute fetch: device.cu:164-173
                const float4* tp = reinterpret_cast<const float4*>(P.tris + h.slot);
                float4 a = tp[0], b = tp[1], c = tp[2];
                float bx = h.u, by = h.v;
                float bw = 1.0f - bx - by;
                v3 v_p = interp3(bw, bx, by, V(a.x, a.y, a.z), V(a.w, b.x, b.y), V(b.z, b.w, c.x));
                v3 v_n = normalize(interp3(bw, bx, by, V(s0.x, s0.y, s0.z), V(s0.w, s1.x, s1.y), V(s1.z, s1.w, s2.x)));
                if (tex_slot >= 0) { // device.cu:75-94
                    float tu = fma_(by, s3.z, fma_(bx, s3.x, bw * s2.z));
                    float tv = fma_(by, s3.w, fma_(bx, s3.y, bw * s2.w));
                    PtTexDesc td = P.textures[tex_slot];
                    mat.base_color = tex_nearest(td.texels, td.width, td.height, tu, tv);
                }
                if (COUNT) ++c_scat;

                // device.cu:176-190 (wo = -normalize(ray direction), device.cu:267-268)
                v3 wo = -normalize(dir);
                v3 T, B;
                onb(v_n, T, B);
                v3 local_wo = to_local(T, B, v_n, wo);
                v3 local_wi = vs(0.0f);
                float pdf = 0.0f;
                v3 f = sample_disney(mat, local_wo, rng, local_wi, pdf, lobe);
                v3 wi = to_world(T, B, v_n, local_wi);

                if (pdf < 1e-5f) { // device.cu:193
                    end_path = true;
                } else if (isinf_(f.x) || isinf_(f.y) || isinf_(f.z) || isnan_(f.x) || isnan_(f.y) || isnan_(f.z)) {
                    // device.cu:196-201: "--depth; continue" -> same ray again with fresh draws
                    if (COUNT) ++c_retry;
                    // safety net (also in the oracle): a hit whose BSDF is NaN for every draw would spin forever
                    if (++retries > 64) end_path = true;
                    else retry = true;
                } else {
                    retries = 0;
                    float aci = abs_(cos_theta(local_wi));
                    throughput = throughput * ((f * aci) / pdf); // device.cu:204
                    org = v_p;                                   // device.cu:205 (no normal offset)
                    dir = wi;
                    // device.cu:209-214: inverted, uncompensated Russian roulette
                    float beta_max = max_(throughput.x, max_(throughput.y, throughput.z));
                    if (lobe != kLobeGlass && depth > 3) {
                        float q = max_(0.05f, 1.0f - beta_max);
                        if (rng_next(rng) > q) end_path = true;
                    }
                    ++depth;
                    if (depth >= P.max_depth) end_path = true; // loop bound, device.cu:130
                }
            }
        }

        if (end_path) {
            color = color + radiance * throughput; // device.cu:217,243
            if (COUNT) ++c_samples;
            ++s;
            new_path = true;
            if (s == P.sample_count) { // pixel finished for this launch
                if (P.sample_begin + P.sample_count >= P.max_samples) {
                    v3 out = color * (1.0f / (float)P.max_samples); // device.cu:247
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
                have_pixel = false;
            }
        }
    }

    if (COUNT) {
        // wave reduction, one atomic per counter per wave
        unsigned long long v[7] = {c_samples, c_rays, c_nodes, c_tris, c_scat, c_env, c_retry};
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            v[k] = x;
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&P.counters->samples, v[0]);
            atomicAdd(&P.counters->rays, v[1]);
            atomicAdd(&P.counters->nodes, v[2]);
            atomicAdd(&P.counters->tris, v[3]);
            atomicAdd(&P.counters->scatters, v[4]);
            atomicAdd(&P.counters->env_misses, v[5]);
            atomicAdd(&P.counters->nan_retries, v[6]);
        }
    }
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
        uint32_t a = 0, b = 0;
        closest_hit<false>(P, stack, V(x[0], x[1], x[2]), V(x[3], x[4], x[5]), h, a, b);
        y[0] = h.slot >= 0 ? 1.0f : 0.0f; y[1] = h.t; y[2] = h.u; y[3] = h.v; y[4] = __int_as_float(h.slot >= 0 ? h.id : -1);
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

// ---- launchers (called from pt_api.cpp) --------------------------------------------------------------------

extern "C" hipError_t pt_launch_render(const PtKernelParams* p, int grid, size_t lds_bytes, hipStream_t stream, int count)
{
    if (count) hipLaunchKernelGGL(pt_render_kernel<true>, dim3(grid), dim3(PT_BLOCK), lds_bytes, stream, *p);
    else hipLaunchKernelGGL(pt_render_kernel<false>, dim3(grid), dim3(PT_BLOCK), lds_bytes, stream, *p);
    return hipGetLastError();
}

extern "C" hipError_t pt_launch_debug(const PtKernelParams* p, int op, const float* in, int in_stride, float* out, int out_stride, long long n,
                                      size_t lds_bytes, hipStream_t stream)
{
    int grid = (int)((n + PT_BLOCK - 1) / PT_BLOCK);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(pt_debug_kernel, dim3(grid), dim3(PT_BLOCK), lds_bytes, stream, *p, op, in, in_stride, out, out_stride, n);
    return hipGetLastError();
}

extern "C" hipError_t pt_kernel_attributes(int count, int* vgprs, int* sgprs, int* static_lds, int* max_blocks_per_cu, size_t lds_bytes)
{
    hipFuncAttributes fa;
    const void* fn = count ? (const void*)pt_render_kernel<true> : (const void*)pt_render_kernel<false>;
    hipError_t e = hipFuncGetAttributes(&fa, fn);
    if (e != hipSuccess) return e;
    *vgprs = fa.numRegs;
    *sgprs = 0;
    *static_lds = (int)fa.sharedSizeBytes;
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, PT_BLOCK, lds_bytes);
    *max_blocks_per_cu = nb;
    return e;
}

extern "C" int pt_kernel_block(void) { return PT_BLOCK; }
