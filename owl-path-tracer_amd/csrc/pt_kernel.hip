// pt_kernel.hip -- the render megakernel for gfx950 (MI355X): ray generation, BVH2 traversal with LDS-resident
// per-lane stacks, Moeller-Trumbore intersection, Disney BSDF sampling and the Russian-roulette bounce loop of the
// reference (path_tracer/src/device/device.cu:113-254).
//
// Two schedulers over the same device functions (DESIGN.md "kernel"):
//
//  pt_render_wave_kernel (default, "wavefront-scheduled"): one wave64 per workgroup owns PT_NS path slots in LDS
//    (a slot = one pixel in flight with its RNG stream, accumulators and current ray) plus two slot queues:
//    rays waiting for traversal and hits waiting for shading.  The wave alternates between
//      * a TRAVERSAL phase: every lane walks one ray; each step the wave executes either one BVH-node step or one
//        triangle test, whichever more lanes are waiting for (ballot majority); finished lanes retire their hit into
//        the hit queue and immediately pull the next ray, so the traversal loop runs with ~64 busy lanes;
//      * a SHADING phase over 64 queued hits at a time (miss / emitter / BSDF sample / Russian roulette / next
//        camera ray / next pixel), which emits the continuation rays back into the ray queue.
//    Lanes are workers, not pixel owners: a ray's lane is unrelated to the lane that shades its hit.  The per-pixel
//    sample order -- hence the reference's per-pixel RNG stream (device.cu:226-243) -- is preserved because a slot
//    has at most one ray in flight.
//
//  pt_render_kernel (option kernel=1): the simple persistent lane-per-pixel form kept for A/B measurements.
//
// The traversal stack is stack[level][lane] in LDS: bank = lane % 32 whatever the level, conflict-free.
#include "pt_device.h"
#include "pt_types.h"

using namespace ptd;

#define PT_BLOCK 256
#define PT_DONE (-1) // ~0: a leaf reference with count 0 never occurs
#define PT_WAVE 64
#define PT_NS 128          // path slots per wave (power of two)
#define PT_RETIRE_MIN 8    // finished lanes that trigger a retire/refill pass

namespace {

struct Hit {
    float t, u, v;
    int slot; // leaf-order index of the triangle
    int id;   // global triangle id (tie-break + shading record)
};

struct Counters {
    uint32_t rays = 0, nodes = 0, tris = 0, scat = 0, env = 0, samples = 0, retry = 0;
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

// One BVH-node step for a lane: test both children, descend into the nearer hit child, push the other.
template <int STRIDE>
__device__ __forceinline__ void node_step(const PtNode* __restrict__ nodes, uint32_t* stack, v3 o, v3 inv, float tbest, int& cur, int& sp)
{
    const float4* np = reinterpret_cast<const float4*>(nodes + cur);
    float4 a = np[0], b = np[1], c = np[2];
    int4 ch = reinterpret_cast<const int4*>(np)[3];
    float tl, tr;
    bool hl = box_test(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, tbest, tl);
    bool hr = box_test(b.z, b.w, c.x, c.y, c.z, c.w, o, inv, tbest, tr);
    if (hl && hr) {
        bool swap = tr < tl;
        int nearc = swap ? ch.y : ch.x;
        int farc = swap ? ch.x : ch.y;
        stack[sp * STRIDE] = (uint32_t)farc;
        ++sp;
        cur = nearc;
    } else if (hl) {
        cur = ch.x;
    } else if (hr) {
        cur = ch.y;
    } else if (sp > 0) {
        --sp;
        cur = (int)stack[sp * STRIDE];
    } else {
        cur = PT_DONE;
    }
}

template <bool COUNT>
__device__ __forceinline__ void closest_hit(const PtKernelParams& P, uint32_t* stack, v3 o, v3 d, Hit& h, Counters& cn)
{
    h.t = kTMax; h.u = 0.0f; h.v = 0.0f; h.id = 0x7fffffff; h.slot = -1;
    const v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int sp = 0;
    int cur = P.root;
    const PtNode* __restrict__ nodes = P.nodes;
    const PtTri* __restrict__ tris = P.tris;
    for (;;) {
        while (cur >= 0) { // internal nodes: runs until every lane of the wave is at a leaf or finished
            if (COUNT) ++cn.nodes;
            node_step<PT_BLOCK>(nodes, stack, o, inv, h.t, cur, sp);
        }
        if (cur == PT_DONE) break;
        uint32_t code = ~(uint32_t)cur;
        int first = (int)(code >> 3), count = (int)(code & 7u);
        for (int i = 0; i < count; ++i) {
            if (COUNT) ++cn.tris;
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

struct PathState {
    uint32_t rng;
    v3 org, dir, throughput;
    int depth, lobe, retries;
};

enum { SR_CONTINUE = 0, SR_END = 1, SR_RETRY = 2 };

// One iteration of the reference's path loop after owl::traceRay returned (device.cu:136-214).
// tslot < 0 = miss.  Returns SR_CONTINUE (ps.org/dir/throughput/depth updated, trace again), SR_END (radiance set; the
// sample contributes radiance * throughput, device.cu:217) or SR_RETRY (NaN/Inf f: shade the same hit again, :196-201).
template <bool COUNT>
__device__ __forceinline__ int shade_hit(const PtKernelParams& P, const float* mats, int tslot, float hu, float hv, PathState& ps, v3& radiance,
                                         Counters& cn)
{
    radiance = vs(0.0f);
    if (tslot < 0) { // miss: device.cu:136-148
        if (P.env_use_map && P.env_map.width > 0) {
            float tu, tv;
            uv_on_sphere(ps.dir, tu, tv);
            radiance = radiance + tex_nearest(P.env_map.texels, P.env_map.width, P.env_map.height, tu, tv);
            if (COUNT) ++cn.env;
        } else if (P.env_use_auto) {
            radiance = radiance + lerp3(vs(1.0f), V(0.5f, 0.7f, 1.0f), 0.5f * (ps.dir.y + 1.0f));
        } else {
            radiance = radiance + V(P.env_color[0], P.env_color[1], P.env_color[2]);
        }
        radiance = radiance * P.env_intensity;
        return SR_END;
    }
    const float4* tp = reinterpret_cast<const float4*>(P.tris + tslot);
    float4 a = tp[0], b = tp[1], c = tp[2];
    const int tid = __float_as_int(c.y);
    const float4* sp4 = reinterpret_cast<const float4*>(P.shade + tid);
    float4 s0 = sp4[0], s1 = sp4[1], s2 = sp4[2], s3 = sp4[3];
    int mi = __float_as_int(s2.y);
    Material mat = material_default(); // device.cu:150-154
    int tex_slot = -1;
    if (mi >= 0) {
        const float* mp = mats + mi * PT_MAT_STRIDE;
        mat = material_load(mp);
        tex_slot = __float_as_int(mp[17]);
    }
    if (mat.emission > 0.0f) { // device.cu:157-161: assignment, white, two-sided
        radiance = vs(mat.emission);
        return SR_END;
    }
    // attribute fetch: device.cu:164-173
    float bx = hu, by = hv;
    float bw = 1.0f - bx - by;
    v3 v_p = interp3(bw, bx, by, V(a.x, a.y, a.z), V(a.w, b.x, b.y), V(b.z, b.w, c.x));
    v3 v_n = normalize(interp3(bw, bx, by, V(s0.x, s0.y, s0.z), V(s0.w, s1.x, s1.y), V(s1.z, s1.w, s2.x)));
    if (tex_slot >= 0) { // device.cu:75-94
        float tu = fma_(by, s3.z, fma_(bx, s3.x, bw * s2.z));
        float tv = fma_(by, s3.w, fma_(bx, s3.y, bw * s2.w));
        PtTexDesc td = P.textures[tex_slot];
        mat.base_color = tex_nearest(td.texels, td.width, td.height, tu, tv);
    }
    if (COUNT) ++cn.scat;

    // device.cu:176-190 (wo = -normalize(ray direction), device.cu:267-268)
    v3 wo = -normalize(ps.dir);
    v3 T, B;
    onb(v_n, T, B);
    v3 local_wo = to_local(T, B, v_n, wo);
    v3 local_wi = vs(0.0f);
    float pdf = 0.0f;
    v3 f = sample_disney(mat, local_wo, ps.rng, local_wi, pdf, ps.lobe);
    v3 wi = to_world(T, B, v_n, local_wi);

    if (pdf < 1e-5f) return SR_END; // device.cu:193
    if (isinf_(f.x) || isinf_(f.y) || isinf_(f.z) || isnan_(f.x) || isnan_(f.y) || isnan_(f.z)) {
        // device.cu:196-201: "--depth; continue" -> same ray again with fresh draws
        if (COUNT) ++cn.retry;
        // safety net (also in the oracle): a hit whose BSDF is NaN for every draw would spin forever
        if (++ps.retries > 64) return SR_END;
        return SR_RETRY;
    }
    ps.retries = 0;
    float aci = abs_(cos_theta(local_wi));
    ps.throughput = ps.throughput * ((f * aci) / pdf); // device.cu:204
    ps.org = v_p;                                      // device.cu:205 (no normal offset)
    ps.dir = wi;
    // device.cu:209-214: inverted, uncompensated Russian roulette
    float beta_max = max_(ps.throughput.x, max_(ps.throughput.y, ps.throughput.z));
    if (ps.lobe != kLobeGlass && ps.depth > 3) {
        float q = max_(0.05f, 1.0f - beta_max);
        if (rng_next(ps.rng) > q) return SR_END;
    }
    ++ps.depth;
    if (ps.depth >= P.max_depth) return SR_END; // loop bound, device.cu:130 (radiance stays 0)
    return SR_CONTINUE;
}

// Camera ray for the next sample of pixel (px, py): device.cu:231-241
__device__ __forceinline__ void gen_camera_ray(const PtKernelParams& P, int px, int py, PathState& ps)
{
    float rx = rng_next(ps.rng);
    float ry = rng_next(ps.rng);
    float su = ((float)px + rx) / (float)P.width;
    float sv = ((float)py + ry) / (float)P.height;
    const v3 cam_origin = V(P.cam[0], P.cam[1], P.cam[2]);
    const v3 cam_llc = V(P.cam[3], P.cam[4], P.cam[5]);
    const v3 cam_hor = V(P.cam[6], P.cam[7], P.cam[8]);
    const v3 cam_ver = V(P.cam[9], P.cam[10], P.cam[11]);
    ps.org = cam_origin;
    ps.dir = normalize(((cam_llc + cam_hor * su) + cam_ver * sv) - cam_origin);
    ps.throughput = vs(1.0f);
    ps.depth = 0;
    ps.lobe = kLobeNone;
    ps.retries = 0;
}

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

template <bool COUNT>
__device__ __forceinline__ void flush_counters(const PtKernelParams& P, const Counters& cn)
{
    if (!COUNT) return;
    unsigned long long v[7] = {cn.samples, cn.rays, cn.nodes, cn.tris, cn.scat, cn.env, cn.retry};
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

__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }
// number of set bits of m below this lane
__device__ __forceinline__ int rank_in(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

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

// =====================================================================================================================
// v2: wavefront-scheduled megakernel, one wave64 per workgroup (default)
// =====================================================================================================================

// slot fields (SoA in LDS: field f of slot s at slotf[f * PT_NS + s])
enum {
    F_PIX = 0, F_RNG, F_PACK, F_RETRY, F_COLX, F_COLY, F_COLZ, F_THRX, F_THRY, F_THRZ, F_ORGX, F_ORGY, F_ORGZ, F_DIRX, F_DIRY, F_DIRZ,
    F_HU, F_HV, F_HT, F_NFIELDS
};
// F_PACK: bits 0-19 sample index within the launch, 20-25 depth, 26-28 lobe+1, 29 fresh
#define PT_PACK(s, depth, lobe, fresh) ((uint32_t)(s) | ((uint32_t)(depth) << 20) | ((uint32_t)((lobe) + 1) << 26) | ((uint32_t)(fresh) << 29))
// park area fields (per lane)
enum { K_PSLOT = 0, K_CUR, K_SP, K_BT, K_BU, K_BV, K_BSLOT, K_BID, K_NFIELDS };

static inline int pt_wave_lds_words(int stack_entries) { return stack_entries * PT_WAVE + K_NFIELDS * PT_WAVE + F_NFIELDS * PT_NS + 2 * PT_NS; }

template <bool COUNT>
__global__ void __launch_bounds__(PT_WAVE) pt_render_wave_kernel(const PtKernelParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    uint32_t* stack = lds + lane;                                     // stack[level * 64]
    uint32_t* park = lds + P.stack_entries * PT_WAVE + lane;          // park[field * 64]
    uint32_t* slotf = lds + (P.stack_entries + K_NFIELDS) * PT_WAVE;  // slotf[field * PT_NS + slot]
    uint32_t* rayq = slotf + F_NFIELDS * PT_NS;
    uint32_t* hitq = rayq + PT_NS;
    const PtNode* __restrict__ nodes = P.nodes;
    const PtTri* __restrict__ tris = P.tris;

#define SF(f, s) slotf[(f) * PT_NS + (s)]
#define SFF(f, s) __uint_as_float(slotf[(f) * PT_NS + (s)])

    // every slot starts "fresh" (needs a pixel) and sits in the hit queue so that the first shading passes start them
    for (int i = lane; i < PT_NS; i += PT_WAVE) {
        hitq[i] = (uint32_t)i;
        SF(F_PACK, i) = PT_PACK(0, 0, kLobeNone, 1);
    }
    park[K_PSLOT * PT_WAVE] = 0xffffffffu;
    int ray_head = 0, ray_count = 0, hit_head = 0, hit_count = PT_NS, n_dead = 0, n_parked = 0;
    Counters cn;

    while (n_dead < PT_NS) {
        const bool do_shade = hit_count >= PT_WAVE || (hit_count > 0 && ray_count == 0 && n_parked == 0);
        if (do_shade) {
            // ======================= SHADING PHASE: up to 64 queued hits =======================================
            const int n = hit_count < PT_WAVE ? hit_count : PT_WAVE;
            const bool mine = lane < n;
            int ps_slot = 0;
            bool to_ray = false, to_hit = false, died = false;
            if (mine) {
                ps_slot = (int)hitq[(hit_head + lane) & (PT_NS - 1)];
                uint32_t pack = SF(F_PACK, ps_slot);
                bool fresh = (pack >> 29) & 1u;
                int s = (int)(pack & 0xfffffu);
                PathState ps;
                uint32_t pid = 0;
                int px = 0, py = 0;
                v3 color = vs(0.0f);
                bool need_gen = fresh;
                bool have_pixel = !fresh;
                if (!fresh) {
                    pid = SF(F_PIX, ps_slot);
                    px = (int)(pid % (uint32_t)P.width);
                    py = (int)(pid / (uint32_t)P.width);
                    ps.rng = SF(F_RNG, ps_slot);
                    ps.depth = (int)((pack >> 20) & 63u);
                    ps.lobe = (int)((pack >> 26) & 7u) - 1;
                    ps.retries = (int)SF(F_RETRY, ps_slot);
                    ps.throughput = V(SFF(F_THRX, ps_slot), SFF(F_THRY, ps_slot), SFF(F_THRZ, ps_slot));
                    ps.org = V(SFF(F_ORGX, ps_slot), SFF(F_ORGY, ps_slot), SFF(F_ORGZ, ps_slot));
                    ps.dir = V(SFF(F_DIRX, ps_slot), SFF(F_DIRY, ps_slot), SFF(F_DIRZ, ps_slot));
                    color = V(SFF(F_COLX, ps_slot), SFF(F_COLY, ps_slot), SFF(F_COLZ, ps_slot));
                    if (COUNT) ++cn.rays;
                    v3 radiance;
                    int r = shade_hit<COUNT>(P, P.materials, (int)SF(F_HT, ps_slot), SFF(F_HU, ps_slot), SFF(F_HV, ps_slot), ps, radiance, cn);
                    if (r == SR_RETRY) {
                        to_hit = true; // same hit, fresh draws (device.cu:196-201)
                    } else if (r == SR_END) {
                        color = color + radiance * ps.throughput; // device.cu:217,243
                        if (COUNT) ++cn.samples;
                        ++s;
                        need_gen = true;
                        if (s == P.sample_count) {
                            finish_pixel(P, pid, px, py, ps.rng, color);
                            have_pixel = false;
                        }
                    } else {
                        to_ray = true;
                    }
                }
                if (need_gen) {
                    if (!have_pixel) {
                        have_pixel = fetch_pixel(P, pid, px, py, ps.rng, color);
                        s = 0;
                    }
                    if (have_pixel) {
                        gen_camera_ray(P, px, py, ps);
                        to_ray = true;
                    } else {
                        died = true;
                    }
                }
                if (!died) {
                    SF(F_PIX, ps_slot) = pid;
                    SF(F_RNG, ps_slot) = ps.rng;
                    SF(F_PACK, ps_slot) = PT_PACK(s, ps.depth, ps.lobe, 0);
                    SF(F_RETRY, ps_slot) = (uint32_t)ps.retries;
                    SF(F_COLX, ps_slot) = __float_as_uint(color.x);
                    SF(F_COLY, ps_slot) = __float_as_uint(color.y);
                    SF(F_COLZ, ps_slot) = __float_as_uint(color.z);
                    SF(F_THRX, ps_slot) = __float_as_uint(ps.throughput.x);
                    SF(F_THRY, ps_slot) = __float_as_uint(ps.throughput.y);
                    SF(F_THRZ, ps_slot) = __float_as_uint(ps.throughput.z);
                    SF(F_ORGX, ps_slot) = __float_as_uint(ps.org.x);
                    SF(F_ORGY, ps_slot) = __float_as_uint(ps.org.y);
                    SF(F_ORGZ, ps_slot) = __float_as_uint(ps.org.z);
                    SF(F_DIRX, ps_slot) = __float_as_uint(ps.dir.x);
                    SF(F_DIRY, ps_slot) = __float_as_uint(ps.dir.y);
                    SF(F_DIRZ, ps_slot) = __float_as_uint(ps.dir.z);
                }
            }
            hit_head = (hit_head + n) & (PT_NS - 1);
            hit_count -= n;
            const unsigned long long m_ray = __ballot(to_ray), m_hit = __ballot(to_hit), m_dead = __ballot(died);
            if (to_ray) rayq[(ray_head + ray_count + rank_in(m_ray)) & (PT_NS - 1)] = (uint32_t)ps_slot;
            if (to_hit) hitq[(hit_head + hit_count + rank_in(m_hit)) & (PT_NS - 1)] = (uint32_t)ps_slot;
            ray_count += popc64(m_ray);
            hit_count += popc64(m_hit);
            n_dead += popc64(m_dead);
        } else {
            // ======================= TRAVERSAL PHASE ==============================================================
            int pslot = (int)park[K_PSLOT * PT_WAVE];
            int cur = PT_DONE, sp = 0;
            Hit h;
            h.t = kTMax; h.u = h.v = 0.0f; h.slot = -1; h.id = 0x7fffffff;
            v3 o = vs(0.0f), d = vs(1.0f), inv = vs(1.0f);
            if (pslot >= 0) { // resume a traversal parked by the previous phase switch
                cur = (int)park[K_CUR * PT_WAVE];
                sp = (int)park[K_SP * PT_WAVE];
                h.t = __uint_as_float(park[K_BT * PT_WAVE]);
                h.u = __uint_as_float(park[K_BU * PT_WAVE]);
                h.v = __uint_as_float(park[K_BV * PT_WAVE]);
                h.slot = (int)park[K_BSLOT * PT_WAVE];
                h.id = (int)park[K_BID * PT_WAVE];
                o = V(SFF(F_ORGX, pslot), SFF(F_ORGY, pslot), SFF(F_ORGZ, pslot));
                d = V(SFF(F_DIRX, pslot), SFF(F_DIRY, pslot), SFF(F_DIRZ, pslot));
                inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            }
            bool first = true;
            for (;;) {
                const unsigned long long m_done = __ballot(pslot >= 0 && cur == PT_DONE);
                const unsigned long long m_node = __ballot(pslot >= 0 && cur >= 0);
                const unsigned long long m_leaf = __ballot(pslot >= 0 && cur < PT_DONE);
                const int n_done = popc64(m_done);
                if (first || n_done >= PT_RETIRE_MIN || (m_node | m_leaf) == 0ull) {
                    first = false;
                    // ---- retire finished rays into the hit queue ----
                    if (pslot >= 0 && cur == PT_DONE) {
                        SF(F_HU, pslot) = __float_as_uint(h.u);
                        SF(F_HV, pslot) = __float_as_uint(h.v);
                        SF(F_HT, pslot) = (uint32_t)h.slot;
                        hitq[(hit_head + hit_count + rank_in(m_done)) & (PT_NS - 1)] = (uint32_t)pslot;
                        pslot = -1;
                    }
                    hit_count += n_done;
                    // ---- refill idle lanes from the ray queue ----
                    const unsigned long long m_idle = __ballot(pslot < 0);
                    const int n_idle = popc64(m_idle);
                    const int take = n_idle < ray_count ? n_idle : ray_count;
                    if (take > 0) {
                        const int rk = rank_in(m_idle);
                        if (pslot < 0 && rk < take) {
                            pslot = (int)rayq[(ray_head + rk) & (PT_NS - 1)];
                            o = V(SFF(F_ORGX, pslot), SFF(F_ORGY, pslot), SFF(F_ORGZ, pslot));
                            d = V(SFF(F_DIRX, pslot), SFF(F_DIRY, pslot), SFF(F_DIRZ, pslot));
                            inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                            cur = P.root;
                            sp = 0;
                            h.t = kTMax; h.u = 0.0f; h.v = 0.0f; h.slot = -1; h.id = 0x7fffffff;
                        }
                        ray_head = (ray_head + take) & (PT_NS - 1);
                        ray_count -= take;
                    }
                    const unsigned long long m_busy = __ballot(pslot >= 0);
                    if (m_busy == 0ull) break;                           // nothing in flight (the ray queue is empty too)
                    if (ray_count == 0 && hit_count >= PT_WAVE) break;   // a full shading batch is waiting and no ray is queued
                    continue;
                }
                // ---- one step for the majority: a BVH node step or a triangle test ----
                if (popc64(m_node) >= popc64(m_leaf)) {
                    if (pslot >= 0 && cur >= 0) {
                        if (COUNT) ++cn.nodes;
                        node_step<PT_WAVE>(nodes, stack, o, inv, h.t, cur, sp);
                    }
                } else {
                    if (pslot >= 0 && cur < PT_DONE) {
                        uint32_t code = ~(uint32_t)cur;
                        int firstt = (int)(code >> 3), count = (int)(code & 7u);
                        if (COUNT) ++cn.tris;
                        tri_test(tris, firstt, o, d, h);
                        if (count > 1) {
                            cur = (int)~(((uint32_t)(firstt + 1) << 3) | (uint32_t)(count - 1));
                        } else if (sp > 0) {
                            --sp;
                            cur = (int)stack[sp * PT_WAVE];
                        } else {
                            cur = PT_DONE;
                        }
                    }
                }
            }
            // ---- park unfinished traversals until the next traversal phase ----
            park[K_PSLOT * PT_WAVE] = (uint32_t)pslot;
            if (pslot >= 0) {
                park[K_CUR * PT_WAVE] = (uint32_t)cur;
                park[K_SP * PT_WAVE] = (uint32_t)sp;
                park[K_BT * PT_WAVE] = __float_as_uint(h.t);
                park[K_BU * PT_WAVE] = __float_as_uint(h.u);
                park[K_BV * PT_WAVE] = __float_as_uint(h.v);
                park[K_BSLOT * PT_WAVE] = (uint32_t)h.slot;
                park[K_BID * PT_WAVE] = (uint32_t)h.id;
            }
            n_parked = popc64(__ballot(pslot >= 0));
        }
    }
#undef SF
#undef SFF
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

extern "C" hipError_t pt_launch_render(const PtKernelParams* p, int variant, int grid, size_t lds_bytes, hipStream_t stream, int count)
{
    if (variant == 1) {
        if (count) hipLaunchKernelGGL(pt_render_kernel<true>, dim3(grid), dim3(PT_BLOCK), lds_bytes, stream, *p);
        else hipLaunchKernelGGL(pt_render_kernel<false>, dim3(grid), dim3(PT_BLOCK), lds_bytes, stream, *p);
    } else {
        if (count) hipLaunchKernelGGL(pt_render_wave_kernel<true>, dim3(grid), dim3(PT_WAVE), lds_bytes, stream, *p);
        else hipLaunchKernelGGL(pt_render_wave_kernel<false>, dim3(grid), dim3(PT_WAVE), lds_bytes, stream, *p);
    }
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

// Launch geometry of a render variant: block size, dynamic LDS bytes, pixels a block keeps in flight, registers, occupancy.
extern "C" hipError_t pt_kernel_geometry(int variant, int count, int stack_entries, int* block, size_t* lds_bytes, int* pixels_per_block,
                                         int* vgprs, int* max_blocks_per_cu)
{
    const void* fn;
    if (variant == 1) {
        fn = count ? (const void*)pt_render_kernel<true> : (const void*)pt_render_kernel<false>;
        *block = PT_BLOCK;
        *lds_bytes = (size_t)stack_entries * PT_BLOCK * 4;
        *pixels_per_block = PT_BLOCK;
    } else {
        fn = count ? (const void*)pt_render_wave_kernel<true> : (const void*)pt_render_wave_kernel<false>;
        *block = PT_WAVE;
        *lds_bytes = (size_t)pt_wave_lds_words(stack_entries) * 4;
        *pixels_per_block = PT_NS;
    }
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
