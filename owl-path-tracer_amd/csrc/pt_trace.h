// pt_trace.h -- device code shared by the render kernels (pt_kernel.hip: the wavefront-scheduled product kernel; pt_kernel_aux.hip: the
// lane-per-pixel variant and the validation kernel): closest-hit building blocks (slab test, Moeller-Trumbore, binary node step), the
// reference's path loop body after owl::traceRay (shade_hit: device.cu:136-214), camera rays (device.cu:231-241) and the work counters.
// Everything is in an anonymous namespace: each translation unit gets its own inlined copy.
#pragma once
#include <cstdlib>

#include "pt_device.h"
#include "pt_types.h"

using namespace ptd;

#define PT_BLOCK 256 // threads per workgroup of the lane-per-pixel and validation kernels
#define PT_DONE (-1) // ~0: a leaf reference with count 0 never occurs
#define PT_WAVE 64

namespace {

struct Hit {
    float t, u, v;
    int slot; // leaf-order index of the triangle
    int id;   // global triangle id (tie-break + shading record)
};

struct Counters {
    uint32_t rays = 0, nodes = 0, tris = 0, scat = 0, env = 0, samples = 0, retry = 0;
    unsigned long long cyc[8] = {}; // COUNT build: shader-clock cycles per phase {node steps, tri steps, retire, hit pass, miss pass, park/resume, sleep, total}
    uint32_t sched[32] = {}; // wave-uniform scheduler census (wavefront kernel)
    uint32_t depth[4] = {};  // per lane: pushes, pushes at stack depth >= 8 / 12 / 16
    uint32_t grp[6] = {};    // wave-uniform census of the group walk: phases, iterations, busy groups, node groups, leaf groups, rays
    unsigned long long grp_cyc = 0;
    uint32_t lobe[16] = {};  // wave-uniform census of the hit passes by sampled lobe (PtCounters::lobes)
    uint32_t cull_nohit = 0, cull_beyond = 0, leaf_noimp = 0; // per lane: quad steps that enter no child; of those: node beyond the best hit; leaf steps that do not improve the hit
};

// Pointers read out of the parameter block are generic; every buffer is hipMalloc memory, so all accesses below go through
// address_space(1) pointers: global_load/global_store (vmcnt only) instead of flat_* (vmcnt + lgkmcnt, aperture check).
#define PT_AS1 __attribute__((address_space(1)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ T PT_AS1* gp(T* p) { return (T PT_AS1*)p; }
__device__ __forceinline__ f32x4 ldg4(const void* base, size_t byte_off) { return *(const f32x4 PT_AS1*)((const char PT_AS1*)base + byte_off); }
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

// The same test with the slab distances as fma(bound, inv, -(o * inv)): see node4_step (pt_kernel.hip) for the error bound and when the
// host asks for the subtracting form instead.
__device__ __forceinline__ bool box_test_fma(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, v3 o, v3 inv, float tbest, float& tnear)
{
    const float nx = -(o.x * inv.x), ny = -(o.y * inv.y), nz = -(o.z * inv.z);
    float t0x = fma_(bminx, inv.x, nx), t1x = fma_(bmaxx, inv.x, nx);
    float t0y = fma_(bminy, inv.y, ny), t1y = fma_(bmaxy, inv.y, ny);
    float t0z = fma_(bminz, inv.z, nz), t1z = fma_(bmaxz, inv.z, nz);
    float tn = fmax_hw(fmax_hw(fmin_hw(t0x, t1x), fmin_hw(t0y, t1y)), fmax_hw(fmin_hw(t0z, t1z), kTMin));
    float tf = fmin_hw(fmin_hw(fmax_hw(t0x, t1x), fmax_hw(t0y, t1y)), fmin_hw(fmax_hw(t0z, t1z), tbest));
    tnear = tn;
    return tn <= tf * 1.0000004f;
}

// Reciprocal direction of a ray FOR THE SLAB TESTS of the wavefront kernel: v_rcp_f32 (1 ulp) instead of the correctly rounded division
// (ten instructions each, three per ray, in the refill step every lane runs through).  A slab distance is then off by a relative
// 2^-23 at most, i.e. a plane seems displaced by < 1.2e-7 x its distance from the ray origin - every box is padded by 1e-5 x the scene
// extent (pt_bvh.cpp), eighty times that - so the test stays conservative with respect to every hit the triangle test can report,
// and the triangle test itself (which decides t, u, v and the image) does not use it.
// The reciprocal is CLAMPED to +-PT_INV_MAX: a direction component of exactly 0 (a horizontal camera's middle row: the jitter is absorbed
// when `lower_left + v * vertical - origin` is formed, ~3e-5 of that row's samples; mirror bounces keep it) must not become inf in the
// fma form of the slab test - fma(plane, inf, -(o * inf)) is NaN only where plane and o have the same sign and -inf otherwise, which
// culled every box that straddles 0 on that axis, the root included (3 pixels of C2's 262 144 differed from the oracle at 256 spp:
// profiles/r04_notes.md 7; found by the whole-frame comparison at full spp).  With a finite reciprocal both forms give
// (plane - o) * 1e18 up to rounding: no constraint from a slab the origin is inside of, a cull for one it is outside of, and a ray that
// runs within 1e-8 of a box face - which is padded, so >= 1e-5 extents away from every triangle of the box - may be culled either way.
#ifndef PT_FAST_RAY_INV
#define PT_FAST_RAY_INV 1
#endif
#ifndef PT_INV_MAX
#define PT_INV_MAX 1e18f // (tests build a variant with infinity here to see the regression tests fail)
#endif
__device__ __forceinline__ v3 ray_inv(v3 d)
{
#if PT_FAST_RAY_INV
    const float x = __builtin_amdgcn_rcpf(d.x), y = __builtin_amdgcn_rcpf(d.y), z = __builtin_amdgcn_rcpf(d.z);
#else
    const float x = 1.0f / d.x, y = 1.0f / d.y, z = 1.0f / d.z;
#endif
    return V(__builtin_amdgcn_fmed3f(x, -PT_INV_MAX, PT_INV_MAX), __builtin_amdgcn_fmed3f(y, -PT_INV_MAX, PT_INV_MAX), __builtin_amdgcn_fmed3f(z, -PT_INV_MAX, PT_INV_MAX));
}

// Moeller-Trumbore, two-sided, kTMin < t; ties in t go to the lower global id (order independent result).
// tri_eval works on a record that is already in registers, so that a leaf step can issue the loads of all its triangles
// before the first test (one memory round trip per leaf instead of one per triangle).
__device__ __forceinline__ void tri_eval(const f32x4 a, const f32x4 b, const f32x4 c, int slot, v3 o, v3 d, Hit& h)
{
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
__device__ __forceinline__ void tri_test(const PtTri* __restrict__ tris, int slot, v3 o, v3 d, Hit& h)
{
    const size_t tb = (size_t)(uint32_t)slot * sizeof(PtTri);
    const f32x4 a = ldg4(tris, tb), b = ldg4(tris, tb + 16), c = ldg4(tris, tb + 32);
    tri_eval(a, b, c, slot, o, d, h);
}
// All triangles of one leaf: the records of the first PT_LEAF_PREFETCH triangles are requested together, the tests follow.
#ifndef PT_LEAF_PREFETCH
#define PT_LEAF_PREFETCH 4
#endif
__device__ __forceinline__ void leaf_test(const PtTri* __restrict__ tris, int first, int count, v3 o, v3 d, Hit& h)
{
#if PT_LEAF_PREFETCH == 0
    for (int k = 0; k < count; ++k) tri_test(tris, first + k, o, d, h);
#else
    f32x4 ra[PT_LEAF_PREFETCH], rb[PT_LEAF_PREFETCH], rc[PT_LEAF_PREFETCH];
    // unconditional loads (lanes with fewer triangles re-read their last one): one basic block, so all requests are in flight
    // before the first wait; with a per-triangle predicate the compiler waits inside each predicated block
#pragma unroll
    for (int k = 0; k < PT_LEAF_PREFETCH; ++k) {
        const int kk = k < count ? k : count - 1;
        const size_t tb = (size_t)(uint32_t)(first + kk) * sizeof(PtTri);
        ra[k] = ldg4(tris, tb); rb[k] = ldg4(tris, tb + 16); rc[k] = ldg4(tris, tb + 32);
    }
#pragma unroll
    for (int k = 0; k < PT_LEAF_PREFETCH; ++k) {
        if (k < count) tri_eval(ra[k], rb[k], rc[k], first + k, o, d, h);
    }
    for (int k = PT_LEAF_PREFETCH; k < count; ++k) tri_test(tris, first + k, o, d, h); // leaf_size > PT_LEAF_PREFETCH only
#endif
}

// One BVH-node step for a lane: test both children, descend into the nearer hit child, push the other.
// Stack entry i lives in LDS (stack[i * STRIDE]) for i < LDS_ENTRIES, else in the lane's HBM overflow column
// (ovf[(i - LDS_ENTRIES) * STRIDE]): on C4 0.006 % of the binary walk's pushes go deeper than 12 (census of the instrumented build: profiles/r01_summary.md), so a 12-entry LDS
// stack halves the LDS a wave needs for BVHs of any depth.  LDS_ENTRIES = 0x7fffffff: everything in LDS.
template <int STRIDE, int LDS_ENTRIES>
__device__ __forceinline__ void stack_push(uint32_t* stack, uint32_t PT_AS1* ovf, int sp, uint32_t v)
{
    if (LDS_ENTRIES == 0x7fffffff || sp < LDS_ENTRIES) stack[sp * STRIDE] = v;
    else ovf[(sp - LDS_ENTRIES) * STRIDE] = v;
}
template <int STRIDE, int LDS_ENTRIES>
__device__ __forceinline__ uint32_t stack_pop(uint32_t* stack, uint32_t PT_AS1* ovf, int sp)
{
    if (LDS_ENTRIES == 0x7fffffff || sp < LDS_ENTRIES) return stack[sp * STRIDE];
    return ovf[(sp - LDS_ENTRIES) * STRIDE];
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int STRIDE, int LDS_ENTRIES>
__device__ __forceinline__ void node_step(const PtNode* __restrict__ nodes, uint32_t* stack, uint32_t PT_AS1* ovf, v3 o, v3 inv, float tbest, int& cur,
                                          int& sp, uint32_t* depth_census = nullptr)
{
    const size_t nb = (size_t)(uint32_t)cur * sizeof(PtNode);
    const f32x4 a = ldg4(nodes, nb), b = ldg4(nodes, nb + 16), c = ldg4(nodes, nb + 32), chf = ldg4(nodes, nb + 48);
    const int chl = __float_as_int(chf.x), chr = __float_as_int(chf.y);
    // slab test of both children at once, {left, right} in the two halves of packed-f32 registers.  Same IEEE operations as
    // box_test: (bound - o) * inv, min/max ignoring NaN, far side scaled by 1.0000004.
    const f32x2 lox = {a.x, a.y}, loy = {a.z, a.w}, loz = {b.x, b.y}, hix = {b.z, b.w}, hiy = {c.x, c.y}, hiz = {c.z, c.w};
    const f32x2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z}, ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    const f32x2 t0x = (lox - ox) * ix, t1x = (hix - ox) * ix;
    const f32x2 t0y = (loy - oy) * iy, t1y = (hiy - oy) * iy;
    const f32x2 t0z = (loz - oz) * iz, t1z = (hiz - oz) * iz;
    const float tl = fmax_hw(fmax_hw(fmin_hw(t0x.x, t1x.x), fmin_hw(t0y.x, t1y.x)), fmax_hw(fmin_hw(t0z.x, t1z.x), kTMin));
    const float tr = fmax_hw(fmax_hw(fmin_hw(t0x.y, t1x.y), fmin_hw(t0y.y, t1y.y)), fmax_hw(fmin_hw(t0z.y, t1z.y), kTMin));
    f32x2 tf = {fmin_hw(fmin_hw(fmax_hw(t0x.x, t1x.x), fmax_hw(t0y.x, t1y.x)), fmin_hw(fmax_hw(t0z.x, t1z.x), tbest)),
                fmin_hw(fmin_hw(fmax_hw(t0x.y, t1x.y), fmax_hw(t0y.y, t1y.y)), fmin_hw(fmax_hw(t0z.y, t1z.y), tbest))};
    const f32x2 pad = {1.0000004f, 1.0000004f};
    tf = tf * pad;
    const bool hl = tl <= tf.x, hr = tr <= tf.y;
    // near child first; the far one is pushed only when both are hit
    const bool right_first = hr && (!hl || tr < tl);
    const int nearc = right_first ? chr : chl;
    const int farc = right_first ? chl : chr;
    if (hl && hr) {
        stack_push<STRIDE, LDS_ENTRIES>(stack, ovf, sp, (uint32_t)farc);
        if (depth_census) { depth_census[0] += 1; depth_census[1] += sp >= 8; depth_census[2] += sp >= 12; depth_census[3] += sp >= 16; }
        ++sp;
    }
    if (hl || hr) {
        cur = nearc;
    } else if (sp > 0) {
        --sp;
        cur = (int)stack_pop<STRIDE, LDS_ENTRIES>(stack, ovf, sp);
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
            node_step<PT_BLOCK, 0x7fffffff>(nodes, stack, nullptr, o, inv, h.t, cur, sp);
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
            radiance = radiance + tex_nearest(gp(P.env_map.texels), P.env_map.width, P.env_map.height, tu, tv);
            if (COUNT) ++cn.env;
        } else if (P.env_use_auto) {
            radiance = radiance + lerp3(vs(1.0f), V(0.5f, 0.7f, 1.0f), 0.5f * (ps.dir.y + 1.0f));
        } else {
            radiance = radiance + V(P.env_color[0], P.env_color[1], P.env_color[2]);
        }
        radiance = radiance * P.env_intensity;
        return SR_END;
    }
    const size_t tb = (size_t)(uint32_t)tslot * sizeof(PtTri);
    // triangle and shading record share the leaf-order index, and the triangle record repeats the material index: two dependent
    // round trips (records, then material) instead of three
    const f32x4 a = ldg4(P.tris, tb), b = ldg4(P.tris, tb + 16), c = ldg4(P.tris, tb + 32);
    const size_t sb = (size_t)(uint32_t)tslot * sizeof(PtShade);
    const f32x4 s0 = ldg4(P.shade, sb), s1 = ldg4(P.shade, sb + 16), s2 = ldg4(P.shade, sb + 32), s3 = ldg4(P.shade, sb + 48);
    int mi = __float_as_int(c.z);
    Material mat = material_default(); // device.cu:150-154
    int tex_slot = -1;
    if (mi >= 0) {
        const float PT_AS1* mp = gp(mats) + mi * PT_MAT_STRIDE;
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
        const PtTexDesc PT_AS1* tdp = gp(P.textures) + tex_slot;
        mat.base_color = tex_nearest(gp(tdp->texels), tdp->width, tdp->height, tu, tv);
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
#pragma unroll
        for (int k = 0; k < 24; ++k) atomicAdd(&P.counters->sched[k], (unsigned long long)cn.sched[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&P.counters->sched[24 + k], cn.cyc[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&P.counters->grp[k], (unsigned long long)cn.grp[k]);
        atomicAdd(&P.counters->grp[6], cn.grp_cyc);
#pragma unroll
        for (int k = 0; k < 16; ++k) atomicAdd(&P.counters->lobes[k], (unsigned long long)cn.lobe[k]);
    }
    {
        unsigned long long x0 = cn.cull_nohit, x1 = cn.cull_beyond, x3 = cn.leaf_noimp;
        for (int off = 32; off > 0; off >>= 1) { x0 += __shfl_down(x0, off, 64); x1 += __shfl_down(x1, off, 64); x3 += __shfl_down(x3, off, 64); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&P.counters->trav[0], x0); atomicAdd(&P.counters->trav[1], x1); atomicAdd(&P.counters->trav[3], x3); }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned long long x = cn.depth[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&P.counters->sched[19 + k], x);
    }
}

// 32-bit result in an SGPR: with __builtin_popcountll the compiler keeps wave-uniform counts as 64-bit values and compares them on the VALU
__device__ __forceinline__ int popc64(unsigned long long m)
{
    int r;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(r) : "s"(m) : "scc");
    return r;
}
// number of set bits of m below this lane
__device__ __forceinline__ int rank_in(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

} // namespace
