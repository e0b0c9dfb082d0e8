// pt_api.cpp -- implementation of the C-ABI declared in include/mi355pt.h (host side, HIP runtime).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pt_internal.h"
#include "pt_launch.h"
#include "pt_tiers.h"

// Slivers are never hit (part of the closest-hit definition; the oracle applies the same rule in its own words): a triangle
// whose height over its longest edge is below 1e-5 of that edge - |e1 x e2|^2 <= 1e-10 * max|e|^4, in double - is collapsed to its first
// vertex for the BVH and the triangle test (det = 0: the Moeller-Trumbore test rejects it for every ray).  Why: for such a needle the
// test's u, v, t are rounding noise and it reports "hits" far outside the triangle's bounding box, which a BVH walk does or does not see
// depending on the order in which it visits the leaves (found by tests/test_gpu_fuzz.py, seed 794689: the oracle's walk and this library's
// disagreed on one ray of 1.7e4 random scenes).  A height of < 100 ulp of the coordinates carries no geometry anyway.
static inline void pt_collapse_sliver(float* p)
{
    const double e1[3] = {(double)p[3] - (double)p[0], (double)p[4] - (double)p[1], (double)p[5] - (double)p[2]};
    const double e2[3] = {(double)p[6] - (double)p[0], (double)p[7] - (double)p[1], (double)p[8] - (double)p[2]};
    const double e3[3] = {(double)p[6] - (double)p[3], (double)p[7] - (double)p[4], (double)p[8] - (double)p[5]};
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double n2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    const double l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2], l3 = e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2];
    const double L2 = l1 > l2 ? (l1 > l3 ? l1 : l3) : (l2 > l3 ? l2 : l3);
    if (!(n2 > 1e-10 * L2 * L2)) { // also NaN / infinite vertices
        for (int k = 3; k < 9; ++k) p[k] = p[k % 3];
    }
}
#define PT_AUTO_PLOC_TRIS 64000000 // builder 3: the device PLOC builder beyond this many triangles (see pt_upload_scene)

namespace {
std::string g_create_error;
} // namespace

namespace pti {

int fail(pt_ctx* c, int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

int ensure(pt_ctx* c, DevBuf& b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return PT_OK;
    if (b.p) HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    HIP_TRY(c, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return PT_OK;
}

} // namespace pti
using pti::ensure;
using pti::fail;

namespace {

int upload(pt_ctx* c, DevBuf& b, const void* src, size_t bytes)
{
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    if (bytes) HIP_TRY(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return PT_OK;
}

void release(DevBuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

void copy_env(pt_ctx* c, const pt_env* env)
{
    c->env = *env;
    c->env_map.w = c->env_map.h = 0;
    c->env_map.px.clear();
    if (env->map.width > 0 && env->map.height > 0 && env->map.rgba8) {
        c->env_map.w = env->map.width;
        c->env_map.h = env->map.height;
        c->env_map.px.assign(env->map.rgba8, env->map.rgba8 + (size_t)env->map.width * env->map.height);
    }
    c->env.map.rgba8 = nullptr;
}

int upload_env(pt_ctx* c)
{
    if (c->host_only) return PT_OK;
    if (c->env_map.w > 0) return upload(c, c->d_env, c->env_map.px.data(), c->env_map.px.size() * 4);
    return PT_OK;
}

// Lobe thresholds of one material for the lobe bins of the hit pass (pt_kernel.hip, LOBE-COHERENT HIT PASSES): the cumulative
// selection probabilities of sample_disney in ITS order (disney.cuh:15-29,44-63: metallic, clearcoat, diffuse; glass is what is left)
// in units of 1/512 of the draw, 10 bits each; bit 30: emitter (device.cu:157-161 ends the path before any lobe runs), bit 31: the
// material has a glass lobe (force_btdf, disney.cuh:39).  A prediction aid only - the shader decides from the exact values.
uint32_t pt_lobe_code(const float* m)
{
    const float metallic = m[4], clearcoat = m[11], transmission = m[14], emission = m[16];
    const float wd = (1.0f - transmission) * (1.0f - metallic), wm = metallic, wc = 0.25f * clearcoat, wg = (1.0f - metallic) * transmission;
    const float sum = wm + wg + wd + wc;
    auto q = [&](float x) {
        const float v = sum > 0.0f ? x / sum * 512.0f + 0.5f : 0.0f;
        return (uint32_t)(v != v || v < 0.0f ? 0.0f : (v > 512.0f ? 512.0f : v));
    };
    const uint32_t t0 = q(wm), t1 = std::max(t0, q(wm + wc)), t2 = std::max(t1, q(wm + wc + wd));
    return t0 | (t1 << 10) | (t2 << 20) | (emission > 0.0f ? 0x40000000u : 0u) | (wg > 0.0f ? 0x80000000u : 0u);
}

int upload_materials(pt_ctx* c)
{
    // the lobe-code table (index = material + 1; 0 = material_data{} defaults, device.cu:150-154) and whether the scene can sample
    // more than one lobe at all (one lobe: the bins would only cost)
    c->lobe_mask = 0;
    std::memset(c->lobe_codes, 0, sizeof(c->lobe_codes));
    const bool fits = c->n_materials + 1 <= PT_LOBE_TABLE;
    static const float def_mat[PT_MAT_FLOATS] = {0.8f, 0.8f, 0.8f, 0.0f, 0.0f, 0.5f, 1.0f, 0.5f, 0.0f, 0.0f, 1.0f, 0.0f, 0.03f, 1.45f, 0.0f, 0.0f, 0.0f};
    for (int i = 0; fits && i <= c->n_materials; ++i) {
        const uint32_t code = pt_lobe_code(i == 0 ? def_mat : &c->materials[(size_t)(i - 1) * PT_MAT_STRIDE]);
        c->lobe_codes[i] = code;
        if (i == 0 && !c->uses_default_material) continue;
        if (code & 0x40000000u) continue; // emitter: no lobe
        const uint32_t t0 = code & 0x3ffu, t1 = (code >> 10) & 0x3ffu, t2 = (code >> 20) & 0x3ffu;
        c->lobe_mask |= (t0 > 0 ? 4u : 0u) | (t1 > t0 ? 2u : 0u) | (t2 > t1 ? 1u : 0u) | (t2 < 512u ? 8u : 0u);
    }
    if (!fits) c->lobe_mask = 0;
    if (c->host_only) return PT_OK;
    int rc = upload(c, c->d_lobe, c->lobe_codes, sizeof(c->lobe_codes));
    if (rc) return rc;
    return upload(c, c->d_materials, c->materials.data(), c->materials.size() * sizeof(float));
}

void pack_materials(pt_ctx* c, const float* materials, int n)
{
    c->n_materials = n;
    c->materials.assign((size_t)n * PT_MAT_STRIDE, 0.0f);
    for (int i = 0; i < n; ++i) {
        float* dst = &c->materials[(size_t)i * PT_MAT_STRIDE];
        std::memcpy(dst, materials + (size_t)i * PT_MAT_FLOATS, PT_MAT_FLOATS * sizeof(float));
        int32_t slot = (i < (int)c->material_texture.size()) ? c->material_texture[i] : -1;
        std::memcpy(dst + 17, &slot, 4);
    }
}

// Pixel ids owned by (rank, world): tile x tile tiles dealt round-robin on (tx + ty) % world; inside a tile the
// ids are emitted in 8x8 blocks so that the 64 lanes of a wave start on one compact screen patch.
int64_t shard_pixels(int W, int H, int tile, int rank, int world, uint32_t* ids, int64_t cap)
{
    if (W <= 0 || H <= 0 || world < 1 || rank < 0 || rank >= world) return -1;
    if (tile < 8) tile = 8;
    tile = (tile + 7) & ~7;
    int ntx = (W + tile - 1) / tile, nty = (H + tile - 1) / tile;
    int64_t n = 0;
    for (int ty = 0; ty < nty; ++ty)
        for (int tx = 0; tx < ntx; ++tx) {
            if ((tx + ty) % world != rank) continue;
            for (int by = 0; by < tile; by += 8)
                for (int bx = 0; bx < tile; bx += 8)
                    for (int y = 0; y < 8; ++y)
                        for (int x = 0; x < 8; ++x) {
                            int px = tx * tile + bx + x, py = ty * tile + by + y;
                            if (px >= W || py >= H) continue;
                            if (ids && n < cap) ids[n] = (uint32_t)px + (uint32_t)W * (uint32_t)py;
                            ++n;
                        }
        }
    return n;
}

int ensure_queue(pt_ctx* c, int W, int H)
{
    if (c->queue_valid && c->q_w == W && c->q_h == H && c->q_rank == c->rank && c->q_world == c->world && c->q_tile == c->tile) return PT_OK;
    int64_t n = shard_pixels(W, H, c->tile, c->rank, c->world, nullptr, 0);
    if (n < 0) return fail(c, PT_E_INVALID, "invalid pixel shard (%d of %d)", c->rank, c->world);
    std::vector<uint32_t> ids((size_t)n);
    shard_pixels(W, H, c->tile, c->rank, c->world, ids.data(), n);
    int rc = upload(c, c->d_pixels, ids.data(), ids.size() * 4);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // ids is a local
    c->n_pixels = (uint32_t)n;
    c->q_w = W; c->q_h = H; c->q_rank = c->rank; c->q_world = c->world; c->q_tile = c->tile;
    c->queue_valid = true;
    return PT_OK;
}

void fill_params(pt_ctx* c, PtKernelParams& P)
{
    std::memset(&P, 0, sizeof(P));
    P.nodes = (const PtNode*)c->d_nodes.p;
    P.tris = (const PtTri*)c->d_tris.p;
    P.shade = (const PtShade*)c->d_shade.p;
    P.materials = (const float*)c->d_materials.p;
    P.textures = (const PtTexDesc*)c->d_texdesc.p;
    P.env_map.texels = c->env_map.w > 0 ? (const uint32_t*)c->d_env.p : nullptr;
    P.env_map.width = c->env_map.w;
    P.env_map.height = c->env_map.h;
    for (int i = 0; i < 3; ++i) P.env_color[i] = c->env.color[i];
    P.env_intensity = c->env.intensity;
    P.env_use_map = c->env.use_map;
    P.env_use_auto = c->env.use_auto;
    P.root = c->bvh.root;
    P.n_tris = (int)c->bvh.tris.size();
    P.n_materials = c->n_materials;
    P.stack_entries = c->bvh.depth < 1 ? 1 : c->bvh.depth;
    for (int i = 0; i < 8; ++i) P.tune[i] = c->tune[i];
    P.lobe_codes = (const uint32_t*)c->d_lobe.p;
    P.hit_slot_mask = c->tri_packed ? 0x00ffffffu : 0xffffffffu;
    // lobe bins (wavefront kernel; off by default - measured: they cost what they save, profiles/r04_notes.md): 1 = whenever the scene
    // allows, -1 = when its materials can sample two or more different lobes
    const int n_lobes = __builtin_popcount(c->lobe_mask);
    P.lobe_bins = (c->kernel == 2 && c->tri_packed && c->lobe_mask != 0 && (c->lobe_bins > 0 || (c->lobe_bins < 0 && n_lobes >= 2))) ? 1 : 0;
}

} // namespace

namespace pti {
// After the render stream has drained: did a wave's watchdog fire (pt_kernel.hip, PT_WATCHDOG_ROUNDS)?  The image is then incomplete.
int check_watchdog(pt_ctx* c)
{
    if (c->kernel != 2 || !c->d_laps.p || !c->flag_pending) return PT_OK;
    uint32_t wd = 0;
    HIP_TRY(c, hipMemcpy(&wd, c->d_laps.p, 4, hipMemcpyDeviceToHost));
    c->flag_pending = false;
    c->watchdog_fired = wd != 0;
    if (wd) return fail(c, PT_E_HIP, "render kernel watchdog fired (scheduler made no progress); the image is incomplete");
    return PT_OK;
}
} // namespace pti
using pti::check_watchdog;

extern "C" {

int pt_abi_version(void) { return PT_ABI_VERSION; }

const char* pt_last_error(const pt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

pt_ctx* pt_create(const pt_config* cfg)
{
    int dev = cfg ? cfg->device : 0;
    pt_ctx* c = new pt_ctx();
    c->device = dev;
    if (dev < 0) { // host-only validation context: scene/BVH/sharding work, every render call fails loudly
        c->host_only = true;
        return c;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || dev >= n) {
        fail(nullptr, PT_E_NO_DEVICE, "no usable HIP device (count=%d, requested=%d): %s", n, dev, hipGetErrorString(e));
        delete c;
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipSetDevice(dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        fail(nullptr, PT_E_NO_DEVICE, "hipSetDevice/hipGetDeviceProperties failed for device %d", dev);
        delete c;
        return nullptr;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(nullptr, PT_E_NO_DEVICE, "device %d is %s; this library ships gfx950 (MI355X) code only", dev, prop.gcnArchName);
        delete c;
        return nullptr;
    }
    c->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->evm) != hipSuccess || hipEventCreate(&c->evr) != hipSuccess ||
        hipEventCreate(&c->evd) != hipSuccess) {
        fail(nullptr, PT_E_HIP, "stream/event creation failed");
        delete c;
        return nullptr;
    }
    return c;
}

void pt_destroy(pt_ctx* c)
{
    if (!c) return;
    if (!c->host_only) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)pt_comm_destroy(c);
        DevBuf* bufs[] = {&c->d_nodes8, &c->d_nodes4, &c->d_nodes, &c->d_tris, &c->d_shade, &c->d_materials, &c->d_texdesc, &c->d_env, &c->d_pixels, &c->d_heads,
                          &c->d_rng, &c->d_accum, &c->d_out, &c->d_out8, &c->d_counters, &c->d_dbg_in, &c->d_dbg_out, &c->d_slots, &c->d_laps, &c->d_ring, &c->d_params, &c->d_cost, &c->d_sorted, &c->d_sort_scratch, &c->d_dbg_start, &c->d_bucket, &c->d_tiers, &c->d_lobe};
        for (DevBuf* b : bufs) release(*b);
        for (void* p : c->d_textures) (void)hipFree(p);
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        if (c->evm) (void)hipEventDestroy(c->evm);
        if (c->evr) (void)hipEventDestroy(c->evr);
        if (c->evd) (void)hipEventDestroy(c->evd);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

int pt_set_option(pt_ctx* c, const char* key, int64_t value)
{
    if (!c || !key) return PT_E_INVALID;
    std::string k(key);
    if (k == "spp_per_launch") c->spp_per_launch = (int)(value < 0 ? 0 : value);
    else if (k == "count") c->count = value ? 1 : 0;
    else if (k == "blocks_per_cu") c->blocks_per_cu = (int)(value < 0 ? 0 : value);
    else if (k == "leaf_size") c->leaf_size = (int)value;
    else if (k == "max_bvh_depth") c->max_bvh_depth = (int)value;
    else if (k == "sticky_pct") c->sticky_pct = (int)(value < 1 ? -1 : (value > 100 ? 100 : value)); // < 1: automatic
    else if (k == "cost_radius") c->cost_radius = (int)(value < 0 ? 0 : (value > 8 ? 8 : value));
    else if (k == "timeline") c->timeline = value != 0;
    else if (k == "latency") c->latency = (int)value;
    else if (k == "census_mode") c->census_mode = (int)value;
    else if (k == "schedule") c->schedule = value == 0 ? 0 : 1;
    else if (k == "prepass_spp") c->prepass_spp = (int)(value < 0 ? 0 : (value > 64 ? 64 : value)); // 0: automatic (8; 16 when a tier plan is prepared)
    else if (k == "chunk_tail_min") c->chunk_tail_min = (int)(value < 0 ? -1 : (value > 65535 ? 65535 : value)); // -1: automatic
    else if (k == "chunk_spp") c->chunk_spp = (int)(value < 1 ? 1 : (value > 65535 ? 65535 : value));
    else if (k == "slots_per_wave") c->slots_per_wave = (int)(value < 0 ? 0 : value);
    else if (k == "adaptive") c->tune[5] = value ? 1 : 2;
    else if (k == "bvh_builder") {
        if (value < 0 || value > 3) return fail(c, PT_E_INVALID, "bvh_builder must be 0 (host binned SAH), 1 (device LBVH), 2 (device PLOC) or 3 (by triangle count)");
        c->bvh_builder = (int)value;
    }
    else if (k == "wide_leaves") c->wide_leaves = value != 0; // oct nodes: subtrees of <= 7 triangles become one leaf (before pt_upload_scene)
    else if (k == "fallback") c->fallback = value != 0; // force the wavefront kernel's 168-VGPR fallback instance (tests)
    else if (k == "express_permille") c->express_permille = (int)(value < 0 ? -1 : (value > 500 ? 500 : value)); // -1: automatic
    else if (k == "whole") c->whole = (int)(value < 0 ? -1 : (value > 1 ? 1 : value)); // whole-pixel schedule by cost class when every pixel can have a path slot: -1 the plan decides (default), 0 never, 1 always
    else if (k == "ns_express") c->ns_express = (int)(value < 1 ? 1 : (value > 64 ? 64 : value));
    else if (k == "groups") c->groups = (int)(value < 0 ? 0 : (value > 2 ? 2 : value)); // group walk: 0 never, 1 sparse waves (default), 2 always
    else if (k == "ploc_radius") c->ploc_radius = (int)(value < 1 ? 1 : (value > 64 ? 64 : value)); // bvh_builder 2: neighbours searched on either side
    else if (k == "lobe_bins") {
        if (value && !pt_kernel_lobe_bins()) return fail(c, PT_E_INVALID, "option 'lobe_bins': this build has no lobe bins (make -C owl-path-tracer_amd/csrc lobebins)");
        c->lobe_bins = (int)(value < 0 ? -1 : (value > 1 ? 1 : value));
    } // hit passes by predicted lobe: 0 never (default), 1 whenever possible, -1 when the scene has two or more lobes
    else if (k == "box_exact") c->box_exact = (int)(value < 0 ? -1 : (value > 0 ? 1 : 0)); // slab test form: -1 automatic (fma unless the camera is far outside the scene), 0 fma, 1 subtracting
    else if (k == "quad") c->quad = value != 0; // wavefront kernel: quad nodes (two binary levels per fetch), next pt_render
    else if (k == "node_pairs") c->node_pairs = value != 0;
    else if (k == "leaf_align") c->leaf_align = (int)(value < 1 ? 1 : (value > 8 ? 8 : value));
    else if (k.size() == 5 && k.compare(0, 4, "tune") == 0 && k[4] >= '0' && k[4] <= '7') c->tune[k[4] - '0'] = (int)value;
    else if (k == "kernel") {
        if (value != 1 && value != 2) return fail(c, PT_E_INVALID, "kernel must be 1 (lane-per-pixel) or 2 (wavefront-scheduled)");
        c->kernel = (int)value;
    }
    else return fail(c, PT_E_INVALID, "unknown option '%s'", key);
    return PT_OK;
}

int pt_upload_scene(pt_ctx* c, const pt_mesh* meshes, int32_t n_meshes, const float* materials, int32_t n_materials,
                    const pt_texture* textures, int32_t n_textures, const int32_t* material_texture, const pt_env* env)
{
    if (!c) return PT_E_INVALID;
    if (n_meshes < 0 || n_materials < 0 || n_textures < 0 || (n_meshes > 0 && !meshes) || (n_materials > 0 && !materials) ||
        (n_textures > 0 && !textures))
        return fail(c, PT_E_INVALID, "pt_upload_scene: null array with non-zero count");
    // everything that can be checked without touching the context is checked first; from here on the context has NO scene until the
    // upload has succeeded (a failure half way must not leave the previous scene's flag over new host arrays)
    for (int i = 0; i < n_textures; ++i)
        if (textures[i].width <= 0 || textures[i].height <= 0 || !textures[i].rgba8) return fail(c, PT_E_INVALID, "texture %d is empty", i);
    if (material_texture)
        for (int i = 0; i < n_materials; ++i)
            if (material_texture[i] >= n_textures) return fail(c, PT_E_INVALID, "material %d: texture index %d out of range (%d textures)", i, material_texture[i], n_textures);
    c->have_scene = false;
    if (!c->host_only) HIP_TRY(c, hipSetDevice(c->device));
    // PT_UPLOAD_TRACE=1: phase times of this call on stderr
    const bool trace = getenv("PT_UPLOAD_TRACE") && getenv("PT_UPLOAD_TRACE")[0] == '1';
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (trace) fprintf(stderr, "pt_upload_scene: %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_phase).count());
        t_phase = now;
    };

    // ---- flatten entities to one record per triangle, global order = entity order then face order ----
    size_t n_tris = 0;
    for (int m = 0; m < n_meshes; ++m) {
        if (meshes[m].n_triangles < 0) return fail(c, PT_E_INVALID, "mesh %d: negative triangle count", m);
        n_tris += (size_t)meshes[m].n_triangles;
    }
    if (n_tris > (size_t)(1u << 28)) return fail(c, PT_E_LIMIT, "too many triangles (%zu)", n_tris);
    std::vector<float> pos(n_tris * 9);
    std::vector<size_t> mesh_first((size_t)n_meshes + 1, 0); // global id of a mesh's first triangle
    c->material_texture.assign((size_t)n_materials, -1);
    if (material_texture)
        for (int i = 0; i < n_materials; ++i) c->material_texture[i] = material_texture[i];
    size_t g = 0;
    c->uses_default_material = false;
    for (int m = 0; m < n_meshes; ++m) {
        const pt_mesh& ms = meshes[m];
        if (ms.material_index < 0 && ms.n_triangles > 0) c->uses_default_material = true;
        if (ms.n_triangles > 0 && (!ms.vertices || !ms.indices)) return fail(c, PT_E_INVALID, "mesh %d: null vertices/indices", m);
        if (ms.material_index >= n_materials) return fail(c, PT_E_INVALID, "mesh %d: material index %d out of range", m, ms.material_index);
        if (ms.texture_index >= n_textures) return fail(c, PT_E_INVALID, "mesh %d: texture index %d out of range", m, ms.texture_index);
        if (ms.texture_index >= 0 && ms.material_index >= 0 && !material_texture) c->material_texture[ms.material_index] = ms.texture_index;
        const bool textured = ms.texture_index >= 0;
        // (the triangles of a large mesh are flattened by all build threads; bad: 1 = vertex index, 2 = normal, 3 = texcoord, + 4 * vertex)
        std::atomic<long long> bad{0};
        const size_t g0 = g;
        mesh_first[(size_t)m] = g0;
        pt_parallel_ranges((size_t)ms.n_triangles, [&](size_t t_lo, size_t t_hi) {
            for (size_t t = t_lo; t < t_hi; ++t) {
                for (int k = 0; k < 3; ++k) {
                    const int32_t vi = ms.indices[t * 3 + (size_t)k];
                    if (vi < 0 || vi >= ms.n_vertices) { bad.store(1 + 4ll * vi); return; }
                    // the reference traps on an out-of-bounds normal/texcoord fetch (macros.hpp:5-11)
                    if (!ms.normals || vi >= ms.n_normals) { bad.store(2 + 4ll * vi); return; }
                    if (textured && (!ms.texcoords || vi >= ms.n_texcoords)) { bad.store(3 + 4ll * vi); return; }
                    std::memcpy(&pos[(g0 + t) * 9 + (size_t)k * 3], ms.vertices + (size_t)vi * 3, 12);
                }
                pt_collapse_sliver(&pos[(g0 + t) * 9]);
            }
        });
        if (const long long b = bad.load()) {
            const int what = (int)(b & 3), vi = (int)(b >> 2);
            if (what == 1) return fail(c, PT_E_INVALID, "mesh %d: vertex index out of range", m);
            if (what == 2) return fail(c, PT_E_INVALID, "mesh %d: no normal for vertex %d", m, vi);
            return fail(c, PT_E_INVALID, "mesh %d: no texcoord for vertex %d", m, vi);
        }
        g += (size_t)ms.n_triangles;
    }
    mesh_first[(size_t)n_meshes] = g;

    phase("flatten entities");
    // ---- BVH (replaces owlGroupBuildAccel, application.cpp:135-139) ----
    auto t0 = std::chrono::steady_clock::now();
    const int leaf_sz = std::max(1, std::min(7, c->leaf_size));
    // builder 3 = automatic: the host SAH tree walks fastest (C4: 478 ms against 606 for PLOC and 697 for the Karras tree) and, since the
    // builder runs on all host threads (round 4), is built as fast as the device PLOC tree comes down and is laid out: 0.9 M triangles 58-83 ms
    // against 88, 4 M triangles 350 against 360 ms (and walks 4 % faster there) - PLOC only takes over where the host builder's memory would
    // become the limit (PT_AUTO_PLOC_TRIS)
    const int builder = c->bvh_builder == 3 ? (n_tris > (size_t)PT_AUTO_PLOC_TRIS ? 2 : 0) : c->bvh_builder;
    if (builder == 2 && !c->host_only && n_tris > (size_t)leaf_sz) {
        // device PLOC (pt_lbvh.hip): the hierarchy comes down, the host lays it out (pt_bvh_from_hierarchy)
        const int n = (int)n_tris;
        DevBuf d_pos, d_ws;
        int rc;
        auto cleanup = [&]() { release(d_pos); release(d_ws); };
        if ((rc = upload(c, d_pos, pos.data(), pos.size() * sizeof(float))) || (rc = ensure(c, d_ws, pt_ploc_workspace_bytes(n)))) {
            cleanup();
            return rc;
        }
        std::vector<int32_t> h_child(2 * (size_t)n * 2), h_count(2 * (size_t)n);
        std::vector<float> h_box(2 * (size_t)n * 6);
        std::vector<uint32_t> h_order((size_t)n);
        int32_t root = -1, rounds = 0;
        hipError_t e = pt_ploc_build_device((const float*)d_pos.p, n, c->ploc_radius, d_ws.p, d_ws.cap, h_child.data(), h_box.data(), h_count.data(), h_order.data(), &root,
                                            &rounds, c->stream);
        cleanup();
        if (e != hipSuccess) return fail(c, PT_E_HIP, "device PLOC build failed: %s", hipGetErrorString(e));
        if (!pt_bvh_from_hierarchy(pos.data(), (int32_t)n_tris, h_child.data(), h_box.data(), h_count.data(), h_order.data(), root, c->leaf_size, c->max_bvh_depth, &c->bvh))
            pt_bvh_build(pos.data(), (int32_t)n_tris, c->leaf_size, c->max_bvh_depth, &c->bvh); // deeper than the stack allows: the host builder caps the depth
    } else if (builder == 1 && !c->host_only && n_tris > (size_t)leaf_sz) {
        // device LBVH (pt_lbvh.hip): positions up, nodes + sorted order down - the host keeps its copy for the validation hooks
        // and for the shading records, which follow the triangles into leaf order below
        const int n = (int)n_tris;
        DevBuf d_pos, d_ws, d_order;
        int rc;
        auto cleanup = [&]() { release(d_pos); release(d_ws); release(d_order); };
        if ((rc = upload(c, d_pos, pos.data(), pos.size() * sizeof(float))) || (rc = ensure(c, d_ws, pt_lbvh_workspace_bytes(n))) ||
            (rc = ensure(c, d_order, (size_t)n * 4)) || (rc = ensure(c, c->d_nodes, (size_t)(n - 1) * sizeof(PtNode)))) {
            cleanup();
            return rc;
        }
        int32_t root = -1, n_nodes = 0, height = 0, max_leaf = 0;
        float pad = 0.0f;
        hipError_t e = pt_lbvh_build_device((const float*)d_pos.p, n, leaf_sz, d_ws.p, d_ws.cap, (PtNode*)c->d_nodes.p, (uint32_t*)d_order.p, &root, &n_nodes,
                                            &height, &max_leaf, &pad, c->stream);
        if (e != hipSuccess) {
            cleanup();
            return fail(c, PT_E_HIP, "device BVH build failed: %s", hipGetErrorString(e));
        }
        c->bvh.nodes.resize((size_t)n_nodes);
        std::vector<uint32_t> order((size_t)n);
        hipError_t e1 = hipMemcpy(c->bvh.nodes.data(), c->d_nodes.p, (size_t)n_nodes * sizeof(PtNode), hipMemcpyDeviceToHost);
        hipError_t e2 = hipMemcpy(order.data(), d_order.p, (size_t)n * 4, hipMemcpyDeviceToHost);
        cleanup();
        if (e1 != hipSuccess || e2 != hipSuccess) return fail(c, PT_E_HIP, "device BVH read-back failed");
        if (height > std::min(c->max_bvh_depth, (int)PT_MAX_STACK)) {
            // the Karras tree has no depth control (clustered or duplicate centroids give long chains): the host builder, which
            // caps the depth, takes over instead of failing the upload
            pt_bvh_build(pos.data(), (int32_t)n_tris, c->leaf_size, c->max_bvh_depth, &c->bvh);
        } else {
            c->bvh.root = root;
            c->bvh.depth = height;
            c->bvh.max_leaf = max_leaf;
            c->bvh.pad = pad;
            c->bvh.tris.resize((size_t)n);
            for (int i = 0; i < n; ++i) {
                PtTri& t = c->bvh.tris[(size_t)i];
                std::memcpy(t.p0, &pos[(size_t)order[(size_t)i] * 9], 36);
                t.id = (int32_t)order[(size_t)i];
                t.material = -1;
                t.pad = 0;
            }
        }
    } else {
        pt_bvh_build(pos.data(), (int32_t)n_tris, c->leaf_size, c->max_bvh_depth, &c->bvh);
    }
    auto t1 = std::chrono::steady_clock::now();
    c->stats.bvh_build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    c->stats.bvh_nodes = c->bvh.nodes.size();
    c->stats.bvh_depth = (uint64_t)c->bvh.depth;
    c->stats.n_triangles = n_tris;
    if (c->bvh.depth > PT_MAX_STACK) return fail(c, PT_E_LIMIT, "BVH depth %d exceeds %d", c->bvh.depth, PT_MAX_STACK);
    phase("BVH build");
    pt_bvh_layout(&c->bvh, c->node_pairs, c->leaf_align);
    c->stats.bvh_nodes = c->bvh.nodes.size();
    { // the two collapses only read the binary tree: side by side
        std::thread oct([&] { pt_bvh_collapse8(c->bvh, c->wide_leaves, &c->nodes8, &c->root8, &c->depth8); });
        pt_bvh_collapse4(c->bvh, &c->nodes4, &c->root4, &c->depth4);
        oct.join();
    }
    if (3 * c->depth4 + 1 > PT_MAX_STACK) c->nodes4.clear(); // the quad walk could need more stack than the kernel has: binary walk instead
    if (7 * c->depth8 + 1 > PT_GROUP_STACK) c->nodes8.clear(); // a group's stack (eight LDS stack columns) could overflow: no group walk
    phase("quad + oct nodes");
    { // shading records in leaf order (padding slots included), gathered from the caller's arrays: the three vertex normals and
      // texcoords of triangle `id` (device.cu:63-94) and its material, which the triangle record carries as well
        const size_t n_slots = c->bvh.tris.size();
        c->shade.resize(n_slots);
        pt_parallel_ranges(n_slots, [&](size_t lo, size_t hi) {
            int m = 0;
            for (size_t i = lo; i < hi; ++i) {
                PtShade& sh = c->shade[i];
                std::memset(&sh, 0, sizeof(sh));
                const int32_t id = c->bvh.tris[i].id;
                if (id == 0x7fffffff) { sh.material = -1; continue; }
                if (!((size_t)id >= mesh_first[(size_t)m] && (size_t)id < mesh_first[(size_t)m + 1]))
                    m = (int)(std::upper_bound(mesh_first.begin(), mesh_first.end(), (size_t)id) - mesh_first.begin()) - 1;
                const pt_mesh& ms = meshes[m];
                const size_t t = (size_t)id - mesh_first[(size_t)m];
                sh.material = ms.material_index;
                c->bvh.tris[i].material = ms.material_index;
                for (int k = 0; k < 3; ++k) {
                    const int32_t vi = ms.indices[t * 3 + (size_t)k]; // validated above
                    float* nd = k == 0 ? sh.n0 : (k == 1 ? sh.n1 : sh.n2);
                    std::memcpy(nd, ms.normals + (size_t)vi * 3, 12);
                    if (ms.texcoords && vi < ms.n_texcoords) std::memcpy(&sh.tc[k * 2], ms.texcoords + (size_t)vi * 2, 8);
                }
            }
        });
    }
    phase("shading records");
    // ---- textures, materials, environment ----
    c->textures.assign((size_t)n_textures, HostTexture{});
    for (int i = 0; i < n_textures; ++i) {
        c->textures[i].w = textures[i].width;
        c->textures[i].h = textures[i].height;
        c->textures[i].px.assign(textures[i].rgba8, textures[i].rgba8 + (size_t)textures[i].width * textures[i].height);
    }
    pack_materials(c, materials, n_materials);
    pt_env def{};
    copy_env(c, env ? env : &def);
    if (!c->host_only) {
        const int urc = pti::upload_scene_to_device(c);
        if (urc) return urc;
    }
    phase("textures, upload to HBM");
    c->have_scene = true;
    return PT_OK;
}

} // extern "C"

namespace pti {

// Host copies of the scene (BVH, quad nodes, triangles, shading records, textures, materials, environment) -> this context's GPU.
int upload_scene_to_device(pt_ctx* c)
{
    HIP_TRY(c, hipSetDevice(c->device));
    const int n_textures = (int)c->textures.size();
    int rc;
    if ((rc = upload(c, c->d_nodes, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(PtNode)))) return rc;
    if ((rc = upload(c, c->d_nodes4, c->nodes4.data(), c->nodes4.size() * sizeof(PtNode4)))) return rc;
    if ((rc = upload(c, c->d_nodes8, c->nodes8.data(), c->nodes8.size() * sizeof(PtNode8)))) return rc;
    {   // the device copy of the triangle records carries the material with the id (PtTri::id): below 2^23 triangle slots a hit's
        // triangle slot leaves room for it in the word the kernel keeps per hit, and id << 8 stays a positive int (same tie-break order)
        // (packed on the device after the copy: pt_pack_tri_ids_kernel)
        c->tri_packed = pt_kernel_lobe_bins() && c->bvh.tris.size() < ((size_t)1 << 23); // (only builds with the lobe bins use it)
        if ((rc = upload(c, c->d_tris, c->bvh.tris.data(), c->bvh.tris.size() * sizeof(PtTri)))) return rc;
        if (c->tri_packed) HIP_TRY(c, pt_launch_pack_tri_ids((PtTri*)c->d_tris.p, (long long)c->bvh.tris.size(), c->stream));
    }
    if ((rc = upload(c, c->d_shade, c->shade.data(), c->shade.size() * sizeof(PtShade)))) return rc;
    for (void* p : c->d_textures) (void)hipFree(p);
    c->d_textures.clear();
    std::vector<PtTexDesc> descs((size_t)n_textures);
    for (int i = 0; i < n_textures; ++i) {
        void* p = nullptr;
        size_t bytes = c->textures[i].px.size() * 4;
        HIP_TRY(c, hipMalloc(&p, bytes));
        c->d_textures.push_back(p);
        HIP_TRY(c, hipMemcpyAsync(p, c->textures[i].px.data(), bytes, hipMemcpyHostToDevice, c->stream));
        descs[i].texels = (const uint32_t*)p;
        descs[i].width = c->textures[i].w;
        descs[i].height = c->textures[i].h;
    }
    if ((rc = upload(c, c->d_texdesc, descs.data(), descs.size() * sizeof(PtTexDesc)))) return rc;
    if ((rc = upload_materials(c))) return rc;
    if ((rc = upload_env(c))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

// The scene of `src` (built once: BVH, leaf-order records, quad nodes) copied into `dst` and uploaded to dst's GPU: the replicas
// of a multi-GPU group (pt_group_upload_scene) do not each rebuild the BVH on the host.
int clone_scene(pt_ctx* dst, const pt_ctx* src)
{
    if (!src->have_scene) return fail(dst, PT_E_NO_SCENE, "clone_scene: the source context has no scene");
    dst->bvh = src->bvh;
    dst->nodes4 = src->nodes4;
    dst->root4 = src->root4;
    dst->depth4 = src->depth4;
    dst->nodes8 = src->nodes8;
    dst->root8 = src->root8;
    dst->depth8 = src->depth8;
    dst->shade = src->shade;
    dst->materials = src->materials;
    dst->n_materials = src->n_materials;
    dst->material_texture = src->material_texture;
    dst->uses_default_material = src->uses_default_material;
    dst->textures = src->textures;
    dst->env = src->env;
    dst->env_map = src->env_map;
    dst->stats.bvh_build_ms = src->stats.bvh_build_ms;
    dst->stats.bvh_nodes = src->stats.bvh_nodes;
    dst->stats.bvh_depth = src->stats.bvh_depth;
    dst->stats.n_triangles = src->stats.n_triangles;
    dst->have_scene = true;
    dst->queue_valid = false;
    if (dst->host_only) return PT_OK;
    return upload_scene_to_device(dst);
}

} // namespace pti

extern "C" {

int pt_set_materials(pt_ctx* c, const float* materials, int32_t n_materials)
{
    if (!c || !materials || n_materials < 0) return PT_E_INVALID;
    if (!c->have_scene) return fail(c, PT_E_NO_SCENE, "pt_set_materials before pt_upload_scene");
    if (n_materials != c->n_materials) return fail(c, PT_E_INVALID, "material count changed (%d -> %d); re-upload the scene", c->n_materials, n_materials);
    pack_materials(c, materials, n_materials);
    if (c->host_only) return PT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int rc = upload_materials(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_set_environment(pt_ctx* c, const pt_env* env)
{
    if (!c || !env) return PT_E_INVALID;
    copy_env(c, env);
    if (c->host_only) return PT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int rc = upload_env(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_set_pixel_shard(pt_ctx* c, int32_t rank, int32_t world_size, int32_t tile)
{
    if (!c) return PT_E_INVALID;
    if (world_size < 1 || rank < 0 || rank >= world_size || tile < 1) return fail(c, PT_E_INVALID, "bad shard %d/%d tile %d", rank, world_size, tile);
    c->rank = rank;
    c->world = world_size;
    c->tile = tile;
    return PT_OK;
}

int64_t pt_shard_pixels(int32_t width, int32_t height, int32_t tile, int32_t rank, int32_t world_size, uint32_t* ids, int64_t cap)
{
    return shard_pixels(width, height, tile, rank, world_size, ids, cap);
}

int pt_render_device(pt_ctx* c, const pt_camera* cam, int32_t W, int32_t H, int32_t max_samples, int32_t max_depth, void* d_out_rgb,
                     void* d_out_rgba8, void* stream_v)
{
    if (!c || !cam || !d_out_rgb) return PT_E_INVALID;
    if (c->host_only) return fail(c, PT_E_NO_DEVICE, "host-only context: the HIP render path is required and there is no CPU fallback");
    if (!c->have_scene) return fail(c, PT_E_NO_SCENE, "no geometries (pt_upload_scene not called)");
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535 || max_samples <= 0 || max_depth < 0 || max_depth > 63 || (int64_t)W * H > (int64_t)0x7fffffff)
        return fail(c, PT_E_INVALID, "bad render size %dx%d spp %d depth %d (depth must be 0..63)", W, H, max_samples, max_depth);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : c->stream;
    int rc = ensure_queue(c, W, H);
    if (rc) return rc;
    if (c->n_pixels == 0) {
        // A rank that owns no tile (fewer tiles than ranks, e.g. 64x64 / tile 16 at world 8): its frame is all zeros and no kernel
        // runs.  (Round-3 advisor finding: the main launch of such a rank read a tier table nobody had written - the sort and the
        // plan kernel return early for an empty queue.)
        HIP_TRY(c, hipEventRecord(c->ev0, stream));
        HIP_TRY(c, hipMemsetAsync(d_out_rgb, 0, (size_t)W * H * 3 * sizeof(float), stream));
        if (d_out_rgba8) HIP_TRY(c, hipMemsetAsync(d_out_rgba8, 0, (size_t)W * H * 4, stream));
        HIP_TRY(c, hipEventRecord(c->ev1, stream));
        c->ev_pending = true;
        c->flag_pending = false;
        c->last_stream = stream;
        c->last_launches = 0;
        c->last_sorted = false;
        c->last_w = W;
        c->last_h = H;
        c->stats.express_pixels = 0;
        c->stats.whole_pixels = 0;
        c->stats.prepass_spp = 0;
        c->stats.grid = 0;
        if (c->count && c->d_counters.p) HIP_TRY(c, hipMemsetAsync(c->d_counters.p, 0, sizeof(PtCounters), stream));
        return PT_OK;
    }

    // kernel 1 (lane-per-pixel): optional spp chunks = separate launches.  kernel 2 (wavefront): ONE persistent launch that
    // walks (pixel, chunk) tickets; spp_per_launch, if set, becomes its chunk size so that the resumability tests cover it.
    PtKernelParams P;
    fill_params(c, P);
    {   // the fma form of the slab test (pt_kernel.hip, node4_step) displaces a plane by |o| 2^-24; the boxes are padded by 1e-5 x the scene
        // extent (bvh.pad): exact form when the camera is so far from the origin that this would eat a quarter of the padding
        const float reach = c->bvh.pad * 4194304.0f; // pad x 2^22 = 42 scene extents
        float far_o = 0.0f;
        for (int a = 0; a < 3; ++a) far_o = std::max(far_o, std::fabs(cam->origin[a]));
        P.box_exact = (c->box_exact > 0 || (c->box_exact < 0 && !(far_o <= reach))) ? 1 : 0;
    }
    if (c->kernel == 2 && c->quad && !c->nodes4.empty()) { // the wavefront kernel walks the quad nodes: own root and stack bound
        P.nodes4 = (const PtNode4*)c->d_nodes4.p;
        P.root = c->root4;
        P.stack_entries = 3 * c->depth4 + 1;
    }
    if (c->kernel == 2 && c->groups && !c->nodes8.empty()) { // group walk of sparse waves (oct nodes)
        P.nodes8 = (const PtNode8*)c->d_nodes8.p;
        P.root8 = c->root8;
        P.groups = c->groups;
    }

    // Launch geometry.  The wavefront kernel keeps `ns` pixels in flight per wave; shrink ns when the image is too small to give
    // every resident wave a full set (e.g. 512x512 over 4096 waves), otherwise use the default.
    const int group_entries = P.nodes8 ? 7 * c->depth8 + 1 : 0;
    size_t lds = 0, state_words = 0;
    int vg = 0, sg = 0, slds = 0, occ = 0, block = 0, ns = 0;
    static const int default_ns = [] { const char* e = getenv("PT_DEFAULT_NS"); const int v = e ? atoi(e) : 0; return v >= 16 && v <= 252 ? v : 96; }(); // (A/B builds of tools/)
    int want_ns = c->slots_per_wave > 0 ? c->slots_per_wave : default_ns; // 16 waves/CU up to 104; 64..255 swept on C4 (profiles/r01_summary.md), 88..104 within 1 %
    // variant of the launch: the wavefront kernel's product instance (2) unless it needs scratch in this build - then its fallback
    // instance with the larger register budget (3): slower (12 instead of 16 waves per CU), the same arithmetic
    // (the one-level walk over PtNode[] - option quad = 0, or a tree too deep for the quad walk's stack bound - is compiled into the
    // instrumented instance only: the product instance is kept small for the instruction cache)
    const int use_count = (c->count || (c->kernel == 2 && !P.nodes4)) ? 1 : 0;
    int variant = c->kernel == 2 && c->fallback && !use_count ? 3 : c->kernel;
    {
        hipError_t ge = pt_kernel_geometry(variant, use_count, P.stack_entries, group_entries, want_ns, P.lobe_bins, P.box_exact, &block, &lds, &ns, &state_words, &vg, &occ, &P.lds_levels);
        if (ge == hipErrorInvalidConfiguration && variant == 2 && !use_count) {
            variant = 3;
            ge = pt_kernel_geometry(variant, use_count, P.stack_entries, group_entries, want_ns, P.lobe_bins, P.box_exact, &block, &lds, &ns, &state_words, &vg, &occ, &P.lds_levels);
        }
        if (ge == hipErrorInvalidConfiguration)
            return fail(c, PT_E_LIMIT, use_count ? "the instrumented instance of the render kernel needs scratch in this build; such builds rendered wrong pixels and are refused (pt_kernel.hip; tests/test_abi_host.py reads hipcc's resource report)"
                                                 : "this build of the render kernel spills registers to scratch even in its fallback instance; such builds rendered wrong pixels and are refused (pt_kernel.hip)");
        HIP_TRY(c, ge);
    }
    if (c->kernel == 2 && c->slots_per_wave == 0 && occ > 0) {
        // small images: fewer slots per wave so that at least 8 waves per CU have pixels (never below 64)
        long fit = (long)c->n_pixels / ((long)c->num_cus * 8);
        if (fit < want_ns) {
            want_ns = (int)std::max(64L, fit);
            HIP_TRY(c, pt_kernel_geometry(variant, use_count, P.stack_entries, group_entries, want_ns, P.lobe_bins, P.box_exact, &block, &lds, &ns, &state_words, &vg, &occ, &P.lds_levels));
        }
    }
    if (occ < 1) return fail(c, PT_E_LIMIT, "render kernel does not fit a CU (LDS %zu bytes, BVH depth %d)", lds, c->bvh.depth);
    int bpc = c->blocks_per_cu > 0 ? std::min(c->blocks_per_cu, occ) : occ;
    long want = ((long)c->n_pixels + ns - 1) / ns; // never more path slots than pixels: a pixel's chunks are sequential
    if (c->kernel == 1) want = ((long)c->n_pixels + block - 1) / block;
    int grid = (int)std::max(1L, std::min(want, (long)c->num_cus * bpc));
    int pre = c->prepass_spp > 0 ? c->prepass_spp : 8; // samples of the cost pre-pass
    const bool sorted = c->kernel == 2 && c->schedule == 1 && c->spp_per_launch == 0 && max_samples >= 4 * pre && max_samples <= 65535;
    // Whole-pixel schedule (pt_kernel.hip, TIERS): when every pixel can have a path slot from the start, the main launch hands out
    // pixels instead of (pixel, chunk) tickets, and every wave serves one cost class with as few pixels as that class needs - the plan
    // is made on the device from the histogram of the counting sort (pt_plan_tiers_kernel).  Option "whole": -1 automatic, 0 never.
    bool tiers = false;
    int ring_grid = 0;
    if (sorted && P.nodes8 && c->whole != 0 && c->slots_per_wave == 0 && occ > 0) {
        const long capacity = (long)c->num_cus * bpc;
        for (int nsd = 96; nsd <= 104 && !tiers; nsd += 8) { // 16 waves per CU up to 104 slots
            if (c->whole < 1 && (long)c->n_pixels + (long)PT_MAX_TIERS * nsd > capacity * nsd) continue; // (one partly filled wave per class)
            want_ns = nsd;
            HIP_TRY(c, pt_kernel_geometry(variant, use_count, P.stack_entries, group_entries, want_ns, P.lobe_bins, P.box_exact, &block, &lds, &ns, &state_words, &vg, &occ, &P.lds_levels));
            if (occ >= bpc && ns == nsd) {
                tiers = true;
                ring_grid = (int)std::max(1L, std::min(((long)c->n_pixels + ns - 1) / ns, capacity)); // what the ring schedule would launch
                grid = (int)capacity;
            }
        }
        if (!tiers) {
            want_ns = default_ns;
            HIP_TRY(c, pt_kernel_geometry(variant, use_count, P.stack_entries, group_entries, want_ns, P.lobe_bins, P.box_exact, &block, &lds, &ns, &state_words, &vg, &occ, &P.lds_levels));
        }
    }
    // the tier plan lives on the cost estimate: twice the samples (1/8 shard of C4 218 -> 201 ms; a throughput-bound frame gains nothing)
    if (tiers && c->prepass_spp == 0 && max_samples >= 4 * 16) pre = 16;
    uint32_t n_express = 0;
    int express_waves = 0;
    if (state_words) {
        if ((rc = ensure(c, c->d_slots, state_words * 4 * (size_t)grid))) return rc;
        P.slot_state = (uint32_t*)c->d_slots.p;
    }
    P.ns = ns;

    // Chunk schedule of one wavefront launch over `total` samples per pixel: n_full chunks of `chunk` samples, then the rest in
    // halving chunks (rem/2, rem/4, ... >= chunk_tail_min).  A frame ends when its slowest in-flight work item ends, so the
    // last items must be short (profiles/r01_summary.md, "wind-down").
    struct Schedule { int chunk = 0, n_full = 0, n_chunks = 1, tail_len[PT_MAX_TAIL_CHUNKS] = {}; };
    int tail_min = c->chunk_tail_min >= 0 ? c->chunk_tail_min : 16; // smallest chunk of the halving tail (automatic: set per schedule below)
    auto make_schedule = [&](int total, int chunk, int rem_min) {
        Schedule sc;
        sc.chunk = std::max(1, std::min(chunk, total));
        sc.n_full = total / sc.chunk;
        int rem = total - sc.n_full * sc.chunk;
        if (tail_min > 0 && sc.n_full > 0 && rem < rem_min) { --sc.n_full; rem += sc.chunk; }
        int n_tail = 0;
        while (rem > 0) {
            int len = rem;
            if (tail_min > 0 && rem > tail_min && n_tail < PT_MAX_TAIL_CHUNKS - 1) len = std::max(tail_min, (rem + 1) / 2);
            sc.tail_len[n_tail++] = len;
            rem -= len;
        }
        sc.n_chunks = sc.n_full + n_tail;
        return sc;
    };
    // kernel 1 (lane-per-pixel): optional spp chunks = separate launches.
    // Wavefront kernel, schedule 1 (default): a short cost pre-pass (prepass_spp samples of every pixel, rays counted), a
    // counting sort of the pixel queue by that cost, then ONE persistent launch over the cost-ordered queue whose first chunk
    // is sticky_pct % of the remaining samples (a slot keeps its pixel, no hand-offs, expensive pixels start first) and whose
    // last samples go round in halving chunks through the per-chunk rings, so that the frame ends on ~n_pixels short work items.
    // schedule 0 (or spp_per_launch, which the resumability tests use): one launch, chunk_spp chunks + halving tail.
    // All schedules give the same image bit for bit (a pixel's stream does not depend on who renders it, or when).
    int S, n_launch;
    Schedule main_sc;
    if (c->kernel == 1) {
        S = c->spp_per_launch > 0 ? std::min(c->spp_per_launch, max_samples) : max_samples;
        S = std::min(S, 65535);
        n_launch = (max_samples + S - 1) / S;
    } else {
        S = max_samples;
        n_launch = sorted ? 2 : 1;
        if (sorted) {
            const int rest = max_samples - pre;
            // Share of a pixel's remaining samples that its first slot renders in one go.  With many more pixels than slots the
            // frame is throughput-bound and hand-offs are pure overhead: 80 %; with fewer pixels per slot 50-65 % (a launch whose
            // pixels all have a slot and whose costs have a tail runs the tier schedule instead).
            int sticky = c->sticky_pct;
            if (sticky < 1) {
                const double ratio = (double)c->n_pixels / ((double)c->num_cus * (double)bpc * (double)ns);
                sticky = (int)std::min(80.0, 50.0 + 6.0 * ratio); // (rounds 1-2, queue ordered by rays: 10-75, rising with the ratio; re-swept with the
                                                                 // queue ordered by time: C4 80, 1/2 shard and C3 65, 1/4 shard 50-65, C2 60 - r3_ab57/58.log)
            }
            // With the queue ordered by the TIME of a pixel's samples the hand-offs of the tail buy little and every lap is a barrier of
            // sorts: the last quarter goes in two chunks, not five (C4 528 -> 500 ms, C3 161 -> 153; smallest tail chunk 16 / 32 / 64 / 128
            // of 1016 samples: 528 / 515 / 505 / 498 ms; profiles/r03_logs/r3_ab54.log).
            if (c->chunk_tail_min < 0) tail_min = std::max(16, rest / 8);
            const int big = std::max(1, (int)((int64_t)rest * sticky / 100));
            main_sc = make_schedule(rest, big, rest - big);
            if ((rc = ensure(c, c->d_cost, (size_t)W * H))) return rc; // cost image; zero where this rank owns nothing
            if ((rc = ensure(c, c->d_bucket, (size_t)c->n_pixels))) return rc;
            HIP_TRY(c, hipMemsetAsync(c->d_cost.p, 0, (size_t)W * H, stream));
            if ((rc = ensure(c, c->d_sorted, (size_t)c->n_pixels * 4))) return rc;
            if ((rc = ensure(c, c->d_sort_scratch, pt_sort_scratch_bytes(c->n_pixels)))) return rc;
        } else {
            const int chunk = std::min(c->spp_per_launch > 0 ? c->spp_per_launch : c->chunk_spp, std::min(max_samples, 65535));
            main_sc = make_schedule(max_samples, chunk, 1);
        }
        const int n_chunks = main_sc.n_chunks;
        if ((uint64_t)c->n_pixels * (uint64_t)n_chunks >= 0xfff00000ull) return fail(c, PT_E_LIMIT, "too many (pixel, chunk) tickets");
        if (n_chunks > 255 || c->n_pixels >= (1u << 24)) return fail(c, PT_E_LIMIT, "the wavefront kernel needs n_chunks <= 255 and < 2^24 pixels per rank (raise chunk_spp)");
        // one ring of ready pixels per chunk index: ring c holds, in completion order of chunk c - 1, the pixels whose chunk c
        // may start.  d_laps = watchdog flag + one fill counter per ring.
        // Layout (never a plain store in a cache line that also holds device-scope atomics): [0] watchdog flag | +256 B: ring fill
        // counters (n_chunks + 1, + the pre-pass's spare) | 256-B aligned: diagnostics timelines of the two launches.
        c->lap_ticks_ofs = ((256 + (size_t)(n_chunks + 3) * 4 + 255) / 256) * 256;
        const size_t laps_bytes = c->lap_ticks_ofs + ((size_t)PT_LAP_REGION(n_chunks) + (size_t)PT_LAP_REGION(1)) * 8;
        if ((rc = ensure(c, c->d_laps, laps_bytes))) return rc;
        if ((rc = ensure(c, c->d_ring, (size_t)c->n_pixels * 4 * (size_t)n_chunks))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->d_laps.p, 0, laps_bytes, stream));
        if (n_chunks > 1) HIP_TRY(c, hipMemsetAsync(c->d_ring.p, 0, (size_t)c->n_pixels * 4 * (size_t)n_chunks, stream));
    }
    const int n_chunks = main_sc.n_chunks;
    if ((rc = ensure(c, c->d_heads, (size_t)n_launch * PT_HEADS_WORDS * 4))) return rc; // per launch: the ticket counter, the express counter 256 bytes on, the tier counters
    HIP_TRY(c, hipMemsetAsync(c->d_heads.p, 0, (size_t)n_launch * PT_HEADS_WORDS * 4, stream));
    if (tiers && (rc = ensure(c, c->d_tiers, (1 + PT_MAX_TIERS * PT_TIER_WORDS) * 4))) return rc;
    HIP_TRY(c, hipMemsetAsync(d_out_rgb, 0, (size_t)W * H * 3 * sizeof(float), stream));
    if (d_out_rgba8) HIP_TRY(c, hipMemsetAsync(d_out_rgba8, 0, (size_t)W * H * 4, stream));
    if (n_launch > 1 || n_chunks > 1) {
        if ((rc = ensure(c, c->d_rng, (size_t)W * H * 4))) return rc;
        if ((rc = ensure(c, c->d_accum, (size_t)W * H * 12))) return rc;
    }
    if (use_count) {
        if ((rc = ensure(c, c->d_counters, sizeof(PtCounters)))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->d_counters.p, 0, sizeof(PtCounters), stream));
    }

    std::memcpy(P.cam, cam, sizeof(float) * 12);
    P.pixel_ids = (const uint32_t*)c->d_pixels.p;
    P.n_pixels = c->n_pixels;
    P.rng_state = (uint32_t*)c->d_rng.p;
    P.accum = (float*)c->d_accum.p;
    P.out_rgb = (float*)d_out_rgb;
    P.out_rgba8 = (uint32_t*)d_out_rgba8;
    P.counters = use_count ? (PtCounters*)c->d_counters.p : nullptr;
    P.width = W;
    P.height = H;
    P.max_samples = max_samples;
    P.max_depth = max_depth;
    P.ring = (uint32_t*)c->d_ring.p;
    P.ring_tail = c->d_laps.p ? (uint32_t*)c->d_laps.p + 64 : nullptr;
    P.error_flag = (uint32_t*)c->d_laps.p;
    P.lap_ticks = (unsigned long long*)((char*)c->d_laps.p + c->lap_ticks_ofs);
    c->last_chunks = n_chunks;
    P.cost_out = nullptr;
    P.timeline = c->timeline;
    P.dbg_start = nullptr;
    P.dbg_cost = nullptr;
    if (c->latency && sorted) {
        if ((rc = ensure(c, c->d_dbg_start, (size_t)W * H * 8))) return rc; // + rays per pixel (instrumented instance)
        HIP_TRY(c, hipMemsetAsync(c->d_dbg_start.p, 0, (size_t)W * H * 8, stream));
        P.dbg_cost = (uint8_t*)c->d_cost.p;
    }
    P.census_mode = c->census_mode;
    P.chunk_spp = main_sc.chunk;
    P.n_chunks = n_chunks;
    P.n_tickets = c->n_pixels * (uint32_t)n_chunks;
    // Express pixels (pt_kernel.hip, take_ticket): the most expensive entries of the cost-ordered queue get waves of their own when the
    // frame is bound by its longest sample chains, i.e. when (nearly) every pixel is in flight from the start - few pixels per path slot
    // (a shard of a multi-GPU frame, a small image).  A throughput-bound frame (many pixels per slot) has none: sparse waves would only
    // take wave slots from it.  Options "express_permille" (-1 = automatic: 10 per mille up to 1.5 pixels per slot, 0 from 4),
    // "ns_express" (8 pixels per express wave), at most an eighth of the waves.
    if (sorted && P.nodes8 && n_chunks <= 254 && (uint64_t)c->n_pixels * (uint64_t)n_chunks < 0xE0000000ull) {
        int& rgrid = tiers ? ring_grid : grid; // the workgroups of the ring schedule (with a tier plan the launch has all resident ones)
        const double ratio = (double)c->n_pixels / ((double)rgrid * (double)ns);
        double permille = c->express_permille >= 0 ? (double)c->express_permille : (ratio <= 1.5 ? 10.0 : (ratio >= 4.0 ? 0.0 : 10.0 * (4.0 - ratio) / 2.5));
        const int nse = std::max(1, std::min(c->ns_express, ns));
        uint64_t want = (uint64_t)((double)c->n_pixels * permille / 1000.0);
        const int capacity = c->num_cus * bpc;
        // at most an eighth of the ring schedule's waves, or what the bulk leaves free.  (Round 4 tried up to 60 % of the waves for 1-15 % of the
        // pixels at 8-48 per wave, also with the grid oversubscribed by the express waves: world 2 351 -> 400-470 ms, world 4 269 -> 300-370,
        // profiles/r04_notes.md 5.)
        const int cap_waves = std::max(rgrid / 8, std::min(capacity / 2, capacity - rgrid));
        want = std::min<uint64_t>(want, (uint64_t)cap_waves * (uint64_t)nse);
        if (want > 0 && want < c->n_pixels) {
            n_express = (uint32_t)want;
            express_waves = (int)((want + (uint64_t)nse - 1) / (uint64_t)nse);
            P.ns_express = nse;
            // wave slots the bulk does not fill (a shard, a small image) hold the express waves on top of the bulk's
            rgrid = std::min(capacity, (int)(((long)c->n_pixels - (long)n_express + ns - 1) / ns) + express_waves);
            if (state_words) {
                if ((rc = ensure(c, c->d_slots, state_words * 4 * (size_t)grid))) return rc;
                P.slot_state = (uint32_t*)c->d_slots.p;
            }
        }
    }
    P.n_full = main_sc.n_full;
    for (int i = 0; i < PT_MAX_TAIL_CHUNKS; ++i) P.tail_len[i] = main_sc.tail_len[i];

    HIP_TRY(c, hipEventRecord(c->ev0, stream));
    if (c->kernel == 2 && (rc = ensure(c, c->d_params, sizeof(PtKernelParams) * (size_t)n_launch))) return rc;
    for (int l = 0; l < n_launch; ++l) {
        P.queue_head = (uint32_t*)c->d_heads.p + PT_HEADS_WORDS * l;
        P.sample_begin = l * S;
        P.sample_count = std::min(S, max_samples - l * S);
        if (sorted) { // launch 0: cost pre-pass in queue order; launch 1: everything else, expensive pixels first
            P.sample_begin = l == 0 ? 0 : pre;
            P.sample_count = l == 0 ? pre : max_samples - pre;
            P.pixel_ids = l == 0 ? (const uint32_t*)c->d_pixels.p : (const uint32_t*)c->d_sorted.p;
            P.cost_out = l == 0 ? (uint8_t*)c->d_cost.p : nullptr;
            P.dbg_start = (l == 1 && c->latency) ? (uint32_t*)c->d_dbg_start.p : nullptr;
            P.lap_ticks = (unsigned long long*)((char*)c->d_laps.p + c->lap_ticks_ofs) + (l == 0 ? PT_LAP_REGION(n_chunks) : 0); // the pre-pass's block follows the main launch's
            P.ring_tail = (uint32_t*)c->d_laps.p + 64 + (l == 0 ? n_chunks : 0); // the pre-pass only uses its [1]: the spare counter
            if (l == 0) { // one chunk per pixel
                P.chunk_spp = P.sample_count;
                P.n_chunks = 1;
                P.n_full = 1;
                P.n_tickets = c->n_pixels;
                P.n_express = 0;
                P.express_waves = 0;
                P.tiers = nullptr;
            } else {
                P.chunk_spp = main_sc.chunk;
                P.n_chunks = n_chunks;
                P.n_full = main_sc.n_full;
                P.n_express = n_express;
                P.express_waves = express_waves;
                P.n_tickets = (c->n_pixels - n_express) * (uint32_t)n_chunks;
                if (tiers) { // the plan decides on the device: pixels by cost class (then none of the above is used), or the ring schedule as prepared
                    P.tiers = (const uint32_t*)c->d_tiers.p;
                    P.ring_grid = ring_grid;
                }
            }
            if (l == 1) {
                HIP_TRY(c, pt_launch_sort_pixels((const uint8_t*)c->d_cost.p, W, H, c->cost_radius, (const uint32_t*)c->d_pixels.p, (uint32_t*)c->d_sorted.p,
                                                 c->n_pixels, (uint32_t)pre, (uint32_t*)c->d_sort_scratch.p, (uint8_t*)c->d_bucket.p, stream));
                if (tiers) HIP_TRY(c, pt_launch_plan_tiers((const uint32_t*)c->d_sort_scratch.p, c->n_pixels, grid, ns, c->whole > 0, (uint32_t*)c->d_tiers.p, stream));
                HIP_TRY(c, hipEventRecord(c->evm, stream));
            }
        }
        const PtKernelParams* dP = (const PtKernelParams*)c->d_params.p + l; // one block per launch: launch l+1's copy never races launch l
        if (c->kernel == 2) HIP_TRY(c, pt_launch_store_params(&P, (PtKernelParams*)dP, stream)); // by value: P is reused for the next launch
        // (with a tier plan prepared only the main launch has every resident workgroup; the pre-pass measures the pixels' costs in waves
        // as dense as the ring schedule's - C2 74.3 -> 71 ms, 1/8 shard 198 -> 194)
        HIP_TRY(c, pt_launch_render(&P, dP, variant, (tiers && sorted && l == 0) ? ring_grid : grid, lds, stream, use_count));
    }
    HIP_TRY(c, hipEventRecord(c->ev1, stream));
    c->ev_pending = true;
    c->flag_pending = true;
    c->last_stream = stream;
    c->last_launches = n_launch;
    c->last_sorted = sorted;
    c->last_w = W;
    c->last_h = H;
    c->stats.vgprs = vg;
    c->stats.kernel_variant = variant;
    c->stats.express_pixels = n_express;
    c->stats.whole_pixels = tiers ? (int32_t)c->n_pixels : 0;
    c->stats.prepass_spp = sorted ? pre : 0;
    c->stats.sgprs = sg;
    c->stats.lds_bytes = (int)lds + slds;
    c->stats.block = block;
    c->stats.grid = grid;
    c->stats.stack_entries = P.stack_entries;
    return PT_OK;
}


int pt_synchronize(pt_ctx* c)
{
    if (!c) return PT_E_INVALID;
    if (c->host_only) return PT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->last_stream && c->last_stream != c->stream) HIP_TRY(c, hipStreamSynchronize(c->last_stream)); // pt_render_device on a caller's stream
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return check_watchdog(c);
}

int pt_render(pt_ctx* c, const pt_camera* cam, int32_t W, int32_t H, int32_t max_samples, int32_t max_depth, float* out_rgb, uint32_t* out_rgba8)
{
    // with a communicator attached (pt_comm_init_rank) only rank 0 receives the frame; the other ranks may pass NULL
    const bool root = !c || !c->comm || c->comm_rank == 0;
    if (!c || !cam || (root && !out_rgb)) return PT_E_INVALID;
    if (c->host_only) return fail(c, PT_E_NO_DEVICE, "host-only context: the HIP render path is required and there is no CPU fallback");
    if (W <= 0 || H <= 0) return fail(c, PT_E_INVALID, "bad render size %dx%d", W, H);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    size_t npx = (size_t)W * H;
    if ((rc = ensure(c, c->d_out, npx * 12))) return rc;
    if (out_rgba8 && (rc = ensure(c, c->d_out8, npx * 4))) return rc;
    // with a communicator the RGBA8 image is made from the reduced float frame on the root (pt_reduce_framebuffer): every rank
    // enqueues the same single collective whatever buffers its caller passed
    rc = pt_render_device(c, cam, W, H, max_samples, max_depth, c->d_out.p, (out_rgba8 && !c->comm) ? c->d_out8.p : nullptr, nullptr);
    if (rc) return rc;
    // N ranks: the one collective of the path - RCCL sum-reduce of the float3 framebuffer onto rank 0 (pt_comm.cpp)
    if (c->comm && (rc = pt_reduce_framebuffer(c, c->d_out.p, (root && out_rgba8) ? c->d_out8.p : nullptr, (int64_t)npx, nullptr))) return rc;
    HIP_TRY(c, hipEventRecord(c->evr, c->stream)); // kernels end (ev1) .. here: this rank's share of the reduce, incl. waiting for the slowest rank
    if (root) {
        HIP_TRY(c, hipMemcpyAsync(out_rgb, c->d_out.p, npx * 12, hipMemcpyDeviceToHost, c->stream));
        if (out_rgba8) HIP_TRY(c, hipMemcpyAsync(out_rgba8, c->d_out8.p, npx * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipEventRecord(c->evd, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    {
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev1, c->evr));
        c->stats.reduce_ms = ms;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->evr, c->evd));
        c->stats.d2h_ms = ms;
    }
    return check_watchdog(c);
}

int pt_get_stats(pt_ctx* c, pt_stats* out)
{
    if (!c || !out) return PT_E_INVALID;
    if (!c->host_only && c->ev_pending) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipEventSynchronize(c->ev1));
        int wrc = check_watchdog(c);
        if (wrc) return wrc;
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->stats.kernel_ms = ms;
        c->stats.prepass_ms = 0.0;
        if (c->last_sorted) {
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->evm));
            c->stats.prepass_ms = ms;
        }
        c->stats.launches = c->last_launches;
        c->ev_pending = false;
        if (c->count && c->d_counters.p) {
            PtCounters h;
            HIP_TRY(c, hipMemcpy(&h, c->d_counters.p, sizeof(h), hipMemcpyDeviceToHost));
            c->stats.samples = h.samples; c->stats.rays = h.rays; c->stats.nodes = h.nodes; c->stats.tris = h.tris;
            c->stats.scatters = h.scatters; c->stats.env_misses = h.env_misses; c->stats.nan_retries = h.nan_retries;
            for (int i = 0; i < 32; ++i) c->stats.sched[i] = h.sched[i];
            for (int i = 0; i < 8; ++i) c->stats.groups[i] = h.grp[i];
            for (int i = 0; i < 16; ++i) c->stats.lobes[i] = h.lobes[i];
            for (int i = 0; i < 4; ++i) c->stats.trav[i] = h.trav[i];
        }
    }
    *out = c->stats;
    return PT_OK;
}

void pt_to_camera_data(const float look_from[3], const float look_at[3], const float look_up[3], float vfov, int32_t w, int32_t h, pt_camera* out)
{
    // camera.cpp:3-21; host code, host libm tan as in the reference.  dot/cross use the same fused forms as the device code.
    auto dot = [](const float* a, const float* b) { return std::fma(a[2], b[2], std::fma(a[1], b[1], a[0] * b[0])); };
    auto cross = [](const float* a, const float* b, float* r) {
        r[0] = std::fma(a[1], b[2], -(a[2] * b[1]));
        r[1] = std::fma(a[2], b[0], -(a[0] * b[2]));
        r[2] = std::fma(a[0], b[1], -(a[1] * b[0]));
    };
    auto normalize = [&](float* v) {
        float s = 1.0f / std::sqrt(dot(v, v));
        v[0] *= s; v[1] *= s; v[2] *= s;
    };
    const float pi = 3.14159265358979323f;
    float aspect = (float)w / (float)h;
    float theta = vfov * pi / 180.0f;
    float hh = std::tan(theta / 2);
    float vh = 2.0f * hh, vw = aspect * vh;
    float W[3] = {look_from[0] - look_at[0], look_from[1] - look_at[1], look_from[2] - look_at[2]};
    normalize(W);
    float U[3], Vv[3];
    cross(look_up, W, U);
    normalize(U);
    cross(W, U, Vv);
    normalize(Vv);
    for (int i = 0; i < 3; ++i) {
        out->origin[i] = look_from[i];
        out->horizontal[i] = vw * U[i];
        out->vertical[i] = vh * Vv[i];
        out->llc[i] = look_from[i] - out->horizontal[i] / 2.0f - out->vertical[i] / 2.0f - W[i];
    }
}

int pt_debug_closest_hit_host(pt_ctx* c, const float org[3], const float dir[3], float tmin, float tmax, float* t, float* u, float* v, int32_t* prim)
{
    if (!c || !c->have_scene) return PT_E_NO_SCENE;
    return pt_bvh_closest_hit_host(c->bvh, org, dir, tmin, tmax, t, u, v, prim) ? 1 : 0;
}

int pt_debug_clone_scene(pt_ctx* dst, const pt_ctx* src)
{
    if (!dst || !src || dst == src) return PT_E_INVALID;
    return pti::clone_scene(dst, src);
}

int pt_debug_quad_info(pt_ctx* c, int64_t out[8])
{
    if (!c || !out) return PT_E_INVALID;
    if (!c->have_scene) return fail(c, PT_E_NO_SCENE, "pt_debug_quad_info before pt_upload_scene");
    // {quad nodes, depth, leaf slots, triangles in leaf slots, empty slots, internal slots, binary nodes, binary leaf references}
    int64_t leaf_slots = 0, tris = 0, empty = 0, internal = 0, bin_leaves = 0;
    for (const PtNode4& q : c->nodes4) {
        for (int k = 0; k < 4; ++k) {
            const int32_t r = q.child[k];
            if (r >= 0) ++internal;
            else if (r == -1) {
                ++empty;
                for (int a = 0; a < 3; ++a)
                    if (!(q.lo[a][k] == INFINITY && q.hi[a][k] == INFINITY)) return fail(c, PT_E_LIMIT, "quad node: empty slot with a finite box");
            } else {
                ++leaf_slots;
                tris += (int64_t)(~(uint32_t)r & 7u);
            }
        }
    }
    for (const PtNode& nd : c->bvh.nodes) {
        if (nd.left < -1) ++bin_leaves;
        if (nd.right < -1) ++bin_leaves;
    }
    out[0] = (int64_t)c->nodes4.size(); out[1] = c->depth4; out[2] = leaf_slots; out[3] = tris; out[4] = empty; out[5] = internal;
    out[6] = (int64_t)c->bvh.nodes.size(); out[7] = bin_leaves;
    return PT_OK;
}

int pt_debug_oct_info(pt_ctx* c, int64_t out[8])
{
    if (!c || !out) return PT_E_INVALID;
    if (!c->have_scene) return fail(c, PT_E_NO_SCENE, "pt_debug_oct_info before pt_upload_scene");
    // {oct nodes, depth, leaf slots, triangles in leaf slots, empty slots, internal slots, largest leaf, triangle slots of the scene}
    int64_t leaf_slots = 0, tris = 0, empty = 0, internal = 0, max_leaf = 0;
    std::vector<uint8_t> seen(c->bvh.tris.size(), 0);
    std::vector<uint8_t> referenced(c->nodes8.size(), 0);
    for (const PtNode8& q : c->nodes8) {
        for (int k = 0; k < 8; ++k) {
            const int32_t r = q.c[k].ref;
            if (r >= 0) {
                if ((size_t)r >= c->nodes8.size() || referenced[(size_t)r]++) return fail(c, PT_E_LIMIT, "oct node: child %d out of range or referenced twice", r);
                ++internal;
            } else if (r == -1) {
                ++empty;
                for (int a = 0; a < 3; ++a)
                    if (!(q.c[k].lo[a] == INFINITY && q.c[k].hi[a] == INFINITY)) return fail(c, PT_E_LIMIT, "oct node: empty slot with a finite box");
            } else {
                const uint32_t code = ~(uint32_t)r, first = code >> 3, count = code & 7u;
                ++leaf_slots;
                tris += count;
                max_leaf = std::max<int64_t>(max_leaf, count);
                for (uint32_t t = first; t < first + count; ++t) {
                    if (t >= seen.size() || seen[t]++) return fail(c, PT_E_LIMIT, "oct node: triangle slot %u out of range or in two leaves", t);
                    // the leaf's box must hold its triangles (wide leaves take the box of the subtree they replace)
                    const PtTri& tr = c->bvh.tris[t];
                    if (tr.id == 0x7fffffff) continue; // leaf_align padding
                    for (int a = 0; a < 3; ++a) {
                        const float lo = std::min(tr.p0[a], std::min(tr.p1[a], tr.p2[a])), hi = std::max(tr.p0[a], std::max(tr.p1[a], tr.p2[a]));
                        if (lo < q.c[k].lo[a] || hi > q.c[k].hi[a]) return fail(c, PT_E_LIMIT, "oct node: triangle slot %u sticks out of its leaf box", t);
                    }
                }
            }
        }
    }
    out[0] = (int64_t)c->nodes8.size(); out[1] = c->depth8; out[2] = leaf_slots; out[3] = tris; out[4] = empty; out[5] = internal;
    out[6] = max_leaf; out[7] = (int64_t)c->bvh.tris.size();
    return PT_OK;
}

int64_t pt_debug_read_queue(pt_ctx* c, uint32_t* queue_ids, uint32_t* input_ids, uint8_t* cost, int64_t cap)
{
    if (!c) return PT_E_INVALID;
    if (c->host_only) return fail(c, PT_E_NO_DEVICE, "host-only context: pt_debug_read_queue needs the GPU");
    if (!c->last_sorted) return 0;
    const int64_t n = std::min<int64_t>(cap, c->n_pixels);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (queue_ids) HIP_TRY(c, hipMemcpy(queue_ids, c->d_sorted.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (input_ids) HIP_TRY(c, hipMemcpy(input_ids, c->d_pixels.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (cost) { // cost image -> cost of each input queue entry
        std::vector<uint8_t> img((size_t)c->last_w * (size_t)c->last_h);
        std::vector<uint32_t> ids((size_t)n);
        HIP_TRY(c, hipMemcpy(img.data(), c->d_cost.p, img.size(), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(ids.data(), c->d_pixels.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) cost[i] = ids[(size_t)i] < img.size() ? img[ids[(size_t)i]] : 0;
    }
    return n;
}

int64_t pt_debug_read_laps(pt_ctx* c, uint64_t* ticks, int64_t cap)
{
    if (!c || !ticks) return PT_E_INVALID;
    if (c->host_only) return fail(c, PT_E_NO_DEVICE, "host-only context: pt_debug_read_laps needs the GPU");
    if (c->kernel != 2 || !c->d_laps.p) return 0;
    const int nt = 3 * (c->last_chunks + 1);
    std::vector<uint64_t> blk((size_t)PT_LAP_REGION(c->last_chunks));
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(blk.data(), (char*)c->d_laps.p + c->lap_ticks_ofs, blk.size() * 8, hipMemcpyDeviceToHost));
    int64_t n = 0; // the timeline, then the 64 latency accumulators, without the padding between them
    for (int i = 0; i < nt && n < cap; ++i) ticks[n++] = blk[(size_t)i];
    for (int i = 0; i < 64 && n < cap; ++i) ticks[n++] = blk[(size_t)PT_LAP_DIAG_OFS(c->last_chunks) + i];
    return n;
}

int64_t pt_debug_plan_tiers(const uint32_t* bucket_pixels, int32_t capacity, int32_t ns, int32_t force, uint32_t* words, int64_t cap)
{
    if (!bucket_pixels || !words || capacity < 1 || ns < 4 || ns > 255 || cap < 1 + PT_MAX_TIERS * PT_TIER_WORDS) return PT_E_INVALID;
    uint32_t start[PT_SORT_BUCKETS + 1];
    uint64_t run = 0;
    for (int b = 0; b < PT_SORT_BUCKETS; ++b) { start[b] = (uint32_t)run; run += bucket_pixels[b]; }
    if (run == 0 || run >= (1ull << 31)) return PT_E_INVALID;
    start[PT_SORT_BUCKETS] = (uint32_t)run;
    std::memset(words, 0, (size_t)(1 + PT_MAX_TIERS * PT_TIER_WORDS) * 4);
    pt_plan_tiers(start, capacity, ns, force, words); // the code of pt_plan_tiers_kernel, on the host
    return 1 + (int64_t)words[0] * PT_TIER_WORDS;
}

int64_t pt_debug_read_tiers(pt_ctx* c, uint32_t* words, int64_t cap)
{
    if (!c || !words) return PT_E_INVALID;
    if (c->host_only) return fail(c, PT_E_NO_DEVICE, "host-only context: pt_debug_read_tiers needs the GPU");
    if (!c->stats.whole_pixels || !c->d_tiers.p) return 0;
    const int64_t n = std::min<int64_t>(cap, 1 + PT_MAX_TIERS * PT_TIER_WORDS);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(words, c->d_tiers.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return n;
}

int64_t pt_debug_read_finish(pt_ctx* c, uint32_t* ticks, int64_t cap)
{
    if (!c || !ticks) return PT_E_INVALID;
    if (c->host_only) return fail(c, PT_E_NO_DEVICE, "host-only context: pt_debug_read_finish needs the GPU");
    if (c->kernel != 2 || !c->latency || !c->last_sorted || !c->d_dbg_start.p) return 0;
    const int64_t n = std::min<int64_t>(cap, 2 * (int64_t)c->last_w * c->last_h);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(ticks, c->d_dbg_start.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return n;
}

int pt_debug_eval(pt_ctx* c, int32_t op, const float* in, int32_t in_stride, float* out, int32_t out_stride, int64_t n)
{
    if (!c || !in || !out || n < 0 || in_stride < 1 || out_stride < 1) return PT_E_INVALID;
    if (c->host_only) return fail(c, PT_E_NO_DEVICE, "host-only context: pt_debug_eval needs the GPU");
    if (n == 0) return PT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if ((rc = upload(c, c->d_dbg_in, in, (size_t)n * in_stride * 4))) return rc;
    if ((rc = ensure(c, c->d_dbg_out, (size_t)n * out_stride * 4))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->d_dbg_out.p, 0, (size_t)n * out_stride * 4, c->stream));
    PtKernelParams P;
    fill_params(c, P);
    if (!c->have_scene) { P.root = -1; P.stack_entries = 1; }
    size_t lds = (size_t)P.stack_entries * pt_debug_block() * 4;
    HIP_TRY(c, pt_launch_debug(&P, op, (const float*)c->d_dbg_in.p, in_stride, (float*)c->d_dbg_out.p, out_stride, (long long)n, lds, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, c->d_dbg_out.p, (size_t)n * out_stride * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

} // extern "C"
