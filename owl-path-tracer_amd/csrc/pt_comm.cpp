// pt_comm.cpp -- the ONE collective of the render path, inside the library: an RCCL sum-reduce of the float3 framebuffer onto
// rank 0 over xGMI (the optional RGBA8 image is quantised from the reduced frame on the root), for both ways of driving N GPUs:
//   * one process per GPU (torchrun / mpirun style): pt_comm_get_unique_id on rank 0, the 128 bytes travel by whatever the
//     launcher offers, every rank calls pt_comm_init_rank; after that pt_render() renders the rank's pixel shard, reduces, and
//     rank 0 receives the complete frame;
//   * one process, N GPUs (the C++ entry point, `pt_main --gpus N`): pt_group_create / pt_group_render.
// The reference is single-GPU (create_context(nullptr, 1), path_tracer/src/application.cpp:62; render + read-back
// application.cpp:363-369): this replaces that render call for N > 1.  Every pixel has exactly one non-zero contributor
// (pt_set_pixel_shard), so the sum is exact and the N-GPU frame is bit-identical to the 1-GPU frame.
//
// librccl.so.1 is resolved at the first communicator call (dlopen): a single-GPU user never maps the 570 MB library, and a
// process that already holds an RCCL (PyTorch-ROCm bundles one under the same SONAME) keeps exactly one copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "pt_internal.h"
#include "pt_launch.h"

namespace {

// Types and prototypes come from the RCCL header; the functions themselves are resolved with dlsym (see the file comment).
static_assert(PT_COMM_ID_BYTES == sizeof(ncclUniqueId), "PT_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    // PT_RCCL_PATH names the library to use (and nothing else is tried): a launcher that ships its own RCCL, or a test that wants
    // the "RCCL unavailable" error
    const char* forced = getenv("PT_RCCL_PATH");
    const char* defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string first_error;
    for (int i = 0; i < (forced && forced[0] ? 1 : 3); ++i) {
        const char* n = (forced && forced[0]) ? forced : defaults[i];
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
        const char* e = dlerror(); // once: the call clears the error
        if (first_error.empty()) first_error = e ? e : n;
    }
    if (!g_rccl.handle) {
        g_rccl.error = std::string(forced && forced[0] ? "PT_RCCL_PATH: " : "librccl.so.1 not found: ") + first_error;
        return;
    }
    struct { void** fn; const char* name; } syms[] = {
        {(void**)&g_rccl.GetUniqueId, "ncclGetUniqueId"},   {(void**)&g_rccl.CommInitRank, "ncclCommInitRank"}, {(void**)&g_rccl.CommInitAll, "ncclCommInitAll"},
        {(void**)&g_rccl.CommDestroy, "ncclCommDestroy"},   {(void**)&g_rccl.Reduce, "ncclReduce"},             {(void**)&g_rccl.GroupStart, "ncclGroupStart"},
        {(void**)&g_rccl.GroupEnd, "ncclGroupEnd"},         {(void**)&g_rccl.GetErrorString, "ncclGetErrorString"}};
    for (auto& s : syms) {
        *s.fn = dlsym(g_rccl.handle, s.name);
        if (!*s.fn) { g_rccl.error = std::string("librccl.so.1 lacks ") + s.name; return; }
    }
}

// 0 or PT_E_HIP with the reason in ctx (or the creation-error slot when ctx is null)
int need_rccl(pt_ctx* c)
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.error.empty()) return pti::fail(c, PT_E_HIP, "RCCL unavailable: %s", g_rccl.error.c_str());
    return PT_OK;
}

#define RCCL_TRY(c, call)                                                                                            \
    do {                                                                                                             \
        ncclResult_t r__ = (call);                                                                                            \
        if (r__ != ncclSuccess) return pti::fail(c, PT_E_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(r__));   \
    } while (0)

} // namespace

struct pt_group {
    std::vector<pt_ctx*> ctx;
    std::vector<void*> d_rgb, d_rgba8; // per device: W*H*3 floats / W*H uint32
    size_t cap_px = 0;
    float* pinned_rgb = nullptr;       // staging for the root's D2H (pinned: the reference's framebuffer is pinned host memory, owl.hpp:108-111)
    uint32_t* pinned_rgba8 = nullptr;
    std::string err;
};

extern "C" {

int pt_comm_get_unique_id(uint8_t id[PT_COMM_ID_BYTES])
{
    if (!id) return PT_E_INVALID;
    int rc = need_rccl(nullptr);
    if (rc) return rc;
    ncclUniqueId u;
    RCCL_TRY(nullptr, g_rccl.GetUniqueId(&u));
    std::memcpy(id, u.internal, PT_COMM_ID_BYTES);
    return PT_OK;
}

int pt_comm_init_rank(pt_ctx* c, const uint8_t id[PT_COMM_ID_BYTES], int32_t rank, int32_t world_size)
{
    if (!c || !id || world_size < 1 || rank < 0 || rank >= world_size) return PT_E_INVALID;
    if (c->host_only) return pti::fail(c, PT_E_NO_DEVICE, "host-only context: a communicator needs the GPU");
    if (c->comm) return pti::fail(c, PT_E_INVALID, "this context already has a communicator (pt_comm_destroy first)");
    int rc = need_rccl(c);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, PT_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    RCCL_TRY(c, g_rccl.CommInitRank(&comm, world_size, u, rank));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world_size;
    return pt_set_pixel_shard(c, rank, world_size, c->tile > 0 ? c->tile : 16);
}

int pt_comm_destroy(pt_ctx* c)
{
    if (!c) return PT_E_INVALID;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)g_rccl.CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    c->comm_rank = 0;
    c->comm_world = 1;
    return pt_set_pixel_shard(c, 0, 1, c->tile > 0 ? c->tile : 16); // the context renders every pixel again
}

int pt_reduce_framebuffer(pt_ctx* c, void* d_rgb, void* d_rgba8, int64_t n_pixels, void* stream_v)
{
    if (!c || !d_rgb || n_pixels <= 0) return PT_E_INVALID;
    if (!c->comm) return PT_OK; // a single rank owns every pixel: nothing to add
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : c->stream;
    // ONE collective, in place on every rank; only the root's buffer holds the sum afterwards.  The RGBA8 image is not reduced: every
    // pixel has exactly one non-zero contributor, so the root quantises the reduced float frame and gets bit for bit what the owning
    // rank would have stored - and the number of collectives a rank enqueues never depends on the buffers its caller happened to pass
    // (round 2 reduced d_rgba8 only where it was non-null: a root with an RGBA8 buffer and peers without one deadlocked).
    RCCL_TRY(c, g_rccl.Reduce(d_rgb, d_rgb, (size_t)n_pixels * 3, ncclFloat32, ncclSum, 0, (ncclComm_t)c->comm, stream));
    if (d_rgba8 && c->comm_rank == 0) HIP_TRY(c, pt_launch_pack_rgba8((const float*)d_rgb, (uint32_t*)d_rgba8, (long long)n_pixels, stream));
    return PT_OK;
}

void* pt_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void pt_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

// ---- one process, N GPUs -----------------------------------------------------------------------------------------------------

pt_group* pt_group_create(const int32_t* devices, int32_t n)
{
    if (n < 1 || n > 64) { pti::fail(nullptr, PT_E_INVALID, "pt_group_create: %d devices", n); return nullptr; }
    pt_group* g = new pt_group();
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) devs[(size_t)i] = devices ? devices[i] : i;
    for (int i = 0; i < n; ++i) {
        pt_config cfg{devs[(size_t)i], 0};
        pt_ctx* c = pt_create(&cfg);
        if (!c) { pt_group_destroy(g); return nullptr; } // pt_last_error(NULL) has the reason
        g->ctx.push_back(c);
    }
    if (n > 1) {
        if (need_rccl(nullptr)) { pt_group_destroy(g); return nullptr; }
        std::vector<ncclComm_t> comms((size_t)n, nullptr);
        ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, devs.data());
        if (r != ncclSuccess) {
            pti::fail(nullptr, PT_E_HIP, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
            pt_group_destroy(g);
            return nullptr;
        }
        for (int i = 0; i < n; ++i) {
            g->ctx[(size_t)i]->comm = comms[(size_t)i];
            g->ctx[(size_t)i]->comm_rank = i;
            g->ctx[(size_t)i]->comm_world = n;
        }
    }
    for (int i = 0; i < n; ++i) (void)pt_set_pixel_shard(g->ctx[(size_t)i], i, n, 16);
    g->d_rgb.assign((size_t)n, nullptr);
    g->d_rgba8.assign((size_t)n, nullptr);
    return g;
}

void pt_group_destroy(pt_group* g)
{
    if (!g) return;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        pt_ctx* c = g->ctx[i];
        (void)hipSetDevice(c->device);
        (void)pt_comm_destroy(c);
        if (i < g->d_rgb.size() && g->d_rgb[i]) (void)hipFree(g->d_rgb[i]);
        if (i < g->d_rgba8.size() && g->d_rgba8[i]) (void)hipFree(g->d_rgba8[i]);
        pt_destroy(c);
    }
    pt_host_free(g->pinned_rgb);
    pt_host_free(g->pinned_rgba8);
    delete g;
}

int32_t pt_group_size(const pt_group* g) { return g ? (int32_t)g->ctx.size() : 0; }
pt_ctx* pt_group_ctx(pt_group* g, int32_t i) { return (g && i >= 0 && i < (int32_t)g->ctx.size()) ? g->ctx[(size_t)i] : nullptr; }
const char* pt_group_last_error(const pt_group* g) { return g ? g->err.c_str() : pt_last_error(nullptr); }

#define GROUP_EACH(g, expr)                                                                   \
    do {                                                                                      \
        for (pt_ctx * c_ : (g)->ctx) {                                                        \
            int rc_ = (expr);                                                                 \
            if (rc_) { (g)->err = pt_last_error(c_); return rc_; }                            \
        }                                                                                     \
    } while (0)

int pt_group_upload_scene(pt_group* g, const pt_mesh* meshes, int32_t n_meshes, const float* materials, int32_t n_materials, const pt_texture* textures,
                          int32_t n_textures, const int32_t* material_texture, const pt_env* env)
{
    if (!g) return PT_E_INVALID;
    // the BVH is built once (device 0's context), every other device gets a copy: full replica per GPU
    pt_ctx* c0 = g->ctx[0];
    int rc = pt_upload_scene(c0, meshes, n_meshes, materials, n_materials, textures, n_textures, material_texture, env);
    if (rc) { g->err = pt_last_error(c0); return rc; }
    for (size_t i = 1; i < g->ctx.size(); ++i) {
        rc = pti::clone_scene(g->ctx[i], c0);
        if (rc) { g->err = pt_last_error(g->ctx[i]); return rc; }
    }
    return PT_OK;
}

int pt_group_set_materials(pt_group* g, const float* materials, int32_t n_materials)
{
    if (!g) return PT_E_INVALID;
    GROUP_EACH(g, pt_set_materials(c_, materials, n_materials));
    return PT_OK;
}

int pt_group_set_option(pt_group* g, const char* key, int64_t value)
{
    if (!g) return PT_E_INVALID;
    GROUP_EACH(g, pt_set_option(c_, key, value));
    return PT_OK;
}

int pt_group_render(pt_group* g, const pt_camera* cam, int32_t W, int32_t H, int32_t max_samples, int32_t max_depth, float* out_rgb, uint32_t* out_rgba8)
{
    if (!g || !cam || !out_rgb || W <= 0 || H <= 0) return PT_E_INVALID;
    const size_t npx = (size_t)W * (size_t)H;
    const int n = (int)g->ctx.size();
    auto bad = [&](pt_ctx* c, int rc) { g->err = pt_last_error(c); return rc; };
    if (npx > g->cap_px) { // per-device framebuffers + pinned staging on the host
        for (int i = 0; i < n; ++i) {
            pt_ctx* c = g->ctx[(size_t)i];
            if (hipSetDevice(c->device) != hipSuccess) return bad(c, pti::fail(c, PT_E_HIP, "hipSetDevice(%d) failed", c->device));
            if (g->d_rgb[(size_t)i]) (void)hipFree(g->d_rgb[(size_t)i]);
            if (g->d_rgba8[(size_t)i]) (void)hipFree(g->d_rgba8[(size_t)i]);
            g->d_rgb[(size_t)i] = g->d_rgba8[(size_t)i] = nullptr;
            if (hipMalloc(&g->d_rgb[(size_t)i], npx * 12) != hipSuccess || hipMalloc(&g->d_rgba8[(size_t)i], npx * 4) != hipSuccess)
                return bad(c, pti::fail(c, PT_E_HIP, "framebuffer allocation on device %d failed", c->device));
        }
        pt_host_free(g->pinned_rgb);
        pt_host_free(g->pinned_rgba8);
        g->pinned_rgb = (float*)pt_host_alloc(npx * 12);
        g->pinned_rgba8 = (uint32_t*)pt_host_alloc(npx * 4);
        if (!g->pinned_rgb || !g->pinned_rgba8) return bad(g->ctx[0], pti::fail(g->ctx[0], PT_E_HIP, "pinned host framebuffer allocation failed"));
        g->cap_px = npx;
    }
    // on any failure below: remember the reason in the group, drain every device's stream (launches of the other devices may be in
    // flight), then return
    auto fail_all = [&](pt_ctx* c, int rc) {
        g->err = pt_last_error(c);
        for (pt_ctx* o : g->ctx) {
            (void)hipSetDevice(o->device);
            (void)hipStreamSynchronize(o->stream);
        }
        return rc;
    };
    // every device renders its own tiles (asynchronous launches, one stream per device); the RGBA8 image is made on device 0 from
    // the reduced float frame ...
    for (int i = 0; i < n; ++i) {
        pt_ctx* c = g->ctx[(size_t)i];
        int rc = pt_render_device(c, cam, W, H, max_samples, max_depth, g->d_rgb[(size_t)i], (out_rgba8 && n == 1) ? g->d_rgba8[(size_t)i] : nullptr, nullptr);
        if (rc) return fail_all(c, rc);
    }
    // ... then ONE reduce onto device 0 (grouped: one host thread drives all ranks of the communicator)
    pt_ctx* c0 = g->ctx[0];
    if (n > 1) {
        ncclResult_t r = g_rccl.GroupStart();
        if (r != ncclSuccess) return fail_all(c0, pti::fail(c0, PT_E_HIP, "ncclGroupStart failed: %s", g_rccl.GetErrorString(r)));
        pt_ctx* bad_ctx = nullptr;
        for (int i = 0; i < n && r == ncclSuccess; ++i) {
            pt_ctx* c = g->ctx[(size_t)i];
            r = g_rccl.Reduce(g->d_rgb[(size_t)i], g->d_rgb[(size_t)i], npx * 3, ncclFloat32, ncclSum, 0, (ncclComm_t)c->comm, c->stream);
            if (r != ncclSuccess) bad_ctx = c;
        }
        // the group is closed even after a failed call (an open group would swallow every later call of this thread); a reduce that
        // only some ranks joined cannot complete, which the drain in fail_all would wait for: the communicator is unusable after such
        // an error and the message says so
        const ncclResult_t re = g_rccl.GroupEnd();
        if (r != ncclSuccess) {
            g->err = std::string("ncclReduce failed on device ") + std::to_string(bad_ctx->device) + ": " + g_rccl.GetErrorString(r) + " (destroy the group)";
            (void)pti::fail(bad_ctx, PT_E_HIP, "%s", g->err.c_str());
            return PT_E_HIP;
        }
        if (re != ncclSuccess) return fail_all(c0, pti::fail(c0, PT_E_HIP, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(re)));
    }
    if (hipSetDevice(c0->device) != hipSuccess) return fail_all(c0, pti::fail(c0, PT_E_HIP, "hipSetDevice(%d) failed", c0->device));
    if (out_rgba8 && n > 1 && pt_launch_pack_rgba8((const float*)g->d_rgb[0], (uint32_t*)g->d_rgba8[0], (long long)npx, c0->stream) != hipSuccess)
        return fail_all(c0, pti::fail(c0, PT_E_HIP, "RGBA8 pack launch failed"));
    if (hipMemcpyAsync(g->pinned_rgb, g->d_rgb[0], npx * 12, hipMemcpyDeviceToHost, c0->stream) != hipSuccess ||
        (out_rgba8 && hipMemcpyAsync(g->pinned_rgba8, g->d_rgba8[0], npx * 4, hipMemcpyDeviceToHost, c0->stream) != hipSuccess))
        return fail_all(c0, pti::fail(c0, PT_E_HIP, "framebuffer read-back failed"));
    {
        int first_rc = PT_OK;
        pt_ctx* first_bad = nullptr;
        for (int i = 0; i < n; ++i) { // drains EVERY device's stream and reads its watchdog flag
            const int rc = pt_synchronize(g->ctx[(size_t)i]);
            if (rc && !first_rc) { first_rc = rc; first_bad = g->ctx[(size_t)i]; }
        }
        if (first_rc) return bad(first_bad, first_rc);
    }
    std::memcpy(out_rgb, g->pinned_rgb, npx * 12);
    if (out_rgba8) std::memcpy(out_rgba8, g->pinned_rgba8, npx * 4);
    return PT_OK;
}

} // extern "C"
