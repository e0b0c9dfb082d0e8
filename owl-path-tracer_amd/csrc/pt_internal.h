// pt_internal.h -- private to the library: the context behind the opaque pt_ctx of include/mi355pt.h, shared by pt_api.cpp
// (scene, render) and pt_comm.cpp (RCCL reduce, multi-GPU group).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mi355pt.h"
#include "pt_bvh.h"
#include "pt_types.h"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct HostTexture {
    int w = 0, h = 0;
    std::vector<uint32_t> px;
};

struct pt_ctx {
    int device = -1;
    bool host_only = false;
    int num_cus = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evm = nullptr, evr = nullptr, evd = nullptr; // evm: after the cost pre-pass and the queue sort; evr / evd: after the reduce / the read-back of pt_render
    bool ev_pending = false;
    bool flag_pending = false, watchdog_fired = false; // the watchdog flag of the last render has not been looked at yet / was set
    hipStream_t last_stream = nullptr;                  // stream of the last pt_render_device (may be the caller's)
    std::string err;

    // host copies
    PtBvh bvh;
    std::vector<PtNode4> nodes4; // two-level collapse of bvh.nodes for the wavefront kernel (pt_bvh_collapse4)
    int32_t root4 = -1;
    int depth4 = 0;
    std::vector<PtNode8> nodes8; // three-level collapse for the group walk of sparse waves (pt_bvh_collapse8)
    int32_t root8 = -1;
    int depth8 = 0;
    std::vector<PtShade> shade;
    std::vector<float> materials; // n * PT_MAT_STRIDE
    int n_materials = 0;
    std::vector<int32_t> material_texture;
    uint32_t lobe_codes[PT_LOBE_TABLE] = {}; // pt_lobe_code per material (index = material + 1), upload_materials
    uint32_t lobe_mask = 0;                 // lobes the scene's materials can sample: bit 0 diffuse, 1 clearcoat, 2 metallic, 3 glass (0: too many materials for the table)
    bool uses_default_material = false;     // some mesh has material_index < 0
    bool tri_packed = false;                // the device triangle records carry id << 8 | (material + 1) (upload_scene_to_device)
    std::vector<HostTexture> textures;
    pt_env env{};
    HostTexture env_map;
    bool have_scene = false;

    // device
    DevBuf d_nodes8, d_nodes4, d_nodes, d_tris, d_shade, d_materials, d_texdesc, d_env, d_pixels, d_heads, d_rng, d_accum, d_out, d_out8, d_counters, d_dbg_in, d_dbg_out, d_slots, d_laps, d_ring, d_params, d_cost, d_sorted, d_sort_scratch, d_dbg_start, d_bucket, d_tiers, d_lobe;
    std::vector<void*> d_textures;

    // pixel queue
    int q_w = 0, q_h = 0, q_rank = 0, q_world = 1, q_tile = 16;
    int rank = 0, world = 1, tile = 16;
    uint32_t n_pixels = 0;
    bool queue_valid = false;

    // options
    int spp_per_launch = 0, count = 0, blocks_per_cu = 0, leaf_size = 4, max_bvh_depth = 48, kernel = 2, slots_per_wave = 0, chunk_spp = 64, chunk_tail_min = -1, schedule = 1, prepass_spp = 0, census_mode = 0, sticky_pct = -1, latency = 0, cost_radius = 2, timeline = 0, node_pairs = 0, leaf_align = 1, bvh_builder = 3, quad = 1, groups = 1, wide_leaves = 1, fallback = 0, ploc_radius = 16, express_permille = -1, ns_express = 8, whole = -1, lobe_bins = 0, box_exact = -1;
    int tune[8] = {};

    void* comm = nullptr;   // ncclComm_t once pt_comm_init_rank / pt_group_create attached one (pt_comm.cpp)
    int comm_rank = 0, comm_world = 1;

    pt_stats stats{};
    int last_launches = 0;
    bool last_sorted = false;
    int last_w = 0, last_h = 0;
    size_t lap_ticks_ofs = 0;
    int last_chunks = 0;
};

namespace pti {
int fail(pt_ctx* c, int code, const char* fmt, ...);
int ensure(pt_ctx* c, DevBuf& b, size_t bytes);
int check_watchdog(pt_ctx* c);
int upload_scene_to_device(pt_ctx* c);
int clone_scene(pt_ctx* dst, const pt_ctx* src);
} // namespace pti

#define HIP_TRY(c, call)                                                                                   \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        if (e__ != hipSuccess) return pti::fail(c, PT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)
