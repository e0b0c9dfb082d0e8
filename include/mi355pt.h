/*
 * mi355pt.h -- C-ABI of the MI355X-native path-tracing render loop (libmi355pt.so).
 *
 * Drop-in boundary for the hot path of jctemp/owl-path-tracer: everything between
 * `owlLaunch2D(ray_gen, W, H, lp)` (path_tracer/src/application.cpp:366) and the framebuffer being
 * readable, i.e. ray_gen / trace_path / triangle_hit / miss (path_tracer/src/device/device.cu:113-293),
 * the OptiX traversal they call, and the host->device contract that application.cpp fills in
 * (launch_params_data, ray_gen_data, entity_data: path_tracer/src/device/device_global.hpp:38-74).
 * The reference has no FFI of its own; each entry point below names the reference call(s) it replaces.
 *
 * Plain C, POD structs, plain pointers and sizes; no C++/torch types cross this boundary.  Every call
 * returns 0 on success or a negative PT_E_* code (no exception crosses the ABI; message via
 * pt_last_error).  A context is bound to one GPU and is not thread-safe (same as the reference: single
 * host thread, application.cpp).  Caller owns every input array (copied during the call) and every
 * output array; the library owns device memory, the BVH and its stream.
 */
#ifndef MI355PT_H
#define MI355PT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 5
#define PT_MAT_FLOATS 17 /* material_data: device_global.hpp:19-36, 68 bytes, field order kept */

enum {
    PT_OK = 0,
    PT_E_INVALID = -1,   /* bad argument (the reference would throw std::runtime_error / trap) */
    PT_E_NO_DEVICE = -2, /* no usable gfx950 device / HIP runtime failure at create */
    PT_E_HIP = -3,       /* a HIP call failed (message has the HIP error string) */
    PT_E_NO_SCENE = -4,  /* render before upload ("no geometries", application.cpp:133) */
    PT_E_LIMIT = -5      /* scene exceeds an internal limit (BVH depth / index range) */
};

typedef struct pt_ctx pt_ctx;

typedef struct pt_config {
    int32_t device;   /* HIP device ordinal (reference: create_context(nullptr, 1), application.cpp:62) */
    int32_t reserved; /* must be 0 */
} pt_config;

/* One entity = one OBJ object that matched a material (application.cpp:166-179, :186-247).
 * Arrays as produced by create_mesh (utils/mesh_loader.cpp:9-83). */
typedef struct pt_mesh {
    const float* vertices;   /* n_vertices * 3   (vertex_buffer, application.cpp:197) */
    const float* normals;    /* n_normals * 3    (normal_buffer, :198) -- must cover every vertex index */
    const float* texcoords;  /* n_texcoords * 2  (texcoords_buffer, :200) or NULL */
    const int32_t* indices;  /* n_triangles * 3  (index_buffer, :199) */
    int32_t n_vertices, n_normals, n_texcoords, n_triangles;
    int32_t material_index;  /* entity_data.material_index (:212); <0 => material_data{} defaults (device.cu:150-154) */
    int32_t texture_index;   /* index into textures[] or <0 (entity_data.has_texture/texture, :236-243) */
} pt_mesh;

/* RGBA8 image, row 0 = v = 0, i.e. AFTER the vertical flip the reference applies at load
 * (application.cpp:229-234, utils/image_buffer.cpp:50-55); sampled nearest / clamp / normalised
 * coordinates (owl.hpp:248-257). */
typedef struct pt_texture {
    int32_t width, height;
    const uint32_t* rgba8;
} pt_texture;

/* launch_params_data environment fields (device_global.hpp:59-63, application.cpp:285-289) */
typedef struct pt_env {
    int32_t use_map;    /* environment_use  (only effective with a non-empty map, device.cu:138) */
    int32_t use_auto;   /* environment_auto */
    float color[3];     /* environment_color */
    float intensity;    /* environment_intensity */
    pt_texture map;     /* environment_map; width == 0 => none */
} pt_env;

/* camera_data (camera.hpp:14-20), 48 bytes, produced by to_camera_data (camera.cpp:3-21) */
typedef struct pt_camera {
    float origin[3], llc[3], horizontal[3], vertical[3];
} pt_camera;

/* Work counters of the last counted render (pt_set_option "count" = 1). One sample = one camera path. */
typedef struct pt_stats {
    double kernel_ms;       /* first launch to last kernel end of the last pt_render*, HIP events on the launch stream */
    int32_t launches;       /* render-kernel launches in the last pt_render* */
    int32_t vgprs, sgprs, lds_bytes, block, grid, stack_entries; /* launch geometry of the render kernel */
    uint64_t samples, rays, nodes, tris, scatters, env_misses, nan_retries; /* valid when counted */
    uint64_t bvh_nodes, bvh_depth, n_triangles;
    double bvh_build_ms;
    /* counted renders, wavefront kernel: {node steps, lanes in them, triangle steps, lanes, retire passes, lanes retired,
     * hit-shading passes, items, miss-shading passes, items, traversal phases, parked lanes,
     * scheduler iterations, sum of idle lanes, sum of finished lanes awaiting retirement, sum of node+leaf lanes} */
    uint64_t sched[32];
    double prepass_ms;      /* part of kernel_ms spent in the cost pre-pass launch + queue sort (0 when the schedule has none) */
    /* counted renders, group walk of sparse waves (eight lanes per ray): {phases, iterations, sum of busy groups, of groups at a
     * node, of groups at a leaf, rays traced, shader-clock cycles, 0} */
    uint64_t groups[8];
    /* last pt_render (not pt_render_device): time on the stream between the end of the render kernels and the end of the RCCL reduce
     * (0 without a communicator; includes waiting for the slowest rank), and of the read-back into the caller's buffers (root) */
    double reduce_ms, d2h_ms;
    int32_t kernel_variant; /* 1 lane per pixel, 2 wavefront kernel, 3 its fallback instance with the larger register budget (chosen when
                             * the 128-VGPR instance of this build would need scratch, or by option "fallback") */
    int32_t express_pixels; /* pixels of the last cost-ordered launch that were rendered as express pixels (waves of their own) */
    int32_t whole_pixels;   /* other pixels of that launch that kept their path slot for all samples (whole-pixel schedule: every pixel had a slot from the start); 0: ring schedule */
    int32_t prepass_spp;    /* samples per pixel of the cost pre-pass launch of the last render (0: the render did not sort) */
    /* counted renders, hit-shading passes by sampled lobe (disney.cuh:9-13: 0 diffuse, 1 clearcoat, 2 metallic, 3 glass; 4 = emitter hit,
     * 5 = NaN retry): [0..5] items, [7] passes that shaded one lobe bin alone (option "lobe_bins"), [8..13] passes in which at least one
     * item took that branch, [14] passes that ran two or more BSDF bodies, [15] passes whose items all took the same branch */
    uint64_t lobes[16];
    /* counted renders, traversal (lane-steps): [0] quad-node steps that enter no child, [1] of those: the node lies beyond the best hit
     * found so far, [2] unused, [3] leaf steps that do not improve the hit */
    uint64_t trav[4];
} pt_stats;

/* ---- lifecycle (replaces init_owl_data/destroy_context: application.cpp:59-128, Main.cpp:30) ---- */
pt_ctx* pt_create(const pt_config* cfg);          /* NULL on failure; pt_last_error(NULL) has the reason */
void pt_destroy(pt_ctx* ctx);
const char* pt_last_error(const pt_ctx* ctx);     /* ctx may be NULL (creation errors) */
int pt_abi_version(void);

/* ---- scene (replaces bind_sbt_data + init_owl_world: application.cpp:184-294, :131-140) ----
 * Copies the meshes, builds the BVH2 on the host and uploads everything to HBM.
 * Closest hit = the minimum over all triangles of the Moeller-Trumbore t (ties: lower global triangle index), independent of the hierarchy;
 * sliver triangles - height below 1e-5 of the longest edge, i.e. below ~100 ulp of their coordinates - are never hit (their test result
 * would be rounding noise).  OptiX's watertight test gives such triangles a vanishing cross-section as well. */
int pt_upload_scene(pt_ctx* ctx, const pt_mesh* meshes, int32_t n_meshes, const float* materials, int32_t n_materials,
                    const pt_texture* textures, int32_t n_textures, const int32_t* material_texture, const pt_env* env);
/* material_texture: optional n_materials ints (texture index per material or <0).  NULL => derived from
 * pt_mesh.texture_index of the entities using the material (the reference ties the texture to the
 * material's filename, parser.cpp:32-35 / application.cpp:214-243). */

/* replaces reset_field (application.cpp:297-304): re-upload the material table, BVH untouched */
int pt_set_materials(pt_ctx* ctx, const float* materials, int32_t n_materials);
int pt_set_environment(pt_ctx* ctx, const pt_env* env);

/* ---- pixel ownership (multi-GPU sharding; no reference counterpart: the reference is single-GPU) ----
 * Default: this context renders every pixel.  Launch-index pixel id = x + W*y (ray_gen's pixelId). */
int pt_set_pixel_shard(pt_ctx* ctx, int32_t rank, int32_t world_size, int32_t tile);
/* Host-only helper (no GPU needed): ids owned by `rank` when tile x tile pixel tiles are dealt
 * round-robin in row-major tile order.  Returns the count (writes at most cap ids); <0 on error. */
int64_t pt_shard_pixels(int32_t width, int32_t height, int32_t tile, int32_t rank, int32_t world_size, uint32_t* ids, int64_t cap);

/* ---- render (replaces owlLaunch2D + framebuffer read-back: application.cpp:363-369) ----
 * Blocking.  out_rgb: W*H*3 floats, linear, averaged over max_samples, stored at
 * x + W*(H-1-y) like the reference framebuffer (device.cu:251); pixels not owned by this context
 * are 0.  out_rgba8 (optional): owl::make_rgba of the same values (device.cu:252-253). */
int pt_render(pt_ctx* ctx, const pt_camera* cam, int32_t width, int32_t height, int32_t max_samples, int32_t max_path_depth,
              float* out_rgb, uint32_t* out_rgba8);
/* Same render, asynchronous on `stream` (a hipStream_t; NULL = the context's own stream), result left
 * in HBM at d_out_rgb (device pointer, W*H*3 floats) for the caller's RCCL reduce.  d_out_rgba8 optional. */
int pt_render_device(pt_ctx* ctx, const pt_camera* cam, int32_t width, int32_t height, int32_t max_samples, int32_t max_path_depth,
                     void* d_out_rgb, void* d_out_rgba8, void* stream);
/* Waits for the last pt_render_device (its stream and the context's) and returns PT_E_HIP if a wave's watchdog fired during
 * it (the image is then incomplete); pt_get_stats reports the same. */
int pt_synchronize(pt_ctx* ctx);

/* ---- N GPUs: pixel tiles sharded over ranks + ONE RCCL sum-reduce of the float3 framebuffer onto rank 0 (pt_comm.cpp) ----
 * No reference counterpart (the reference is single-GPU: create_context(nullptr, 1), application.cpp:62); for N > 1 these
 * replace the render + read-back of application.cpp:363-369.  Every pixel has exactly one non-zero contributor, so the
 * N-GPU frame is bit-identical to the 1-GPU frame.  The library owns the communicator (librccl.so.1, resolved on first use).
 *
 * (a) One process per GPU.  Rank 0 calls pt_comm_get_unique_id; the 128 bytes reach the other ranks by the launcher's
 *     means; every rank calls pt_comm_init_rank (collective; also sets the rank's pixel shard, tile 16).  From then on
 *     pt_render renders the shard, reduces, and fills out_rgb / out_rgba8 on rank 0 only (other ranks may pass NULL). */
#define PT_COMM_ID_BYTES 128
int pt_comm_get_unique_id(uint8_t id[PT_COMM_ID_BYTES]);
int pt_comm_init_rank(pt_ctx* ctx, const uint8_t id[PT_COMM_ID_BYTES], int32_t rank, int32_t world_size);
int pt_comm_destroy(pt_ctx* ctx);
/* The reduce by itself, asynchronous on `stream` (NULL = the context's stream), in place on the device buffer pt_render_device
 * filled (n_pixels*3 floats): exactly ONE collective on every rank, whatever else is passed.  d_rgba8 (optional, n_pixels uint32)
 * is an OUTPUT on rank 0 - owl::make_rgba of the reduced frame, bit for bit what the owning ranks would have stored (one non-zero
 * contributor per pixel) - and ignored on the other ranks.  A no-op without a communicator.
 * Environment: PT_RCCL_PATH = the RCCL library to load (default librccl.so.1 by the usual search). */
int pt_reduce_framebuffer(pt_ctx* ctx, void* d_rgb, void* d_rgba8, int64_t n_pixels, void* stream);
/* Pinned host memory for the frame, like the reference's framebuffer (owlBufferGetPointer, owl.hpp:108-111). */
void* pt_host_alloc(size_t bytes);
void pt_host_free(void* p);

/* (b) One process, N GPUs (`pt_main --gpus N`): N contexts + ncclCommInitAll; scene calls fan out to every device (full
 *     replica each), pt_group_render = shards + one reduce + read-back from device 0.  devices NULL = 0..n-1. */
typedef struct pt_group pt_group;
pt_group* pt_group_create(const int32_t* devices, int32_t n);   /* NULL on failure; pt_last_error(NULL) has the reason */
void pt_group_destroy(pt_group* g);
int32_t pt_group_size(const pt_group* g);
pt_ctx* pt_group_ctx(pt_group* g, int32_t i);
const char* pt_group_last_error(const pt_group* g);
int pt_group_upload_scene(pt_group* g, const pt_mesh* meshes, int32_t n_meshes, const float* materials, int32_t n_materials,
                          const pt_texture* textures, int32_t n_textures, const int32_t* material_texture, const pt_env* env);
int pt_group_set_materials(pt_group* g, const float* materials, int32_t n_materials);
int pt_group_set_option(pt_group* g, const char* key, int64_t value);
int pt_group_render(pt_group* g, const pt_camera* cam, int32_t width, int32_t height, int32_t max_samples, int32_t max_path_depth,
                    float* out_rgb, uint32_t* out_rgba8);

/* Tuning / test options (all have working defaults; none changes an image):
 *   "kernel" 2 (default, wavefront-scheduled) | 1 (lane per pixel);  "count" 0/1: instrumented kernel that fills pt_stats;
 *   "leaf_size", "max_bvh_depth": BVH builder, next pt_upload_scene;  "bvh_builder" 3 (default: by triangle count - 0 up to 64 M
 *   triangles, 2 beyond) | 0 (binned SAH on all host threads: the tree that walks fastest, 0.9 M triangles in 58-83 ms) | 1 (linear BVH built on
 *   the device, csrc/pt_lbvh.hip: 50 ms, 1.4x slower to walk) | 2 (PLOC on the device: 88 ms, 1.04-1.25x slower to walk; "ploc_radius" 16; a tree
 *   deeper than max_bvh_depth falls back to 0);  "blocks_per_cu", "slots_per_wave": launch geometry;
 *   "schedule" 1 (default: cost pre-pass + cost-ordered queue, from 4 x prepass_spp samples per pixel) | 0 (chunks only);
 *   "prepass_spp" (0 = automatic: 8, or 16 when a tier plan is prepared), "cost_radius" (2: the cost of a pixel - the time its pre-pass
 *   samples took - is de-noised by the mean over the look-alike pixels of its (2r+1)^2 neighbourhood), "whole" -1 (default: a launch
 *   whose pixels can all have a path slot from the start hands out whole pixels by cost class if a plan made on the device says so)
 *   | 0 (never: ring schedule) | 1 (always), "sticky_pct" (automatic: min(80, 50 + 6 x pixels per path slot) %: share of the
 *   remaining samples a pixel gets in its first chunk), "chunk_spp" (64, schedule 0), "chunk_tail_min" (-1 = automatic: an eighth of the
 *   samples after the pre-pass, at least 16: smallest of the halving tail chunks; 0 = no tail), "spp_per_launch" (kernel 1: samples per launch; kernel 2: forces schedule 0 with this
 *   chunk size - the resumability tests use it);  "census_mode", "latency": diagnostics of the instrumented build;
 *   "groups" 1 (default: a wave with few rays to trace walks them eight lanes per ray over oct nodes) | 0 (never) | 2 (always: tests),
 *   "wide_leaves" 1 (oct nodes: subtrees of <= 7 triangles are one leaf step; next pt_upload_scene), "tune6" / "tune7" (16 / 24: ray-queue
 *   level and running pixels up to which a wave counts as sparse);  "tune0" (8: a shading pass with idle lanes also takes the entries of the other
 *   queue when that holds at least this many; > 64 = never; off by itself when an environment map is bound);
 *   "quad" 1 (default: two binary levels per 128-byte record) | 0;  "box_exact" -1 (default: slab distances by one fma per plane, the
 *   subtracting form when the camera is more than 42 scene extents from the origin) | 0 | 1;  "lobe_bins" 1 | -1: lobe-coherent hit passes (a hit pass shades the hits of ONE predicted
 *   lobe at a time; -1: only when the materials can sample two or more lobes) - exists in `make lobebins` builds only (validated bit-exact, costs
 *   what it saves: profiles/r04_notes.md); the product build returns PT_E_INVALID; "tune4" (24: hits of one lobe that make a pass of their own);  "fallback" 1: use the wavefront kernel's 168-VGPR instance (what
 *   the library does by itself when the 128-VGPR instance of a build needs scratch). */
int pt_set_option(pt_ctx* ctx, const char* key, int64_t value);
int pt_get_stats(pt_ctx* ctx, pt_stats* out);

/* ---- host utilities ---- */
/* to_camera_data (camera.cpp:3-21) */
void pt_to_camera_data(const float look_from[3], const float look_at[3], const float look_up[3], float vertical_fov_deg,
                       int32_t width, int32_t height, pt_camera* out);

/* ---- validation hooks (used by tests/ only; never on the render path) ---- */
/* Closest hit through the PRODUCT BVH walked on the host: validates the host builder without a GPU. */
int pt_debug_closest_hit_host(pt_ctx* ctx, const float org[3], const float dir[3], float tmin, float tmax,
                              float* t, float* u, float* v, int32_t* prim);
/* Batched device-side evaluation of the kernel's building blocks on the GPU (op codes in pt_kernel.hip):
 * lets the parity tests compare them bit-for-bit with the oracle.  in/out are host arrays. */
int pt_debug_eval(pt_ctx* ctx, int32_t op, const float* in, int32_t in_stride, float* out, int32_t out_stride, int64_t n);

/* What pt_group_upload_scene does for devices 1..n-1: the scene `src` holds (BVH built once) copied into `dst` and uploaded to
 * dst's GPU.  Exposed so that a one-GPU box can test it with two contexts on the same device. */
int pt_debug_clone_scene(pt_ctx* dst, const pt_ctx* src);

/* Structure of the quad nodes the wavefront kernel walks (host side, no GPU needed): out = {quad nodes, depth, leaf slots,
 * triangles in leaf slots, empty slots, internal slots, binary nodes, binary leaf references}.  Every leaf of the binary tree
 * must appear in exactly one quad slot; an empty slot must carry the never-hit box. */
int pt_debug_quad_info(pt_ctx* ctx, int64_t out[8]);

/* The same for the oct nodes of the group walk (PtNode8): out = {oct nodes, depth, leaf slots, triangles in leaf slots, empty slots,
 * internal slots, largest leaf, triangle slots of the scene}.  Fails (PT_E_LIMIT) if a triangle slot is in no or in two leaves, a
 * triangle sticks out of its leaf's box, a node is referenced twice or an empty slot has a finite box. */
int pt_debug_oct_info(pt_ctx* ctx, int64_t out[8]);

/* The pixel queue of the last pt_render* call with the cost-ordered schedule: queue_ids[i] = pixel id (x + width * y) of
 * entry i of the cost-ordered queue, input_ids[i] / cost[i] = entry i of the shard's input queue and its cost class: 16 log2 of
 * the microseconds its first prepass_spp samples took (1..255).  Any pointer may be NULL.  Returns the number of entries (0: the last
 * render did not sort), or a negative error. */
int64_t pt_debug_read_queue(pt_ctx* ctx, uint32_t* queue_ids, uint32_t* input_ids, uint8_t* cost, int64_t cap);

/* Timeline of the last wavefront launch, three arrays of n_chunks + 1 values: ticks[0] = constant 100 MHz clock (s_memrealtime)
 * at kernel entry, ticks[1 + c] = when the last pixel finished chunk c; then [0] unused, [1 + c] = when the last work item of
 * chunk c started; then [0] unused, [1 + c] = queue entry that finished chunk c last.  Returns the number of values written. */
int64_t pt_debug_read_laps(pt_ctx* ctx, uint64_t* ticks, int64_t cap);

/* With option "latency" = 1: per pixel (index x + width * y), ticks of the 100 MHz clock from the entry of the last cost-ordered main
 * launch to the moment the pixel's last sample was stored (0 for pixels outside this rank's shard); then, if cap allows, a second
 * array of the same size: rays traced per pixel in that launch (option "count" = 1, else zeros).  Returns the number of values
 * written, 0 if the last render was not such a launch. */
int64_t pt_debug_read_finish(pt_ctx* ctx, uint32_t* ticks, int64_t cap);

/* Tier table of the last launch with the whole-pixel schedule (pt_stats.whole_pixels != 0): words[0] = number of tiers, then 8 words
 * per tier: first queue entry, entries, pixels per wave, first workgroup, workgroups, cost class, 2 unused.  Returns the number of
 * words written (at most 257), 0 if the last render used the ring schedule. */
int64_t pt_debug_read_tiers(pt_ctx* ctx, uint32_t* words, int64_t cap);

/* The tier plan for a launch whose cost-ordered queue holds bucket_pixels[b] pixels in cost bucket b (32 buckets, 0 = most expensive,
 * each 19 % cheaper than the one before), with `capacity` resident workgroups and `ns` path slots per wave: exactly what the device
 * computes after the counting sort, run on the host (no GPU needed).  force = 1: plan even when the cost distribution has no tail.
 * Same output as pt_debug_read_tiers (words[0] = 0: the plan declines and the launch runs the ring schedule); cap >= 257. */
int64_t pt_debug_plan_tiers(const uint32_t* bucket_pixels, int32_t capacity, int32_t ns, int32_t force, uint32_t* words, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* MI355PT_H */
