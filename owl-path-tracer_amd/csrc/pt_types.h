// pt_types.h -- HBM data layout shared by the host builder and the HIP kernel (DESIGN.md "data layout").
#pragma once
#include <stdint.h>

// One BVH2 node = both children's boxes + both child references: 64 bytes, 64-byte aligned,
// fetched by one lane as 4 x global_load_dwordx4 (one half cache line).  The planes of the LEFT and RIGHT child are
// interleaved pairwise -- lo[axis] = {left.min, right.min}, hi[axis] = {left.max, right.max} -- so that the slab test of both
// children runs on packed f32 math (v_pk_add_f32 / v_pk_mul_f32: 12 instructions instead of 24).
// child >= 0: index of an internal node.  child < 0: leaf, ~child = (first_slot << 3) | count (count 1..7).
struct PtNode {
    float lo[3][2]; // lo[axis][0] = left child's min, lo[axis][1] = right child's min
    float hi[3][2]; // hi[axis][0] = left child's max, hi[axis][1] = right child's max
    int32_t left, right;
    uint32_t pad[2];
};
static_assert(sizeof(PtNode) == 64, "PtNode must be 64 bytes");

// Two binary levels per record ("quad node", wavefront kernel): the boxes of the four GRANDchildren of binary node i, so that one
// 128-byte record - one cache line, one memory round trip - takes a ray two levels down the binary tree.  Slot 2 * side + s holds
// child s of the left (side 0) / right (side 1) child of node i; where a child of node i is itself a leaf it occupies slot
// 2 * side with its own box and slot 2 * side + 1 is empty.  Built by pt_bvh_collapse4 from the PtNode[] tree (same boxes, same
// leaves, same topology - the intermediate level's boxes are simply not stored); planes are laid out per axis for the packed-f32
// slab test: lo[axis] = {slot 0, 1, 2, 3}.
// child >= 0: quad node index; < -1: leaf code as in PtNode; -1: empty slot (its box is {+inf, +inf}: never hit).
struct PtNode4 {
    float lo[3][4];
    float hi[3][4];
    int32_t child[4];
    uint32_t pad[4];
};
static_assert(sizeof(PtNode4) == 128, "PtNode4 must be 128 bytes");

// Three binary levels per record ("oct node", round 3): up to eight descendants of a binary node, each with its own box - the
// record of the group walk (pt_kernel.hip, traverse_groups), where eight lanes test the eight children of ONE ray's node at once:
// lane k of a group reads child k (2 x global_load_dwordx4; the eight lanes cover the record's two cache lines), so a ray goes down
// the tree in a third of the dependent steps of the binary walk.  Built by pt_bvh_collapse8 from the PtNode[] tree (same boxes and
// leaves; the slot with the largest box is opened until eight are used).  ref >= 0: oct node index; < -1: leaf code as in PtNode;
// -1: empty slot (box {+inf, +inf}: never hit).
struct PtNode8 {
    struct Child {
        float lo[3], hi[3];
        int32_t ref;
        uint32_t pad;
    } c[8];
};
static_assert(sizeof(PtNode8) == 256, "PtNode8 must be 256 bytes");

// One triangle in LEAF order: the three vertices (the reference's vertex_buffer values, fetched through
// index_buffer: device.cu:42-61) + its global id (entity order, then face order).  48 bytes = 3 x dwordx4.
struct PtTri {
    float p0[3], p1[3], p2[3];
    int32_t id;       // global triangle id (the closest hit's tie-break).  On the device, scenes below 2^23 triangle slots carry
                      // id << 8 | min(material + 1, 255) here (same order: ids are unique) - the hit's material then reaches the
                      // lobe bins of the hit pass without a fetch (pt_api.cpp, upload_scene_to_device)
    int32_t material; // copy of the shading record's material index: the material fetch need not wait for that record
    uint32_t pad;
};
static_assert(sizeof(PtTri) == 48, "PtTri must be 48 bytes");

// Shading record per triangle, in LEAF order like PtTri (one index serves both): the three vertex normals (device.cu:63-73), the material index
// (entity_data.material_index) and the three texcoords (device.cu:75-94).  64 bytes = 4 x dwordx4.
// Flattened per triangle instead of the reference's buffer-of-buffers double indirection.
struct PtShade {
    float n0[3], n1[3], n2[3];
    int32_t material;
    float tc[6];
};
static_assert(sizeof(PtShade) == 64, "PtShade must be 64 bytes");

#define PT_MAT_STRIDE 20 // material_data (17 floats) + texture slot + 2 pad: 80 bytes, 16-byte aligned rows
#define PT_MAX_STACK 64
// Tier table of the whole-pixel schedule (uint32 words): [0] number of tiers, then PT_TIER_WORDS per tier:
// first queue entry, entries, pixels per wave, first workgroup, workgroups, cost class (diagnostics), 2 unused.  The ticket counter of tier t is
// queue_head + PT_TIER_COUNTER(t) (a cache line of its own).
#define PT_MAX_TIERS 32
#define PT_TIER_WORDS 8
#define PT_TIER_COUNTER(t) (128 + 32 * (t))
#define PT_HEADS_WORDS (128 + 32 * PT_MAX_TIERS) // per launch: ticket counter, +64 express counter, then the tier counters
#define PT_GROUP_STACK 96 // entries of a group's stack in the group walk: eight columns of the 12-level LDS stack area

struct PtTexDesc {
    const uint32_t* texels;
    int32_t width, height;
};

struct PtCounters {
    unsigned long long samples, rays, nodes, tris, scatters, env_misses, nan_retries;
    // scheduler census of the wavefront kernel (wave-level events and the lanes that took part in them)
    unsigned long long sched[32];
    // group walk (sparse waves): {phases, iterations, sum of busy groups, sum of node groups, sum of leaf groups, rays traced, shader-clock cycles, unused}
    unsigned long long grp[8];
    // hit passes by sampled lobe (disney.cuh:9-13 order: 0 diffuse, 1 clearcoat, 2 metallic, 3 glass; 4 = emitter hit, 5 = NaN retry):
    // [0..5] items, [7] hit passes over one lobe bin alone, [8..13] hit passes in which at least one item took that branch (executions of that body), [14] passes with two or
    // more BSDF bodies, [15] passes whose items all took one branch
    unsigned long long lobes[16];
    // traversal census (lane-steps): [0] quad-node steps that enter no child, [1] of those: the ray does cross a child's box, but beyond the
    // best hit so far (an entry distance kept with the stack entry would have skipped the fetch), [2] unused, [3] leaf steps that do not
    // improve the hit
    unsigned long long trav[4];
};

#define PT_MAX_TAIL_CHUNKS 20
// Diagnostics block of one launch (u64 units, 256-byte aligned): the chunk timeline (3 x (n_chunks + 1) values, plain stores), then -
// on cache lines of their own, because they are updated with device-scope atomics - 64 first-chunk latency accumulators.
#define PT_LAP_DIAG_OFS(nc) (((3 * ((nc) + 1)) + 31) & ~31)
#define PT_LAP_REGION(nc) (PT_LAP_DIAG_OFS(nc) + 64)

struct PtKernelParams {
    const PtNode* nodes;
    const PtTri* tris;
    const PtShade* shade;
    const float* materials; // n_materials * PT_MAT_STRIDE
    const PtTexDesc* textures;
    const uint32_t* pixel_ids; // work queue: launch-index pixel ids owned by this context
    uint32_t* queue_head;      // ticket counter, one per launch
    uint32_t* rng_state;       // per pixel (launch-index order), carried between spp chunks
    float* accum;              // per pixel * 3, carried between spp chunks
    float* out_rgb;            // W*H*3, framebuffer order
    uint32_t* out_rgba8;       // optional
    PtCounters* counters;      // optional (instrumented build)
    uint32_t* slot_state;      // wavefront kernel: per-wave path-slot state + park area (pt_wave_state_words each)
    uint32_t* ring;            // wavefront kernel: ring[c * n_pixels + i] = 1 + id of the pixel whose chunk c may start (0: not yet), in completion order of chunk c - 1
    uint32_t* ring_tail;       // ring_tail[c] = entries published to ring c so far
    unsigned long long* lap_ticks; // [0] = s_memrealtime (100 MHz) at kernel entry, [c + 1] = when the last pixel finished chunk c (diagnostics)
    uint32_t* error_flag;      // set to 1 by a wave whose scheduler watchdog fired
    uint32_t* dbg_start;       // diagnostics (option "latency"): per pixel, clock at the start of its first chunk; null otherwise
    uint8_t* dbg_cost;         // diagnostics: per pixel cost of the pre-pass
    uint8_t* cost_out;         // cost pre-pass only (n_chunks == 1): cost image, rays traced per pixel id (saturating at 255)
    PtTexDesc env_map;
    float cam[12];
    float env_color[3];
    float env_intensity;
    int32_t env_use_map, env_use_auto;
    int32_t root;              // root child reference (leaf if the scene is tiny)
    int32_t n_tris;
    int32_t n_materials;
    uint32_t n_pixels;         // queue length
    int32_t width, height;
    int32_t max_samples;       // total spp of the frame (normalisation)
    int32_t sample_begin, sample_count; // this launch covers [sample_begin, sample_begin + sample_count)
    int32_t max_depth;
    int32_t stack_entries;
    int32_t lds_levels;        // wavefront kernel: stack levels kept in LDS (pt_wave_lds_stack)
    int32_t ns;                // wavefront kernel: path slots per wave (64..255)
    int32_t chunk_spp, n_chunks; // wavefront kernel: samples per (pixel, chunk) ticket and chunks per pixel
    uint32_t n_tickets;        // (n_pixels - n_express) * n_chunks
    uint32_t n_express;        // the first n_express entries of the (cost-ordered) queue are express pixels (pt_kernel.hip, take_ticket); 0: none
    int32_t express_waves;     // workgroups [0, express_waves) render express pixels only,
    int32_t ns_express;        // ... this many at a time
    const uint32_t* tiers;     // != null: whole-pixel schedule by cost class (pt_kernel.hip, TIERS): the table pt_plan_tiers_kernel wrote for this launch
    int32_t ring_grid;         // ... and if that table is empty (the plan chose the ring schedule): workgroups beyond this one have nothing to do
    int32_t timeline;          // diagnostics: record the chunk timeline (pt_debug_read_laps); costs one more atomic per finished pixel
    int32_t census_mode;       // instrumented build: 1 = the scheduler census covers only a wave's wind-down (after its first failed ticket)
    int32_t n_full;            // chunks [0, n_full) have chunk_spp samples; the rest follow tail_len[] (shrinking chunks: short frame tail)
    int32_t tail_len[PT_MAX_TAIL_CHUNKS];
    const PtNode4* nodes4;     // wavefront kernel: quad nodes (null: walk PtNode[] one level per step); root / stack_entries then refer to them
    const PtNode8* nodes8;     // wavefront kernel: oct nodes of the group walk (null: no group walk)
    int32_t root8;             // root reference into nodes8 (leaf code if the scene is tiny)
    int32_t groups;            // group walk: 0 = never, 1 = when a wave has few rays to trace (sparse wave), 2 = always (tests)
    int32_t tune[8];           // scheduler knobs (pt_set_option "tune0".."tune7"; 0 = built-in default), see pt_kernel.hip
    int32_t box_exact;         // wavefront kernel: 1 = slab distances as (plane - o) * (1 / d) instead of the fma form (camera farther than 40 scene extents from the origin)
    const uint32_t* lobe_codes; // PT_LOBE_TABLE words: lobe thresholds per material (pt_lobe_code), index = material + 1
    int32_t lobe_bins;         // wavefront kernel: 1 = hit passes shade one predicted lobe at a time (pt_kernel.hip, LOBE-COHERENT HIT PASSES)
    uint32_t hit_slot_mask;    // 0x00ffffff when PtTri::id is packed as id << 8 | (material + 1) (then a hit's material rides in the top byte of its
                               // triangle slot), else 0xffffffff
};
#define PT_LOBE_TABLE 32       // materials + 1 the lobe bins can tell apart (scenes with more: no bins)
