// pt_kernel.hip -- the render megakernel for gfx950 (MI355X): ray generation, BVH2 traversal with LDS-resident
// per-lane stacks, Moeller-Trumbore intersection, Disney BSDF sampling and the Russian-roulette bounce loop of the
// reference (path_tracer/src/device/device.cu:113-254).
//
// Two schedulers over the same device functions (DESIGN.md "kernel"):
//
//  pt_render_wave_kernel (default, "wavefront-scheduled"): one wave64 per workgroup owns ns path slots
//    (a slot = one pixel in flight with its RNG stream, accumulators and current ray) plus three slot queues:
//    rays waiting for traversal, hits and misses waiting for shading.  The wave alternates between
//      * a TRAVERSAL phase: every lane walks one ray (speculatively: one stashed leaf per lane); each step the wave
//        executes either a burst of BVH-node steps or one whole-leaf triangle step, whichever more lanes are waiting for
//        (ballot majority); finished lanes retire in batches into the hit / miss queue and pull the next ray.  A node step
//        reads the binary tree TWO levels at a time from 128-byte quad-node records (node4_step; option quad = 0: one level
//        per step from the 64-byte records, node_step); a leaf step requests all triangles of the leaf before the first test;
//      * a SHADING pass over up to 64 queued hits (emitter / BSDF sample / Russian roulette) or misses (environment,
//        sample accumulation, next camera ray / next work item), which emits the continuation rays into the ray queue.
//    The batch sizes follow the number of pixels the wave still runs ("adaptive"), so that a sparse wave does not wait for
//    its slowest ray before it shades anything.
//    Lanes are workers, not pixel owners: a ray's lane is unrelated to the lane that shades its hit.  The per-pixel
//    sample order -- hence the reference's per-pixel RNG stream (device.cu:226-243) -- is preserved because a slot
//    has at most one ray in flight.  Work items are (pixel, sample chunk) tickets; the host orders the pixel queue by a
//    cost pre-pass (pt_api.cpp, "schedule").
//
//  (The simple persistent lane-per-pixel form, option kernel=1, and the validation kernel live in pt_kernel_aux.hip; the queue sort,
//  the tier plan and the small helper kernels in pt_schedule.hip; the device code all of them share in pt_trace.h.)
//
// The traversal stack is stack[level][lane] in LDS: bank = lane % 32 whatever the level, so the stack itself is conflict-free (the LDS
// bank conflicts rocprof reports - a third of the LDS cycles, at ~6 % LDS utilisation - come from the slot-indexed state arrays,
// lray / lstate[field * ns + slot], where lanes hold arbitrary slots).
#include "pt_launch.h"
#include "pt_tiers.h"
#include "pt_trace.h"
#ifndef PT_WAVES_PER_EU
#define PT_WAVES_PER_EU 4
#endif
#ifndef PT_COUNT_WAVES_PER_EU
#define PT_COUNT_WAVES_PER_EU 2 // the instrumented instance gets 256 VGPRs (see pt_render_wave_kernel)
#endif
#ifndef PT_WATCHDOG_ROUNDS
#define PT_WATCHDOG_ROUNDS 20000000 // wave-loop rounds (shading passes + traversal phases) before a wave gives up; a C5 frame needs ~2e5 per wave
#endif
#ifndef PT_RAY_LOW
#define PT_RAY_LOW 32      // ray-queue level below which a partial shading batch is worth it
#endif
#ifndef PT_MIN_BATCH
#define PT_MIN_BATCH 32    // smallest shading batch taken early
#endif
#ifndef PT_NODE_KEEP
#define PT_NODE_KEEP 24  // > 0: keep doing bursts of node steps (up to PT_NODE_BURSTS) while at least this many lanes want one
#endif
#ifndef PT_NODE_BURSTS
#define PT_NODE_BURSTS 3
#endif
#ifndef PT_NODE_REPS
#define PT_NODE_REPS 2 // consecutive node steps per scheduler iteration in the common (LDS-stack-only) instance
#endif
#ifndef PT_QUAD_REPS
#define PT_QUAD_REPS 1 // quad-node steps per burst (a quad step is two binary levels and up to three pushes)
#endif
#ifndef PT_WITH_LOBE_BINS
#define PT_WITH_LOBE_BINS 0 // lobe-coherent hit passes (option "lobe_bins"): validated bit-exact, but they cost what they save, and their 2 KB of code cost
                            // the product instance 1 % even when switched off (profiles/r04_notes.md) - `make lobebins` builds the library with them
#endif
#ifndef PT_TOPUP_MIN
#define PT_TOPUP_MIN 8 // a shading pass with idle lanes also takes entries of the other queue when that holds at least this many (option "tune0"; > 64 = never):
                       // fewer, fuller passes - C4 497-500 -> 491-492 ms, C2 72.6-73.2 -> 70.8 (profiles/r04_notes.md); not with an environment map, whose
                       // miss shader (atan2, asin, a texel) is too long to run for a few topped-up lanes (C5: 3 194 -> 3 219)
#endif
#ifndef PT_PURE_MIN
#define PT_PURE_MIN 24     // lanes the fullest lobe bin must fill for a hit pass over that bin alone (option "tune4")
#endif
#ifndef PT_RETIRE_MIN
#define PT_RETIRE_MIN 16   // finished lanes that trigger a retire/refill pass (8..24 swept: +-1 %)
#endif


namespace {

// Quad-node step (PtNode4, pt_types.h): the four grandchild boxes of a binary node in one 128-byte record - one cache line and
// one memory round trip per TWO levels of the binary tree.  Slab test as in node_step, two packed pairs; the lane continues with
// the nearest hit child and pushes the other hit children (in slot order; any visiting order gives the same closest hit).
// The lane reads its own record with 7 x global_load_dwordx4 = 7 vL1D accesses per lane and step.  (Two validated experiments that did not
// pay - whole-line cooperative fetches through an LDS staging area, 64-byte records with 8-bit planes - live in variants/.)
template <int STRIDE, int LDS_ENTRIES, bool CENSUS = false>
__device__ __forceinline__ void node4_step(const PtNode4* __restrict__ nodes4, uint32_t* stack, uint32_t PT_AS1* ovf, v3 o, v3 inv, float tbest, int& cur, int& sp,
                                           const bool exact, uint32_t* n_nohit = nullptr, uint32_t* n_beyond = nullptr)
{
    bool crossed = false; // CENSUS: the ray crosses some slot's box when the bound of the best hit is ignored
    const size_t nb = (size_t)(uint32_t)cur * sizeof(PtNode4);
    const f32x4 lx = ldg4(nodes4, nb), ly = ldg4(nodes4, nb + 16), lz = ldg4(nodes4, nb + 32);
    const f32x4 hx = ldg4(nodes4, nb + 48), hy = ldg4(nodes4, nb + 64), hz = ldg4(nodes4, nb + 80), cf = ldg4(nodes4, nb + 96);
    // Slab distances (round 4): fma(plane, 1/d, -(o/d)) - one packed fma per pair of planes instead of a subtraction and a multiplication
    // (24 of the step's ~150 VALU operations; C4 492 -> 484-486 ms, C3 149 -> 146).  Against (plane - o) * (1/d) the distance is off by
    // |o/d| 2^-24, i.e. the plane seems displaced by |o| 2^-24 in space: rays start on surfaces or at the camera, the boxes are padded by
    // 1e-5 x the scene extent, and the host launches the instances with the subtracting form (`exact`) for a camera farther than 42
    // extents from the origin (pt_api.cpp) - the test stays conservative with respect to every hit the triangle test can report, and
    // the triangle test decides the image.  The reciprocals are finite (ray_inv clamps them: a direction component of exactly 0 would
    // otherwise give -inf or NaN here depending on the signs of plane and origin, and cull boxes the ray runs through).
    // (Both forms behind a run-time switch in ONE instance cost 4 %: C4 486 -> 508 ms, profiles/r04_notes.md.)
    const f32x2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z}, ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    const float nx = -(o.x * inv.x), ny = -(o.y * inv.y), nz = -(o.z * inv.z);
    const f32x2 nox = {nx, nx}, noy = {ny, ny}, noz = {nz, nz};
    const f32x2 pad = {1.0000004f, 1.0000004f};
    float tn[4];
    bool hit[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) { // slots {0, 1} and {2, 3}
        const f32x2 lox = p ? (f32x2){lx.z, lx.w} : (f32x2){lx.x, lx.y}, loy = p ? (f32x2){ly.z, ly.w} : (f32x2){ly.x, ly.y};
        const f32x2 loz = p ? (f32x2){lz.z, lz.w} : (f32x2){lz.x, lz.y}, hix = p ? (f32x2){hx.z, hx.w} : (f32x2){hx.x, hx.y};
        const f32x2 hiy = p ? (f32x2){hy.z, hy.w} : (f32x2){hy.x, hy.y}, hiz = p ? (f32x2){hz.z, hz.w} : (f32x2){hz.x, hz.y};
        f32x2 t0x, t1x, t0y, t1y, t0z, t1z;
        if (exact) { // a compile-time constant in the product instances (see pt_render_wave_kernel: EXACT)
            t0x = (lox - ox) * ix; t1x = (hix - ox) * ix;
            t0y = (loy - oy) * iy; t1y = (hiy - oy) * iy;
            t0z = (loz - oz) * iz; t1z = (hiz - oz) * iz;
        } else {
            t0x = __builtin_elementwise_fma(lox, ix, nox); t1x = __builtin_elementwise_fma(hix, ix, nox);
            t0y = __builtin_elementwise_fma(loy, iy, noy); t1y = __builtin_elementwise_fma(hiy, iy, noy);
            t0z = __builtin_elementwise_fma(loz, iz, noz); t1z = __builtin_elementwise_fma(hiz, iz, noz);
        }
        const float ta = fmax_hw(fmax_hw(fmin_hw(t0x.x, t1x.x), fmin_hw(t0y.x, t1y.x)), fmax_hw(fmin_hw(t0z.x, t1z.x), kTMin));
        const float tb = fmax_hw(fmax_hw(fmin_hw(t0x.y, t1x.y), fmin_hw(t0y.y, t1y.y)), fmax_hw(fmin_hw(t0z.y, t1z.y), kTMin));
        f32x2 tf = {fmin_hw(fmin_hw(fmax_hw(t0x.x, t1x.x), fmax_hw(t0y.x, t1y.x)), fmin_hw(fmax_hw(t0z.x, t1z.x), tbest)),
                    fmin_hw(fmin_hw(fmax_hw(t0x.y, t1x.y), fmax_hw(t0y.y, t1y.y)), fmin_hw(fmax_hw(t0z.y, t1z.y), tbest))};
        tf = tf * pad;
        tn[2 * p] = ta; tn[2 * p + 1] = tb;
        hit[2 * p] = ta <= tf.x; hit[2 * p + 1] = tb <= tf.y;
        if (CENSUS) { // instrumented instance: would the slot be hit without the bound of the best hit so far?
            const float fa = fmin_hw(fmin_hw(fmax_hw(t0x.x, t1x.x), fmax_hw(t0y.x, t1y.x)), fmax_hw(t0z.x, t1z.x)) * 1.0000004f;
            const float fb = fmin_hw(fmin_hw(fmax_hw(t0x.y, t1x.y), fmax_hw(t0y.y, t1y.y)), fmax_hw(t0z.y, t1z.y)) * 1.0000004f;
            crossed = crossed || ta <= fa || tb <= fb;
        }
    }
    const int r0 = __float_as_int(cf.x), r1 = __float_as_int(cf.y), r2 = __float_as_int(cf.z), r3 = __float_as_int(cf.w);
    const float far_key = kTMax * 2.0f; // beyond every valid entry distance
    const float k0 = hit[0] ? tn[0] : far_key, k1 = hit[1] ? tn[1] : far_key, k2 = hit[2] ? tn[2] : far_key, k3 = hit[3] ? tn[3] : far_key;
    const bool b01 = k1 < k0, b23 = k3 < k2;
    const float m01 = b01 ? k1 : k0, m23 = b23 ? k3 : k2;
    const int r01 = b01 ? r1 : r0, r23 = b23 ? r3 : r2;
    const bool in_b = m23 < m01; // the nearest hit is in slot pair {2, 3}
    const int nearc = in_b ? r23 : r01;
    const bool any = hit[0] || hit[1] || hit[2] || hit[3];
    if (CENSUS) { // steps that enter nothing, and of those the ones where the ray does cross a box: the node lies beyond the best hit
        *n_nohit += any ? 0u : 1u;
        *n_beyond += (!any && crossed) ? 1u : 0u;
    }
    // push order: the two slots of the OTHER pair first, the nearest's sibling last (popped first): siblings share a parent box, so
    // the sibling is usually the next nearest.  Three pushes at most.
    // (of the other pair, the farther slot first)
    const bool swap_o = in_b ? b01 : b23; // the other pair's second slot is the nearer one: push it second
    const int oa = in_b ? r0 : r2, ob = in_b ? r1 : r3;
    const bool ha = in_b ? hit[0] : hit[2], hb = in_b ? hit[1] : hit[3];
    const int x1 = swap_o ? oa : ob, x2 = swap_o ? ob : oa;
    const bool h1 = swap_o ? ha : hb, h2 = swap_o ? hb : ha;
    const bool first_of_pair = in_b ? !b23 : !b01; // the nearest is the first slot of its pair
    const int x3 = in_b ? (first_of_pair ? r3 : r2) : (first_of_pair ? r1 : r0);
    const bool h3 = in_b ? (first_of_pair ? hit[3] : hit[2]) : (first_of_pair ? hit[1] : hit[0]);
    // (strictly by entry distance - the partner inserted where its distance puts it - costs 12 more instructions per step and saves
    // 0.1 % of the triangle tests: C4 497 -> 515 ms, C5 3 211 -> 3 353, profiles/r04_notes.md)
    if (LDS_ENTRIES == 0x7fffffff) {
        // common instance (the caller made sure sp + 3 stays inside the LDS part): store above the top whether or not the entry is
        // pushed - the slot is free either way - and advance sp by the predicate; no branches
        stack[sp * STRIDE] = (uint32_t)x1; sp += h1 ? 1 : 0;
        stack[sp * STRIDE] = (uint32_t)x2; sp += h2 ? 1 : 0;
        stack[sp * STRIDE] = (uint32_t)x3; sp += h3 ? 1 : 0;
    } else {
        if (h1) { stack_push<STRIDE, LDS_ENTRIES>(stack, ovf, sp, (uint32_t)x1); ++sp; }
        if (h2) { stack_push<STRIDE, LDS_ENTRIES>(stack, ovf, sp, (uint32_t)x2); ++sp; }
        if (h3) { stack_push<STRIDE, LDS_ENTRIES>(stack, ovf, sp, (uint32_t)x3); ++sp; }
    }
    if (any) {
        cur = nearc;
    } else if (sp > 0) {
        --sp;
        cur = (int)stack_pop<STRIDE, LDS_ENTRIES>(stack, ovf, sp);
    } else {
        cur = PT_DONE;
    }
}

// ---- (pixel, spp-chunk) work items of the wavefront kernel ----------------------------------------------------------------
// Ring schedule: the frame is cut into n_chunks chunks per pixel (chunk_spp samples each, shrinking at the end).  A slot renders
// ONE chunk c of a pixel, publishes the pixel's (rng, accum) state and appends the pixel to ring c + 1; then it takes the next
// ticket.  Ticket t < n_pixels is chunk 0 of queue entry t (always ready); ticket t = c * n_pixels + i is the i-th pixel that
// finished chunk c - 1.  Tickets are handed out in order, so every pixel finishes chunk c before the bulk of chunk c + 1 starts
// and the end of the frame still has ~n_pixels independent work items.  A pixel whose chunk ran long lands at the END of the
// next ring, where the ticket counter already waits for it: laggards are never queued behind faster pixels (a single FIFO
// over all laps let expensive pixels fall ~20 ms further behind per lap; a slot keeping its pixel for all samples leaves
// 25-28 % of the wave time to a wind-down with ~8 busy lanes - profiles/r01_summary.md).
//
// Cross-wave hand-off.  This is the sc1 form of MI355X_MICROARCH.md "Valid forms" (first row of its table), not a C++ release /
// acquire pair, and it is valid only because every condition of that row holds here:
//   * the handed-off bytes (rng_state[pid], accum[3 pid ..]) are written ONLY by 4-byte agent-scope stores (global_store_dword sc1,
//     write-through to the fabric) and read ONLY by 4-byte agent-scope loads to registers (global_load_dword sc1: never served from
//     this CU's vL1D, never flat_): st_agent / ld_agent below; hipMalloc memory; one wave per workgroup;
//   * the storing wave executes s_waitcnt vmcnt(0) after those stores (the inline asm in finish_chunk; a workgroup is one wave,
//     so "every storing wave" is this wave), and only then stores the flag - the ring cell - again sc1;
//   * the consumer learns of it by an sc1 load poll of THAT cell, and the polling lane issues its loads of the bytes only after
//     its poll has matched (start_chunk returns before them otherwise: a control dependency in the same lane).
// The guide measured that row at ONE workgroup per CU; this kernel runs 16 one-wave workgroups per CU.  That it holds here as well is
// an empirical statement: tests/test_gpu_parity.py::test_c4_dragon_standin_full_size hands every sample of every pixel of C4 on (64
// one-sample chunks per pixel: 1.3e8 hand-offs in one launch) and compares the frame bit for bit with the one-chunk frame, and
// bench.py asserts the crc of every timed frame (~4 M hand-offs each).
// An agent-scope acquire per poll would also be correct but is 2-3x slower per hop and, with hundreds of pollers, costs the
// whole chip bandwidth (same section, "Invalid forms"); a release would write back the XCD's L2.  Counters that are updated with
// device-scope atomics (ticket heads, ring fills, the diagnostics accumulators) sit on cache lines that nothing stores to
// plainly (pt_api.cpp lays them out in 256-byte blocks; PT_LAP_DIAG_OFS) - a plain store into such a line made publications
// disappear in round 1.  No spinning: a slot whose cell is not published yet keeps its ticket and polls again in a later pass.
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(gp(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(gp(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t take_agent(uint32_t* p) { return __hip_atomic_fetch_add(gp(p), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#define PT_NO_TICKET 0xfffffffeu // take_ticket: the queue is exhausted
struct WaveTier {      // where this wave's pixels come from (wave-uniform)
    uint32_t* counter; // tier schedule: the ticket counter of the wave's tier, else null
    uint32_t q0, count; // ... and the tier's queue entries [q0, q0 + count)
    bool express;      // ring schedule: this wave renders express pixels only
};
// EXPRESS PIXELS (round 3).  A pixel is one sequential chain of rays (device.cu:226-243) and a frame whose pixels all run at once -
// one rank's shard of a multi-GPU frame, a small image - ends when its most expensive pixel does: chain length x per-ray turnaround,
// which in a wave that is busy with 96 pixels is 35-55 us.  The n_express most expensive pixels of the cost-ordered queue (its
// first entries) are therefore not mixed with the others: the first express_waves workgroups take ONLY them, ns_express at a time,
// each for all of its samples in one go (chunk PT_WHOLE: no hand-offs), at raised wave priority (s_setprio) and - having at most
// ns_express rays - with the group walk: a turnaround of 10-15 us.  The other waves share the remaining pixels as before and help
// with express pixels once their own tickets are gone.  The image cannot change: a pixel's stream does not depend on who renders it.
#define PT_EXPRESS 0xE0000000u  // ticket = PT_EXPRESS | index into the first n_express queue entries (the host keeps the bulk tickets below)
#define PT_WHOLE 255u           // chunk index of "every sample of this launch" (n_chunks <= 254 when express pixels exist)
// Next work item: tickets are handed out in order (hipcc aggregates the per-lane atomics of a pass into one per wave).
// TIERS (whole-pixel schedule).  When every pixel of the launch can have a path slot from the start (a shard, a small image) the ring
// schedule degenerates: tickets are handed out lap by lap, so no pixel starts chunk c + 1 before a slot holds the ticket of every pixel
// still in chunk c - 1, and the cheap pixels wait at every lap for the expensive ones (75 % of the rays of a 1/8 shard of C4 were
// traced by waves with idle slots, 842 M ring polls).  Such a frame is not bound by throughput but by its sample chains - rays of a
// pixel x turnaround of a ray in the wave that holds it - and the turnaround is a property of the wave: 25-28 us with 4-8 pixels (group
// walk), 35 / 43 with 12 / 16, ~50 with 32-48, 60-66 with 96 (measured on that shard, profiles/r03_logs/r3_ab37.log).  So the launch
// hands out pixels, not chunks - a slot keeps its pixel for all samples - and every wave serves ONE cost class of the cost-ordered
// queue with as few pixels as that class needs to finish with the others: pt_plan_tiers_kernel (below) turns the histogram of the
// counting sort into a tier table on the device, and a wave looks up its tier by workgroup index.
__device__ __forceinline__ uint32_t take_ticket(const PtKernelParams& P, const WaveTier& tier)
{
    if (tier.counter != nullptr) {
        const uint32_t e = take_agent(tier.counter);
        return e < tier.count ? PT_EXPRESS | (tier.q0 + e) : PT_NO_TICKET;
    }
    if (!tier.express) {
        const uint32_t t = take_agent(P.queue_head);
        if (t < P.n_tickets) return t;
    }
    if (P.n_express) {
        const uint32_t e = take_agent(P.queue_head + 64); // the express counter: 256 bytes after the queue head
        if (e < P.n_express) return PT_EXPRESS | e;
    }
    return PT_NO_TICKET;
}

// Try to start the work item of `ticket`.  Returns false if its ring cell is not published yet.
__device__ __forceinline__ bool start_chunk(const PtKernelParams& P, uint32_t ticket, uint32_t& c, int& px, int& py, uint32_t& rng, v3& color)
{
    uint32_t pid;
    const uint32_t n_bulk = P.n_pixels - P.n_express; // queue entries [n_express, n_pixels) go through the chunk tickets
    if ((ticket & 0xF0000000u) == PT_EXPRESS) {
        pid = gp(P.pixel_ids)[ticket & 0x0FFFFFFFu];
        c = PT_WHOLE;
    } else if (ticket < n_bulk) {
        pid = gp(P.pixel_ids)[P.n_express + ticket];
        c = 0;
    } else {
        const uint32_t e = ld_agent(P.ring + ticket); // ring c = ticket / n_bulk, entry ticket % n_bulk; cell = pixel id + 1
        if (e == 0u) return false;                    // chunk c - 1 of that pixel is still running somewhere
        pid = e - 1u;
        c = ticket / n_bulk;
    }
    if (P.timeline && c != PT_WHOLE && ticket % n_bulk == n_bulk - 1u) gp(P.lap_ticks)[(P.n_chunks + 1) + 1 + c] = wall_clock64(); // last start of chunk c
    px = (int)(pid % (uint32_t)P.width);
    py = (int)(pid / (uint32_t)P.width);
    if (c == 0 && P.dbg_start) gp(P.dbg_start)[pid] = (uint32_t)wall_clock64();
    if ((c == 0 || c == PT_WHOLE) && P.sample_begin == 0) {
        rng = rng_init((uint32_t)px, (uint32_t)py); // device.cu:226
        color = vs(0.0f);
    } else {
        rng = ld_agent(P.rng_state + pid);
        const uint32_t* a = reinterpret_cast<const uint32_t*>(P.accum) + 3 * (size_t)pid;
        color = V(__uint_as_float(ld_agent(a)), __uint_as_float(ld_agent(a + 1)), __uint_as_float(ld_agent(a + 2)));
    }
    return true;
}

__device__ __forceinline__ int chunk_len(const PtKernelParams& P, uint32_t c)
{
    if (c == PT_WHOLE) return P.sample_count;
    return (int)c < P.n_full ? P.chunk_spp : P.tail_len[(int)c - P.n_full];
}

// The slot finished chunk c of its pixel: write the framebuffer (device.cu:246-253) or hand the pixel on.
__device__ __forceinline__ void finish_chunk(const PtKernelParams& P, uint32_t c, int px, int py, uint32_t rng, v3 color)
{
    const bool last_chunk = (int)c + 1 >= P.n_chunks; // (PT_WHOLE included)
    const uint32_t n_bulk = P.n_pixels - P.n_express;
    const uint32_t pid = (uint32_t)px + (uint32_t)P.width * (uint32_t)py;
    if (c == 0 && P.dbg_start) { // diagnostics (tools/ab_bench.py latency=1): first-chunk duration by cost class of the pixel
        const uint32_t dt = (uint32_t)wall_clock64() - gp(P.dbg_start)[pid];
        const uint32_t cls = gp(P.dbg_cost)[pid] >> 2 < 31u ? gp(P.dbg_cost)[pid] >> 2 : 31u;
        atomicAdd(P.lap_ticks + PT_LAP_DIAG_OFS(P.n_chunks) + cls, (unsigned long long)dt);
        atomicAdd(P.lap_ticks + PT_LAP_DIAG_OFS(P.n_chunks) + 32 + cls, 1ull);
    }
    if (last_chunk && P.dbg_start) gp(P.dbg_start)[pid] = (uint32_t)(wall_clock64() - gp(P.lap_ticks)[0]); // diagnostics: when the pixel was done (pt_debug_read_finish)
    if (last_chunk && P.timeline && c != PT_WHOLE) { // ring_tail[n_chunks] only counts; the timeline is a diagnostic (pt_debug_read_laps)
        if (take_agent(P.ring_tail + P.n_chunks) == n_bulk - 1u) {
            gp(P.lap_ticks)[P.n_chunks] = wall_clock64();
            gp(P.lap_ticks)[2 * (P.n_chunks + 1) + P.n_chunks] = pid;
        }
    }
    if (last_chunk && P.sample_begin + P.sample_count >= P.max_samples) {
        v3 out = color * (1.0f / (float)P.max_samples);                          // device.cu:247
        size_t ofs = (size_t)px + (size_t)P.width * (size_t)(P.height - 1 - py); // device.cu:251
        float PT_AS1* orgb = gp(P.out_rgb);
        orgb[3 * ofs] = out.x;
        orgb[3 * ofs + 1] = out.y;
        orgb[3 * ofs + 2] = out.z;
        if (P.out_rgba8) gp(P.out_rgba8)[ofs] = make_rgba(out);
    } else {
        st_agent(P.rng_state + pid, rng);
        uint32_t* a = reinterpret_cast<uint32_t*>(P.accum) + 3 * (size_t)pid;
        st_agent(a, __float_as_uint(color.x));
        st_agent(a + 1, __float_as_uint(color.y));
        st_agent(a + 2, __float_as_uint(color.z));
        if (last_chunk) return; // the next launch resumes the pixel (cost pre-pass -> main pass)
        // the ring position is taken while the state stores are still in flight (one round trip for both); the entry itself
        // leaves only after the state has left this CU
        const uint32_t pos = take_agent(P.ring_tail + (c + 1)); // completion order of chunk c = start order of chunk c + 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_agent(P.ring + (size_t)(c + 1) * n_bulk + pos, pid + 1u);
        if (P.timeline && pos == n_bulk - 1u) {
            gp(P.lap_ticks)[c + 1] = wall_clock64();
            gp(P.lap_ticks)[2 * (P.n_chunks + 1) + c + 1] = pid;
        }
    }
}

} // namespace

// =====================================================================================================================
// v2: wavefront-scheduled megakernel, one wave64 per workgroup (default)
// =====================================================================================================================
//
// Per-wave storage (DESIGN.md "data layout"):
//   LDS    stack[level][lane]           top PT_LDS_STACK levels of the traversal stacks (deeper levels: HBM overflow columns)
//          lray[field][slot]            L_DIR* ray direction; L_A* ray origin while the slot waits for / is in traversal,
//                                       then the hit (u, v, triangle slot) once traversal has retired it
//          lstate[field][slot]          S_*: the slot's pixel / path state, touched once per shading pass
//          rayq / hitq / missq          ring buffers of slot ids (one byte each)
//   HBM/L2 park[field][lane]            traversal registers of lanes interrupted by a phase switch (7.8 MB chip-wide: L2 resident)
//          ovf[level][lane]             stack levels >= PT_LDS_STACK
// Round-1 history: with lstate in HBM (15 waves/CU) the slot state alone cost 1.3 TB of fabric traffic per C4 frame -- 60 % of
// all L2 misses, at 70 % of the chip's measured random-64-B-gather rate (profiles/r01_fetch_calibration.md); with a full-depth
// LDS stack and lstate in LDS only 8 waves fit a CU.  The short stack pays for keeping the state on chip.

#ifndef PT_LDS_STACK
#define PT_LDS_STACK 12
#endif
enum { L_DIRX = 0, L_DIRY, L_DIRZ, L_AX, L_AY, L_AZ, L_NFIELDS };
// S_RNG holds the ticket while the slot waits for its work item (S_PIX == PT_FRESH); S_QKC = chunk index (cost pre-pass: clock at the start of the pixel)
enum { S_PIX = 0, S_RNG, S_PACK, S_COLX, S_COLY, S_COLZ, S_THRX, S_THRY, S_THRZ, S_QKC, S_NFIELDS };
// S_PACK: bits 0-15 sample index within the chunk, 16-21 depth, 22-24 lobe+1, 25-31 consecutive NaN retries
#define PT_PACK(s, depth, lobe, retries) ((uint32_t)(s) | ((uint32_t)(depth) << 16) | ((uint32_t)((lobe) + 1) << 22) | ((uint32_t)(retries) << 25))
#define PT_FRESH 0xffffffffu // S_PIX marker: slot has no (pixel, chunk) running; S_RNG then holds the ticket it waits on, or PT_FRESH
enum { K_PSLOT = 0, K_CUR, K_SP, K_BT, K_BU, K_BV, K_BSLOT, K_BID, K_PEND, K_NFIELDS };

// LDS stack levels of a wave: the top PT_LDS_STACK levels of the per-lane stacks (fewer for a shallow tree), but at least what the
// group walk needs (group_entries entries in eight columns per group; <= PT_GROUP_STACK = 8 * PT_LDS_STACK)
static inline int pt_wave_lds_stack(int stack_entries, int group_entries)
{
    const int a = stack_entries < PT_LDS_STACK ? stack_entries : PT_LDS_STACK, b = (group_entries + 7) / 8;
    return a > b ? a : b;
}
// bins: the four lobe bins of the hit pass (ns bytes each) + the lobe-code table (LOBE-COHERENT HIT PASSES)
static inline size_t pt_wave_lds_bytes(int stack_entries, int group_entries, int ns, int bins)
{
    return ((size_t)pt_wave_lds_stack(stack_entries, group_entries) * PT_WAVE + (size_t)(L_NFIELDS + S_NFIELDS) * ns) * 4 +
           (((size_t)(bins ? 7 : 3) * ns + (bins ? PT_LOBE_TABLE * 4 : 0) + 15) & ~(size_t)15);
}
static_assert(PT_GROUP_STACK <= 8 * PT_LDS_STACK, "a group's stack is eight columns of the LDS stack area");
static inline size_t pt_wave_state_words(int stack_entries)
{
    int ovf = stack_entries > PT_LDS_STACK ? stack_entries - PT_LDS_STACK : 0;
    return (size_t)(K_NFIELDS + ovf) * PT_WAVE;
}

namespace {

struct WaveCtx {
    uint32_t* lray;   // LDS
    uint32_t* lstate; // LDS
    uint8_t *rayq, *hitq, *missq;
    uint8_t* binq;          // lobe bins of the hit pass: binq[b * ns + i], b = predicted lobe (LOBE-COHERENT HIT PASSES, below)
    const uint32_t* ltab;   // LDS copy of the per-material lobe codes (PT_LOBE_TABLE words)
    uint32_t bin_head, bin_count; // ring heads / fills of the four bins, one byte each (ns <= 255)
    int binned;             // items in the bins (hit_count counts the unclassified ones in hitq)
    int ns;
    int ray_head, ray_count, hit_head, hit_count, miss_head, miss_count, n_dead;
    bool miss_blocked; // the last miss pass only polled tickets whose predecessor chunk is still running
    int ray_low, min_batch, full_batch; // shading-batch thresholds (PT_RAY_LOW, PT_MIN_BATCH, 64; pt_set_option "tune1".."tune3")
    WaveTier tier;  // take_ticket
    int topup_min;  // a shading pass with idle lanes also takes entries of the other queue when that holds at least this many (0 or > 64: never)
    bool topup;     // ... at all
    int n_run;      // slots of this wave with a (pixel, chunk) running
    int adapt;      // 1: the thresholds follow n_run: a wave with few running pixels shades small batches instead of waiting for its slowest ray, and
                    // keeps stepping nodes while half of the lanes that started a burst still want to (1/8 shard of C4 391 -> 350 ms, 1/64 251 -> 214 ms)
    __device__ __forceinline__ void retune(int mb0, int rl0, int fb0)
    {
        if (!adapt) return;
        const int third = n_run / 3;
        const int lo = third < 4 ? 4 : third;
        min_batch = lo < mb0 ? lo : mb0;
        ray_low = lo < rl0 ? lo : rl0;
        const int two = 2 * third < 8 ? 8 : 2 * third;
        full_batch = two < fb0 ? two : fb0;
    }
    __device__ __forceinline__ int wrap(int i) const { return i >= ns ? i - ns : i; } // i < 2 * ns
};

// Which shading pass the wave should run next (0: none -> traverse).  Used both at the top of the wave loop and as the exit test
// of the traversal phase, so the two can never disagree (a disagreement is a livelock: leave traversal, shade nothing, re-enter).
// COST OF A PIXEL = TIME.  The cost pre-pass renders a few samples of every pixel and records how long they took in the slot that
// rendered them (100 MHz clock; the slot's wave is dense and serves its pixels' rays in turn, so the duration is the pixel's sample
// chain: rays x turnaround of ITS rays).  Rounds 1-3 counted rays instead; but a ray that grazes the dragon walks five times as many
// nodes as one that leaves the floor for the sky, and two pixels with the same 2.2 rays per sample differ by 2x in the time their
// chains take (the late half of a cost class of the 1/8 shard of C4: same rays, 60 instead of 30 us per ray).  The cost image holds
// the duration as a class, 16 per doubling: class = 16 log2(duration / 1 us), 1..255 (1 us .. 63 ms); 0 = not this rank's pixel.
__device__ __forceinline__ uint8_t pt_cost_class(uint32_t ticks)
{
    const float us = (float)ticks * 0.01f;
    const int k = us > 1.0f ? (int)(16.0f * __log2f(us)) : 0;
    return (uint8_t)(k < 1 ? 1 : (k > 255 ? 255 : k));
}

enum { PICK_NONE = 0, PICK_HIT = 1, PICK_MISS = 2 };
__device__ __forceinline__ int pick_pass(const WaveCtx& w, bool starving)
{
    const bool miss_ok = !w.miss_blocked;
    const int hits = w.hit_count + w.binned; // hits waiting for shading: not yet classified + in the lobe bins
    if (hits >= w.full_batch) return PICK_HIT;
    if (miss_ok && w.miss_count >= w.full_batch) return PICK_MISS;
    if (w.ray_count < w.ray_low) {
        // the ray queue is about to run dry: a half-full shading pass is cheaper than idle traversal lanes (traversal is ~80 %
        // of a wave's time, shading ~12 %)
        const bool h = hits >= w.min_batch, m = miss_ok && w.miss_count >= w.min_batch;
        if (h && (!m || hits >= w.miss_count)) return PICK_HIT;
        if (m) return PICK_MISS;
    }
    if (starving) { // traversal has nothing to do: shade whatever is queued
        if (hits > 0 && (hits >= w.miss_count || !miss_ok)) return PICK_HIT;
        if (miss_ok && w.miss_count > 0) return PICK_MISS;
    }
    return PICK_NONE;
}

// One shading pass over up to 64 entries of the hit queue (IS_MISS = false) or the miss queue (IS_MISS = true):
// device.cu:136-214 for the hit/miss, then sample accumulation, next camera ray / next pixel (device.cu:229-254).
// (one instance serves both queues - is_miss is wave-uniform: the "next sample / next work item" half of the pass is the same code
// either way, and two copies of it cost 4 KB of an instruction cache the kernel fills)
template <bool COUNT>
__device__ __forceinline__ void shade_pass(const PtKernelParams& P, WaveCtx& w, int lane, Counters& cn, const bool IS_MISS)
{
    const int ns = w.ns;
    uint32_t* lray = w.lray;
    uint32_t* gstate = w.lstate;
#define LF(f, s) lray[(f) * ns + (s)]
#define LFF(f, s) __uint_as_float(lray[(f) * ns + (s)])
#define GF(f, s) gstate[(f) * ns + (s)]
#define GFF(f, s) __uint_as_float(gstate[(f) * ns + (s)])
    int n, n1 = 0, n2 = 0; // items of the pass: n1 from the queue it was called for, n2 topped up from the other one
    bool lane_miss = IS_MISS;
    int ps_slot = 0;
    if (PT_WITH_LOBE_BINS && !IS_MISS && P.lobe_bins) {
        // ---- LOBE-COHERENT HIT PASSES (round 4) ----------------------------------------------------------------------------------
        // sample_disney picks ONE of four lobe bodies per hit from one draw against thresholds that depend on the material only
        // (disney.cuh:31-63), and the bodies are long (GGX sampling + three Smith terms; GTR1 with pow / log; rough glass), so a hit
        // pass over a mixed batch runs all of them back to back, each with the few lanes that chose it (C2: two bodies per pass at
        // ~19 of 64 lanes each).  Here a hit is first CLASSIFIED - the material index travels with the hit (packed with the triangle
        // id: tri_eval, retire), its lobe thresholds come from a 32-word LDS table, and the draw the shader will make next is peeked
        // from the slot's RNG state without advancing it - and appended to the bin of its predicted lobe; the pass then shades ONE
        // bin when that bin alone fills enough lanes (>= pure_min), else everything that is queued, bin by bin.  The shader itself is
        // untouched and still decides everything from its own draw: a wrong prediction (9-bit thresholds; force_btdf, which needs
        // the shading normal) costs divergence, never a different result, and a slot still has one item in flight, so the
        // per-pixel RNG order (device.cu:226-243) is what it was.
        uint32_t bh = w.bin_head, bc = w.bin_count;
        while (w.hit_count > 0) { // classify the new arrivals (at most ns: two rounds)
            const int nc = w.hit_count < PT_WAVE ? w.hit_count : PT_WAVE;
            int lobe = -1, cslot = 0;
            if (lane < nc) {
                cslot = (int)w.hitq[w.wrap(w.hit_head + lane)];
                const uint32_t code = w.ltab[LF(L_AZ, cslot) >> 24]; // material + 1 (0: material_data{} defaults), <= PT_LOBE_TABLE - 1
                const uint32_t q9 = (16807u * GF(S_RNG, cslot) + 1013904223u) >> 23; // top 9 bits of the next draw (random.hpp:61-69)
                const int prev = (int)((GF(S_PACK, cslot) >> 22) & 7u) - 1;
                lobe = q9 < (code & 0x3ffu) ? kLobeMetallic : (q9 < ((code >> 10) & 0x3ffu) ? kLobeClearcoat : (q9 < ((code >> 20) & 0x3ffu) ? kLobeDiffuse : kLobeGlass));
                if ((code & 0x80000000u) && prev == kLobeGlass) lobe = kLobeGlass; // inside a glass object: force_btdf (disney.cuh:39)
                if (code & 0x40000000u) lobe = kLobeDiffuse;                       // emitter: no body runs at all - with the cheapest bin
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned long long mb = __ballot(lobe == b);
                const int cnt_b = (int)((bc >> (8 * b)) & 0xffu), head_b = (int)((bh >> (8 * b)) & 0xffu);
                if (lobe == b) w.binq[b * ns + w.wrap(w.wrap(head_b + cnt_b) + rank_in(mb))] = (uint8_t)cslot;
                bc += (uint32_t)popc64(mb) << (8 * b);
            }
            w.hit_head = w.wrap(w.hit_head + nc);
            w.hit_count -= nc;
            w.binned += nc;
        }
        // the fullest bin alone if it fills pure_min lanes, else all bins in turn
        const int c0 = (int)(bc & 0xffu), c1 = (int)((bc >> 8) & 0xffu), c2 = (int)((bc >> 16) & 0xffu), c3 = (int)(bc >> 24);
        int best = 0, cbest = c0;
        if (c1 > cbest) { best = 1; cbest = c1; }
        if (c2 > cbest) { best = 2; cbest = c2; }
        if (c3 > cbest) { best = 3; cbest = c3; }
        const int pure_min = P.tune[4] > 0 ? P.tune[4] : PT_PURE_MIN;
        const bool pure = cbest >= pure_min && cbest >= w.min_batch;
        int t0 = c0, t1 = c1, t2 = c2, t3 = c3; // items taken from each bin
        if (pure) {
            t0 = best == 0 ? (c0 < PT_WAVE ? c0 : PT_WAVE) : 0; t1 = best == 1 ? (c1 < PT_WAVE ? c1 : PT_WAVE) : 0;
            t2 = best == 2 ? (c2 < PT_WAVE ? c2 : PT_WAVE) : 0; t3 = best == 3 ? (c3 < PT_WAVE ? c3 : PT_WAVE) : 0;
        } else {
            int room = PT_WAVE;
            t0 = t0 < room ? t0 : room; room -= t0;
            t1 = t1 < room ? t1 : room; room -= t1;
            t2 = t2 < room ? t2 : room; room -= t2;
            t3 = t3 < room ? t3 : room;
        }
        n = n1 = t0 + t1 + t2 + t3;
        if (lane < n) {
            const int b = lane < t0 ? 0 : (lane < t0 + t1 ? 1 : (lane < t0 + t1 + t2 ? 2 : 3));
            const int i = lane - (b > 0 ? t0 : 0) - (b > 1 ? t1 : 0) - (b > 2 ? t2 : 0);
            ps_slot = (int)w.binq[b * ns + w.wrap((int)((bh >> (8 * b)) & 0xffu) + i)];
        }
        const uint32_t h0 = (uint32_t)w.wrap((int)(bh & 0xffu) + t0), h1 = (uint32_t)w.wrap((int)((bh >> 8) & 0xffu) + t1),
                       h2 = (uint32_t)w.wrap((int)((bh >> 16) & 0xffu) + t2), h3 = (uint32_t)w.wrap((int)(bh >> 24) + t3);
        w.bin_head = h0 | (h1 << 8) | (h2 << 16) | (h3 << 24);
        w.bin_count = bc - ((uint32_t)t0 | ((uint32_t)t1 << 8) | ((uint32_t)t2 << 16) | ((uint32_t)t3 << 24));
        w.binned -= n;
        if (COUNT) cn.lobe[7] += pure;
    } else {
        // the queue the pass was called for, topped up from the other one when that holds at least topup_min entries: the second half of
        // a pass (sample accumulation, next camera ray / work item, state write-back) is the same for both kinds and costs as much
        // as the hit shader itself, so lanes a half-empty batch leaves idle may as well serve the other queue (option "tune0")
        uint8_t* q = IS_MISS ? w.missq : w.hitq;
        uint8_t* q2 = IS_MISS ? w.hitq : w.missq;
        const int q_head = IS_MISS ? w.miss_head : w.hit_head, q2_head = IS_MISS ? w.hit_head : w.miss_head;
        const int q_count = IS_MISS ? w.miss_count : w.hit_count, q2_count = IS_MISS ? w.hit_count : w.miss_count;
        n1 = q_count < PT_WAVE ? q_count : PT_WAVE;
        const int topup_min = w.topup_min;
        if (topup_min > 0 && topup_min <= PT_WAVE && q2_count >= topup_min && (IS_MISS || !w.miss_blocked)) n2 = q2_count < PT_WAVE - n1 ? q2_count : PT_WAVE - n1;
        n = n1 + n2;
        if (lane < n1) ps_slot = (int)q[w.wrap(q_head + lane)];
        else if (lane < n) { ps_slot = (int)q2[w.wrap(q2_head + lane - n1)]; lane_miss = !IS_MISS; }
        const int hit_taken = IS_MISS ? n2 : n1, miss_taken = IS_MISS ? n1 : n2;
        w.hit_head = w.wrap(w.hit_head + hit_taken); w.hit_count -= hit_taken;
        w.miss_head = w.wrap(w.miss_head + miss_taken); w.miss_count -= miss_taken;
    }
    const bool mine = lane < n;
    if (COUNT) { // (all four with constant indices and selected values: a dynamic index sends the counter block to scratch)
        const uint32_t m1 = IS_MISS ? 1u : 0u, nm = (uint32_t)(IS_MISS ? n1 : n2);
        cn.sched[6] += 1u - m1; cn.sched[7] += (uint32_t)n - nm; cn.sched[8] += m1; cn.sched[9] += nm;
    }
    bool to_ray = false, to_hit = false, to_wait = false, died = false, started = false, ended = false;
    int branch = -1; // COUNT: which branch of the hit shader the item took (0..3 sampled lobe, 4 emitter, 5 NaN retry)
    if (mine) {
        uint32_t pid = GF(S_PIX, ps_slot);
        const bool running = pid != PT_FRESH;
        uint32_t ticket = running ? PT_FRESH : GF(S_RNG, ps_slot);
        const uint32_t qkc = running ? GF(S_QKC, ps_slot) : 0u;
        // S_QKC: chunk index, or in the cost pre-pass (one chunk per pixel) the clock at which the slot started the pixel
        uint32_t chunk = P.cost_out ? 0u : qkc, cost = P.cost_out ? qkc : 0u;
        uint32_t pack = running ? GF(S_PACK, ps_slot) : 0u;
        int s = (int)(pack & 0xffffu);
        PathState ps;
        ps.rng = 0; ps.org = vs(0.0f); ps.dir = vs(0.0f); ps.throughput = vs(1.0f); ps.depth = 0; ps.lobe = kLobeNone; ps.retries = 0;
        int px = 0, py = 0;
        v3 color = vs(0.0f);
        bool need_gen = !running;
        bool have_pixel = running;
        if (running) {
            ps.rng = GF(S_RNG, ps_slot);
            ps.throughput = V(GFF(S_THRX, ps_slot), GFF(S_THRY, ps_slot), GFF(S_THRZ, ps_slot));
            color = V(GFF(S_COLX, ps_slot), GFF(S_COLY, ps_slot), GFF(S_COLZ, ps_slot));
            px = (int)(pid & 0xffffu); // S_PIX holds x | y << 16
            py = (int)(pid >> 16);
            ps.depth = (int)((pack >> 16) & 63u);
            ps.lobe = (int)((pack >> 22) & 7u) - 1;
            ps.retries = (int)(pack >> 25);
            ps.dir = V(LFF(L_DIRX, ps_slot), LFF(L_DIRY, ps_slot), LFF(L_DIRZ, ps_slot));
            if (COUNT) ++cn.rays;
            if (P.dbg_start && !P.cost_out) atomicAdd(P.dbg_start + (size_t)P.width * (size_t)P.height + (uint32_t)px + (uint32_t)P.width * (uint32_t)py, 1u); // diagnostics: rays per pixel
            v3 radiance;
            const int tslot = lane_miss ? -1 : (int)(LF(L_AZ, ps_slot) & (PT_WITH_LOBE_BINS ? P.hit_slot_mask : 0xffffffffu)); // (the hit's material rides in the top byte: retire)
            const uint32_t scat0 = cn.scat;
            int r = shade_hit<COUNT>(P, P.materials, tslot, LFF(L_AX, ps_slot), LFF(L_AY, ps_slot), ps, radiance, cn);
            if (COUNT && !lane_miss) branch = r == SR_RETRY ? 5 : (cn.scat != scat0 ? ps.lobe : 4);
            if (r == SR_RETRY) {
                to_hit = true; // same hit, fresh draws (device.cu:196-201); L_A* still hold the hit
            } else if (r == SR_END) {
                color = color + radiance * ps.throughput; // device.cu:217,243
                if (COUNT) ++cn.samples;
                ++s;
                need_gen = true;
                if (s == chunk_len(P, chunk)) {
                    if (P.cost_out) { // cost image, indexed by pixel id: how long the pre-pass's samples of this pixel took (pt_cost_class)
                        gp(P.cost_out)[(uint32_t)px + (uint32_t)P.width * (uint32_t)py] = pt_cost_class((uint32_t)wall_clock64() - cost);
                        cost = 0u;
                    }
                    ticket = take_ticket(P, w.tier); // in flight while finish_chunk stores the pixel's state
                    finish_chunk(P, chunk, px, py, ps.rng, color);
                    have_pixel = false;
                    ended = true;
                }
            } else {
                to_ray = true;
            }
        }
        if (need_gen) {
            if (!have_pixel) {
                if (ticket == PT_FRESH) ticket = take_ticket(P, w.tier);
                if (ticket == PT_NO_TICKET) {
                    died = true;
                } else if (start_chunk(P, ticket, chunk, px, py, ps.rng, color)) {
                    have_pixel = true;
                    started = true;
                    s = 0;
                    if (P.cost_out) cost = (uint32_t)wall_clock64();
                } else {
                    to_wait = true; // predecessor chunk still running somewhere: keep the ticket, poll again later
                }
            }
            if (have_pixel) {
                gen_camera_ray(P, px, py, ps);
                to_ray = true;
            }
        }
        if (!died) {
            if (to_wait) {
                GF(S_PIX, ps_slot) = PT_FRESH;
                GF(S_RNG, ps_slot) = ticket;
            } else {
                GF(S_PIX, ps_slot) = (uint32_t)px | ((uint32_t)py << 16);
                GF(S_QKC, ps_slot) = P.cost_out ? cost : chunk;
                GF(S_RNG, ps_slot) = ps.rng;
                GF(S_PACK, ps_slot) = PT_PACK(s, ps.depth, ps.lobe, ps.retries);
                GF(S_COLX, ps_slot) = __float_as_uint(color.x);
                GF(S_COLY, ps_slot) = __float_as_uint(color.y);
                GF(S_COLZ, ps_slot) = __float_as_uint(color.z);
                GF(S_THRX, ps_slot) = __float_as_uint(ps.throughput.x);
                GF(S_THRY, ps_slot) = __float_as_uint(ps.throughput.y);
                GF(S_THRZ, ps_slot) = __float_as_uint(ps.throughput.z);
                if (to_ray) {
                    LF(L_DIRX, ps_slot) = __float_as_uint(ps.dir.x);
                    LF(L_DIRY, ps_slot) = __float_as_uint(ps.dir.y);
                    LF(L_DIRZ, ps_slot) = __float_as_uint(ps.dir.z);
                    LF(L_AX, ps_slot) = __float_as_uint(ps.org.x);
                    LF(L_AY, ps_slot) = __float_as_uint(ps.org.y);
                    LF(L_AZ, ps_slot) = __float_as_uint(ps.org.z);
                }
            }
        }
    }
    if (COUNT && (!IS_MISS || n2 > 0)) {
        int bodies = 0, branches = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const unsigned long long mb = __ballot(branch == b);
            cn.lobe[b] += (uint32_t)popc64(mb);
            cn.lobe[8 + b] += mb != 0ull;
            bodies += (b < 4 && mb != 0ull) ? 1 : 0;
            branches += mb != 0ull ? 1 : 0;
        }
        cn.lobe[14] += bodies >= 2;
        cn.lobe[15] += branches == 1;
    }
    const unsigned long long m_ray = __ballot(to_ray), m_hit = __ballot(to_hit), m_wait = __ballot(to_wait), m_dead = __ballot(died);
    if (to_ray) w.rayq[w.wrap(w.wrap(w.ray_head + w.ray_count) + rank_in(m_ray))] = (uint8_t)ps_slot;
    if (to_hit) w.hitq[w.wrap(w.wrap(w.hit_head + w.hit_count) + rank_in(m_hit))] = (uint8_t)ps_slot;
    if (to_wait) w.missq[w.wrap(w.wrap(w.miss_head + w.miss_count) + rank_in(m_wait))] = (uint8_t)ps_slot;
    w.ray_count += popc64(m_ray);
    w.hit_count += popc64(m_hit);
    w.miss_count += popc64(m_wait);
    w.n_dead += popc64(m_dead);
    w.n_run += popc64(__ballot(started)) - popc64(__ballot(ended));
    if (COUNT) { cn.sched[16] += popc64(m_wait); cn.sched[17] += (w.n_dead > 0) ? n : 0; } // [17]: rays shaded after the queue ran dry (wind-down)
    // a pass that only polled unpublished tickets must not be repeated before the wave has done something else
    w.miss_blocked = n > 0 && popc64(m_wait) == n; // (only entries of the miss queue can be slots that wait for a ticket)
#undef LF
#undef LFF
#undef GF
#undef GFF
}
} // namespace

// =====================================================================================================================
// Group walk: the traversal of a SPARSE wave (round 3)
// =====================================================================================================================
// A wave with few rays to trace wastes its lanes in the per-lane walk: a node step costs the same wave-instructions for 4 busy
// lanes as for 64, the vector L1 spends ~25 cycles on a load instruction however few lanes it has, and a ray still needs its
// ~12 dependent quad-node steps plus leaf steps - 25-40 us per ray where a dense wave manages 10 (profiles/r02_summary.md).  That
// regime is the tail of every frame and most of a frame whose pixels all start at once (one rank's shard of a multi-GPU frame,
// small images), and a pixel is one sequential chain of rays (device.cu:226-243), so the frame waits for it.
// Here EIGHT LANES WALK ONE RAY: the tree is read as oct nodes (PtNode8: up to eight descendants of a binary node, three binary
// levels per record), lane k of a group tests child k (two 16-byte loads: the eight lanes fetch the record's two cache lines
// coalesced), the group finds the nearest hit child with three DPP steps, pushes the other hit children on the group's stack
// and continues with the nearest; at a leaf, lane k tests triangle k and the group reduces (t, id) lexicographically.  A ray goes
// down the tree in a third of the dependent steps, every step is one memory round trip for up to eight rays, and a wave needs
// eight rays, not sixty-four, to fill its lanes.  Node groups and leaf groups are served in the SAME iteration (their loads are in
// flight together), so no ray waits for a majority.
// The closest hit does not depend on the visiting order (tri_eval's tie-break on the triangle id, conservative boxes), so the
// image is bit-identical to the per-lane walk's - the parity suite runs with the group walk forced on as well (option groups = 2).
// A group's stack is eight adjacent columns of the per-lane LDS stack area: entry e at stack[(e >> 3) * 64 + 8 * group + (e & 7)]
// (capacity 8 * levels; the host enables the group walk only if 7 * depth8 + 1 entries fit, pt_api.cpp).
#ifndef PT_GROUP_ORDERED
#define PT_GROUP_ORDERED 1 // the hit children that are not entered go on the group's stack farthest first (0: in lane order)
#endif
#ifndef PT_GROUP_MAX_RAYS
#define PT_GROUP_MAX_RAYS 16 // ray-queue level up to which a traversal phase uses the group walk (option groups = 1)
#endif
#ifndef PT_GROUP_MAX_RUN
#define PT_GROUP_MAX_RUN 24  // running pixels of the wave up to which it counts as sparse
#endif

namespace {
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); }  // quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true); }  // quad_perm [2,3,0,1]
__device__ __forceinline__ uint32_t dpp_xor3(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xf, 0xf, true); }  // quad_perm [3,2,1,0]
__device__ __forceinline__ uint32_t dpp_mir8(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true); } // row_half_mirror
__device__ __forceinline__ uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }

// (bound_ctrl with a zero `old`: every lane of these patterns has a source lane, and in this form the compiler folds the DPP fetch into the
// consuming instruction instead of a v_mov_b32_dpp per fetch)
// one reduction stage of the leaf step: take the partner's candidate if it is better ((t, id) lexicographic)
#define PT_HIT_STAGE(DPP)                                                                                           \
    {                                                                                                               \
        const float t2 = __uint_as_float(DPP(__float_as_uint(hc.t)));                                               \
        const int id2 = (int)DPP((uint32_t)hc.id);                                                                  \
        const uint32_t u2 = DPP(__float_as_uint(hc.u)), v2 = DPP(__float_as_uint(hc.v)), s2 = DPP((uint32_t)hc.slot); \
        const bool better = t2 < hc.t || (t2 == hc.t && id2 < hc.id);                                               \
        hc.t = better ? t2 : hc.t; hc.id = better ? id2 : hc.id;                                                    \
        hc.u = better ? __uint_as_float(u2) : hc.u; hc.v = better ? __uint_as_float(v2) : hc.v;                     \
        hc.slot = better ? (int)s2 : hc.slot;                                                                       \
    }

// park: the wave's HBM park area (as for the per-lane walk): a group phase ends as soon as a shading batch is due, the unfinished
// groups keep their registers there and their stacks in LDS, and the next traversal phase of the wave is a group phase again
// (n_parked > 0 on entry: resume).  Returns the number of parked lanes.
template <bool COUNT>
__device__ __forceinline__ int traverse_groups(const PtKernelParams& P, WaveCtx& w, int lane, uint32_t* stack0, uint32_t PT_AS1* park, int n_parked, Counters& cn, const bool box_exact)
{
    const int ns = w.ns;
    uint32_t* lray = w.lray;
#define LF(f, s) lray[(f) * ns + (s)]
#define LFF(f, s) __uint_as_float(lray[(f) * ns + (s)])
    const PtNode8* __restrict__ nodes8 = P.nodes8;
    const PtTri* __restrict__ tris = P.tris;
    const int sub = lane & 7, gbase = lane & 56;
    const bool lead = sub == 0;
    uint32_t* gstack = stack0 + gbase; // entry e: gstack[(e >> 3) * 64 + (e & 7)]
    int pslot = -1, cur = PT_DONE, sp = 0;
    Hit h;
    h.t = kTMax; h.u = h.v = 0.0f; h.slot = -1; h.id = 0x7fffffff;
    v3 o = vs(0.0f), d = vs(1.0f), inv = vs(1.0f);
    if (n_parked > 0) { // resume the groups parked by the previous group phase
        pslot = (int)park[K_PSLOT * PT_WAVE];
        if (pslot >= 0) {
            cur = (int)park[K_CUR * PT_WAVE];
            sp = (int)park[K_SP * PT_WAVE];
            h.t = __uint_as_float(park[K_BT * PT_WAVE]);
            h.u = __uint_as_float(park[K_BU * PT_WAVE]);
            h.v = __uint_as_float(park[K_BV * PT_WAVE]);
            h.slot = (int)park[K_BSLOT * PT_WAVE];
            h.id = (int)park[K_BID * PT_WAVE];
            o = V(LFF(L_AX, pslot), LFF(L_AY, pslot), LFF(L_AZ, pslot));
            d = V(LFF(L_DIRX, pslot), LFF(L_DIRY, pslot), LFF(L_DIRZ, pslot));
            inv = ray_inv(d);
        }
    }
    int n_iter = 0;
    bool first = true;
    for (;;) {
        if (++n_iter > (1 << 22)) { // as in the per-lane walk: a corrupt reference must not spin
            if (lane == 0) gp(P.error_flag)[0] = 1u;
            pslot = -1;
            break;
        }
        // ---- retire finished groups (the leader writes the hit over the ray origin), refill idle groups from the ray queue ----
        const bool fin = pslot >= 0 && cur == PT_DONE;
        if (first || __ballot(fin) != 0ull) {
            first = false;
            const bool fin_hit = fin && lead && h.slot >= 0, fin_miss = fin && lead && h.slot < 0;
            const unsigned long long m_fh = __ballot(fin_hit), m_fm = __ballot(fin_miss);
            if (fin_hit) {
                LF(L_AX, pslot) = __float_as_uint(h.u);
                LF(L_AY, pslot) = __float_as_uint(h.v);
                LF(L_AZ, pslot) = (uint32_t)h.slot | (PT_WITH_LOBE_BINS ? (uint32_t)h.id << 24 & ~P.hit_slot_mask : 0u); // + material code of the hit (PtTri::id, packed)
                w.hitq[w.wrap(w.wrap(w.hit_head + w.hit_count) + rank_in(m_fh))] = (uint8_t)pslot;
            }
            if (fin_miss) w.missq[w.wrap(w.wrap(w.miss_head + w.miss_count) + rank_in(m_fm))] = (uint8_t)pslot;
            if (fin) pslot = -1;
            w.hit_count += popc64(m_fh);
            w.miss_count += popc64(m_fm);
            const unsigned long long m_idle = __ballot(pslot < 0); // whole groups
            const int n_idle = popc64(m_idle) >> 3;
            const int take = n_idle < w.ray_count ? n_idle : w.ray_count;
            if (take > 0) {
                const int rk = rank_in(m_idle) >> 3; // rank of this lane's group among the idle groups
                if (pslot < 0 && rk < take) {
                    pslot = (int)w.rayq[w.wrap(w.ray_head + rk)];
                    o = V(LFF(L_AX, pslot), LFF(L_AY, pslot), LFF(L_AZ, pslot));
                    d = V(LFF(L_DIRX, pslot), LFF(L_DIRY, pslot), LFF(L_DIRZ, pslot));
                    inv = ray_inv(d);
                    cur = P.root8; // an oct node, or the leaf code of a scene of <= leaf_size triangles
                    sp = 0;
                    h.t = kTMax; h.u = 0.0f; h.v = 0.0f; h.slot = -1; h.id = 0x7fffffff;
                }
                w.ray_head = w.wrap(w.ray_head + take);
                w.ray_count -= take;
                if (COUNT) cn.grp[5] += take;
            }
            if (__ballot(pslot >= 0) == 0ull) break; // every ray of this phase is in the hit or the miss queue
            // a shading batch is due (the thresholds follow the wave's running pixels): park the unfinished groups and go - the
            // idle groups get their next rays from that pass
            if (pick_pass(w, false) != PICK_NONE) break;
        }
        // ---- one step for every group: the loads of the node groups and of the leaf groups are in flight together ----
        const bool is_node = cur >= 0, is_leaf = cur < PT_DONE;
        const unsigned long long m_nodeg = __ballot(is_node), m_leafg = __ballot(is_leaf);
        if (COUNT) {
            cn.grp[1] += 1; cn.grp[2] += popc64(__ballot(pslot >= 0)) >> 3;
            cn.grp[3] += popc64(m_nodeg) >> 3; cn.grp[4] += popc64(m_leafg) >> 3;
        }
        const uint32_t code = ~(uint32_t)cur;
        const int first = (int)(code >> 3), count = is_leaf ? (int)(code & 7u) : 0;
        const bool tri_on = sub < count;
        // unconditional loads inside the wave-uniform branches (idle lanes read record 0): every request is in flight before the first wait
        f32x4 c0 = {0.0f, 0.0f, 0.0f, 0.0f}, c1 = c0, ta = c0, tb4 = c0, tc = c0;
        if (m_nodeg != 0ull) {
            const size_t nb = is_node ? (size_t)(uint32_t)cur * sizeof(PtNode8) + (size_t)sub * 32 : 0;
            c0 = ldg4(nodes8, nb); c1 = ldg4(nodes8, nb + 16);
        }
        if (m_leafg != 0ull) {
            const size_t tb = tri_on ? (size_t)(uint32_t)(first + sub) * sizeof(PtTri) : 0;
            ta = ldg4(tris, tb); tb4 = ldg4(tris, tb + 16); tc = ldg4(tris, tb + 32);
        }
        if (COUNT && lead) { cn.nodes += is_node ? 4u : 0u; cn.tris += (uint32_t)count; } // an oct node is four 64-byte units
        int next = PT_DONE;   // node groups with a hit child: the nearest one
        bool descend = false;
        if (m_nodeg != 0ull) {
            // -- node groups: lane k tests child k
            float tn = 0.0f;
            const int ref = __float_as_int(c1.z);
            bool hit = false;
            if (is_node) hit = box_exact ? box_test(c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, o, inv, h.t, tn) : box_test_fma(c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, o, inv, h.t, tn);
            // entry distance with the child index in its low bits: unique within the group, ordered like the distance (tn >= kTMin > 0)
            const uint32_t key = hit ? ((__float_as_uint(tn) & ~7u) | (uint32_t)sub) : 0xffffffffu;
            uint32_t kmin = umin_(key, dpp_xor1(key));
            kmin = umin_(kmin, dpp_xor2(kmin));
            kmin = umin_(kmin, dpp_mir8(kmin));
            descend = kmin != 0xffffffffu;
            const bool push = hit && key != kmin;
            // the other hit children go on the group's stack FARTHEST FIRST (the nearest of them is popped first): a pushing lane's
            // position is the number of pushing lanes with a larger key - seven cross-lane fetches: the three other lanes of its
            // quad, the mirror lane and the three other lanes of the mirror lane's quad
            const uint32_t pbits = (uint32_t)(__ballot(push) >> gbase) & 0xffu; // this group's pushing lanes
#if PT_GROUP_ORDERED
            const uint32_t pk = push ? key : 0u;
            const uint32_t pm = dpp_mir8(pk);
            const int prank = (int)(dpp_xor1(pk) > key) + (int)(dpp_xor2(pk) > key) + (int)(dpp_xor3(pk) > key) + (int)(pm > key) + (int)(dpp_xor1(pm) > key) +
                              (int)(dpp_xor2(pm) > key) + (int)(dpp_xor3(pm) > key);
#else
            const int prank = __builtin_popcount(pbits & ((1u << sub) - 1u)); // lane order
#endif
            if (push) {
                const int e = sp + prank;
                gstack[(e >> 3) * PT_WAVE + (e & 7)] = (uint32_t)ref;
            }
            next = __builtin_amdgcn_ds_bpermute((gbase | (int)(kmin & 7u)) << 2, ref);
            if (is_node) sp += __builtin_popcount(pbits);
        }
        if (m_leafg != 0ull) {
            // -- leaf groups: lane k tests triangle k; (t, id)-lexicographic minimum over the group, payload (u, v, slot)
            Hit hc = h;
            if (tri_on) tri_eval(ta, tb4, tc, first + sub, o, d, hc);
            PT_HIT_STAGE(dpp_xor1)
            PT_HIT_STAGE(dpp_xor2)
            PT_HIT_STAGE(dpp_mir8)
            h = hc; // unchanged where the group is not at a leaf: all its lanes held the same candidate
        }
        // -- next reference
        if (is_node && descend) {
            cur = next;
        } else if (is_node || is_leaf) {
            if (sp > 0) {
                --sp;
                cur = (int)gstack[(sp >> 3) * PT_WAVE + (sp & 7)];
            } else {
                cur = PT_DONE;
            }
        }
    }
    // ---- park unfinished groups until the next traversal phase ----
    const int n_left = popc64(__ballot(pslot >= 0));
    if (n_left > 0) {
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
    }
    return n_left;
#undef LF
#undef LFF
}
} // namespace

// a lane that arrives at a leaf stashes it and continues with the next node on its stack (speculative traversal)
#define PT_STASH_LEAF(ENTRIES)                                                       \
    if (cur < PT_DONE && pend == PT_DONE) {                                          \
        pend = cur;                                                                  \
        if (sp > 0) {                                                                \
            --sp;                                                                    \
            cur = (int)stack_pop<PT_WAVE, ENTRIES>(stack, ovf, sp);                  \
        } else {                                                                     \
            cur = PT_DONE;                                                           \
        }                                                                            \
    }

// The instrumented instance gets 256 VGPRs (2 waves/SIMD): with 128 it spills ~50 registers to scratch, and a spilling build of this
// kernel rendered wrong pixels in round 1 (both instrumented instances, identical source otherwise; never the 128-VGPR product
// instance, which does not spill).  pt_render refuses to launch any instance that needs scratch (pt_kernel_geometry).
// Instances (WAVES = minimum waves per SIMD the register allocator must allow = 512 / VGPR budget):
//   <false, PT_WAVES_PER_EU = 4>   the product instance: 128 VGPRs, 16 waves per CU;
//   <false, PT_FALLBACK_WAVES = 3> the same source with a 168-VGPR budget (12 waves per CU): chosen automatically when the 128-VGPR
//                                  instance of THIS build needs scratch (a compiler bump, a local edit) - slower, still exact,
//                                  instead of refusing to render (pt_kernel_geometry; option "fallback" forces it);
//   <true, PT_COUNT_WAVES_PER_EU = 2> the instrumented instance.
#ifndef PT_FALLBACK_WAVES
#define PT_FALLBACK_WAVES 3
#endif
//   EXACT: the slab tests use the subtracting form (camera far outside the scene: node4_step) - instances of their own, so that the product
//   instances carry one form only; the instrumented instance switches at run time (P.box_exact).
template <bool COUNT, int WAVES, bool EXACT>
__global__ void __launch_bounds__(PT_WAVE, WAVES) pt_render_wave_kernel(const PtKernelParams* __restrict__ Pp)
{
    // The parameter block lives in HBM and is read with scalar loads where it is used.  Passed by value it arrives as
    // s_load_dwordx16 tuples that stay live for the whole kernel; the register allocator then spilled them to VGPR lanes
    // and re-read all 16 with v_readlane in EVERY node step just to get the `nodes` pointer.
    const PtKernelParams& P = *Pp;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const int ns = P.ns;
    const int lds_stack = P.lds_levels; // pt_wave_lds_stack(), from the host
    uint32_t* lds0 = lds;
    uint32_t* stack = lds0 + lane;                   // stack[level * 64], levels < PT_LDS_STACK
    uint32_t* lray = lds0 + lds_stack * PT_WAVE;     // lray[field * ns + slot]
    uint32_t* lstate = lray + L_NFIELDS * ns;       // lstate[field * ns + slot]
    const int ovf_levels = P.stack_entries > PT_LDS_STACK ? P.stack_entries - PT_LDS_STACK : 0;
    uint32_t PT_AS1* wave_state = gp(P.slot_state) + (size_t)blockIdx.x * ((size_t)(K_NFIELDS + ovf_levels) * PT_WAVE);
    uint32_t PT_AS1* park = wave_state + lane; // park[field * 64]
    uint32_t PT_AS1* ovf = park + K_NFIELDS * PT_WAVE;                                                                   // ovf[level * 64]
    WaveCtx w;
    w.lray = lray;
    w.lstate = lstate;
    w.ns = ns;
    w.rayq = reinterpret_cast<uint8_t*>(lstate + S_NFIELDS * ns);
    w.hitq = w.rayq + ns;
    w.missq = w.hitq + ns;
    w.binq = w.missq + ns;
    uint32_t* ltab = reinterpret_cast<uint32_t*>(w.binq + ((PT_WITH_LOBE_BINS && P.lobe_bins) ? 4 * ns : 0)); // (4-byte aligned: ns is a multiple of 4 when the bins exist)
    w.ltab = ltab;
    if (PT_WITH_LOBE_BINS && P.lobe_bins && lane < PT_LOBE_TABLE) ltab[lane] = gp(P.lobe_codes)[lane];
    w.bin_head = w.bin_count = 0u;
    w.binned = 0;
    const PtNode* __restrict__ nodes = P.nodes;
    const PtNode4* __restrict__ nodes4 = P.nodes4;
    const bool quad = !COUNT || nodes4 != nullptr; // the product instances walk quad nodes only (pt_api.cpp launches the instrumented one otherwise): a compile-time fact there
    const bool box_exact = EXACT || (COUNT && P.box_exact != 0); // node4_step / group walk: the subtracting slab form (camera far outside the scene)
    const PtTri* __restrict__ tris = P.tris;

#define LF(f, s) lray[(f) * ns + (s)]
#define LFF(f, s) __uint_as_float(lray[(f) * ns + (s)])

    // every slot starts "fresh" (needs a pixel) and sits in the miss queue so that the first shading passes start them
    if (blockIdx.x == 0 && lane == 0) gp(P.lap_ticks)[0] = wall_clock64(); // diagnostics timeline origin
    for (int i = lane; i < ns; i += PT_WAVE) {
        w.missq[i] = (uint8_t)i;
        lstate[S_PIX * ns + i] = PT_FRESH;
        lstate[S_RNG * ns + i] = PT_FRESH;
    }
    // express waves (take_ticket): few pixels, all of them long sample chains - the wave gets the SIMD when it asks for it
    w.tier.counter = nullptr;
    w.tier.q0 = w.tier.count = 0u;
    w.tier.express = P.n_express != 0u && blockIdx.x < (uint32_t)P.express_waves;
    int ns_live = w.tier.express ? (P.ns_express < ns ? P.ns_express : ns) : ns; // slots that ever get a pixel
    if (P.tiers != nullptr) { // tier schedule: which cost class does this workgroup serve, and how many of its pixels at a time?
        const uint32_t PT_AS1* tt = gp(P.tiers);
        const uint32_t n_tiers = tt[0];
        if (n_tiers == 0u) { // the plan chose the ring schedule (pt_plan_tiers_kernel): the launch has more workgroups than that wants
            if (blockIdx.x >= (uint32_t)P.ring_grid) ns_live = 0;
        } else {
        ns_live = 0; // a workgroup beyond the plan has nothing to do and must not touch a tier's counter
        w.tier.counter = P.queue_head + PT_TIER_COUNTER(0);
        for (uint32_t t = 0; t < n_tiers; ++t) {
            const uint32_t PT_AS1* e = tt + 1 + PT_TIER_WORDS * t;
            if (blockIdx.x - e[3] < e[4]) {
                w.tier.q0 = e[0];
                w.tier.count = e[1];
                ns_live = (int)e[2] < ns ? (int)e[2] : ns;
                w.tier.counter = P.queue_head + PT_TIER_COUNTER(t);
                break;
            }
        }
        w.tier.express = ns_live <= PT_GROUP_MAX_RAYS;
        }
    }
    if (w.tier.express) __builtin_amdgcn_s_setprio(3);
    w.miss_blocked = false;
    w.topup_min = P.tune[0] > 0 ? P.tune[0] : ((P.env_use_map && P.env_map.width > 0) ? 0 : PT_TOPUP_MIN);
    w.topup = w.topup_min > 0 && w.topup_min <= PT_WAVE;
    w.min_batch = P.tune[1] > 0 ? P.tune[1] : PT_MIN_BATCH;
    w.ray_low = P.tune[2] > 0 ? P.tune[2] : PT_RAY_LOW;
    w.full_batch = P.tune[3] > 0 ? P.tune[3] : PT_WAVE;
    const int mb0 = w.min_batch, rl0 = w.ray_low, fb0 = w.full_batch;
    w.n_run = 0;
    w.adapt = P.tune[5] != 2; // option "adaptive" (default on; 2 = off)
    w.ray_head = 0; w.ray_count = 0; w.hit_head = 0; w.hit_count = 0; w.miss_head = 0; w.miss_count = ns_live; w.n_dead = ns - ns_live;
    int n_parked = 0, n_rounds = 0;
    bool parked_groups = false; // what is parked belongs to a group phase (oct nodes, group stacks): it can only be resumed by one
    const int grp_max_rays = P.tune[6] > 0 ? P.tune[6] : PT_GROUP_MAX_RAYS, grp_max_run = P.tune[7] > 0 ? P.tune[7] : PT_GROUP_MAX_RUN; // options "tune6", "tune7"
    Counters cn;

    unsigned long long t_begin = 0;
    if (COUNT) t_begin = __builtin_amdgcn_s_memtime();
    unsigned long long t_dead = 0; // COUNT: when the first slot of this wave found the work queue exhausted
    while (w.n_dead < ns) {
        unsigned long long t0 = 0;
        if (COUNT) t0 = __builtin_amdgcn_s_memtime();
        if (COUNT && w.n_dead > 0 && t_dead == 0) {
            t_dead = t0;
            if (P.census_mode == 1) { // census of the wind-down only
#pragma unroll
                for (int k = 0; k < 7; ++k) cn.cyc[k] = 0;
#pragma unroll
                for (int k = 0; k < 19; ++k) cn.sched[k] = 0;
                cn.nodes = cn.tris = cn.rays = 0;
            }
        }
        const bool starving = w.ray_count == 0 && n_parked == 0;
        const int pick = pick_pass(w, starving);
        // scheduler bug guard: never hang the GPU.  n_rounds counts the rounds since this wave last shaded a ray (waiting for another
        // wave's chunk is legitimate and can take long: the cap is ~30 s of nothing but polls and sleeps)
        if (++n_rounds > PT_WATCHDOG_ROUNDS) {
            if (lane == 0) gp(P.error_flag)[0] = 1u;
            break;
        }
        if (pick != PICK_NONE) { // one call site: one copy of the pass in the instruction cache
            const bool miss_pass = pick == PICK_MISS;
            shade_pass<COUNT>(P, w, lane, cn, miss_pass);
            if (!miss_pass || !w.miss_blocked) n_rounds = 0; // hits are always running paths; a miss pass may have done nothing but poll unpublished tickets
            w.retune(mb0, rl0, fb0);
            if (COUNT) { const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0; cn.cyc[3] += miss_pass ? 0ull : dt; cn.cyc[4] += miss_pass ? dt : 0ull; }
        } else if (starving) {
            // every live slot of this wave waits for a work item that another wave is still rendering
            __builtin_amdgcn_s_sleep(64);
            w.miss_blocked = false;
            if (COUNT) { cn.sched[18] += 1; cn.cyc[6] += __builtin_amdgcn_s_memtime() - t0; }
        } else if (P.nodes8 != nullptr && (n_parked == 0 ? (P.groups == 2 || (w.ray_count <= grp_max_rays && w.n_run <= grp_max_run)) : parked_groups)) {
            // sparse wave: eight lanes per ray; unfinished groups stay parked for the wave's next traversal phase, which is then a group phase too
            n_parked = traverse_groups<COUNT>(P, w, lane, lds0, park, n_parked, cn, box_exact);
            parked_groups = true;
            if (COUNT) { cn.grp[0] += 1; cn.grp_cyc += __builtin_amdgcn_s_memtime() - t0; }
        } else {
            // ======================= TRAVERSAL PHASE ==============================================================
            int pslot = -1;
            int cur = PT_DONE, sp = 0;
            int pend = PT_DONE; // pending leaf reference (< PT_DONE) or PT_DONE for none
            Hit h;
            h.t = kTMax; h.u = h.v = 0.0f; h.slot = -1; h.id = 0x7fffffff;
            v3 o = vs(0.0f), d = vs(1.0f), inv = vs(1.0f);
            if (n_parked > 0) { // resume traversals parked by the previous phase switch
                pslot = (int)park[K_PSLOT * PT_WAVE];
                if (pslot >= 0) {
                    cur = (int)park[K_CUR * PT_WAVE];
                    sp = (int)park[K_SP * PT_WAVE];
                    h.t = __uint_as_float(park[K_BT * PT_WAVE]);
                    h.u = __uint_as_float(park[K_BU * PT_WAVE]);
                    h.v = __uint_as_float(park[K_BV * PT_WAVE]);
                    h.slot = (int)park[K_BSLOT * PT_WAVE];
                    h.id = (int)park[K_BID * PT_WAVE];
                    pend = (int)park[K_PEND * PT_WAVE];
                    o = V(LFF(L_AX, pslot), LFF(L_AY, pslot), LFF(L_AZ, pslot));
                    d = V(LFF(L_DIRX, pslot), LFF(L_DIRY, pslot), LFF(L_DIRZ, pslot));
                    inv = ray_inv(d);
                }
            }
            bool first = true;
            int n_retire_passes = 0, n_inner = 0;
            unsigned long long t1 = 0;
            if (COUNT) { t1 = __builtin_amdgcn_s_memtime(); cn.cyc[5] += t1 - t0; }
            for (;;) {
                // every iteration steps or retires at least one lane, and a ray needs a few thousand steps at most: far beyond that
                // the traversal state is inconsistent (e.g. a corrupt node reference) - give up loudly instead of spinning
                if (++n_inner > (1 << 22)) {
                    if (lane == 0) gp(P.error_flag)[0] = 1u;
                    n_rounds = PT_WATCHDOG_ROUNDS;
                    break;
                }
                // idle lanes (pslot < 0) always hold cur == PT_DONE and pend == PT_DONE
                const unsigned long long m_leaf = __ballot(pend < PT_DONE);               // lanes with a stashed leaf to test
                const unsigned long long m_done = __ballot(cur == PT_DONE) & __ballot(pslot >= 0) & ~m_leaf;
                const unsigned long long m_node = __ballot(cur >= 0);
                const int n_done = popc64(m_done);
                if (COUNT) {
                    const int n_idle_c = popc64(__ballot(pslot < 0));
                    cn.sched[12] += 1; cn.sched[13] += n_idle_c; cn.sched[14] += n_done; cn.sched[15] += popc64(m_node | m_leaf);
                    if (w.n_dead > 0) { cn.sched[10] += 1; cn.sched[11] += n_idle_c; } // wind-down: the pixel queue is exhausted
                }
                if (first || n_done >= PT_RETIRE_MIN || (m_node | m_leaf) == 0ull) {
                    first = false;
                    // ---- retire finished rays into the hit / miss queues (the hit overwrites the ray origin) ----
                    const bool fin = pslot >= 0 && cur == PT_DONE && pend == PT_DONE;
                    const bool fin_hit = fin && h.slot >= 0;
                    const bool fin_miss = fin && h.slot < 0;
                    const unsigned long long m_fh = __ballot(fin_hit), m_fm = __ballot(fin_miss);
                    if (COUNT) { cn.sched[4] += 1; cn.sched[5] += popc64(m_fh | m_fm); }
                    if (fin_hit) {
                        LF(L_AX, pslot) = __float_as_uint(h.u);
                        LF(L_AY, pslot) = __float_as_uint(h.v);
                        LF(L_AZ, pslot) = (uint32_t)h.slot | (PT_WITH_LOBE_BINS ? (uint32_t)h.id << 24 & ~P.hit_slot_mask : 0u); // + material code of the hit (PtTri::id, packed)
                        w.hitq[w.wrap(w.wrap(w.hit_head + w.hit_count) + rank_in(m_fh))] = (uint8_t)pslot;
                    }
                    if (fin_miss) w.missq[w.wrap(w.wrap(w.miss_head + w.miss_count) + rank_in(m_fm))] = (uint8_t)pslot;
                    if (fin) pslot = -1;
                    w.hit_count += popc64(m_fh);
                    w.miss_count += popc64(m_fm);
                    // ---- refill idle lanes from the ray queue ----
                    const unsigned long long m_idle = __ballot(pslot < 0);
                    const int n_idle = popc64(m_idle);
                    const int take = n_idle < w.ray_count ? n_idle : w.ray_count;
                    if (take > 0) {
                        const int rk = rank_in(m_idle);
                        if (pslot < 0 && rk < take) {
                            pslot = (int)w.rayq[w.wrap(w.ray_head + rk)];
                            o = V(LFF(L_AX, pslot), LFF(L_AY, pslot), LFF(L_AZ, pslot));
                            d = V(LFF(L_DIRX, pslot), LFF(L_DIRY, pslot), LFF(L_DIRZ, pslot));
                            inv = ray_inv(d);
                            // a scene of <= leaf_size triangles has a leaf as its root: it starts as the stashed leaf
                            cur = P.root >= 0 ? P.root : PT_DONE;
                            sp = 0;
                            pend = P.root < PT_DONE ? P.root : PT_DONE;
                            h.t = kTMax; h.u = 0.0f; h.v = 0.0f; h.slot = -1; h.id = 0x7fffffff;
                        }
                        w.ray_head = w.wrap(w.ray_head + take);
                        w.ray_count -= take;
                    }
                    const unsigned long long m_busy = __ballot(pslot >= 0);
                    if (COUNT) { unsigned long long t2 = __builtin_amdgcn_s_memtime(); cn.cyc[2] += t2 - t1; t1 = t2; }
                    if (m_busy == 0ull) break;                                                           // nothing in flight
                    if (++n_retire_passes >= 16) w.miss_blocked = false; // time to poll the waiting tickets again
                    // a full shading batch is ready: go and turn it into rays (a miss queue that only holds unpublished tickets does not count)
                    // (a sparse wave stays until its rays are done: nothing parked, so the next phase can be a group walk)
                    if (pick_pass(w, false) != PICK_NONE && !(P.groups == 1 && w.n_run <= grp_max_run && w.ray_count == 0)) break;
                    continue;
                }
                // ---- one step for the majority: a BVH node step or a triangle test (thresholds 12..40 and node+triangle in
                // every iteration were measured 3-40 % slower, profiles/r01_summary.md) ----
                // Speculative traversal (Aila-Laine): a lane arriving at a leaf stashes it in `pend` and keeps walking nodes; a second
                // leaf blocks it (cur stays on that leaf) until a triangle step has drained `pend`.  Closest hit does not depend on
                // visiting order (tie-break on triangle id), so this only changes which lanes are busy, not the result.
                if (popc64(m_node) >= popc64(m_leaf)) {
                    if (COUNT) { cn.sched[0] += 1; cn.sched[1] += popc64(m_node); }
                    // stack levels >= PT_LDS_STACK live in HBM; 0.006 % of the pushes on C4 go there, so the common instance of the
                    // step (chosen by one wave-uniform branch) does not carry that code at all
                    // The scheduler round around a step (three ballots, retire test, majority vote, loop-carried copies) costs about as
                    // much as the step: node steps come in bursts of PT_NODE_REPS, repeated while >= PT_NODE_KEEP lanes still want one
                    // (1 x 1 -> 2 x up to 3: C4 626 -> 599 ms, its 1/8 shard 480 -> 435 ms, C2 98 -> 91 ms).
                    // every push of a burst must stay inside the LDS part of the stack for the common instance: one push per step when
                    // walking PtNode[], up to three per quad-node step
                    const int n_reps = quad ? PT_QUAD_REPS : PT_NODE_REPS;
                    const int sp_lim = quad ? PT_LDS_STACK - 3 * PT_QUAD_REPS : PT_LDS_STACK - PT_NODE_REPS;
                    if (__ballot(sp > sp_lim) == 0ull) {
#if PT_NODE_KEEP > 0
                      // a sparse wave (adapt): the bursts go on while at least half of the lanes that started them want another step, up to twice as many
                      const int n_node0 = popc64(m_node);
                      const int keep = w.adapt && n_node0 < 2 * PT_NODE_KEEP ? (n_node0 + 1) / 2 : PT_NODE_KEEP;
                      const int max_bursts = (w.adapt && n_node0 < PT_NODE_KEEP ? 2 * PT_NODE_BURSTS : PT_NODE_BURSTS) * (PT_NODE_REPS / (quad ? PT_QUAD_REPS : PT_NODE_REPS));
                      for (int burst = 0; burst < max_bursts; ++burst) {
                        if (burst > 0 && (popc64(__ballot(cur >= 0)) < keep || __ballot(sp > sp_lim) != 0ull)) break;
#endif
#pragma unroll
                        for (int rep = 0; rep < PT_NODE_REPS; ++rep) {
                            if (rep >= n_reps) break;
                            if (cur >= 0) {
                                if (COUNT) cn.nodes += quad ? 2 : 1; // a quad node is two binary nodes' worth of boxes (128 B)
                                if (quad) node4_step<PT_WAVE, 0x7fffffff, COUNT>(nodes4, stack, ovf, o, inv, h.t, cur, sp, box_exact, &cn.cull_nohit, &cn.cull_beyond);
                                else if (COUNT) node_step<PT_WAVE, 0x7fffffff>(nodes, stack, ovf, o, inv, h.t, cur, sp, cn.depth); // the one-level walk exists in the instrumented instance only
                                PT_STASH_LEAF(0x7fffffff);
                            }
                        }
#if PT_NODE_KEEP > 0
                      }
#endif
                    } else if (cur >= 0) { // some stack of the wave is about to leave LDS (rare): per-lane fetch, overflow-aware pushes
                        if (COUNT) cn.nodes += quad ? 2 : 1;
                        if (quad) node4_step<PT_WAVE, PT_LDS_STACK>(nodes4, stack, ovf, o, inv, h.t, cur, sp, box_exact);
                        else if (COUNT) node_step<PT_WAVE, PT_LDS_STACK>(nodes, stack, ovf, o, inv, h.t, cur, sp, cn.depth);
                        if (cur < PT_DONE && pend == PT_DONE) {
                            pend = cur;
                            if (sp > 0) {
                                --sp;
                                cur = (int)stack_pop<PT_WAVE, PT_LDS_STACK>(stack, ovf, sp);
                            } else {
                                cur = PT_DONE;
                            }
                        }
                    }
                    if (COUNT) { unsigned long long t2 = __builtin_amdgcn_s_memtime(); cn.cyc[0] += t2 - t1; t1 = t2; }
                } else {
                    if (COUNT) { cn.sched[2] += 1; cn.sched[3] += popc64(m_leaf); }
                    if (pend < PT_DONE) { // every triangle of the pending leaf in one step: the leaf lanes are cleared for good
                        const uint32_t code = ~(uint32_t)pend;
                        const int firstt = (int)(code >> 3), count = (int)(code & 7u);
                        if (COUNT) cn.tris += (uint32_t)count;
                        const float t_before = h.t;
                        leaf_test(tris, firstt, count, o, d, h);
                        if (COUNT) cn.leaf_noimp += h.t == t_before ? 1u : 0u;
                        if (cur < PT_DONE) { // the lane was blocked on a second leaf: it becomes the pending one
                            pend = cur;
                            if (sp > 0) {
                                --sp;
                                cur = (int)stack_pop<PT_WAVE, PT_LDS_STACK>(stack, ovf, sp);
                            } else {
                                cur = PT_DONE;
                            }
                        } else {
                            pend = PT_DONE;
                        }
                    }
                    if (COUNT) { unsigned long long t2 = __builtin_amdgcn_s_memtime(); cn.cyc[1] += t2 - t1; t1 = t2; }
                }
            }
            // ---- park unfinished traversals until the next traversal phase ----
            parked_groups = false;
            n_parked = popc64(__ballot(pslot >= 0));
            if (n_parked > 0) {
                park[K_PSLOT * PT_WAVE] = (uint32_t)pslot;
                if (pslot >= 0) {
                    park[K_CUR * PT_WAVE] = (uint32_t)cur;
                    park[K_SP * PT_WAVE] = (uint32_t)sp;
                    park[K_BT * PT_WAVE] = __float_as_uint(h.t);
                    park[K_BU * PT_WAVE] = __float_as_uint(h.u);
                    park[K_BV * PT_WAVE] = __float_as_uint(h.v);
                    park[K_BSLOT * PT_WAVE] = (uint32_t)h.slot;
                    park[K_BID * PT_WAVE] = (uint32_t)h.id;
                    park[K_PEND * PT_WAVE] = (uint32_t)pend;
                }
            }
        }
    }
#undef LF
#undef LFF
    if (COUNT) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        cn.cyc[7] = P.census_mode == 1 ? (t_dead ? t_end - t_dead : 0ull) : t_end - t_begin;
        cn.sched[23] = t_dead ? (uint32_t)(t_end - t_dead) : 0u; // wind-down time of this wave
    }
    flush_counters<COUNT>(P, cn);
}

// ---- launchers (called from pt_api.cpp) --------------------------------------------------------------------

// d_params: device copy of *p (wavefront kernel reads its parameters from HBM; the caller keeps it stream-ordered)
extern "C" hipError_t pt_launch_render(const PtKernelParams* p, const PtKernelParams* d_params, int variant, int grid, size_t lds_bytes,
                                       hipStream_t stream, int count)
{
    if (variant == 1) return pt_launch_render_lane(p, grid, lds_bytes, stream, count); // pt_kernel_aux.hip
    const bool exact = p->box_exact != 0;
    if (count) hipLaunchKernelGGL((pt_render_wave_kernel<true, PT_COUNT_WAVES_PER_EU, false>), dim3(grid), dim3(PT_WAVE), lds_bytes, stream, d_params);
    else if (variant == 3 && exact) hipLaunchKernelGGL((pt_render_wave_kernel<false, PT_FALLBACK_WAVES, true>), dim3(grid), dim3(PT_WAVE), lds_bytes, stream, d_params);
    else if (variant == 3) hipLaunchKernelGGL((pt_render_wave_kernel<false, PT_FALLBACK_WAVES, false>), dim3(grid), dim3(PT_WAVE), lds_bytes, stream, d_params);
    else if (exact) hipLaunchKernelGGL((pt_render_wave_kernel<false, PT_WAVES_PER_EU, true>), dim3(grid), dim3(PT_WAVE), lds_bytes, stream, d_params);
    else hipLaunchKernelGGL((pt_render_wave_kernel<false, PT_WAVES_PER_EU, false>), dim3(grid), dim3(PT_WAVE), lds_bytes, stream, d_params);
    return hipGetLastError();
}

// Launch geometry of a render variant (1: lane per pixel, 2: wavefront kernel, 3: the wavefront kernel's 168-VGPR fallback
// instance): block size, dynamic LDS bytes, pixels a block keeps in flight (ns is chosen here for the wavefront kernel), per-block
// global state words, registers, occupancy.  hipErrorInvalidConfiguration: the instance needs scratch (see below).
extern "C" hipError_t pt_kernel_geometry(int variant, int count, int stack_entries, int group_entries, int want_ns, int bins, int exact, int* block, size_t* lds_bytes, int* ns,
                                         size_t* state_words_per_block, int* vgprs, int* max_blocks_per_cu, int* lds_levels)
{
    if (variant == 1) { // pt_kernel_aux.hip
        *lds_levels = stack_entries;
        *state_words_per_block = 0;
        return pt_lane_kernel_geometry(count, stack_entries, block, lds_bytes, ns, vgprs, max_blocks_per_cu);
    }
    const void* fn = count ? (const void*)pt_render_wave_kernel<true, PT_COUNT_WAVES_PER_EU, false>
                     : variant == 3 ? (exact ? (const void*)pt_render_wave_kernel<false, PT_FALLBACK_WAVES, true> : (const void*)pt_render_wave_kernel<false, PT_FALLBACK_WAVES, false>)
                                    : (exact ? (const void*)pt_render_wave_kernel<false, PT_WAVES_PER_EU, true> : (const void*)pt_render_wave_kernel<false, PT_WAVES_PER_EU, false>);
    int n = want_ns < 16 ? 16 : (want_ns > 255 ? 255 : want_ns);
    if (bins) n = (n + 3) & ~3; // the lobe-code table follows the byte queues: keep it word aligned
    if (n > 252) n = 252;
    *block = PT_WAVE;
    *ns = n;
    *lds_bytes = pt_wave_lds_bytes(stack_entries, group_entries, n, bins);
    *lds_levels = pt_wave_lds_stack(stack_entries, group_entries);
    *state_words_per_block = pt_wave_state_words(stack_entries);
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, fn);
    if (e != hipSuccess) return e;
    // register spills: see pt_render_wave_kernel.  PT_ALLOW_SCRATCH=1 in the environment lifts the refusal for the experiment that
    // investigates it (profiles/r02_spill_investigation.md); never set it in production
    if (fa.localSizeBytes != 0 && !(getenv("PT_ALLOW_SCRATCH") && getenv("PT_ALLOW_SCRATCH")[0] == '1')) return hipErrorInvalidConfiguration;
    *vgprs = fa.numRegs;
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, *block, *lds_bytes);
    *max_blocks_per_cu = nb;
    return e;
}

// does this build contain the lobe bins of the hit pass (option "lobe_bins")?
extern "C" int pt_kernel_lobe_bins(void) { return PT_WITH_LOBE_BINS; }
