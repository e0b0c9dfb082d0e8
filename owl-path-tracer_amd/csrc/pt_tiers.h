// Tier plan of the whole-pixel schedule (pt_kernel.hip, TIERS): host + device code, so that the CPU tests can exercise exactly what the
// one-thread kernel pt_plan_tiers_kernel runs after the counting sort (pt_debug_plan_tiers in pt_api.cpp).
#pragma once
#include <cmath>
#include <cstdint>
#include "pt_types.h"

#if defined(__HIPCC__)
#define PT_TIERS_FN __host__ __device__ inline
#else
#define PT_TIERS_FN inline
#endif

#define PT_SORT_BUCKETS 32   // cost buckets of the counting sort (pt_kernel.hip, cost_bucket): four pre-pass classes each (19 % wide)
#define PT_COST_TOP 230      // ... from class PT_COST_TOP (20 ms for the pre-pass's samples of one pixel) down; bucket 0 = most expensive
#define PT_TIER_MIN_SPREAD 6 // buckets between the median pixel and the 98th percentile (2.8x in time) from which the tier plan is used

// relative sample-chain duration of a pixel in `bucket`: the middle of the bucket (a pre-pass class is 1/16 of a doubling)
PT_TIERS_FN float pt_bucket_time(int bucket) { return exp2f(((float)(PT_COST_TOP - 4 * bucket) - 1.5f) * 0.0625f); }

// Turnaround of a ray in a wave that holds n pixels, relative to a wave with 96: measured on a 1/8 shard of C4 with waves that hold
// nothing but expensive pixels (profiles/r03_logs/r3_ab37.log: 25 / 28 / 35 / 43 / 56 / 48 / 49 us at 4 / 8 / 12 / 16 / 24 / 32 / 48 pixels,
// 60-66 at 96).  Between 16 and 32 pixels a wave is neither: too many rays for the group walk (three phases in a row), too few for
// the per-lane walk - the plan never uses 17..31.  (Under full load the sparse end is less favourable - 33 / 37 / 42 / 48 us at 4 / 8 /
// 12 / 16 - but plans made with flatter curves were slower: r3_ab45.log, r3_ab46.log.)
#ifndef PT_TIER_CURVE
#define PT_TIER_CURVE 0.33f, 0.38f, 0.46f, 0.52f, 0.68f, 0.78f, 0.86f, 1.0f, 1.1f
#endif
PT_TIERS_FN float pt_tier_turnaround(int n)
{
    const float xs[9] = {4.f, 8.f, 12.f, 16.f, 32.f, 48.f, 64.f, 96.f, 128.f};
    const float ys[9] = {PT_TIER_CURVE};
    if (n <= 4) return ys[0];
    for (int i = 1; i < 9; ++i)
        if ((float)n <= xs[i]) return ys[i - 1] + (ys[i] - ys[i - 1]) * ((float)n - xs[i - 1]) / (xs[i] - xs[i - 1]);
    return ys[8];
}
// pixels per wave for a class whose chain takes `t` if the frame is to end at `T` (relative units): the most that still make it
PT_TIERS_FN int pt_tier_pixels(float t, float T, int ns)
{
    int best = 4;
    for (int n = 4; n <= ns; n += (n < 16 ? 2 : (n == 16 ? 16 : 8))) // 4 6 .. 16, 32 40 .. ns
        if (t * pt_tier_turnaround(n) <= T) best = n;
    if (t * pt_tier_turnaround(ns) <= T) best = ns;
    return best;
}
// Workgroups for the `cnt` pixels of a class whose chain takes `t`: `per` pixels per wave (pt_tier_pixels), and when a pixel of the
// class is done well before T its slot takes another one of the class (take_ticket) - as many rounds as fit.
PT_TIERS_FN uint32_t pt_tier_waves(uint32_t cnt, float t, float T, int ns, int* per_out)
{
    const int per = pt_tier_pixels(t, T, ns);
    const float one = t * pt_tier_turnaround(per);
    uint32_t rounds = one < T ? (uint32_t)(T / one) : 1u;
    if (rounds < 1u) rounds = 1u;
    if (rounds > 1024u) rounds = 1024u;
    if (per_out) *per_out = per;
    const uint32_t slots = (uint32_t)per * rounds;
    return (cnt + slots - 1u) / slots;
}
// start[b]: first queue entry of bucket b (start[PT_SORT_BUCKETS] = n, the pixels of the launch).  The frame time is the largest
// chain x turnaround over the buckets; bisect the smallest T whose plan fits the `capacity` resident waves and write the tier table
// (pt_types.h) - or leave it empty (tiers[0] = 0: the launch runs the ring schedule the host prepared alongside).
// The plan pays when the frame has a tail: a cheap majority and an expensive minority whose chains decide when it ends (a shard of the
// dragon: the 98th percentile pixel takes ~10x the median pixel's time).  When all pixels cost about the same (the Cornell box: 1.4x)
// homogeneous waves gain nothing over the ring schedule, which balances the waves' load chunk by chunk (C2: 71 vs 85 ms).  A launch
// with <= 16 pixels per resident wave always uses the plan (the ring schedule would run it in a few dense waves), as does `force`.
PT_TIERS_FN void pt_plan_tiers(const uint32_t* start, int capacity, int ns, int force, uint32_t* tiers)
{
    const uint32_t n = start[PT_SORT_BUCKETS];
    if (!force && (unsigned long long)n > 16ull * (unsigned long long)capacity) {
        int b98 = 0, b50 = 0;
        while (b98 < PT_SORT_BUCKETS - 1 && (unsigned long long)start[b98 + 1] * 50ull < (unsigned long long)n) ++b98;
        while (b50 < PT_SORT_BUCKETS - 1 && (unsigned long long)start[b50 + 1] * 2ull < (unsigned long long)n) ++b50;
        if (b50 - b98 < PT_TIER_MIN_SPREAD) { tiers[0] = 0u; return; }
    }
    // at hi every class runs ns pixels per wave, in as many rounds as the pixels need
    float lo = 0.0f, hi = pt_bucket_time(0) * pt_tier_turnaround(ns) * (2.0f + 2.0f * (float)n / ((float)capacity * (float)ns));
    for (int it = 0; it < 24; ++it) {
        const float T = 0.5f * (lo + hi);
        long waves = 0;
        for (int b = 0; b < PT_SORT_BUCKETS; ++b) {
            const uint32_t cnt = start[b + 1] - start[b];
            if (cnt != 0u) waves += (long)pt_tier_waves(cnt, pt_bucket_time(b), T, ns, nullptr);
        }
        if (waves <= (long)capacity) hi = T; else lo = T;
    }
    uint32_t n_tiers = 0, wave = 0;
    for (int b = 0; b < PT_SORT_BUCKETS; ++b) {
        const uint32_t cnt = start[b + 1] - start[b];
        if (cnt == 0u) continue;
        int per = 0;
        const uint32_t w = pt_tier_waves(cnt, pt_bucket_time(b), hi, ns, &per);
        uint32_t* e = tiers + 1 + PT_TIER_WORDS * n_tiers;
        e[0] = start[b]; e[1] = cnt; e[2] = (uint32_t)per; e[3] = wave; e[4] = w; e[5] = (uint32_t)b; e[6] = 0u; e[7] = 0u;
        wave += w;
        ++n_tiers;
    }
    tiers[0] = wave <= (uint32_t)capacity ? n_tiers : 0u; // (a plan that does not fit cannot happen from 64 resident waves on; if it does: ring schedule)
}
