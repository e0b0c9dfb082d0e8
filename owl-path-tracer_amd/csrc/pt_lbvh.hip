// pt_lbvh.hip -- BVH2 build on the device (replaces owlGroupBuildAccel, path_tracer/src/application.cpp:131-140, whose
// OptiX builder also runs on the GPU).  Option "bvh_builder" = 1; the default stays the host binned-SAH builder (pt_bvh.cpp), whose
// trees traverse faster -- this one is for scenes where host build time matters.
//
// Linear BVH after Karras 2012 ("Maximizing parallelism in the construction of BVHs, octrees and k-d trees"):
//   1. bounds of the triangle centroids (block reduction + ordered-int atomics)
//   2. 30-bit Morton code of every centroid, made unique by the triangle index: key = code << 32 | index; radix sort of the keys
//      (hipcub::DeviceRadixSort - a plain library sort; the kernels below are ours)
//   3. one thread per internal node finds its key range and split from common-prefix lengths (binary radix tree over n leaves)
//   4. bottom-up pass, one thread per leaf: node boxes and heights; the second thread to reach a node continues upwards
//   5. every subtree of at most leaf_size triangles becomes ONE leaf (its triangles are contiguous in sorted order), the
//      surviving internal nodes are compacted (exclusive scan) and written in the layout of pt_types.h: each node stores the
//      padded boxes of its two children
// The result has the same format, padding rule and leaf encoding as pt_bvh_build, so the render kernel does not know which
// builder ran, and - closest hit being independent of the tree - images are bit-identical either way.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdint>
#include <cstring>

#include "pt_launch.h"
#include "pt_types.h"

namespace {

__device__ __forceinline__ uint32_t f2ord(float f)
{ // order-preserving map float -> uint32
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

struct LbvhMeta {
    uint32_t cmin[3], cmax[3]; // centroid bounds (ordered ints)
    uint32_t vmin[3], vmax[3]; // vertex bounds (ordered ints): scene extent for the padding rule
    int32_t root;              // child reference of the root in the FINAL numbering
    int32_t n_nodes;           // surviving internal nodes
    int32_t height;            // deepest chain of surviving internal nodes
    int32_t max_leaf;
};

__global__ void __launch_bounds__(256) k_init_meta(LbvhMeta* m)
{
    if (threadIdx.x < 3) {
        m->cmin[threadIdx.x] = m->vmin[threadIdx.x] = 0xffffffffu;
        m->cmax[threadIdx.x] = m->vmax[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) { m->root = -1; m->n_nodes = 0; m->height = 0; m->max_leaf = 0; }
}

__global__ void __launch_bounds__(256) k_bounds(const float* __restrict__ pos, int n, LbvhMeta* m)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float c[3] = {0, 0, 0}, lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < n) {
        const float* p = pos + (size_t)i * 9;
        for (int a = 0; a < 3; ++a) {
            c[a] = (p[a] + p[3 + a] + p[6 + a]) * (1.0f / 3.0f); // same expression as pt_bvh.cpp
            lo[a] = fminf(p[a], fminf(p[3 + a], p[6 + a]));
            hi[a] = fmaxf(p[a], fmaxf(p[3 + a], p[6 + a]));
        }
    }
    __shared__ uint32_t s[12];
    if (threadIdx.x < 12) s[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xffffffffu : 0u;
    __syncthreads();
    if (i < n) {
        for (int a = 0; a < 3; ++a) {
            atomicMin(&s[a], f2ord(c[a]));
            atomicMax(&s[3 + a], f2ord(c[a]));
            atomicMin(&s[6 + a], f2ord(lo[a]));
            atomicMax(&s[9 + a], f2ord(hi[a]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        atomicMin(&m->cmin[threadIdx.x], s[threadIdx.x]);
        atomicMax(&m->cmax[threadIdx.x], s[3 + threadIdx.x]);
        atomicMin(&m->vmin[threadIdx.x], s[6 + threadIdx.x]);
        atomicMax(&m->vmax[threadIdx.x], s[9 + threadIdx.x]);
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{ // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void __launch_bounds__(256) k_morton(const float* __restrict__ pos, int n, const LbvhMeta* __restrict__ m, unsigned long long* __restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = pos + (size_t)i * 9;
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = ord2f(m->cmin[a]), hi = ord2f(m->cmax[a]);
        const float c = (p[a] + p[3 + a] + p[6 + a]) * (1.0f / 3.0f);
        const float ext = hi - lo;
        float t = ext > 0.0f ? (c - lo) / ext : 0.0f;
        t = fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f);
        q[a] = (uint32_t)t;
    }
    const uint32_t code = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
    keys[i] = ((unsigned long long)code << 32) | (uint32_t)i;
}

// length of the common prefix of keys i and j (keys are unique); -1 outside the array
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// Karras 2012, section 4: internal node i of the binary radix tree.  Children >= 0: internal node index; < 0: ~leaf index.
__global__ void __launch_bounds__(256) k_hierarchy(const unsigned long long* __restrict__ keys, int n, int2* __restrict__ child, int2* __restrict__ range,
                                                  int* __restrict__ parent_int, int* __restrict__ parent_leaf)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    const int left = (first == gamma) ? ~gamma : gamma;
    const int right = (last == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    child[i] = make_int2(left, right);
    range[i] = make_int2(first, last);
    if (left >= 0) parent_int[left] = i; else parent_leaf[~left] = i;
    if (right >= 0) parent_int[right] = i; else parent_leaf[~right] = i;
    if (i == 0) parent_int[0] = -1;
}

struct Box6 { float lo[3], hi[3]; };

__device__ __forceinline__ Box6 tri_box(const float* __restrict__ pos, uint32_t tri)
{
    const float* p = pos + (size_t)tri * 9;
    Box6 b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fminf(p[a], fminf(p[3 + a], p[6 + a]));
        b.hi[a] = fmaxf(p[a], fmaxf(p[3 + a], p[6 + a]));
    }
    return b;
}

// a box another thread wrote earlier in this kernel (after the fences of k_fit): read it from memory, not from a register copy
__device__ __forceinline__ Box6 load_box(const Box6* p)
{
    const volatile float* f = (const volatile float*)p;
    Box6 b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = f[a]; b.hi[a] = f[3 + a]; }
    return b;
}

// Bottom-up: one thread per sorted leaf.  The first thread to arrive at a node leaves; the second one has both children's
// boxes available (the first published its box before the counter increment) and goes on.
__global__ void __launch_bounds__(256) k_fit(const float* __restrict__ pos, const unsigned long long* __restrict__ keys, int n, const int2* __restrict__ child,
                                            const int2* __restrict__ range, const int* __restrict__ parent_int, const int* __restrict__ parent_leaf, int leaf_size,
                                            Box6* node_box, int* node_height, unsigned int* visits)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int node = parent_leaf[i];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&visits[node], 1u) == 0u) return; // sibling subtree not done yet
        __threadfence();
        const int2 ch = child[node];
        Box6 b, c;
        int hl = 0, hr = 0;
        if (ch.x >= 0) { b = load_box(&node_box[ch.x]); hl = *(volatile int*)&node_height[ch.x]; }
        else b = tri_box(pos, (uint32_t)keys[~ch.x]);
        if (ch.y >= 0) { c = load_box(&node_box[ch.y]); hr = *(volatile int*)&node_height[ch.y]; }
        else c = tri_box(pos, (uint32_t)keys[~ch.y]);
        for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(b.lo[a], c.lo[a]); b.hi[a] = fmaxf(b.hi[a], c.hi[a]); }
        node_box[node] = b;
        // height counts only nodes that survive the leaf collapse (range of more than leaf_size triangles)
        const int2 r = range[node];
        node_height[node] = (r.y - r.x + 1 > leaf_size) ? 1 + max(hl, hr) : 0;
        node = parent_int[node];
    }
}

__global__ void __launch_bounds__(256) k_keep(const int2* __restrict__ range, int n, int leaf_size, uint32_t* __restrict__ keep)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int2 r = range[i];
    keep[i] = (r.y - r.x + 1 > leaf_size) ? 1u : 0u;
}

__device__ __forceinline__ int32_t child_ref(int c, const int2* __restrict__ range, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ new_index,
                                             int* max_leaf)
{
    if (c < 0) { atomicMax(max_leaf, 1); return (int32_t)~((((uint32_t)~c) << 3) | 1u); } // a single triangle
    if (keep[c]) return (int32_t)new_index[c];
    const int2 r = range[c];
    const uint32_t cnt = (uint32_t)(r.y - r.x + 1); // collapsed subtree: its triangles are contiguous in sorted order
    atomicMax(max_leaf, (int)cnt);
    return (int32_t)~((((uint32_t)r.x) << 3) | cnt);
}

__device__ __forceinline__ Box6 child_box(int c, const float* __restrict__ pos, const unsigned long long* __restrict__ keys, const Box6* __restrict__ node_box)
{
    return c < 0 ? tri_box(pos, (uint32_t)keys[~c]) : node_box[c];
}

__global__ void __launch_bounds__(256) k_emit(const float* __restrict__ pos, const unsigned long long* __restrict__ keys, int n, const int2* __restrict__ child,
                                             const int2* __restrict__ range, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ new_index,
                                             const Box6* __restrict__ node_box, const int* __restrict__ node_height, float pad, PtNode* __restrict__ out,
                                             LbvhMeta* m)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1 || !keep[i]) return;
    const int2 ch = child[i];
    const Box6 l = child_box(ch.x, pos, keys, node_box), r = child_box(ch.y, pos, keys, node_box);
    PtNode nd;
    for (int a = 0; a < 3; ++a) {
        nd.lo[a][0] = l.lo[a] - pad; nd.hi[a][0] = l.hi[a] + pad;
        nd.lo[a][1] = r.lo[a] - pad; nd.hi[a][1] = r.hi[a] + pad;
    }
    nd.left = child_ref(ch.x, range, keep, new_index, &m->max_leaf);
    nd.right = child_ref(ch.y, range, keep, new_index, &m->max_leaf);
    nd.pad[0] = nd.pad[1] = 0;
    out[new_index[i]] = nd;
    if (i == 0) {
        m->root = (int32_t)new_index[0];
        m->height = node_height[0];
    }
}

__global__ void __launch_bounds__(256) k_order(const unsigned long long* __restrict__ keys, int n, uint32_t* __restrict__ order)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) order[i] = (uint32_t)keys[i];
}

} // namespace

// Workspace the caller provides (all device memory).  Sizes from pt_lbvh_workspace_bytes.
struct PtLbvhWorkspace {
    unsigned long long *keys, *keys_sorted;
    int2 *child, *range;
    int *parent_int, *parent_leaf, *node_height;
    Box6* node_box;
    unsigned int* visits;
    uint32_t *keep, *new_index;
    LbvhMeta* meta;
    void* cub_temp;
    size_t cub_temp_bytes;
};

extern "C" size_t pt_lbvh_workspace_bytes(int n)
{
    size_t sort_tmp = 0, scan_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, n, 0, 62);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, n);
    const size_t tmp = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
    const size_t N = (size_t)n;
    // keys x2, child, range, parent_int, parent_leaf, node_height, node_box, visits, keep, new_index, meta, cub temp; 256-byte slack per array
    return 2 * N * 8 + 2 * N * 8 + 3 * N * 4 + N * sizeof(Box6) + 3 * N * 4 + sizeof(LbvhMeta) + tmp + 16 * 256;
}

// d_pos: n * 9 floats (global triangle order).  Outputs: d_nodes (capacity n - 1 PtNode), d_order (n: sorted position -> global
// triangle index), *h_meta (root, node count, height, largest leaf; read back here - the only synchronisation of the build).
extern "C" hipError_t pt_lbvh_build_device(const float* d_pos, int n, int leaf_size, void* d_workspace, size_t workspace_bytes, PtNode* d_nodes, uint32_t* d_order,
                                           int32_t* h_root, int32_t* h_n_nodes, int32_t* h_height, int32_t* h_max_leaf, float* h_pad, hipStream_t stream)
{
    if (n < 2 || workspace_bytes < pt_lbvh_workspace_bytes(n)) return hipErrorInvalidValue;
    char* w = (char*)d_workspace;
    auto take = [&](size_t bytes) { char* p = w; w += (bytes + 255) & ~(size_t)255; return (void*)p; };
    PtLbvhWorkspace ws;
    const size_t N = (size_t)n;
    ws.keys = (unsigned long long*)take(N * 8);
    ws.keys_sorted = (unsigned long long*)take(N * 8);
    ws.child = (int2*)take(N * 8);
    ws.range = (int2*)take(N * 8);
    ws.parent_int = (int*)take(N * 4);
    ws.parent_leaf = (int*)take(N * 4);
    ws.node_height = (int*)take(N * 4);
    ws.node_box = (Box6*)take(N * sizeof(Box6));
    ws.visits = (unsigned int*)take(N * 4);
    ws.keep = (uint32_t*)take(N * 4);
    ws.new_index = (uint32_t*)take(N * 4);
    ws.meta = (LbvhMeta*)take(sizeof(LbvhMeta));
    ws.cub_temp = (void*)w;
    ws.cub_temp_bytes = workspace_bytes - (size_t)(w - (char*)d_workspace);

    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_init_meta, dim3(1), dim3(256), 0, stream, ws.meta);
    hipLaunchKernelGGL(k_bounds, dim3(nb), dim3(256), 0, stream, d_pos, n, ws.meta);
    hipLaunchKernelGGL(k_morton, dim3(nb), dim3(256), 0, stream, d_pos, n, (const LbvhMeta*)ws.meta, ws.keys);
    size_t tmp = ws.cub_temp_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(ws.cub_temp, tmp, (const unsigned long long*)ws.keys, ws.keys_sorted, n, 0, 62, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_order, dim3(nb), dim3(256), 0, stream, (const unsigned long long*)ws.keys_sorted, n, d_order);
    hipLaunchKernelGGL(k_hierarchy, dim3(nb), dim3(256), 0, stream, (const unsigned long long*)ws.keys_sorted, n, ws.child, ws.range, ws.parent_int, ws.parent_leaf);
    e = hipMemsetAsync(ws.visits, 0, N * 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fit, dim3(nb), dim3(256), 0, stream, d_pos, (const unsigned long long*)ws.keys_sorted, n, (const int2*)ws.child, (const int2*)ws.range,
                       (const int*)ws.parent_int, (const int*)ws.parent_leaf, leaf_size, ws.node_box, ws.node_height, ws.visits);
    hipLaunchKernelGGL(k_keep, dim3(nb), dim3(256), 0, stream, (const int2*)ws.range, n, leaf_size, ws.keep);
    tmp = ws.cub_temp_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(ws.cub_temp, tmp, (const uint32_t*)ws.keep, ws.new_index, n - 1, stream);
    if (e != hipSuccess) return e;
    // scene extent -> padding, same rule as pt_bvh_build: 1e-5 x max(range, |min|, |max|) over the axes
    LbvhMeta hm;
    e = hipMemcpyAsync(&hm, ws.meta, sizeof(hm), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) {
        uint32_t ulo = hm.vmin[a], uhi = hm.vmax[a];
        float lo, hi;
        { uint32_t u = (ulo & 0x80000000u) ? (ulo & 0x7fffffffu) : ~ulo; memcpy(&lo, &u, 4); }
        { uint32_t u = (uhi & 0x80000000u) ? (uhi & 0x7fffffffu) : ~uhi; memcpy(&hi, &u, 4); }
        ext = fmaxf(ext, hi - lo);
        ext = fmaxf(ext, fmaxf(fabsf(lo), fabsf(hi)));
    }
    const float pad = ext * 1e-5f;
    // number of surviving nodes = exclusive sum at the last element + its keep flag
    uint32_t last_idx = 0, last_keep = 0;
    (void)hipMemcpyAsync(&last_idx, ws.new_index + (n - 2), 4, hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(&last_keep, ws.keep + (n - 2), 4, hipMemcpyDeviceToHost, stream);
    hipLaunchKernelGGL(k_emit, dim3(nb), dim3(256), 0, stream, d_pos, (const unsigned long long*)ws.keys_sorted, n, (const int2*)ws.child, (const int2*)ws.range,
                       (const uint32_t*)ws.keep, (const uint32_t*)ws.new_index, (const Box6*)ws.node_box, (const int*)ws.node_height, pad, d_nodes, ws.meta);
    e = hipMemcpyAsync(&hm, ws.meta, sizeof(hm), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    *h_root = hm.root;
    *h_n_nodes = (int32_t)(last_idx + last_keep);
    *h_height = hm.height;
    *h_max_leaf = hm.max_leaf;
    *h_pad = pad;
    return hipGetLastError();
}


// =====================================================================================================================
// PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) - option "bvh_builder" = 2 (round 3)
// =====================================================================================================================
// The Karras tree above splits at Morton-code bits; it is fast to build and 1.4x slower to walk than the host SAH tree
// (profiles/r02_summary.md).  PLOC builds the tree bottom-up instead: the clusters (at first the triangles) stay in Morton
// order; every cluster looks for the neighbour within `radius` positions whose union with it has the smallest surface area;
// mutual nearest neighbours merge into a new node; the clusters are compacted; repeat until one is left.  Each round is three
// small kernels and a scan; ~log n rounds.  The merge criterion is the surface-area heuristic itself, applied locally.
// The device part delivers the hierarchy (children, boxes, triangle counts per node); the host turns it into the layout of
// pt_types.h (pt_bvh_from_hierarchy: leaf collapse, triangles in depth-first order) - a linear pass.
namespace {

__device__ __forceinline__ float union_area(const Box6& a, const Box6& b)
{
    const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
    const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
    const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ void __launch_bounds__(256) k_ploc_init(const float* __restrict__ pos, const unsigned long long* __restrict__ keys, int n, Box6* __restrict__ box,
                                                  int2* __restrict__ child, int* __restrict__ count, int* __restrict__ cid)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    box[i] = tri_box(pos, (uint32_t)keys[i]);
    child[i] = make_int2(-1, -1);
    count[i] = 1;
    cid[i] = i;
}

__global__ void __launch_bounds__(256) k_ploc_nn(const int* __restrict__ cid, int m, int radius, const Box6* __restrict__ box, int* __restrict__ nn)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const Box6 bi = box[cid[i]];
    float best = INFINITY;
    int bj = -1;
    const int lo = max(0, i - radius), hi = min(m - 1, i + radius);
    for (int j = lo; j <= hi; ++j) {
        if (j == i) continue;
        const float a = union_area(bi, box[cid[j]]);
        if (a < best) { best = a; bj = j; } // ties: the smaller position (deterministic topology)
    }
    nn[i] = bj;
}

__global__ void __launch_bounds__(256) k_ploc_merge(const int* __restrict__ cid, int m, const int* __restrict__ nn, Box6* __restrict__ box, int2* __restrict__ child,
                                                   int* __restrict__ count, int n, int* __restrict__ counter, int* __restrict__ out, uint32_t* __restrict__ valid)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int j = nn[i];
    if (j >= 0 && nn[j] == i) {
        if (i < j) {
            const int a = cid[i], b = cid[j];
            const int p = n + atomicAdd(counter, 1);
            const Box6 ba = box[a], bb = box[b];
            Box6 u;
            for (int k = 0; k < 3; ++k) { u.lo[k] = fminf(ba.lo[k], bb.lo[k]); u.hi[k] = fmaxf(ba.hi[k], bb.hi[k]); }
            box[p] = u;
            child[p] = make_int2(a, b);
            count[p] = count[a] + count[b];
            out[i] = p;
            valid[i] = 1u;
        } else {
            valid[i] = 0u;
        }
    } else {
        out[i] = cid[i];
        valid[i] = 1u;
    }
}

__global__ void __launch_bounds__(256) k_ploc_compact(const int* __restrict__ out, const uint32_t* __restrict__ valid, const uint32_t* __restrict__ offs, int m,
                                                     int* __restrict__ cid_next)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m && valid[i]) cid_next[offs[i]] = out[i];
}

} // namespace

extern "C" size_t pt_ploc_workspace_bytes(int n)
{
    size_t sort_tmp = 0, scan_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, n, 0, 62);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, n);
    const size_t tmp = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
    const size_t N = (size_t)n;
    // keys x2, cid x2, nn, out, valid, offs, box[2n], child[2n], count[2n], meta + counter, cub temp; 256-byte slack per array
    return 2 * N * 8 + 6 * N * 4 + 2 * N * (sizeof(Box6) + 8 + 4) + sizeof(LbvhMeta) + 256 + tmp + 16 * 256;
}

// d_pos: n * 9 floats.  Host outputs (capacity 2n - 1 each unless stated): h_child (both -1 for node i < n: triangle h_order[i]), h_box (6 floats per
// node), h_count (triangles below), h_order (n), *h_root, *h_rounds.
extern "C" hipError_t pt_ploc_build_device(const float* d_pos, int n, int radius, void* d_workspace, size_t workspace_bytes, int* h_child, float* h_box, int* h_count,
                                           uint32_t* h_order, int32_t* h_root, int32_t* h_rounds, hipStream_t stream)
{
    if (n < 2 || radius < 1 || workspace_bytes < pt_ploc_workspace_bytes(n)) return hipErrorInvalidValue;
    char* w = (char*)d_workspace;
    auto take = [&](size_t bytes) { char* p = w; w += (bytes + 255) & ~(size_t)255; return (void*)p; };
    const size_t N = (size_t)n;
    unsigned long long* keys = (unsigned long long*)take(N * 8);
    unsigned long long* keys_sorted = (unsigned long long*)take(N * 8);
    int* cid_a = (int*)take(N * 4);
    int* cid_b = (int*)take(N * 4);
    int* nn = (int*)take(N * 4);
    int* out = (int*)take(N * 4);
    uint32_t* valid = (uint32_t*)take(N * 4);
    uint32_t* offs = (uint32_t*)take(N * 4);
    Box6* box = (Box6*)take(2 * N * sizeof(Box6));
    int2* child = (int2*)take(2 * N * 8);
    int* count = (int*)take(2 * N * 4);
    LbvhMeta* meta = (LbvhMeta*)take(sizeof(LbvhMeta));
    int* counter = (int*)take(256);
    void* cub_temp = (void*)w;
    const size_t cub_temp_bytes = workspace_bytes - (size_t)(w - (char*)d_workspace);

    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_init_meta, dim3(1), dim3(256), 0, stream, meta);
    hipLaunchKernelGGL(k_bounds, dim3(nb), dim3(256), 0, stream, d_pos, n, meta);
    hipLaunchKernelGGL(k_morton, dim3(nb), dim3(256), 0, stream, d_pos, n, (const LbvhMeta*)meta, keys);
    size_t tmp = cub_temp_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(cub_temp, tmp, (const unsigned long long*)keys, keys_sorted, n, 0, 62, stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(counter, 0, 4, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_ploc_init, dim3(nb), dim3(256), 0, stream, d_pos, (const unsigned long long*)keys_sorted, n, box, child, count, cid_a);
    int m = n, rounds = 0;
    int* cid = cid_a;
    int* cid_next = cid_b;
    while (m > 1) {
        const int mb = (m + 255) / 256;
        hipLaunchKernelGGL(k_ploc_nn, dim3(mb), dim3(256), 0, stream, (const int*)cid, m, radius, (const Box6*)box, nn);
        hipLaunchKernelGGL(k_ploc_merge, dim3(mb), dim3(256), 0, stream, (const int*)cid, m, (const int*)nn, box, child, count, n, counter, out, valid);
        tmp = cub_temp_bytes;
        e = hipcub::DeviceScan::ExclusiveSum(cub_temp, tmp, (const uint32_t*)valid, offs, m, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_ploc_compact, dim3(mb), dim3(256), 0, stream, (const int*)out, (const uint32_t*)valid, (const uint32_t*)offs, m, cid_next);
        uint32_t last_off = 0, last_valid = 0;
        (void)hipMemcpyAsync(&last_off, offs + (m - 1), 4, hipMemcpyDeviceToHost, stream);
        (void)hipMemcpyAsync(&last_valid, valid + (m - 1), 4, hipMemcpyDeviceToHost, stream);
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        const int m_new = (int)(last_off + last_valid);
        if (m_new >= m || m_new < 1 || ++rounds > 4096) return hipErrorUnknown; // the smallest union is always a mutual pair: every round merges
        m = m_new;
        int* t = cid; cid = cid_next; cid_next = t;
    }
    int root = 0;
    (void)hipMemcpyAsync(&root, cid, 4, hipMemcpyDeviceToHost, stream);
    hipLaunchKernelGGL(k_order, dim3(nb), dim3(256), 0, stream, (const unsigned long long*)keys_sorted, n, (uint32_t*)nn); // nn is free now
    (void)hipMemcpyAsync(h_order, nn, N * 4, hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(h_child, child, (2 * N - 1) * 8, hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(h_box, box, (2 * N - 1) * sizeof(Box6), hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(h_count, count, (2 * N - 1) * 4, hipMemcpyDeviceToHost, stream);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    *h_root = root;
    *h_rounds = rounds;
    return hipGetLastError();
}
