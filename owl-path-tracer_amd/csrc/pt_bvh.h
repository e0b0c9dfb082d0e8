// pt_bvh.h -- host BVH2 builder (replaces owlGroupBuildAccel, path_tracer/src/application.cpp:131-140; OptiX's
// builder is closed source, this one is our own design: binned SAH, child boxes stored in the parent).
#pragma once
#include <stdint.h>

#include <functional>
#include <vector>

#include "pt_types.h"

struct PtBvh {
    std::vector<PtNode> nodes;
    std::vector<PtTri> tris; // leaf order
    int32_t root = -1;       // child reference of the root (PT_DONE-like -1 for an empty scene)
    int depth = 0;           // deepest chain of internal nodes (= stack entries the traversal can need)
    int max_leaf = 0;
    float pad = 0.0f;
};

// positions: n_tris*9 floats in global triangle order.  leaf_size 1..7, max_depth <= PT_MAX_STACK.
void pt_bvh_build(const float* positions, int32_t n_tris, int leaf_size, int max_depth, PtBvh* out);

// Memory layout of the built tree (pt_bvh.cpp): sibling_pairs != 0 re-indexes the nodes so that sibling records share a
// 128-byte line; leaf_align > 1 starts every leaf at a multiple of that many triangle slots (padding slots: id 0x7fffffff).
// After it tris.size() may exceed the triangle count.
void pt_bvh_layout(PtBvh* bvh, int sibling_pairs, int leaf_align);

// Two-level collapse of the binary tree into quad nodes (PtNode4, pt_types.h).  root4 = 0 when the root is an internal node (else the
// binary root reference: leaf code or -1), depth4 = deepest chain of quad nodes (the traversal stack needs 3 * depth4 entries).
void pt_bvh_collapse4(const PtBvh& bvh, std::vector<PtNode4>* out, int32_t* root4, int* depth4);

// fn(begin, end) over [0, n), one contiguous share per build thread (small n: the caller's thread alone)
void pt_parallel_ranges(size_t n, const std::function<void(size_t, size_t)>& fn);

// Expected visits of a long random ray through the root box: quad nodes / leaf slots (surface-area metric; diagnostics).
void pt_bvh_quad_cost(const std::vector<PtNode4>& nodes4, int32_t root4, double* node_visits, double* leaf_visits);

// Host mirror of the kernel's traversal over the product BVH (validation only, never on the render path).
bool pt_bvh_closest_hit_host(const PtBvh& bvh, const float org[3], const float dir[3], float tmin, float tmax, float* t, float* u,
                             float* v, int32_t* prim);

// Three-level collapse into oct nodes (PtNode8, pt_types.h): starting from the two children of a binary node, the internal slot with
// the largest surface area is replaced by its two children until eight slots are used (or none is internal).  root8 = 0 when the
// root is an internal node (else the binary root reference), depth8 = deepest chain of oct nodes (the group walk pushes at most
// seven entries per level).  wide_leaves != 0: a subtree of <= 7 triangles that are contiguous in leaf order becomes one leaf
// (lane k of a group tests triangle k).
void pt_bvh_collapse8(const PtBvh& bvh, int wide_leaves, std::vector<PtNode8>* out, int32_t* root8, int* depth8);

// A binary hierarchy over n triangles built elsewhere (the device PLOC builder, pt_lbvh.hip) turned into the layout of pt_types.h:
// node i < n is the triangle order[i]; node i >= n has child[2i], child[2i + 1]; box[6i..] = {lo xyz, hi xyz}; count[i] = triangles
// below.  Subtrees of at most leaf_size triangles become leaves, triangles go into depth-first order, boxes get the padding of
// pt_bvh_build.  Returns false (out untouched in a usable way) if the tree is deeper than max_depth: the caller then builds on the host.
bool pt_bvh_from_hierarchy(const float* positions, int32_t n_tris, const int32_t* child, const float* box, const int32_t* count, const uint32_t* order, int32_t root,
                           int leaf_size, int max_depth, PtBvh* out);
