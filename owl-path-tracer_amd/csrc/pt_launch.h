// pt_launch.h -- prototypes of the kernel launchers (defined in pt_kernel.hip / pt_lbvh.hip, stubbed in pt_nogpu_stubs.cpp for the
// host-side sanitizer build) as pt_api.cpp / pt_comm.cpp call them.  ONE declaration for definition, stub and caller: the functions
// have C linkage, so a mismatched parameter list would link and then misbehave (round-3 advisor finding).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "pt_types.h"

extern "C" {
hipError_t pt_launch_render(const PtKernelParams* p, const PtKernelParams* d_params, int variant, int grid, size_t lds_bytes, hipStream_t stream, int count);
hipError_t pt_launch_debug(const PtKernelParams* p, int op, const float* in, int in_stride, float* out, int out_stride, long long n, size_t lds_bytes,
                           hipStream_t stream);
size_t pt_sort_scratch_bytes(uint32_t n);
hipError_t pt_launch_plan_tiers(const uint32_t* scratch, uint32_t n, int capacity, int ns, int force, uint32_t* tiers, hipStream_t stream);
hipError_t pt_launch_sort_pixels(const uint8_t* cost_img, int W, int H, int radius, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t c0, uint32_t* scratch,
                                 uint8_t* bucket, hipStream_t stream);
hipError_t pt_kernel_geometry(int variant, int count, int stack_entries, int group_entries, int want_ns, int bins, int exact, int* block, size_t* lds_bytes, int* ns,
                              size_t* state_words_per_block, int* vgprs, int* max_blocks_per_cu, int* lds_levels);
int pt_debug_block(void);
int pt_kernel_lobe_bins(void);
// pt_kernel_aux.hip: the lane-per-pixel variant (pt_launch_render / pt_kernel_geometry forward variant 1 to these)
hipError_t pt_launch_render_lane(const PtKernelParams* p, int grid, size_t lds_bytes, hipStream_t stream, int count);
hipError_t pt_lane_kernel_geometry(int count, int stack_entries, int* block, size_t* lds_bytes, int* ns, int* vgprs, int* max_blocks_per_cu);
hipError_t pt_launch_store_params(const PtKernelParams* p, PtKernelParams* d_dst, hipStream_t stream);
hipError_t pt_launch_pack_tri_ids(PtTri* tris, long long n, hipStream_t stream);
hipError_t pt_launch_pack_rgba8(const float* rgb, uint32_t* out, long long n, hipStream_t stream);
// pt_lbvh.hip
size_t pt_lbvh_workspace_bytes(int n);
hipError_t pt_lbvh_build_device(const float* d_pos, int n, int leaf_size, void* d_workspace, size_t workspace_bytes, PtNode* d_nodes, uint32_t* d_order, int32_t* h_root,
                                int32_t* h_n_nodes, int32_t* h_height, int32_t* h_max_leaf, float* h_pad, hipStream_t stream);
size_t pt_ploc_workspace_bytes(int n);
hipError_t pt_ploc_build_device(const float* d_pos, int n, int radius, void* d_workspace, size_t workspace_bytes, int* h_child, float* h_box, int* h_count,
                                uint32_t* h_order, int32_t* h_root, int32_t* h_rounds, hipStream_t stream);
}
