// pt_nogpu_stubs.cpp -- the kernel launchers of pt_kernel.hip / pt_lbvh.hip as stubs that fail: ONLY for the host-side sanitizer build
// (make -C csrc asan: libmi355pt_asan.so).  That build compiles pt_api.cpp / pt_comm.cpp / pt_bvh.cpp with g++ -fsanitize=address,
// undefined so that the CPU tests and the garbled-input tests run the host paths of the library (scene flattening, BVH build and
// collapse, sharding, validation hooks) under ASan/UBSan; device code cannot be sanitized on this pool and a host-only context never
// reaches these functions (every render path refuses first: "no CPU fallback").  Not part of libmi355pt.so.
#include <hip/hip_runtime.h>

#include "pt_launch.h"

extern "C" {
hipError_t pt_launch_render(const PtKernelParams*, const PtKernelParams*, int, int, size_t, hipStream_t, int) { return hipErrorNotSupported; }
hipError_t pt_launch_debug(const PtKernelParams*, int, const float*, int, float*, int, long long, size_t, hipStream_t) { return hipErrorNotSupported; }
size_t pt_sort_scratch_bytes(uint32_t) { return 16; }
hipError_t pt_launch_plan_tiers(const uint32_t*, uint32_t, int, int, int, uint32_t*, hipStream_t) { return hipErrorNotSupported; }
hipError_t pt_launch_sort_pixels(const uint8_t*, int, int, int, const uint32_t*, uint32_t*, uint32_t, uint32_t, uint32_t*, uint8_t*, hipStream_t) { return hipErrorNotSupported; }
hipError_t pt_kernel_geometry(int, int, int, int, int, int, int, int*, size_t*, int*, size_t*, int*, int*, int*) { return hipErrorNotSupported; }
int pt_debug_block(void) { return 256; }
int pt_kernel_lobe_bins(void) { return 0; }
size_t pt_lbvh_workspace_bytes(int) { return 16; }
hipError_t pt_lbvh_build_device(const float*, int, int, void*, size_t, PtNode*, uint32_t*, int32_t*, int32_t*, int32_t*, int32_t*, float*, hipStream_t) { return hipErrorNotSupported; }
size_t pt_ploc_workspace_bytes(int) { return 16; }
hipError_t pt_ploc_build_device(const float*, int, int, void*, size_t, int*, float*, int*, uint32_t*, int32_t*, int32_t*, hipStream_t) { return hipErrorNotSupported; }
hipError_t pt_launch_store_params(const PtKernelParams*, PtKernelParams*, hipStream_t) { return hipErrorNotSupported; }
hipError_t pt_launch_pack_tri_ids(PtTri*, long long, hipStream_t) { return hipErrorNotSupported; }
hipError_t pt_launch_pack_rgba8(const float*, uint32_t*, long long, hipStream_t) { return hipErrorNotSupported; }
}
