// Launchers of the instantiated path-tracing kernels.  Every instantiation lives in one of the translation units
// under variants/ (explicit specialisations of the templates below, written with the macros of variants/define.h).
//
// Which instantiations exist:
//   scenes whose tables fit in LDS (CACHE) get kernels specialised on what the scene contains:
//     spheres, planes, rectangles only     k_bounce<0, UV, 0, PASS, 1>   k_tail<0, UV, PASS, 1>
//     + single triangles                   k_bounce<1, UV, 0, 1, 1>      k_tail<1, UV, 1, 1>
//     + meshes                             k_bounce<1, UV, 1, PASS, 1>   k_tail<1, UV, 1, 1>   k_bounce_resume<UV, PASS, 1>
//   every other scene runs the general kernels (all features compiled in, tables read from memory):
//                                          k_bounce<1, 1, DEFER, 1, 0>   k_tail<1, 1, 1, 0>    k_bounce_resume<1, 1, 0>
#pragma once
#include <hip/hip_runtime.h>

#include "scene_device.h"

namespace ptk {

template<bool TRIS, bool UV, bool DEFER, bool PASS, bool CACHE>
void launch_bounce_variant(hipStream_t s, dim3 grid, const DevScene& sc, const BatchParams& b, const PathPool& in, const PathPool& out,
                           const SampleResults& res, const uint32_t* list, const BinInfo* info, uint32_t rounds, uint32_t* block_counts,
                           uint32_t count_stride, unsigned long long* statistics, const DeferQueues& queues);

template<bool TRIS, bool UV, bool PASS, bool CACHE>
void launch_tail_variant(hipStream_t s, dim3 grid, const DevScene& sc, const BatchParams& b, const PathPool& in, const PathPool& out,
                         const SampleResults& res, const uint32_t* list, BinInfo* info, unsigned long long* statistics);

template<bool UV, bool PASS, bool CACHE>
void launch_resume_variant(hipStream_t s, dim3 grid, const DevScene& sc, const BatchParams& b, const PathPool& out, const SampleResults& res,
                           unsigned long long* statistics, const DeferQueues& queues);

} // namespace ptk
