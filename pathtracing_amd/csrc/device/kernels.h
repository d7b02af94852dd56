// Host-callable launchers of the kernels in kernels.hip and denoise.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "scene_device.h"

namespace ptk {

// number of blocks k_extend / k_bin_scatter use for a list of n entries (= rows of block_counts used)
uint32_t bin_blocks(uint32_t n);

// In every launcher `bound` is an upper bound of the number of list entries, used only to size the
// grid; the kernels read the true lengths from device memory (*previous: the summary of the list they
// consume; *info: the summary of the sorted list they produce or walk), so the host never has to wait
// for a count before it queues the next bounce.
void launch_generate(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathState& ps, uint32_t* list, BinInfo* first_info);
void launch_extend(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathState& ps, const uint32_t* list, uint32_t bound,
                   const BinInfo* previous, uint32_t* block_counts, uint32_t count_stride, unsigned long long* ray_counter);
// scan + scatter: list_in -> class-sorted list_out, summary in *info
void launch_bin(hipStream_t s, const DevScene& sc, const PathState& ps, const uint32_t* list_in, uint32_t bound, const BinInfo* previous,
                uint32_t* block_counts, uint32_t count_stride, BinInfo* info, uint32_t* list_out);
void launch_shade(hipStream_t s, int cls, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathState& ps,
                  const uint32_t* list, const BinInfo* info);
void launch_connect(hipStream_t s, uint32_t bound, const DevScene& sc, const PathState& ps, const uint32_t* list, const BinInfo* info,
                    unsigned long long* ray_counter);
void launch_accumulate(hipStream_t s, const BatchParams& b, const PathState& ps, double* image, double* position, double* normal,
                       unsigned long long* pixel_counters);
void launch_gather_samples(hipStream_t s, const BatchParams& b, const PathState& ps, size_t pixel_base, double* color,
                           double* position, double* normal);
void launch_primary(hipStream_t s, const DevScene& sc, const BatchParams& b, int32_t* object_index, int32_t* primitive_index,
                    double* distance, double* position, double* normal);

// denoise.hip
void launch_atrous(hipStream_t s, int width, int height, const double* image_in, double* image_out, const double* position,
                   const double* normal, double sigma_rt, double sigma_x, double sigma_n, int hole_width);
void launch_median(hipStream_t s, int width, int height, const double* image_in, double* image_out, int kernel_size,
                   double threshold);

} // namespace ptk
