// Host-callable launchers of the kernels in kernels.hip and denoise.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "scene_device.h"

namespace ptk {

void launch_generate(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathPool& pool, const SampleResults& res);
void launch_extend(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathPool& pool, uint32_t n_active,
                   const HitRecords& hits, uint32_t* queues, uint32_t queue_stride, BounceCounters* counters,
                   const SampleResults& res);
// `bound` is an upper bound of the class's queue length, used only to size the grid; the kernel
// reads the true length from `counters`.
void launch_shade(hipStream_t s, int cls, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathPool& in,
                  const PathPool& out, const HitRecords& hits, const uint32_t* queue, BounceCounters* counters,
                  const ShadowQueue& shadow, const SampleResults& res);
void launch_connect(hipStream_t s, uint32_t bound, const DevScene& sc, const ShadowQueue& shadow, const BounceCounters* counters,
                    const SampleResults& res, unsigned long long* ray_counter);
void launch_accumulate(hipStream_t s, const BatchParams& b, const SampleResults& res, double* image, double* position,
                       double* normal, unsigned long long* pixel_counters);
void launch_gather_samples(hipStream_t s, const BatchParams& b, const SampleResults& res, size_t pixel_base, double* color,
                           double* position, double* normal);
void launch_primary(hipStream_t s, const DevScene& sc, const BatchParams& b, int32_t* object_index, int32_t* primitive_index,
                    double* distance, double* position, double* normal);

// denoise.hip
void launch_atrous(hipStream_t s, int width, int height, const double* image_in, double* image_out, const double* position,
                   const double* normal, double sigma_rt, double sigma_x, double sigma_n, int hole_width);
void launch_median(hipStream_t s, int width, int height, const double* image_in, double* image_out, int kernel_size,
                   double threshold);

} // namespace ptk
