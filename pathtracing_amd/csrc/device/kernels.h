// Host-callable launchers of the kernels in kernels.hip and denoise.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "scene_device.h"

namespace ptk {

// number of whole or partial BIN_CHUNK runs in n records
uint32_t bin_blocks(uint32_t n);
// Chunk size, in rounds of one block (1..BIN_ROUNDS), the bounce that walks a list of at most `bound`
// records should use; the pool it writes is then sorted with the same value.  k_generate always uses
// BIN_ROUNDS.
uint32_t chunk_rounds(uint32_t bound);
// entries per class block_counts needs for pools of up to `records` records under chunk_rounds()
uint32_t chunk_capacity(uint32_t records);

// What a launcher needs to know about the device it launches on and about the process-wide tuning knobs.
struct LaunchContext {
    hipStream_t stream;
    uint32_t compute_units; // of the device the stream belongs to: sizes the grids of the persistent kernels
    bool quad_traverse;     // true: the hierarchy walks of queued rays use four lanes per ray (k_traverse_quad), false: one
    bool specialised;       // false: every scene runs the general kernels (all features compiled in, tables read from
                            // memory) instead of the ones specialised on its contents; the results are the same
};

// In every launcher `bound` is an upper bound of the number of records, used only to size the grid;
// the kernels read the true counts from device memory (*previous: the summary of the bounce that wrote
// the pool they read; *info: the summary of the sorted list they produce or walk), so the host never
// has to wait for a count before it queues the next bounce.
// block_counts: [N_CLASSES][count_stride] per-chunk class counts; count_stride >= chunk_capacity(capacity)
void launch_generate(hipStream_t s, const DevScene& sc, const BatchParams& b, const PathPool& pool, const SampleResults& res,
                     BinInfo* first_info, uint32_t* block_counts, uint32_t count_stride, const DeferQueues& queues);
// scenes with meshes, after launch_generate and launch_traverse: completes the camera rays that were queued
void launch_primary_resume(hipStream_t s, uint32_t n_paths, const DevScene& sc, const PathPool& pool, const SampleResults& res,
                           const DeferQueues& queues);
// scan + scatter: class-sorted index list of the records of `pool` from the counts its writer left, summary in *info
void launch_bin(hipStream_t s, const DevScene& sc, const PathPool& pool, uint32_t bound, uint32_t rounds, const BinInfo* previous,
                uint32_t* block_counts, uint32_t count_stride, BinInfo* info, uint32_t* list_out);
// one vertex of every path; leaves the class counts of `out` in block_counts;
// statistics: [STATISTICS_SLOTS][N_STATISTICS] per-block slots; queues: used by scenes with meshes
void launch_bounce(const LaunchContext& lc, uint32_t bound, uint32_t rounds, const DevScene& sc, const BatchParams& b, const PathPool& in,
                   const PathPool& out, const SampleResults& res, const uint32_t* list, const BinInfo* info, uint32_t* block_counts,
                   uint32_t count_stride, unsigned long long* statistics, const DeferQueues& queues);
// Name (as the profiler prints it) of the k_bounce instantiation launch_bounce uses for this scene.
std::string bounce_kernel_name(const DevScene& sc, bool specialised);
// Every remaining bounce of the paths in `list` (at most `bound` of them) in one launch.  TAIL_RECORDS is the default
// count at which a batch switches to it (scenes with meshes earlier: tail_threshold in ptk_api.hip), measured on one MI355X: the bounces of
// fewer than a million paths no longer fill the device, and three launches per bounce cost more than the sorting
// they buy.
constexpr uint32_t TAIL_RECORDS = 1u << 20;
void launch_tail(const LaunchContext& lc, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathPool& in, const PathPool& out,
                 const SampleResults& res, const uint32_t* list, BinInfo* info, unsigned long long* statistics);
// Scenes with meshes, after launch_bounce: the hierarchy walks of the queued rays (statistics slots: [0] rays,
// [1] nodes visited, [2] triangles tested), then the deferred vertices.
void launch_traverse(const LaunchContext& lc, uint32_t bound, const DevScene& sc, const DeferQueues& queues, unsigned long long* statistics);
void launch_bounce_resume(const LaunchContext& lc, uint32_t bound, const DevScene& sc, const BatchParams& b, const PathPool& out,
                          const SampleResults& res, unsigned long long* statistics, const DeferQueues& queues);
// block_counts[class][chunk] of the records `out` holds for the chunks of the list `info` describes (scenes with meshes)
void launch_count_classes(const LaunchContext& lc, uint32_t bound, uint32_t rounds, const DevScene& sc, const PathPool& out, const BinInfo* info,
                          uint32_t* block_counts, uint32_t count_stride);
void launch_sum_statistics(hipStream_t s, const unsigned long long* slots, unsigned long long* totals);
void launch_accumulate(hipStream_t s, const BatchParams& b, const SampleResults& res, double* image, double* position, double* normal,
                       unsigned long long* pixel_counters);
void launch_gather_samples(hipStream_t s, const BatchParams& b, const SampleResults& res, size_t pixel_base, double* color,
                           double* position, double* normal);
void launch_primary(hipStream_t s, const DevScene& sc, const BatchParams& b, int32_t* object_index, int32_t* primitive_index,
                    double* distance, double* position, double* normal);

// Copies record list[i] of `in` to record i of `out` (i < n, i % 5 != 3) through the LDS staging k_bounce uses;
// `blocks` blocks of 256 threads, `noise` holds noise_mask + 1 doubles, `sink` blocks * 256.
void launch_selftest_staging(hipStream_t s, uint32_t blocks, const PathPool& in, const PathPool& out, const uint32_t* list, uint32_t n,
                             const double* noise, uint32_t noise_mask, double* sink);

// `from` folded into the film whose buffers come first: per pixel the sample-count-weighted mean of the two running
// means, with increment_average's arithmetic (both films on the device of the stream).
void launch_merge_films(hipStream_t s, size_t n_pixels, double* image, double* position, double* normal, unsigned long long* counters,
                        const double* from_image, const double* from_position, const double* from_normal,
                        const unsigned long long* from_counters);

// bvh_build.hip: the hierarchy of `n` triangles (9 doubles each, host memory) built on the current device (linear BVH
// over Morton-ordered triangles, collapsed to 4-wide nodes).  false: not built (fewer than 1024 triangles, a tree
// that needs more than `stack_limit` stack entries, or a HIP error, left in *error) and the caller builds on the
// host.  build_ms: device time of the build kernels.  memory_limit: most device memory the build may allocate (0: no
// limit); a build that needs more fails with hipErrorOutOfMemory, as on a device without room for it.
struct BuiltBvh;
bool build_bvh4_gpu(const double* vertices, int64_t n, int max_leaf, int stack_limit, hipStream_t stream, BuiltBvh& result, double* build_ms,
                    hipError_t* error, size_t memory_limit);

// denoise.hip
void launch_atrous(hipStream_t s, int width, int height, const double* image_in, double* image_out, const double* position,
                   const double* normal, double sigma_rt, double sigma_x, double sigma_n, int hole_width);
void launch_median(hipStream_t s, int width, int height, const double* image_in, double* image_out, int kernel_size,
                   double threshold);

} // namespace ptk
