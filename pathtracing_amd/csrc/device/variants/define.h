// Macros that instantiate one kernel and define its launcher (see ../variants.h).
#pragma once
#include "../path_kernels.h"
#include "../variants.h"

#define PTK_BOUNCE_VARIANT(T, U, D, P, C)                                                                                              \
    template<>                                                                                                                         \
    void launch_bounce_variant<T, U, D, P, C>(hipStream_t s, dim3 grid, const DevScene& sc, const BatchParams& b, const PathPool& in,  \
                                              const PathPool& out, const SampleResults& res, const uint32_t* list, const BinInfo* info, \
                                              uint32_t rounds, uint32_t* block_counts, uint32_t count_stride,                          \
                                              unsigned long long* statistics, const DeferQueues& queues) {                             \
        hipLaunchKernelGGL((k_bounce<T, U, D, P, C>), grid, dim3(BLOCK), 0, s, sc, b, in, out, res, list, info, rounds, block_counts,  \
                           count_stride, statistics, queues);                                                                          \
    }

#define PTK_TAIL_VARIANT(T, U, P, C)                                                                                                   \
    template<>                                                                                                                         \
    void launch_tail_variant<T, U, P, C>(hipStream_t s, dim3 grid, const DevScene& sc, const BatchParams& b, const PathPool& in,       \
                                         const PathPool& out, const SampleResults& res, const uint32_t* list, BinInfo* info,           \
                                         unsigned long long* statistics) {                                                             \
        hipLaunchKernelGGL((k_tail<T, U, P, C>), grid, dim3(BLOCK), 0, s, sc, b, in, out, res, list, info, statistics);                \
    }

#define PTK_RESUME_VARIANT(U, P, C)                                                                                                    \
    template<>                                                                                                                         \
    void launch_resume_variant<U, P, C>(hipStream_t s, dim3 grid, const DevScene& sc, const BatchParams& b, const PathPool& out,       \
                                        const SampleResults& res, unsigned long long* statistics, const DeferQueues& queues) {         \
        hipLaunchKernelGGL((k_bounce_resume<U, P, C>), grid, dim3(BLOCK), 0, s, sc, b, out, res, statistics, queues);                  \
    }
