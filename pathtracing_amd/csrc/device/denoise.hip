// Denoise stencils for gfx950: the edge-avoiding a-trous wavelet filter and the thresholded
// per-channel median of the reference (src/denoise.cpp), one output pixel per lane.
//
// Both filters read float64 [H][W][3] buffers.  A block handles a TILE_W x TILE_H tile of output
// pixels and first stages the centre tile of the three guide buffers in LDS; the 24 dilated taps
// of a pixel are (1 + hole) pixels apart, so neighbouring lanes read neighbouring pixels of the
// same row for every tap and the global loads stay coalesced 24-byte-per-lane runs.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ptk {

namespace {

constexpr int TILE_W = 32;
constexpr int TILE_H = 8;

__device__ __forceinline__ int reflect_coordinate(int x, int size) { // clamp_x/y_coordinate src/denoise.cpp:20-38
    if (x < 0) return -x;
    if (x > size - 1) return 2 * (size - 1) - x;
    return x;
}

__device__ __forceinline__ int expand_kernel_idx(int idx, int hole_width) { // :68-82
    if (idx == 1) return idx + hole_width;
    if (idx == -1) return idx - hole_width;
    if (idx == 2) return idx + 2 * hole_width;
    if (idx == -2) return idx - 2 * hole_width;
    return 0;
}

__device__ __forceinline__ double weight_component(double dx, double dy, double dz, double sigma) { // :40-45
    if (sigma == 0) return 0;
    return exp(-sqrt(dx * dx + dy * dy + dz * dz) / (sigma * sigma));
}

// include/denoise.h:11-14: centre 3/8, inner ring 1/4, outer ring 1/16
__device__ __forceinline__ double kernel_value(int dx, int dy) {
    int r = max(abs(dx), abs(dy));
    return r == 0 ? 3.0 / 8.0 : (r == 1 ? 1.0 / 4.0 : 1.0 / 16.0);
}

// blur_pixel (src/denoise.cpp:84-116) for every pixel; Jacobi update into image_out.
__global__ __launch_bounds__(TILE_W* TILE_H) void k_atrous(int W, int H, const double* __restrict__ image,
                                                           double* __restrict__ image_out, const double* __restrict__ position,
                                                           const double* __restrict__ normal, double sigma_rt, double sigma_x,
                                                           double sigma_n, int hole_width) {
    // centre values of the tile, staged once: every tap of a pixel compares against them
    __shared__ double centre[TILE_H][TILE_W][9];
    const int gx = blockIdx.x * TILE_W + threadIdx.x;
    const int gy = blockIdx.y * TILE_H + threadIdx.y;
    const bool inside = gx < W && gy < H;
    const size_t p = static_cast<size_t>(W) * static_cast<size_t>(inside ? gy : 0) + static_cast<size_t>(inside ? gx : 0);
    double* mine = centre[threadIdx.y][threadIdx.x];
    for (int j = 0; j < 3; j++) {
        mine[j] = image[3 * p + j];
        mine[3 + j] = position[3 * p + j];
        mine[6 + j] = normal[3 * p + j];
    }
    __syncthreads();
    if (!inside) return;
    double acc[3] = {0, 0, 0};
    double normalization = 0;
    for (int dx = -2; dx <= 2; dx++) {
        for (int dy = -2; dy <= 2; dy++) {
            const int x = reflect_coordinate(gx + expand_kernel_idx(dx, hole_width), W);
            const int y = reflect_coordinate(gy + expand_kernel_idx(dy, hole_width), H);
            const size_t q = static_cast<size_t>(W) * static_cast<size_t>(y) + static_cast<size_t>(x);
            const double c0 = image[3 * q], c1 = image[3 * q + 1], c2 = image[3 * q + 2];
            double w = weight_component(mine[0] - c0, mine[1] - c1, mine[2] - c2, sigma_rt);
            w *= weight_component(mine[3] - position[3 * q], mine[4] - position[3 * q + 1], mine[5] - position[3 * q + 2], sigma_x);
            w *= weight_component(mine[6] - normal[3 * q], mine[7] - normal[3 * q + 1], mine[8] - normal[3 * q + 2], sigma_n);
            const double k = kernel_value(dx, dy);
            double t0 = k * c0 * w, t1 = k * c1 * w, t2 = k * c2 * w;
            if (isnan(t0 * t0 + t1 * t1 + t2 * t2)) {
                t0 = t1 = t2 = 0.0;
                w = 0.0;
            }
            acc[0] += t0;
            acc[1] += t1;
            acc[2] += t2;
            normalization += k * w;
        }
    }
    image_out[3 * p] = acc[0] / normalization;
    image_out[3 * p + 1] = acc[1] / normalization;
    image_out[3 * p + 2] = acc[2] / normalization;
}

// 9-element median by a 19-exchange sorting network (kernel_size 3).
// Order of the median's values: ascending, NaN after everything else.  (The reference hands the window to
// std::nth_element, whose result is unspecified once a NaN breaks the ordering; here a window's median is NaN only
// when more than half of it is.)
__device__ __forceinline__ bool comes_before(double a, double b) { return a < b || (!isnan(a) && isnan(b)); }
__device__ __forceinline__ void cswap(double& a, double& b) {
    const bool swap = comes_before(b, a);
    const double lo = swap ? b : a, hi = swap ? a : b;
    a = lo;
    b = hi;
}
__device__ __forceinline__ double median9(double v[9]) {
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[1]); cswap(v[3], v[4]); cswap(v[6], v[7]);
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[3]); cswap(v[5], v[8]); cswap(v[4], v[7]);
    cswap(v[3], v[6]); cswap(v[1], v[4]); cswap(v[2], v[5]);
    cswap(v[4], v[7]); cswap(v[4], v[2]); cswap(v[6], v[4]);
    cswap(v[4], v[2]);
    return v[4];
}

// median_filter (src/denoise.cpp:149-201), kernel_size 3: the tile plus a one-pixel halo is staged
// in LDS, then each lane selects the median of its 3x3 window per channel.
__global__ __launch_bounds__(TILE_W* TILE_H) void k_median3(int W, int H, const double* __restrict__ image,
                                                            double* __restrict__ image_out, double threshold) {
    __shared__ double tile[TILE_H + 2][TILE_W + 2][3];
    const int x0 = blockIdx.x * TILE_W - 1, y0 = blockIdx.y * TILE_H - 1;
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    for (int i = tid; i < (TILE_H + 2) * (TILE_W + 2); i += TILE_W * TILE_H) {
        const int ty = i / (TILE_W + 2), tx = i - ty * (TILE_W + 2);
        int x = x0 + tx, y = y0 + ty;
        // coordinates beyond the far edges only feed lanes that are outside the image
        x = reflect_coordinate(min(x, 2 * (W - 1)), W);
        y = reflect_coordinate(min(y, 2 * (H - 1)), H);
        x = max(x, 0);
        y = max(y, 0);
        const size_t q = static_cast<size_t>(W) * static_cast<size_t>(y) + static_cast<size_t>(x);
        tile[ty][tx][0] = image[3 * q];
        tile[ty][tx][1] = image[3 * q + 1];
        tile[ty][tx][2] = image[3 * q + 2];
    }
    __syncthreads();
    const int gx = blockIdx.x * TILE_W + threadIdx.x;
    const int gy = blockIdx.y * TILE_H + threadIdx.y;
    if (gx >= W || gy >= H) return;
    double med[3];
    for (int c = 0; c < 3; c++) {
        double v[9];
        for (int dy = 0; dy < 3; dy++)
            for (int dx = 0; dx < 3; dx++) v[3 * dy + dx] = tile[threadIdx.y + dy][threadIdx.x + dx][c];
        med[c] = median9(v);
    }
    const double* cur = tile[threadIdx.y + 1][threadIdx.x + 1];
    const double d0 = cur[0] - med[0], d1 = cur[1] - med[1], d2 = cur[2] - med[2];
    const bool replace = sqrt(d0 * d0 + d1 * d1 + d2 * d2) > threshold;
    const size_t p = static_cast<size_t>(W) * static_cast<size_t>(gy) + static_cast<size_t>(gx);
    image_out[3 * p] = replace ? med[0] : cur[0];
    image_out[3 * p + 1] = replace ? med[1] : cur[1];
    image_out[3 * p + 2] = replace ? med[2] : cur[2];
}

// General odd kernel size (<= 15): the (k*k/2)-th smallest value of the window, found by counting
// for each candidate how many window values precede it (ties broken by position).
constexpr int MAX_MEDIAN = 15;
__global__ __launch_bounds__(TILE_W* TILE_H) void k_median_general(int W, int H, const double* __restrict__ image,
                                                                   double* __restrict__ image_out, int kernel_size,
                                                                   double threshold) {
    const int gx = blockIdx.x * TILE_W + threadIdx.x;
    const int gy = blockIdx.y * TILE_H + threadIdx.y;
    if (gx >= W || gy >= H) return;
    const int offset = (kernel_size - 1) / 2;
    const int size = kernel_size * kernel_size;
    const int target = size / 2;
    double med[3];
    for (int c = 0; c < 3; c++) {
        med[c] = 0;
        for (int i = 0; i < size; i++) {
            const int ix = reflect_coordinate(gx + i % kernel_size - offset, W);
            const int iy = reflect_coordinate(gy + i / kernel_size - offset, H);
            const double vi = image[3 * (static_cast<size_t>(W) * iy + ix) + c];
            int rank = 0;
            for (int j = 0; j < size; j++) {
                const int jx = reflect_coordinate(gx + j % kernel_size - offset, W);
                const int jy = reflect_coordinate(gy + j / kernel_size - offset, H);
                const double vj = image[3 * (static_cast<size_t>(W) * jy + jx) + c];
                rank += (comes_before(vj, vi) || (!comes_before(vi, vj) && j < i)) ? 1 : 0;
            }
            if (rank == target) med[c] = vi;
        }
    }
    const size_t p = static_cast<size_t>(W) * static_cast<size_t>(gy) + static_cast<size_t>(gx);
    const double c0 = image[3 * p], c1 = image[3 * p + 1], c2 = image[3 * p + 2];
    const double d0 = c0 - med[0], d1 = c1 - med[1], d2 = c2 - med[2];
    const bool replace = sqrt(d0 * d0 + d1 * d1 + d2 * d2) > threshold;
    image_out[3 * p] = replace ? med[0] : c0;
    image_out[3 * p + 1] = replace ? med[1] : c1;
    image_out[3 * p + 2] = replace ? med[2] : c2;
}

} // namespace

void launch_atrous(hipStream_t s, int width, int height, const double* image_in, double* image_out, const double* position,
                   const double* normal, double sigma_rt, double sigma_x, double sigma_n, int hole_width) {
    dim3 grid((width + TILE_W - 1) / TILE_W, (height + TILE_H - 1) / TILE_H);
    hipLaunchKernelGGL(k_atrous, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, position, normal, sigma_rt,
                       sigma_x, sigma_n, hole_width);
}

void launch_median(hipStream_t s, int width, int height, const double* image_in, double* image_out, int kernel_size,
                   double threshold) {
    dim3 grid((width + TILE_W - 1) / TILE_W, (height + TILE_H - 1) / TILE_H);
    if (kernel_size == 3 && width >= 2 && height >= 2)
        hipLaunchKernelGGL(k_median3, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, threshold);
    else
        hipLaunchKernelGGL(k_median_general, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, kernel_size,
                           threshold);
}

} // namespace ptk
