// Denoise stencils for gfx950: the edge-avoiding a-trous wavelet filter and the thresholded
// per-channel median of the reference (src/denoise.cpp), one output pixel per lane.
//
// Both filters read float64 [H][W][3] buffers.  A block handles a TILE_W x TILE_H tile of output
// pixels and first stages the centre tile of the three guide buffers in LDS; the 24 dilated taps
// of a pixel are (1 + hole) pixels apart, so neighbouring lanes read neighbouring pixels of the
// same row for every tap and the global loads stay coalesced 24-byte-per-lane runs.
#include <hip/hip_runtime.h>

#include "kernels.h"

#include "device_math.h"

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

// include/denoise.h:11-14: centre 3/8, inner ring 1/4, outer ring 1/16
__device__ __forceinline__ double kernel_value(int dx, int dy) {
    int r = max(abs(dx), abs(dy));
    return r == 0 ? 3.0 / 8.0 : (r == 1 ? 1.0 / 4.0 : 1.0 / 16.0);
}

// Euclidean norm.  fsqrt (device_math.h) gives sqrt's bits at two thirds of its instructions for operands in the normal
// range; sums of squares below it (differences under 1e-150) take the library routine.
__device__ __forceinline__ double norm3(double a, double b, double c) {
    const double s = a * a + b * b + c * c;
    return (s >= 0x1p-1000 || s == 0) ? fsqrt(s) : sqrt(s);
}

struct Guide { // what a tap compares: colour, position, normal of one pixel
    double c[3], x[3], n[3];
};

// Weight and weighted colour of one tap (blur_pixel's loop body, src/denoise.cpp:93-108).  The reference multiplies three
// factors exp(-|d| / sigma^2) (compute_weight_component :40-45); here their exponents are added and one exponential is
// taken, which is the same number up to the rounding of exp (relative 1e-13 at most for arguments that do not
// underflow, inside the 1e-10 the filter is checked to) at a third of the transcendental work.  A sigma of zero makes
// its factor, and with it the weight, zero, as in the reference.
__device__ __forceinline__ void add_tap(const Guide& centre, const Guide& tap, double k, double inv_rt, double inv_x, double inv_n, bool any_zero_sigma,
                                        double (&acc)[3], double& normalization) {
    const double dc = norm3(centre.c[0] - tap.c[0], centre.c[1] - tap.c[1], centre.c[2] - tap.c[2]);
    const double dx = norm3(centre.x[0] - tap.x[0], centre.x[1] - tap.x[1], centre.x[2] - tap.x[2]);
    const double dn = norm3(centre.n[0] - tap.n[0], centre.n[1] - tap.n[1], centre.n[2] - tap.n[2]);
    double w = exp(-(dc * inv_rt) - dx * inv_x - dn * inv_n);
    if (any_zero_sigma) w = isnan(w) ? w : 0.0;
    double t0 = k * tap.c[0] * w, t1 = k * tap.c[1] * w, t2 = k * tap.c[2] * w;
    if (isnan(t0 * t0 + t1 * t1 + t2 * t2)) {
        t0 = t1 = t2 = 0.0;
        w = 0.0;
    }
    acc[0] += t0;
    acc[1] += t1;
    acc[2] += t2;
    normalization += k * w;
}

// blur_pixel (src/denoise.cpp:84-116) for every pixel; Jacobi update into image_out.
//
// STAGED: the taps of a tile's pixels lie within REACH pixels of the tile (reach = 2 + 2 * hole_width: the first two
// iterations of the filter, hole widths 0 and 1), so the block first copies the tile and that rim of the three buffers
// into LDS, coordinates reflected at the image border as the reference does (:20-38), and every tap is an LDS read:
// each pixel's 72 bytes are fetched from memory once per block instead of up to 25 times.  The later iterations (hole
// widths 3, 7, ...) reach 8 pixels and more; their taps come from memory (neighbouring lanes read neighbouring
// pixels of a row for every tap), and only the centre values are kept in registers.
constexpr int ATROUS_MAX_REACH = 4;
constexpr int STAGE_W = TILE_W + 2 * ATROUS_MAX_REACH, STAGE_H = TILE_H + 2 * ATROUS_MAX_REACH;

template<bool STAGED>
__global__ __launch_bounds__(TILE_W* TILE_H) void k_atrous(int W, int H, const double* __restrict__ image, double* __restrict__ image_out,
                                                           const double* __restrict__ position, const double* __restrict__ normal,
                                                           double sigma_rt, double sigma_x, double sigma_n, int hole_width) {
    // plane-major: nine planes of STAGE_H x STAGE_W doubles; the lanes of a wave read consecutive doubles of one plane
    __shared__ double stage[STAGED ? 9 * STAGE_H * STAGE_W : 1];
    const int gx = blockIdx.x * TILE_W + threadIdx.x;
    const int gy = blockIdx.y * TILE_H + threadIdx.y;
    const int reach = 2 + 2 * hole_width;
    if (STAGED) {
        const int x0 = blockIdx.x * TILE_W - reach, y0 = blockIdx.y * TILE_H - reach;
        const int sw = TILE_W + 2 * reach, sh = TILE_H + 2 * reach;
        for (int i = threadIdx.y * TILE_W + threadIdx.x; i < sw * sh; i += TILE_W * TILE_H) {
            const int ty = i / sw, tx = i - ty * sw;
            // Coordinates beyond the far edges by more than one reflection only feed lanes outside the image.
            int x = reflect_coordinate(min(x0 + tx, 2 * (W - 1)), W), y = reflect_coordinate(min(y0 + ty, 2 * (H - 1)), H);
            x = max(x, 0);
            y = max(y, 0);
            const size_t q = 3 * (static_cast<size_t>(W) * static_cast<size_t>(y) + static_cast<size_t>(x));
            const int at = ty * STAGE_W + tx;
            for (int j = 0; j < 3; j++) {
                stage[(0 + j) * STAGE_H * STAGE_W + at] = image[q + j];
                stage[(3 + j) * STAGE_H * STAGE_W + at] = position[q + j];
                stage[(6 + j) * STAGE_H * STAGE_W + at] = normal[q + j];
            }
        }
        __syncthreads();
    }
    if (gx >= W || gy >= H) return;
    const size_t p = static_cast<size_t>(W) * static_cast<size_t>(gy) + static_cast<size_t>(gx);
    auto fetch = [&](int x, int y, int tx, int ty) { // pixel (x, y) of the image = entry (tx, ty) of the staged tile
        Guide g;
        if (STAGED) {
            const int at = ty * STAGE_W + tx;
            for (int j = 0; j < 3; j++) {
                g.c[j] = stage[(0 + j) * STAGE_H * STAGE_W + at];
                g.x[j] = stage[(3 + j) * STAGE_H * STAGE_W + at];
                g.n[j] = stage[(6 + j) * STAGE_H * STAGE_W + at];
            }
        }
        else {
            const size_t q = 3 * (static_cast<size_t>(W) * static_cast<size_t>(y) + static_cast<size_t>(x));
            for (int j = 0; j < 3; j++) {
                g.c[j] = image[q + j];
                g.x[j] = position[q + j];
                g.n[j] = normal[q + j];
            }
        }
        return g;
    };
    const Guide centre = fetch(gx, gy, static_cast<int>(threadIdx.x) + reach, static_cast<int>(threadIdx.y) + reach);
    const bool any_zero_sigma = sigma_rt == 0 || sigma_x == 0 || sigma_n == 0;
    // -d / sigma^2 as a product with 1 / sigma^2 (a zero sigma is dealt with by any_zero_sigma)
    const double inv_rt = sigma_rt == 0 ? 0.0 : 1.0 / (sigma_rt * sigma_rt), inv_x = sigma_x == 0 ? 0.0 : 1.0 / (sigma_x * sigma_x),
                 inv_n = sigma_n == 0 ? 0.0 : 1.0 / (sigma_n * sigma_n);
    double acc[3] = {0, 0, 0};
    double normalization = 0;
    for (int dx = -2; dx <= 2; dx++) {
        for (int dy = -2; dy <= 2; dy++) {
            const int ox = expand_kernel_idx(dx, hole_width), oy = expand_kernel_idx(dy, hole_width);
            // the staged tile holds reflected pixels at reflected places: the tap's offset is all that is needed there
            const int x = STAGED ? 0 : reflect_coordinate(gx + ox, W), y = STAGED ? 0 : reflect_coordinate(gy + oy, H);
            const Guide tap = fetch(x, y, static_cast<int>(threadIdx.x) + reach + ox, static_cast<int>(threadIdx.y) + reach + oy);
            add_tap(centre, tap, kernel_value(dx, dy), inv_rt, inv_x, inv_n, any_zero_sigma, acc, normalization);
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

// General odd kernel size (<= 15).  The tile and a rim of (k - 1) / 2 pixels are staged in LDS once (reflected at the
// image border like the reference's window, src/denoise.cpp:160-175), one channel at a time; every lane then selects the
// (k * k / 2)-th smallest value of its k x k window from LDS:
//   k <= 7   by counting, for each candidate, how many window values precede it (ties broken by position): k^4 LDS reads
//   k >= 9   bit by bit, from the most significant bit of an order-preserving 64-bit image of the doubles down: 64 passes
//            that each count the window values agreeing with the bits fixed so far and having a zero next (64 k^2 reads,
//            a quarter of k^4 at k = 15)
// Order: ascending, NaN after everything else (comes_before).
constexpr int MAX_MEDIAN = 15;
constexpr int MEDIAN_STAGE_W = TILE_W + MAX_MEDIAN - 1, MEDIAN_STAGE_H = TILE_H + MAX_MEDIAN - 1;

__device__ __forceinline__ unsigned long long order_key(double v) { // a < b  <=>  key(a) < key(b); NaN above +infinity
    if (isnan(v)) return ~0ull;
    if (v == 0) v = 0.0; // -0.0 and +0.0 compare equal
    const unsigned long long u = static_cast<unsigned long long>(__double_as_longlong(v));
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k) {
    if (k == ~0ull) return __longlong_as_double(0x7FF8000000000000ll);
    const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double(static_cast<long long>(u));
}

// KS: the kernel size when it is 5 or 7 (the window then lives in registers and the counting loops are unrolled), 0: any
template<int KS>
__global__ __launch_bounds__(TILE_W* TILE_H) void k_median_general(int W, int H, const double* __restrict__ image,
                                                                   double* __restrict__ image_out, int kernel_size,
                                                                   double threshold) {
    __shared__ double tile[MEDIAN_STAGE_H * MEDIAN_STAGE_W];
    const int gx = blockIdx.x * TILE_W + threadIdx.x;
    const int gy = blockIdx.y * TILE_H + threadIdx.y;
    const bool inside = gx < W && gy < H;
    const int offset = (kernel_size - 1) / 2;
    const int size = kernel_size * kernel_size;
    const int target = size / 2;
    const int sw = TILE_W + kernel_size - 1, sh = TILE_H + kernel_size - 1;
    const int x0 = blockIdx.x * TILE_W - offset, y0 = blockIdx.y * TILE_H - offset;
    double med[3] = {0, 0, 0}, cur[3] = {0, 0, 0};
    for (int c = 0; c < 3; c++) {
        __syncthreads(); // the previous channel's tile is no longer read
        for (int i = threadIdx.y * TILE_W + threadIdx.x; i < sw * sh; i += TILE_W * TILE_H) {
            const int ty = i / sw, tx = i - ty * sw;
            // coordinates beyond the far edges by more than one reflection only feed lanes outside the image
            int x = reflect_coordinate(min(x0 + tx, 2 * (W - 1)), W), y = reflect_coordinate(min(y0 + ty, 2 * (H - 1)), H);
            x = max(x, 0);
            y = max(y, 0);
            tile[ty * MEDIAN_STAGE_W + tx] = image[3 * (static_cast<size_t>(W) * static_cast<size_t>(y) + static_cast<size_t>(x)) + c];
        }
        __syncthreads();
        if (!inside) continue;
        const double* window = tile + threadIdx.y * MEDIAN_STAGE_W + threadIdx.x; // window value (wx, wy) at [wy * MEDIAN_STAGE_W + wx]
        cur[c] = window[offset * MEDIAN_STAGE_W + offset];
        if (KS > 0) {
            double v[KS > 0 ? KS * KS : 1];
#pragma unroll
            for (int i = 0; i < KS * KS; i++) v[i] = window[(i / KS) * MEDIAN_STAGE_W + i % KS];
#pragma unroll
            for (int i = 0; i < KS * KS; i++) {
                int rank = 0;
#pragma unroll
                for (int j = 0; j < KS * KS; j++) rank += (comes_before(v[j], v[i]) || (!comes_before(v[i], v[j]) && j < i)) ? 1 : 0;
                if (rank == target) med[c] = v[i];
            }
        }
        else if (kernel_size <= 7) {
            for (int i = 0; i < size; i++) {
                const double vi = window[(i / kernel_size) * MEDIAN_STAGE_W + i % kernel_size];
                int rank = 0;
                for (int j = 0; j < size; j++) {
                    const double vj = window[(j / kernel_size) * MEDIAN_STAGE_W + j % kernel_size];
                    rank += (comes_before(vj, vi) || (!comes_before(vi, vj) && j < i)) ? 1 : 0;
                }
                if (rank == target) med[c] = vi;
            }
        }
        else {
            unsigned long long prefix = 0; // the bits of the answer found so far
            int below = 0;                 // window values smaller than anything with that prefix
            for (int bit = 63; bit >= 0; bit--) {
                const unsigned long long mask = ~0ull << bit; // the bits decided after this pass
                int zeros = 0;             // values that agree with the prefix and have a zero at `bit`
                for (int wy = 0; wy < kernel_size; wy++)
                    for (int wx = 0; wx < kernel_size; wx++)
                        zeros += ((order_key(window[wy * MEDIAN_STAGE_W + wx]) & mask) == prefix) ? 1 : 0;
                if (below + zeros <= target) { // the answer is not among them: its bit is one
                    below += zeros;
                    prefix |= 1ull << bit;
                }
            }
            med[c] = key_value(prefix);
        }
    }
    if (!inside) return;
    const size_t p = static_cast<size_t>(W) * static_cast<size_t>(gy) + static_cast<size_t>(gx);
    const double d0 = cur[0] - med[0], d1 = cur[1] - med[1], d2 = cur[2] - med[2];
    const bool replace = sqrt(d0 * d0 + d1 * d1 + d2 * d2) > threshold;
    image_out[3 * p] = replace ? med[0] : cur[0];
    image_out[3 * p + 1] = replace ? med[1] : cur[1];
    image_out[3 * p + 2] = replace ? med[2] : cur[2];
}

} // namespace

void launch_atrous(hipStream_t s, int width, int height, const double* image_in, double* image_out, const double* position,
                   const double* normal, double sigma_rt, double sigma_x, double sigma_n, int hole_width) {
    dim3 grid((width + TILE_W - 1) / TILE_W, (height + TILE_H - 1) / TILE_H);
    // staging needs the rim's reflected coordinates inside the image: images at least as large as the reach
    const int reach = 2 + 2 * hole_width;
    if (reach <= ATROUS_MAX_REACH && width > reach && height > reach)
        hipLaunchKernelGGL(k_atrous<true>, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, position, normal, sigma_rt,
                           sigma_x, sigma_n, hole_width);
    else
        hipLaunchKernelGGL(k_atrous<false>, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, position, normal, sigma_rt,
                           sigma_x, sigma_n, hole_width);
}

void launch_median(hipStream_t s, int width, int height, const double* image_in, double* image_out, int kernel_size,
                   double threshold) {
    dim3 grid((width + TILE_W - 1) / TILE_W, (height + TILE_H - 1) / TILE_H);
    if (kernel_size == 3 && width >= 2 && height >= 2)
        hipLaunchKernelGGL(k_median3, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, threshold);
    else if (kernel_size == 5)
        hipLaunchKernelGGL(k_median_general<5>, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, kernel_size, threshold);
    else if (kernel_size == 7)
        hipLaunchKernelGGL(k_median_general<7>, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, kernel_size, threshold);
    else
        hipLaunchKernelGGL(k_median_general<0>, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, kernel_size, threshold);
}

} // namespace ptk
