// Denoise stencils for gfx950: the edge-avoiding a-trous wavelet filter and the thresholded
// per-channel median of the reference (src/denoise.cpp), one output pixel per lane.
//
// Both filters read float64 [H][W][3] buffers.  A block handles a TILE_W x TILE_H tile of output pixels.  The a-trous
// kernel of the first two iterations (taps at most four pixels away) stages the tile and its rim of all three guide
// buffers in LDS and takes every tap from there; later iterations (taps eight pixels away and more) read their taps
// from memory, where neighbouring lanes read neighbouring pixels of a row for every tap.  The median kernels stage the
// tile plus the rim their window needs, one channel at a time for the general sizes.  What bounds them is fp64
// arithmetic per tap (three square roots and an exponential) and per window value, not memory: the kernels move a
// tenth of what HBM could.
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

// Euclidean norm to a relative 1e-13: v_rsq_f64 (23 bits) and one Newton-Raphson step on the root, five instructions
// where the correctly rounded square root takes about fourteen.  The filter is checked to 1e-10, and an error of the
// norm enters the weight's exponent scaled by norm / sigma^2.  2^-1000 is added to the sum of squares: it changes no
// sum that is not itself that small (differences under 1e-150), keeps the reciprocal root of an exact zero finite, and
// a NaN stays a NaN.  (An infinite sum gives NaN where the root would be infinite; the tap then drops out instead of
// getting the weight zero, which is the same thing.)
__device__ __forceinline__ double norm3(double a, double b, double c) {
    const double s = (a * a + b * b + c * c) + 0x1p-1000;
    const double y = __builtin_amdgcn_rsq(s);
    const double g = s * y, h = 0.5 * y;
    return __builtin_fma(__builtin_fma(-g, g, s), h, g);
}

// 2^t for t <= 0 to a relative 1e-14: t is split into its nearest integer and a fraction in [-1/2, 1/2], whose power
// is a polynomial of degree 11 (the series of e^(f ln 2); truncation error 6e-15 at the ends), scaled by v_ldexp_f64.
// Arguments below -1100 give 0 like the ones just above them, and so does a NaN (the maximum of a NaN and a number is
// the number): a tap whose position or normal is NaN then gets the weight zero where the reference drops it, which
// adds the same nothing to both sums.
__device__ __forceinline__ double exp2_nonpositive(double t) {
    t = fmax(t, -1100.0);
    const double n = rint(t);
    const double f = t - n; // exact
    double p = 0x1.e8cac7351bb22p-32;               // ln2^11 / 11!
    p = __builtin_fma(p, f, 0x1.e4cf5158b8ec7p-28); // ln2^10 / 10!
    p = __builtin_fma(p, f, 0x1.b5253d395e7c1p-24); // ln2^9 / 9!
    p = __builtin_fma(p, f, 0x1.62c0223a5c822p-20); // ln2^8 / 8!
    p = __builtin_fma(p, f, 0x1.ffcbfc588b0c5p-17); // ln2^7 / 7!
    p = __builtin_fma(p, f, 0x1.430912f86c786p-13); // ln2^6 / 6!
    p = __builtin_fma(p, f, 0x1.5d87fe78a6730p-10); // ln2^5 / 5!
    p = __builtin_fma(p, f, 0x1.3b2ab6fba4e77p-7);  // ln2^4 / 4!
    p = __builtin_fma(p, f, 0x1.c6b08d704a0bfp-5);  // ln2^3 / 3!
    p = __builtin_fma(p, f, 0x1.ebfbdff82c58ep-3);  // ln2^2 / 2!
    p = __builtin_fma(p, f, 0x1.62e42fefa39efp-1);  // ln2
    p = __builtin_fma(p, f, 1.0);
    return ldexp(p, static_cast<int>(n));
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
    double w = exp2_nonpositive(-(dc * inv_rt) - dx * inv_x - dn * inv_n); // (inv_*: 1 / sigma^2 times log2 e)
    if (any_zero_sigma) w = isnan(w) ? w : 0.0;
    const double kw = k * w;
    const double t0 = tap.c[0] * kw, t1 = tap.c[1] * kw, t2 = tap.c[2] * kw;
    // a tap with a NaN in it is dropped (:100-104); a branch, not selects: such taps are rare and the lanes that have
    // none skip nothing
    if (!isnan(t0 * t0 + t1 * t1 + t2 * t2)) {
        acc[0] += t0;
        acc[1] += t1;
        acc[2] += t2;
        normalization += kw;
    }
}

// blur_pixel (src/denoise.cpp:84-116) for every pixel; Jacobi update into image_out.
//
// The taps of an iteration lie s = 1 + hole_width pixels apart (s = 1, 2, 4, 8 for the first four iterations), at
// 0, +-s and +-2s in either direction: the pixels of one row residue modulo s never look at rows of another.  STAGED: a
// block takes TILE_W consecutive pixels of TILE_H rows of one residue (rows r + s j) and first copies what their taps
// reach into LDS: those rows and two more on either side (4 rows of the same residue), TILE_W + 4 s pixels wide,
// coordinates reflected at the image border as the reference does (:20-38).  Every tap is then an LDS read, rows are
// staged as runs of consecutive pixels, and a pixel's 72 bytes come from memory 2.25 to 3 times per iteration instead
// of 25 times (the unstaged iterations were bound by the vector cache's 64 bytes per clock).  Iterations beyond the
// fourth (s >= 16) read their taps from memory.
constexpr int ATROUS_MAX_SPACING = 8;
constexpr int STAGE_ROWS = TILE_H + 4;

template<bool STAGED>
__global__ __launch_bounds__(TILE_W* TILE_H) void k_atrous(int W, int H, const double* __restrict__ image, double* __restrict__ image_out,
                                                           const double* __restrict__ position, const double* __restrict__ normal,
                                                           double sigma_rt, double sigma_x, double sigma_n, int hole_width, int tiles_per_residue) {
    // plane-major: nine planes of STAGE_ROWS x stage_w doubles; the lanes of a wave read consecutive doubles of one plane
    extern __shared__ double stage[];
    const int spacing = 1 + hole_width;
    const int stage_w = TILE_W + 4 * spacing, plane = STAGE_ROWS * stage_w;
    const int gx = blockIdx.x * TILE_W + threadIdx.x;
    int gy = blockIdx.y * TILE_H + threadIdx.y;
    if (STAGED) {
        // blockIdx.y = residue * tiles_per_residue + tile: rows residue + spacing * (TILE_H * tile + 0 .. TILE_H - 1)
        const int residue = blockIdx.y / tiles_per_residue, tile = blockIdx.y - residue * tiles_per_residue;
        const int first_row = residue + spacing * (TILE_H * tile); // of the block's own rows
        gy = first_row + spacing * static_cast<int>(threadIdx.y);
        const int x0 = blockIdx.x * TILE_W - 2 * spacing;
        for (int i = threadIdx.y * TILE_W + threadIdx.x; i < stage_w * STAGE_ROWS; i += TILE_W * TILE_H) {
            const int ty = i / stage_w, tx = i - ty * stage_w;
            // Coordinates beyond the far edges by more than one reflection only feed lanes outside the image.
            int x = reflect_coordinate(min(x0 + tx, 2 * (W - 1)), W), y = reflect_coordinate(min(first_row + spacing * (ty - 2), 2 * (H - 1)), H);
            x = max(x, 0);
            y = max(y, 0);
            const size_t q = 3 * (static_cast<size_t>(W) * static_cast<size_t>(y) + static_cast<size_t>(x));
            const int at = ty * stage_w + tx;
            for (int j = 0; j < 3; j++) {
                stage[(0 + j) * plane + at] = image[q + j];
                stage[(3 + j) * plane + at] = position[q + j];
                stage[(6 + j) * plane + at] = normal[q + j];
            }
        }
        __syncthreads();
    }
    if (gx >= W || gy >= H) return;
    const size_t p = static_cast<size_t>(W) * static_cast<size_t>(gy) + static_cast<size_t>(gx);
    auto fetch = [&](int x, int y, int tx, int ty) { // pixel (x, y) of the image = entry (tx, ty) of the staged tile
        Guide g;
        if (STAGED) {
            const int at = ty * stage_w + tx;
            for (int j = 0; j < 3; j++) {
                g.c[j] = stage[(0 + j) * plane + at];
                g.x[j] = stage[(3 + j) * plane + at];
                g.n[j] = stage[(6 + j) * plane + at];
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
    const int cx = static_cast<int>(threadIdx.x) + 2 * spacing, cy = static_cast<int>(threadIdx.y) + 2; // the pixel itself in the staged tile
    const Guide centre = fetch(gx, gy, cx, cy);
    const bool any_zero_sigma = sigma_rt == 0 || sigma_x == 0 || sigma_n == 0;
    // -d / sigma^2 as a product with log2(e) / sigma^2, the weight being a power of two then (a zero sigma is dealt
    // with by any_zero_sigma)
    const double LOG2_E = 1.4426950408889634;
    const double inv_rt = sigma_rt == 0 ? 0.0 : LOG2_E / (sigma_rt * sigma_rt), inv_x = sigma_x == 0 ? 0.0 : LOG2_E / (sigma_x * sigma_x),
                 inv_n = sigma_n == 0 ? 0.0 : LOG2_E / (sigma_n * sigma_n);
    double acc[3] = {0, 0, 0};
    double normalization = 0;
    for (int dx = -2; dx <= 2; dx++) {
        for (int dy = -2; dy <= 2; dy++) {
            const int ox = expand_kernel_idx(dx, hole_width), oy = expand_kernel_idx(dy, hole_width);
            // the staged tile holds reflected pixels at reflected places: the tap's offset is all that is needed there
            // (dx columns of `spacing` pixels, dy staged rows)
            const int x = STAGED ? 0 : reflect_coordinate(gx + ox, W), y = STAGED ? 0 : reflect_coordinate(gy + oy, H);
            const Guide tap = fetch(x, y, cx + ox, cy + dy);
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

// General odd kernel size (5 to 15).  The tile and a rim of (k - 1) / 2 pixels are staged in LDS (reflected at the
// image border like the reference's window, src/denoise.cpp:160-175), one channel at a time, as order-preserving
// 64-bit images of the doubles, so that every comparison is an integer one.  Every lane then selects the
// (k * k / 2)-th smallest value of its k x k window by selection around pivots: a pass counts the window values below
// and equal to the lane's pivot, which either is the median or bounds it from one side, and a second sweep takes the
// next pivot from the values still between the bounds (the first one at or after a place that moves from pass to
// pass, which is as good as a random one on image data).  About 2 ln(k^2) passes of 2 k^2 LDS reads.  (Measured at
// 1080p: 0.5 ms for 5 x 5 where counting ranks in registers, k^4 comparisons, took 1.7; 2.3 ms for 9 x 9 where a
// bit-by-bit selection over the same keys, 64 passes of k^2 reads, took 11.9.)
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

__global__ __launch_bounds__(TILE_W* TILE_H) void k_median_general(int W, int H, const double* __restrict__ image,
                                                                   double* __restrict__ image_out, int kernel_size,
                                                                   double threshold) {
    __shared__ unsigned long long tile[MEDIAN_STAGE_H * MEDIAN_STAGE_W];
    const int gx = blockIdx.x * TILE_W + threadIdx.x;
    const int gy = blockIdx.y * TILE_H + threadIdx.y;
    const bool inside = gx < W && gy < H;
    const int offset = (kernel_size - 1) / 2;
    const int size = kernel_size * kernel_size;
    const int target = size / 2;
    const int sw = TILE_W + kernel_size - 1, sh = TILE_H + kernel_size - 1;
    const int x0 = blockIdx.x * TILE_W - offset, y0 = blockIdx.y * TILE_H - offset;
    double med[3] = {0, 0, 0};
    for (int c = 0; c < 3; c++) {
        __syncthreads(); // the previous channel's tile is no longer read
        for (int i = threadIdx.y * TILE_W + threadIdx.x; i < sw * sh; i += TILE_W * TILE_H) {
            const int ty = i / sw, tx = i - ty * sw;
            // coordinates beyond the far edges by more than one reflection only feed lanes outside the image
            int x = reflect_coordinate(min(x0 + tx, 2 * (W - 1)), W), y = reflect_coordinate(min(y0 + ty, 2 * (H - 1)), H);
            x = max(x, 0);
            y = max(y, 0);
            tile[ty * MEDIAN_STAGE_W + tx] = order_key(image[3 * (static_cast<size_t>(W) * static_cast<size_t>(y) + static_cast<size_t>(x)) + c]);
        }
        __syncthreads();
        if (!inside) continue;
        const unsigned long long* keys = tile + threadIdx.y * MEDIAN_STAGE_W + threadIdx.x; // window value (wx, wy) at [wy * MEDIAN_STAGE_W + wx]
        // the median lies in [lo, hi] (keys; no value has key 0, and ~0 is the key of NaN, the largest)
        unsigned long long lo = 1, hi = ~0ull;
        unsigned long long pivot = keys[offset * MEDIAN_STAGE_W + offset]; // the pixel itself: often close to the median
        unsigned long long answer = pivot;
        bool done = false;
        for (int pass = 0;; pass++) { // (every unfinished lane's interval loses its pivot each pass: at most k^2 passes)
            int less = 0, equal = 0;
            for (int wy = 0; wy < kernel_size; wy++)
                for (int wx = 0; wx < kernel_size; wx++) {
                    const unsigned long long v = keys[wy * MEDIAN_STAGE_W + wx];
                    less += v < pivot ? 1 : 0;
                    equal += v == pivot ? 1 : 0;
                }
            if (!done) {
                if (less <= target && target < less + equal) {
                    answer = pivot;
                    done = true;
                }
                else if (target < less) hi = pivot - 1;
                else lo = pivot + 1;
            }
            if (__all(done ? 1 : 0)) break;
            // the next pivot: the first value within [lo, hi] at or after window place `from`, else the first one at all
            // (there is one: the median itself lies in the interval)
            const int from = (pass * 37 + 11) % size;
            unsigned long long first = 0, after = 0;
            bool have_first = false, have_after = false;
            int place = 0;
            for (int wy = 0; wy < kernel_size; wy++)
                for (int wx = 0; wx < kernel_size; wx++, place++) {
                    const unsigned long long v = keys[wy * MEDIAN_STAGE_W + wx];
                    const bool within = v >= lo && v <= hi;
                    if (within && !have_first) {
                        first = v;
                        have_first = true;
                    }
                    if (within && place >= from && !have_after) {
                        after = v;
                        have_after = true;
                    }
                }
            if (!done) pivot = have_after ? after : first;
        }
        med[c] = key_value(answer);
    }
    if (!inside) return;
    const size_t p = static_cast<size_t>(W) * static_cast<size_t>(gy) + static_cast<size_t>(gx);
    const double cur[3] = {image[3 * p], image[3 * p + 1], image[3 * p + 2]};
    const double d0 = cur[0] - med[0], d1 = cur[1] - med[1], d2 = cur[2] - med[2];
    const bool replace = sqrt(d0 * d0 + d1 * d1 + d2 * d2) > threshold;
    image_out[3 * p] = replace ? med[0] : cur[0];
    image_out[3 * p + 1] = replace ? med[1] : cur[1];
    image_out[3 * p + 2] = replace ? med[2] : cur[2];
}

} // namespace

void launch_atrous(hipStream_t s, int width, int height, const double* image_in, double* image_out, const double* position,
                   const double* normal, double sigma_rt, double sigma_x, double sigma_n, int hole_width) {
    const int spacing = 1 + hole_width, reach = 2 * spacing;
    // staging needs the rim's reflected coordinates inside the image: images larger than the reach
    if (spacing <= ATROUS_MAX_SPACING && width > reach && height > reach) {
        const int rows_per_residue = (height + spacing - 1) / spacing; // (of residue 0: the others have as many or one fewer)
        const int tiles_per_residue = (rows_per_residue + TILE_H - 1) / TILE_H;
        const dim3 grid((width + TILE_W - 1) / TILE_W, spacing * tiles_per_residue);
        const size_t lds = static_cast<size_t>(9) * STAGE_ROWS * (TILE_W + 4 * spacing) * sizeof(double);
        hipLaunchKernelGGL(k_atrous<true>, grid, dim3(TILE_W, TILE_H), lds, s, width, height, image_in, image_out, position, normal, sigma_rt,
                           sigma_x, sigma_n, hole_width, tiles_per_residue);
    }
    else {
        const dim3 grid((width + TILE_W - 1) / TILE_W, (height + TILE_H - 1) / TILE_H);
        hipLaunchKernelGGL(k_atrous<false>, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, position, normal, sigma_rt,
                           sigma_x, sigma_n, hole_width, 1);
    }
}

void launch_median(hipStream_t s, int width, int height, const double* image_in, double* image_out, int kernel_size,
                   double threshold) {
    dim3 grid((width + TILE_W - 1) / TILE_W, (height + TILE_H - 1) / TILE_H);
    if (kernel_size == 3 && width >= 2 && height >= 2)
        hipLaunchKernelGGL(k_median3, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, threshold);
    else
        hipLaunchKernelGGL(k_median_general, grid, dim3(TILE_W, TILE_H), 0, s, width, height, image_in, image_out, kernel_size, threshold);
}

} // namespace ptk
