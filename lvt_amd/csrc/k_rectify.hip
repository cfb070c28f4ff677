// k_rectify.hip -- EuRoC pre-step: cv::initUndistortRectifyMap + cv::remap(INTER_LINEAR) as the
// reference's example runs them before every track() (examples/euroc/euroc_example.cpp:95-107,142-143 there).
//
//   k_rectify_map : one thread per image ROW.  OpenCV accumulates the homogeneous coordinate along the row
//                   (_x += ir[0] per column), so the columns of a row are a sequential fp64 recurrence; rows are
//                   independent.  Runs once per rectifier; writes the two CV_32FC1 maps.
//   k_rectify     : one thread per 4 output pixels: map -> 1/32-px fixed point (round-half-even), 2x2 gather with
//                   BORDER_CONSTANT 0, 15-bit weights, one 32-bit coalesced store.  8 B of map + 1 B out + <= 4 B of source
//                   per pixel: HBM-bound by construction, but one 752x480 image is 4.7 MB -- 0.6 us at 8 TB/s -- so a
//                   single launch is latency-bound like the rest of the single-sequence chain.
#include "lvt_dev.h"

namespace lvt {

struct RectifyArgs {
    double ir[9];             // (Pnew * R)^-1
    double fx, fy, u0, v0;    // distorted camera
    double k1, k2, p1, p2, k3;
    int w, h;
};

__global__ __launch_bounds__(64) void k_rectify_map(RectifyArgs a, float *map1, float *map2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.h) return;
    const double *ir = a.ir;
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    for (int j = 0; j < a.w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double iw = 1. / _w, x = _x * iw, y = _y * iw;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((a.k3 * r2 + a.k2) * r2 + a.k1) * r2) / (1 + ((0 * r2 + 0) * r2 + 0) * r2);
        const double xd = (x * kr + a.p1 * _2xy + a.p2 * (r2 + 2 * x2) + 0 * r2 + 0 * r2 * r2);
        const double yd = (y * kr + a.p1 * (r2 + 2 * y2) + a.p2 * _2xy + 0 * r2 + 0 * r2 * r2);
        const double u = a.fx * 1. * xd + a.u0;
        const double v = a.fy * 1. * yd + a.v0;
        map1[(size_t)i * a.w + j] = (float)u;
        map2[(size_t)i * a.w + j] = (float)v;
    }
}

__device__ __forceinline__ int remap_one(const uint8_t *src, int sw, int sh, int sstep, float mx, float my) {
    const int sxf = __float2int_rn(mx * 32.f), syf = __float2int_rn(my * 32.f);  // cvRound: round half to even
    const int sx = min(max(sxf >> 5, -32768), 32767), sy = min(max(syf >> 5, -32768), 32767), ax = sxf & 31, ay = syf & 31;
    int w0 = (32 - ax) * (32 - ay) * 32, w1 = ax * (32 - ay) * 32, w2 = (32 - ax) * ay * 32, w3 = ax * ay * 32;
    if (ax == 0 && ay == 0) w0 = 32767, w3 = 1;  // initInterTab2D: 32768 saturates to short, the sum fix-up lands on the last weight
    int v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if ((unsigned)sx < (unsigned)max(sw - 1, 0) && (unsigned)sy < (unsigned)max(sh - 1, 0)) {
        const uint8_t *S = src + (size_t)sy * sstep + sx;
        v0 = S[0], v1 = S[1], v2 = S[sstep], v3 = S[sstep + 1];
    } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
        return 0;
    } else {
        const uint8_t *S0 = src + (ptrdiff_t)sy * sstep, *S1 = src + (ptrdiff_t)(sy + 1) * sstep;
        if (sx >= 0 && sy >= 0) v0 = S0[sx];
        if (sx + 1 < sw && sy >= 0) v1 = S0[sx + 1];
        if (sx >= 0 && sy + 1 < sh) v2 = S1[sx];
        if (sx + 1 < sw && sy + 1 < sh) v3 = S1[sx + 1];
    }
    const int r = (v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15;
    return min(max(r, 0), 255);
}

// dst_pitch % 4 == 0; the padding columns of dst (if any) are written as zero
__global__ __launch_bounds__(256) void k_rectify(const uint8_t *src, int sw, int sh, int sstep, const float *map1, const float *map2, int dw, int dh,
                                                 uint8_t *dst, int dst_pitch) {
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x4 >= dst_pitch || y >= dh) return;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = x4 + k;
        if (x < dw) packed |= (uint32_t)remap_one(src, sw, sh, sstep, map1[(size_t)y * dw + x], map2[(size_t)y * dw + x]) << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(dst + (size_t)y * dst_pitch + x4) = packed;
}

}  // namespace lvt
