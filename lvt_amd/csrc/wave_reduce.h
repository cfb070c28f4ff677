// Sums of up to 32 doubles per lane over the 64 lanes of a wavefront as a REDUCE-SCATTER: every level halves the number of live registers
// instead of carrying all of them through six butterfly steps (28 values: 14 + 7 + 4 + 2 + 1 + 1 = 29 additions instead of 168).
//   level 1  lanes l <-> l ^ 32 : v_permlane32_swap (gfx950) puts the two halves of TWO values side by side, one addition finishes the level for both
//   level 2  rows  r <-> r ^ 1  : v_permlane16_swap, the same for 16-lane rows
//   level 3  l <-> l ^ 8        : DPP row_ror:8;  level 4  l <-> 7 - (l & 7) : DPP row_half_mirror;  level 5 / 6  l ^ 2, l ^ 1 : DPP quad_perm
// On return lane l holds the wavefront's total of value wave_rs_index(l) (both lanes of a pair l, l ^ 1 hold the same one).
// The order of the additions is fixed by the lane numbers alone: the result is deterministic.
#pragma once
#include <hip/hip_runtime.h>

namespace lvt {

__device__ __forceinline__ int wave_rs_index(int lane) {
    return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4);
}

typedef unsigned int wave_rs_u2 __attribute__((ext_vector_type(2)));

// x: lanes 0-31 keep their x and receive the x of lane + 32; lanes 32-63 keep their y and receive the y of lane - 32
__device__ __forceinline__ double wave_rs_pair32(double x, double y) {
    const wave_rs_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const wave_rs_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double wave_rs_pair16(double x, double y) {
    const wave_rs_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const wave_rs_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
template <int CTRL>
__device__ __forceinline__ double wave_rs_dpp(double s) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(s), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(s), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// lanes whose `bit` is clear keep x and receive their partner's x, the others keep y and receive their partner's y
template <int CTRL>
__device__ __forceinline__ double wave_rs_pair_dpp(double x, double y, bool bit) {
    const double keep = bit ? y : x, send = bit ? x : y;
    return keep + wave_rs_dpp<CTRL>(send);
}

template <int NV>
__device__ __forceinline__ double wave_reduce_scatter(const double (&v)[NV]) {
    static_assert(NV >= 1 && NV <= 32, "wave_reduce_scatter: at most 32 values");
    const int lane = threadIdx.x & 63;
    constexpr int N1 = (NV + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2;
    double a[N1], b[N2], c[N3], d[N4];
#pragma unroll
    for (int i = 0; i < N1; i++) a[i] = wave_rs_pair32(v[2 * i], (2 * i + 1 < NV) ? v[2 * i + 1] : 0.0);
#pragma unroll
    for (int i = 0; i < N2; i++) b[i] = wave_rs_pair16(a[2 * i], (2 * i + 1 < N1) ? a[2 * i + 1] : 0.0);
#pragma unroll
    for (int i = 0; i < N3; i++) c[i] = wave_rs_pair_dpp<0x128>(b[2 * i], (2 * i + 1 < N2) ? b[2 * i + 1] : 0.0, (lane & 8) != 0);   // row_ror:8
#pragma unroll
    for (int i = 0; i < N4; i++) d[i] = wave_rs_pair_dpp<0x141>(c[2 * i], (2 * i + 1 < N3) ? c[2 * i + 1] : 0.0, (lane & 4) != 0);   // row_half_mirror
    double e = wave_rs_pair_dpp<0x4E>(d[0], (N4 > 1) ? d[N4 > 1 ? 1 : 0] : 0.0, (lane & 2) != 0);                                       // quad_perm [2,3,0,1]
    e += wave_rs_dpp<0xB1>(e);                                                                                                          // quad_perm [1,0,3,2]
    return e;
}

}  // namespace lvt
