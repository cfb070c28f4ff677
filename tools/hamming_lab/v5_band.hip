// v5_band.hip (round 4 experiment, NOT part of the product: measured 231 us against the shipped kernel's 218 -- profiles/r04_hamming_analysis.md)
// -- the radius-mode batched matcher (k_hamming.hip's MODE 0) on a different spatial structure: ONE contiguous candidate
// range per query, no position arithmetic inside the loops.
//
// What the matcher must return (lvt_image_features_struct.cpp:68-120 + cv::BFMatcher::knnMatch(k = 2, mask)): for every query the two
// unflagged train features with the smallest (Hamming distance, index) among those with dx*dx + dy*dy < r2 (fp32, strict).  The reference walks the
// hash cells [cy +- csr] x [cx +- csr] around the query, but with csr = ceil(r / 25) that window always CONTAINS the circle (|dx| < r <= 25 csr), and
// the reference's train features lie inside the image: the candidate set IS the circle, and any spatial index that offers a superset of the circle
// to the exact fp32 test returns the reference's answer.  k_hamming_batched<0, ...> bins by the reference's own 25-px cells, so a query's window is
// three LDS ranges and every candidate costs five instructions of virtual-index -> position arithmetic (13 % of the kernel's VALU issue, the wall
// that launch runs into: profiles/r03_hamming_analysis.md).  Here:
//   * rows of height RH (just above r) and x-bins of width XB = RH / 2; BAND b holds the train features of rows b - 1, b, b + 1 ordered by x-bin
//     (every feature is listed in up to three bands: 3 N entries of x, y, descriptor address = 30 KB at N = 1500);
//   * |ty - py| < r  =>  |row(ty) - row(py)| <= 1  (row() = clamp(floor(y / RH)) is monotone and 1-Lipschitz in units of RH > r), and
//     |tx - px| < r  =>  xbin(px - r) <= xbin(tx) <= xbin(px + r)  (fl() and xbin() are monotone), so the circle of a query lies inside
//     band row(py), bins xbin(px - r) .. xbin(px + r): ONE range of the band's list -- ~15 candidates at KITTI density against 18 in a 3 x 3
//     cell window;
//   * stage A walks that range two candidates per step in SoA form (x and y of neighbours are neighbours in LDS): packed fp32 subtract / multiply /
//     add give both squared distances, their comparison with r2 is the SIGN of fl(d2 - r2) (exact: the difference of two floats within a factor
//     of two of each other is exact, and otherwise far from zero), shifted into the mask by one v_alignbit each: 4 instructions per candidate
//     where the flattened three-range walk needs 10;
//   * stage B walks the set bits as before (descriptor address from a u16 list entry: no position arithmetic either).
// The train descriptors wait in registers while the lists are built and stage A runs -- the list arrays and the descriptors then SHARE their LDS
// (66 KB per problem instead of 76) -- and enter LDS in input order: linear, conflict-free stores instead of a scatter.
// Output identical to k_hamming_batched on every input (tests/test_gpu_primitives.py::test_hamming_match_batched*; tools/hamming_lab/lab.py diffs
// all 8.2 M queries of the bench launch).
#pragma once
#include "k_hamming.hip"
#include <algorithm>
#include <cmath>

namespace lvt {

struct BandArgs {
    HammingArgs h;
    float inv_rh, inv_xb;  // 1 / row height, 1 / x-bin width
    float r_up;            // radius, rounded up
    int nbands, nxb;       // rows (= bands), x-bins per row
};

constexpr int HBN_MASK = 31;     // candidates of a range that fit the stage-A mask (one bit is lost to block alignment); longer ranges are matched in one stage (exact, rare)
constexpr int HBN_BASE_BITS = 13;  // list positions: 3 N <= 8191
constexpr int HBN_Q_BITS = 12;     // query index: M <= 4096

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));  // (arrays of the native vector type are promoted to registers; HIP's uint4 struct was not)

// (bcnt_acc: k_hamming.hip)
// maximum over the wavefront (all lanes active), broadcast: DPP row shifts + row broadcasts, then lane 63
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// 16-byte LDS read at a byte address held in a register (the list entries ARE such addresses)
typedef v4u __attribute__((address_space(3))) *lds_v4u_ptr;
__device__ __forceinline__ v4u lds_v4u(uint32_t addr) { return *(lds_v4u_ptr)addr; }
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

static inline size_t hamming_band_lds_bytes(int N, int M, int nbins) {
    const size_t t3 = 3 * (size_t)N + 2 + 36;  // (+ what the padded walk of the last range may read)
    const size_t region_d = std::max((size_t)32 * N, 8 * t3 + 4 * ((size_t)nbins + 2));
    return ((region_d + 15) & ~(size_t)15) + ((2 * t3 + 15) & ~(size_t)15) + 12 * (size_t)M + 16;
}

// grid of the band lists for a launch: rows just taller than the radius (the 1-Lipschitz argument above needs r * inv_rh < 1 by more than the products'
// rounding), x-bins half as wide, at most 17 x 101 bins however small the radius is.  Returns false when the launch does not fit this kernel.
static inline bool hamming_band_setup(BandArgs &a) {
    const float r2 = a.h.r2;
    if (!(r2 >= 0.0f) || !(r2 < 1e30f) || a.h.N < 1 || a.h.N > 2729 || a.h.M < 1 || a.h.M > HB_MMAX || a.h.img_rows < 1 || a.h.img_cols < 1) return false;
    float r = std::sqrt(r2);
    r = std::nextafter(std::nextafter(r, 1e38f), 1e38f);  // >= the exact square root
    const float rh = std::max(r * 1.0005f, (float)a.h.img_rows / 16.0f);
    const float xb = std::max(rh * 0.5f, (float)a.h.img_cols / 100.0f);
    a.r_up = r;
    a.inv_rh = 1.0f / rh, a.inv_xb = 1.0f / xb;
    a.nbands = std::min(17, (int)std::ceil((double)a.h.img_rows / rh) + 1);
    a.nxb = std::min(101, (int)std::ceil((double)a.h.img_cols / xb) + 1);
    return hamming_band_lds_bytes(a.h.N, a.h.M, a.nbands * a.nxb) <= 80 * 1024 - 512;
}

template <int QPT, int TPT>
__global__ __launch_bounds__(HB_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_hamming_band(BandArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int N = a.h.N, M = a.h.M;
    const int nxb = a.nxb, nbands = a.nbands, nb = nxb * nbands;
    const int T3 = 3 * N + 2 + 36;
    // region D: [t3x | t3y | cnt] while the lists are built and stage A runs, then [dlo | dhi]
    const size_t region_d = ((size_t)((32 * (size_t)N > 8 * (size_t)T3 + 4 * ((size_t)nb + 2)) ? 32 * (size_t)N : 8 * (size_t)T3 + 4 * ((size_t)nb + 2)) + 15) & ~(size_t)15;
    // list entries in blocks of two: [x0 x1 y0 y1]: one 16-byte read hands stage A the x pair and the y pair of two neighbours as register pairs
    float *t3 = reinterpret_cast<float *>(smem);
    int *cnt = reinterpret_cast<int *>(t3 + 2 * T3);
    uint4 *s_dlo = reinterpret_cast<uint4 *>(smem);
    uint4 *s_dhi = s_dlo + N;
    uint16_t *t3i = reinterpret_cast<uint16_t *>(smem + region_d);
    uint32_t *slot = reinterpret_cast<uint32_t *>(smem + region_d + ((2 * (size_t)T3 + 15) & ~(size_t)15));
    __shared__ int s_scan[32];
    __shared__ int s_hist[HB_HIST];

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const v4u *td = reinterpret_cast<const v4u *>(a.h.t_desc + (size_t)b * N * 4);
    const float2 *txy = a.h.t_xy + (size_t)b * N;
    const uint8_t *tf = a.h.t_flag + (size_t)b * N;
    const uint4 *qd = reinterpret_cast<const uint4 *>(a.h.q_desc + (size_t)b * M * 4);
    const float2 *qxy = a.h.q_xy + (size_t)b * M;
    int4 *out = a.h.out + (size_t)b * M;
    const float r2 = a.h.r2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;  // LDS address of the dynamic segment (the low half of its flat address)

    long long *dbg = (a.h.dbg && blockIdx.x == gridDim.x / 2 && tid == 0) ? a.h.dbg : nullptr;
    if (dbg) dbg[0] = clock64();
    __builtin_amdgcn_s_setprio(3);

    // ---- 1. every HBM load of the problem's sort phase, issued back to back (clamped indices, no branches)
    float2 tp[TPT];
    uint8_t tfl[TPT];
    v4u tdlo[TPT], tdhi[TPT];
    float2 qp[QPT];
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        const int jc = max(min(tid + k * HB_THREADS, N - 1), 0);
        tp[k] = txy[jc];
        tfl[k] = tf[jc];
        tdlo[k] = td[2 * jc];
        tdhi[k] = td[2 * jc + 1];
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) qp[k] = qxy[min(tid + k * HB_THREADS, M - 1)];
#if defined(LAB_STOP) && LAB_STOP == 3
    {   // the launch's memory traffic alone: every load of the problem, folded into the stores the real kernel makes (nothing else)
        uint32_t f = 0;
#pragma unroll
        for (int k = 0; k < TPT; k++)
            f ^= tdlo[k].x ^ tdlo[k].y ^ tdlo[k].z ^ tdlo[k].w ^ tdhi[k].x ^ tdhi[k].y ^ tdhi[k].z ^ tdhi[k].w ^ __float_as_uint(tp[k].x) ^ __float_as_uint(tp[k].y) ^ tfl[k];
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            const int q = tid + k * HB_THREADS;
            const v4u w0 = reinterpret_cast<const v4u *>(qd)[2 * min(q, M - 1)], w1 = reinterpret_cast<const v4u *>(qd)[2 * min(q, M - 1) + 1];
            f ^= w0.x ^ w0.y ^ w0.z ^ w0.w ^ w1.x ^ w1.y ^ w1.z ^ w1.w ^ __float_as_uint(qp[k].x) ^ __float_as_uint(qp[k].y);
            if (q < M) out[q] = make_int4((int)f, 0, 0, 0);
        }
        return;
    }
#endif
    for (int i = tid; i <= nb + 1; i += HB_THREADS) cnt[i] = 0;
    if (tid < HB_HIST) s_hist[tid] = 0;
    __syncthreads();
    if (dbg) dbg[1] = clock64();

    // ---- 2. count: a feature of row r is listed in the bands r - 1, r, r + 1; the counting atomic returns its rank inside the (band, x-bin) list
    const float fxmax = (float)(nxb - 1), fbmax = (float)(nbands - 1);
    int tcell[TPT], trow[TPT], trank[TPT][3];
    bool tv[TPT];
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        const float x = tp[k].x, y = tp[k].y;
        // a feature with a non-finite coordinate is nobody's candidate (its dx or dy is inf or NaN: d2 < r2 is false)
        tv[k] = (tid + k * HB_THREADS < N) & (tfl[k] == 0) & (fabsf(x) < __builtin_inff()) & (fabsf(y) < __builtin_inff());
        const int r = (int)fminf(fmaxf(floorf(y * a.inv_rh), 0.0f), fbmax);
        const int xb = (int)fminf(fmaxf(floorf(x * a.inv_xb), 0.0f), fxmax);
        tcell[k] = r * nxb + xb, trow[k] = r;
        trank[k][0] = trank[k][1] = trank[k][2] = 0;
        if (tv[k]) {
            trank[k][1] = atomicAdd(&cnt[tcell[k]], 1);
            if (r > 0) trank[k][0] = atomicAdd(&cnt[tcell[k] - nxb], 1);
            if (r < nbands - 1) trank[k][2] = atomicAdd(&cnt[tcell[k] + nxb], 1);
        }
    }
    __syncthreads();
    if (dbg) dbg[2] = clock64();
    {   // counts -> exclusive starts, in place (two entries per thread: nb + 1 <= 2048); entry nb receives the total
        const int i0 = 2 * tid;
        const int v0 = (i0 <= nb) ? cnt[i0] : 0, v1 = (i0 + 1 <= nb) ? cnt[i0 + 1] : 0;
        int total;
        const int run = block_excl_scan(v0 + v1, s_scan, &total);
        if (i0 <= nb) cnt[i0] = run;
        if (i0 + 1 <= nb) cnt[i0 + 1] = run + v0;
    }
    __syncthreads();
    if (dbg) dbg[3] = clock64();
    // ---- 3. scatter into the band lists; the queries' ranges and their first sort (by range length) need the starts only
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        if (tv[k]) {
            const uint16_t da = (uint16_t)(((tid + k * HB_THREADS) << 4) + lds0);  // LDS address of the descriptor's low half in stage B's array
            const int r = trow[k];
#define LVT_BAND_PUT(pos_)                                            \
    {                                                                 \
        const int pos = (pos_);                                       \
        float *e = t3 + (((pos & ~1) << 1) | (pos & 1));              \
        e[0] = tp[k].x, e[2] = tp[k].y, t3i[pos] = da;                \
    }
            LVT_BAND_PUT(cnt[tcell[k]] + trank[k][1])
            if (r > 0) LVT_BAND_PUT(cnt[tcell[k] - nxb] + trank[k][0])
            if (r < nbands - 1) LVT_BAND_PUT(cnt[tcell[k] + nxb] + trank[k][2])
#undef LVT_BAND_PUT
        }
    }
    int qw[QPT], qkey[QPT], qrank[QPT];
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        const int q = tid + k * HB_THREADS;
        qw[k] = 0, qkey[k] = 0, qrank[k] = 0;
        if (q < M) {
            const float px = qp[k].x, py = qp[k].y;
            const bool fin = (fabsf(px) < __builtin_inff()) & (fabsf(py) < __builtin_inff());
            const int rb = (int)fminf(fmaxf(floorf(py * a.inv_rh), 0.0f), fbmax);
            const int x0 = (int)fminf(fmaxf(floorf((px - a.r_up) * a.inv_xb), 0.0f), fxmax);
            const int x1 = (int)fminf(fmaxf(floorf((px + a.r_up) * a.inv_xb), 0.0f), fxmax);
            const int base = cnt[rb * nxb + x0];
            const int len = fin ? cnt[rb * nxb + x1 + 1] - base : 0;
            const int lc = min(len, HB_HIST - 1);
            qkey[k] = HB_HIST - 1 - lc;
            qw[k] = base | (q << HBN_BASE_BITS) | ((len > HBN_MASK ? 1 : 0) << (HBN_BASE_BITS + HBN_Q_BITS)) | (lc << (HBN_BASE_BITS + HBN_Q_BITS + 1));
            qrank[k] = atomicAdd(&s_hist[qkey[k]], 1);
        }
    }
    __syncthreads();
    if (dbg) dbg[4] = clock64();
    if (wv == 0) {
        const int v = s_hist[lane];
        s_hist[lane] = wave_incl_scan(v) - v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        if (tid + k * HB_THREADS < M) {
            uint32_t *s = slot + 3 * (s_hist[qkey[k]] + qrank[k]);
            s[0] = (uint32_t)qw[k], s[1] = __float_as_uint(qp[k].x), s[2] = __float_as_uint(qp[k].y);
        }
    }
    __syncthreads();
    if (dbg) dbg[5] = clock64();
    __builtin_amdgcn_s_setprio(0);
#if defined(LAB_STOP) && LAB_STOP == 1
    if (tid < M) out[tid] = make_int4((int)slot[3 * tid], 0, 0, 0);
    return;
#endif

    // rounds of 1024 queries in sorted order; odd rounds reverse the wave order so that every wave gets a similar sum
    auto sorted_slot = [&](int j) -> int { return j * HB_THREADS + ((j & 1) ? (HB_WAVES - 1 - wv) : wv) * 64 + lane; };

    // ---- 4a. the radius test over every candidate of the range: candidate v ends up in bit v of the mask.  The wavefront walks its 64 ranges from
    //          a COMMON padded length downwards (the queries are sorted by length: the lanes of a wave differ by a candidate or two), four candidates
    //          per trip: one address register, immediate offsets, no per-lane loop condition; what a lane reads beyond its own range lands in bits
    //          >= len and is cut off.  x and y of candidates i - 1 and i come as register PAIRS (ds_read2_b32 at dword offsets 0 / 2 and 1 / 3 of
    //          the (x, y) array): packed subtract, multiply, add, subtract r2 -- and the comparison is the sign of that last difference.
    if (tid < HB_HIST) s_hist[tid] = 0;
    uint32_t aw[QPT], amask[QPT];
    int akey[QPT], arank[QPT], alen[QPT];
    const v2f vr2 = {r2, r2};
#pragma unroll
    for (int j = 0; j < QPT; j++) {
        const int s = sorted_slot(j);
        const bool live = s < M;
        const uint32_t w = live ? slot[3 * s] : 0u;
        const float px = live ? __uint_as_float(slot[3 * s + 1]) : 0.0f, py = live ? __uint_as_float(slot[3 * s + 2]) : 0.0f;
        const int base = (int)(w & ((1u << HBN_BASE_BITS) - 1));
        const bool lng = (w >> (HBN_BASE_BITS + HBN_Q_BITS)) & 1u;
        int len = (int)(w >> (HBN_BASE_BITS + HBN_Q_BITS + 1));
        // the walk starts on a block boundary: an odd base brings the entry in front of the range along (bit 0, shifted out below)
        const int k0 = base & 1;
        const int padded = (wave_max_u32(lng ? 0u : (uint32_t)(len + k0)) + 3) & ~3;  // wave-uniform, <= 32
        uint32_t mask = 0;
        {
            const v2f vpx = {px, px}, vpy = {py, py};
            const v4f *blk = reinterpret_cast<const v4f *>(t3) + (base >> 1);
#pragma unroll 2
            for (int i = padded - 4; i >= 0; i -= 4) {
                const v4f B1 = blk[(i >> 1) + 1], B0 = blk[i >> 1];
                const v2f dx1 = B1.xy - vpx, dy1 = B1.zw - vpy, dx0 = B0.xy - vpx, dy0 = B0.zw - vpy;
                const v2f t1 = (dx1 * dx1 + dy1 * dy1) - vr2;  // (-ffp-contract=off: mul, mul, add, sub -- the reference's roundings, then the sign)
                const v2f t0 = (dx0 * dx0 + dy0 * dy0) - vr2;
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(t1.y), 31);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(t1.x), 31);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(t0.y), 31);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(t0.x), 31);
            }
            mask = (mask >> k0) & ((1u << len) - 1u);  // (len <= 31 here)
        }
        if (lng) {  // more candidates than the mask holds: stage B matches this query in one stage; it needs the true length
            const int rb = (int)fminf(fmaxf(floorf(py * a.inv_rh), 0.0f), fbmax);
            const int x1 = (int)fminf(fmaxf(floorf((px + a.r_up) * a.inv_xb), 0.0f), fxmax);
            len = cnt[rb * nxb + x1 + 1] - base;
            mask = 0;
        }
        aw[j] = live ? w : 0xFFFFFFFFu, amask[j] = mask, alen[j] = len, arank[j] = 0;
        akey[j] = lng ? 0 : HB_HIST - 1 - min((int)__popc(mask), HB_HIST - 1);
    }
    // ---- 4b. the queries again, sorted by the number of candidates inside the circle; the descriptors move into the lists' LDS
    __syncthreads();  // every stage-A read of the lists and of the slots is done; s_hist is zeroed
#pragma unroll
    for (int j = 0; j < QPT; j++)
        if (aw[j] != 0xFFFFFFFFu) arank[j] = atomicAdd(&s_hist[akey[j]], 1);
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        const int jn = tid + k * HB_THREADS;
        if (jn < N) reinterpret_cast<v4u *>(s_dlo)[jn] = tdlo[k], reinterpret_cast<v4u *>(s_dhi)[jn] = tdhi[k];
    }
    __syncthreads();
    if (wv == 0) {
        const int v = s_hist[lane];
        s_hist[lane] = wave_incl_scan(v) - v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < QPT; j++)
        if (aw[j] != 0xFFFFFFFFu) {
            uint32_t *s = slot + 3 * (s_hist[akey[j]] + arank[j]);
            s[0] = aw[j], s[1] = amask[j], s[2] = (uint32_t)alen[j];
        }
    __syncthreads();
    if (dbg) dbg[7] = clock64();
#if defined(LAB_STOP) && LAB_STOP == 2
    if (tid < M) out[tid] = make_int4((int)slot[3 * tid + 1], 0, 0, 0);
    return;
#endif

    const uint32_t hi_off = (uint32_t)N * 16;  // s_dhi - s_dlo in bytes
#pragma unroll 1
    for (int j = 0; j < QPT; j++) {
        const int s = sorted_slot(j);
        if (s >= M) continue;
        const uint32_t w = slot[3 * s], mk = slot[3 * s + 1];
        const int base = (int)(w & ((1u << HBN_BASE_BITS) - 1));
        const int q = (int)((w >> HBN_BASE_BITS) & ((1u << HBN_Q_BITS) - 1));
        const bool lng = (w >> (HBN_BASE_BITS + HBN_Q_BITS)) & 1u;
        const v4u w0 = reinterpret_cast<const v4u *>(qd)[2 * q], w1 = reinterpret_cast<const v4u *>(qd)[2 * q + 1];
        const uint64_t d0 = ((uint64_t)w0.y << 32) | w0.x, d1 = ((uint64_t)w0.w << 32) | w0.z;
        const uint64_t d2 = ((uint64_t)w1.y << 32) | w1.x, d3 = ((uint64_t)w1.w << 32) | w1.z;
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        const uint16_t *ci = t3i + base;
        if (!lng) {
            uint32_t m = mk;
            if (m) {
                // set bits of the mask = candidates inside the circle, two deep in flight: while candidate c's distance is computed, the descriptor
                // of c + 1 and the list entry (descriptor address) of c + 2 are on their way.  A lane that has run out of bits re-reads its first
                // candidate (valid addresses, results unused).  Two copies of the step with the register sets swapped: no moves at the loop end.
                const int bf = __ffs((int)m) - 1;
                m &= m - 1;
                uint32_t adA = ci[bf], adB, adn;
                bool hasB = m != 0, hasA, hasn;
                {
                    const int bi = hasB ? (__ffs((int)m) - 1) : bf;
                    m &= m - 1;
                    adB = ci[bi];
                }
                v4u a0 = lds_v4u(adA), a1 = lds_v4u(adA + hi_off), b0, b1;
#define LVT_BAND_STEP(C0, C1, ADC, N0, N1, ADN, HASN)                                                                                  \
    {                                                                                                                                  \
        hasn = m != 0;                                                                                                                 \
        const int bi = hasn ? (__ffs((int)m) - 1) : bf;                                                                                \
        m &= m - 1;                                                                                                                    \
        adn = ci[bi];                                                                                                                  \
        N0 = lds_v4u(ADN), N1 = lds_v4u(ADN + hi_off);                                                                                 \
        asm volatile("" ::: "memory"); /* the loads above stay above: in flight while this candidate's distance is formed */            \
        uint32_t d = bcnt_acc(C0.x ^ w0.x, 0u);                                                                                        \
        d = bcnt_acc(C0.y ^ w0.y, d), d = bcnt_acc(C0.z ^ w0.z, d), d = bcnt_acc(C0.w ^ w0.w, d);                                      \
        d = bcnt_acc(C1.x ^ w1.x, d), d = bcnt_acc(C1.y ^ w1.y, d), d = bcnt_acc(C1.z ^ w1.z, d), d = bcnt_acc(C1.w ^ w1.w, d);        \
        const uint32_t key = (d << 16) | ADC; /* (the descriptor's LDS address orders like its index) */                               \
        k2 = umed3(k1, k2, key);                                                                                                       \
        k1 = min(k1, key);                                                                                                             \
    }
                for (;;) {
                    LVT_BAND_STEP(a0, a1, adA, b0, b1, adB, hasB)
                    if (!hasB) break;
                    adA = adn, hasA = hasn;
                    LVT_BAND_STEP(b0, b1, adB, a0, a1, adA, hasA)
                    if (!hasA) break;
                    adB = adn, hasB = hasn;
                }
#undef LVT_BAND_STEP
            }
        } else {  // the whole range, filter and distance in one pass (rare: a range of more than 32 candidates)
            const float2 p = qxy[q];
            const int len = (int)slot[3 * s + 2];
            // (the lists are gone: the coordinates come from HBM again)
            for (int v = 0; v < len; v++) {
                const uint32_t ad = ci[v];
                const float2 r = txy[(ad - lds0) >> 4];
                const float dx = r.x - p.x, dy = r.y - p.y;
                const bool ok = (dx * dx + dy * dy) < r2;
                const v4u a0 = lds_v4u(ad), a1 = lds_v4u(ad + hi_off);
                const int d = __popcll(d0 ^ (((uint64_t)a0.y << 32) | a0.x)) + __popcll(d1 ^ (((uint64_t)a0.w << 32) | a0.z)) +
                              __popcll(d2 ^ (((uint64_t)a1.y << 32) | a1.x)) + __popcll(d3 ^ (((uint64_t)a1.w << 32) | a1.z));
                const uint32_t key = ok ? (((uint32_t)d << 16) | ad) : 0xFFFFFFFFu;
                k2 = umed3(k1, k2, key);
                k1 = min(k1, key);
            }
        }
        int4 o;
        o.x = (k1 == 0xFFFFFFFFu) ? -1 : (int)(((k1 & 0xFFFFu) - lds0) >> 4);
        o.y = (k1 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k1 >> 16);
        o.z = (k2 == 0xFFFFFFFFu) ? -1 : (int)(((k2 & 0xFFFFu) - lds0) >> 4);
        o.w = (k2 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k2 >> 16);
        out[q] = o;
    }
    if (dbg) dbg[6] = clock64();
}

}  // namespace lvt
