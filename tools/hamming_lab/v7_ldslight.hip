// v7_ldslight.hip -- round-6 lab build (tools/hamming_lab/lab.py): the batched masked 2-NN matcher with an LDS that holds NO descriptors.
// LDS per problem: train coordinates in bin order, bin starts, the per-query words, the two u16 permutations (~32 KB at N = 1500, M = 1000), so that FOUR whole
// problems (512-thread workgroups) are resident per CU instead of two.  Stage B gathers a candidate's 32-byte descriptor with two global_load_dwordx4 from
// L2, where one dword per 128-byte line was touched by the problem's first round of loads.  Same HammingArgs, same output, same tie rule as k_hamming.hip.
#include "lvt_dev.h"
#include <type_traits>

namespace lvt {

struct HammingArgs {
    const uint64_t *q_desc;  // [B][M][4]
    const float2 *q_xy;      // [B][M]
    const uint64_t *t_desc;  // [B][N][4]
    const float2 *t_xy;      // [B][N]
    const uint8_t *t_flag;   // [B][N]
    int4 *out;               // [B][M] (idx1, d1, idx2, d2)
    int M, N;
    float r2;
    int img_rows, img_cols;
    int nbx, nby, csr;       // bins: hash cells (mode 0) or rows (mode 1: nbx = 1, nby = rows + 1)
    long long *dbg;          // optional: phase cycle stamps of one workgroup
};

#ifndef LL_THREADS
#define LL_THREADS 512
#endif
#ifndef LL_WPE
#define LL_WPE 8
#endif
#ifndef LL_TOUCH
#define LL_TOUCH 1   // 1: touch the train descriptor lines in the first round of loads, 2: the query descriptor lines too, 0: nothing
#endif
constexpr int HB_THREADS = LL_THREADS;
constexpr int HB_WAVES = HB_THREADS / 64;
constexpr int HB_TPT = 4;          // train features per thread  => N <= 4096 (LDS permitting)
constexpr int HB_QPT = 4;          // queries per thread         => M <= 4096 (LDS permitting)
constexpr int HB_NMAX = HB_THREADS * HB_TPT;
constexpr int HB_MMAX = HB_THREADS * HB_QPT;
constexpr int HB_HIST = 64;        // query classes by candidate count (>= 63 candidates share the first class)

// y / 25.0f, correctly rounded, in three instructions instead of the ~10 of the IEEE division expansion: with
// c = RN(1/25), q0 = RN(y c), r = y - 25 q0 (exact in one fma), RN(q0 + r c) is the correctly rounded quotient
// (Markstein).  Checked against y / 25.0f for every finite float (tests/test_div25.py runs a sample of that sweep).
static_assert(HASH_CELL == 25, "div_cell is specialised to the reference's 25-px hash cell");
__device__ __forceinline__ float div_cell(float y) {
    const float c = 0.04f;
    const float q0 = y * c;
    return __builtin_fmaf(__builtin_fmaf(-25.0f, q0, y), c, q0);
}

// mask = 2 * mask + (d2 < r2): compare into VCC, add-with-carry shifts the mask and appends the bit -- two instructions
// per candidate where (compare, select, or) plus a materialised bit constant cost four.  The candidates are walked from
// the last to the first, so candidate v still lands in bit v.
__device__ __forceinline__ void push_bit(uint32_t &mask, float d2, float r2) {
    asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(d2), "v"(r2) : "vcc");
}

// 256-bit Hamming distance as ONE chain of accumulating popcounts (v_bcnt_u32_b32 adds its third operand): 8 xor + 8 bcnt.  Left to itself the
// compiler splits the sum of eight popcounts into four chains and adds them up again (8 + 8 + 3 instructions), and bcnt / add3 / min / lshl_or issue at
// HALF the rate of xor / add / fma on this machine (4.2 against 2.3 cycles per wave64 instruction and SIMD: tools/lab/int_issue.hip)
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
// the running top-2 (k1 <= k2) takes a new key: k2' is the MEDIAN of (k1, k2, key) -- one v_med3_u32 where min(k2, max(k1, key)) is two half-rate instructions
__device__ __forceinline__ void top2_insert(uint32_t &k1, uint32_t &k2, uint32_t key) {
    uint32_t m;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k1), "v"(k2), "v"(key));
    k2 = m;
    k1 = min(k1, key);
}
__device__ __forceinline__ uint32_t hamming256(uint64_t d0, uint64_t d1, uint64_t d2, uint64_t d3, const uint4 &a0, const uint4 &a1) {
    uint32_t d = bcnt_acc((uint32_t)d0 ^ a0.x, 0u);
    d = bcnt_acc((uint32_t)(d0 >> 32) ^ a0.y, d);
    d = bcnt_acc((uint32_t)d1 ^ a0.z, d);
    d = bcnt_acc((uint32_t)(d1 >> 32) ^ a0.w, d);
    d = bcnt_acc((uint32_t)d2 ^ a1.x, d);
    d = bcnt_acc((uint32_t)(d2 >> 32) ^ a1.y, d);
    d = bcnt_acc((uint32_t)d3 ^ a1.z, d);
    d = bcnt_acc((uint32_t)(d3 >> 32) ^ a1.w, d);
    return d;
}

// NSP = number of candidate ranges a query keeps in registers: 1 (row mode), 3 (csr 1), 5 (csr 2); 0 = any csr, the
// ranges are walked one after the other (no flattening)
template <int MODE, int NSP, int QPT, int TPT>  // QPT = ceil(M / 512) rounds of queries, TPT = ceil(N / 512) train features per thread
__global__ __launch_bounds__(HB_THREADS) __attribute__((amdgpu_waves_per_eu(LL_WPE, LL_WPE))) void k_hamming_batched(HammingArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int N = a.N, M = a.M;
    const int nbins = a.nbx * a.nby;
    // carve: desc [N][2] uint4 | xy [N] float2 | per-query slot [M] 12 B | start [nbins + 1] | idx [N] u16 | order [M] u16
    // descriptor halves as TWO arrays, not one of 32-byte records: a ds_read_b128 serves 16 lanes per LDS cycle over the 16 four-bank
    // groups, and with 32-byte records every read (all low halves, or all high halves) can only reach 8 of them
    float2 *s_xy = reinterpret_cast<float2 *>(smem);
    uint2 *s_mask = reinterpret_cast<uint2 *>(s_xy + N);
    uint32_t *s_q = reinterpret_cast<uint32_t *>(s_mask);  // 12 B per query: packed ranges + candidate mask (see stage 4)
    int *s_start = reinterpret_cast<int *>(s_q + 3 * (size_t)M);
    uint16_t *s_idx = reinterpret_cast<uint16_t *>(s_start + nbins + 1);
    uint16_t *s_order = s_idx + ((N + 1) & ~1);
    __shared__ int s_scan[32];
    __shared__ int s_hist[HB_HIST];
    __shared__ int s_recheck;  // row mode: some train feature sits in a row bin without being an in-range integer row (see the scatter)

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint4 *td = reinterpret_cast<const uint4 *>(a.t_desc + (size_t)b * N * 4);
    const float2 *txy = a.t_xy + (size_t)b * N;
    const uint8_t *tf = a.t_flag + (size_t)b * N;
    const uint4 *qd = reinterpret_cast<const uint4 *>(a.q_desc + (size_t)b * M * 4);
    const float2 *qxy = a.q_xy + (size_t)b * M;
    int4 *out = a.out + (size_t)b * M;

    long long *dbg = (a.dbg && blockIdx.x == gridDim.x / 2 && tid == 0) ? a.dbg : nullptr;
    if (dbg) dbg[0] = clock64();
    __builtin_amdgcn_s_setprio(3);  // the load / sort phases are latency-bound: let them through ahead of the other
                                    // workgroup's VALU-bound query phases

    // ---- 1. everything this thread needs from HBM for the sort, issued back to back (indices clamped: no branches)
    float2 tp[TPT];
    bool tv[TPT];
    uint8_t tfl[TPT];
    float2 qp[QPT];
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        const int j = tid + k * HB_THREADS;
        const int jc = max(min(j, N - 1), 0);
        tfl[k] = 1;
        tp[k] = make_float2(0.f, 0.f);
        if (N > 0) {  // the flag is only LOOKED AT after every load is in flight (a compare here would wait for it)
            tp[k] = txy[jc];
            tfl[k] = tf[jc];
        }
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) qp[k] = qxy[min(tid + k * HB_THREADS, M - 1)];
    uint32_t touch = 0;  // one dword of every 128-byte line of the descriptors: the lines are in L2 when stage B gathers from them
    if (LL_TOUCH >= 1)
        for (int i = tid; i < (N * 32 + 127) / 128; i += HB_THREADS) touch |= reinterpret_cast<const uint32_t *>(td)[i * 32];
    if (LL_TOUCH >= 2)
        for (int i = tid; i < (M * 32 + 127) / 128; i += HB_THREADS) touch |= reinterpret_cast<const uint32_t *>(qd)[i * 32];
    asm volatile("" ::"v"(touch));
#pragma unroll
    for (int k = 0; k < TPT; k++) tv[k] = (tid + k * HB_THREADS < N) & (tfl[k] == 0);
    for (int i = tid; i <= nbins; i += HB_THREADS) s_start[i] = 0;
    if (tid < HB_HIST) s_hist[tid] = 0;
    if (tid == 0) s_recheck = 0;
    __syncthreads();
    if (dbg) dbg[1] = clock64();

    // ---- 2. counting sort of the unflagged train features: the counting atomic returns the rank inside the bin
    auto bin_of = [&](float x, float y) -> int {
        if (MODE == 1) return min(max((int)floorf(y), 0), a.nby - 1);
        const int cy = min(max((int)floorf(div_cell(y)), 0), a.nby - 1);
        const int cx = min(max((int)floorf(div_cell(x)), 0), a.nbx - 1);
        return cy * a.nbx + cx;
    };
    int tbin[TPT], trank[TPT];
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        tbin[k] = bin_of(tp[k].x, tp[k].y);
        trank[k] = 0;
        if (tv[k]) trank[k] = atomicAdd(&s_start[tbin[k]], 1);
    }
    __syncthreads();
    if (dbg) dbg[2] = clock64();
    if (nbins + 1 <= HB_THREADS) {  // (the usual case) one entry per thread: no chunk loops, no index arithmetic
        const int v = (tid <= nbins) ? s_start[tid] : 0;
        int total;
        const int run = block_excl_scan(v, s_scan, &total);
        if (tid <= nbins) s_start[tid] = run;
    } else {  // counts -> exclusive starts, in place; entry nbins receives the total.  One contiguous chunk per thread.
        const int chunk = (nbins + 1 + HB_THREADS - 1) / HB_THREADS;
        const int i0 = min(tid * chunk, nbins + 1), i1 = min(i0 + chunk, nbins + 1);
        int sum = 0;
        for (int i = i0; i < i1; i++) sum += s_start[i];
        int total;
        int run = block_excl_scan(sum, s_scan, &total);
        for (int i = i0; i < i1; i++) {
            const int v = s_start[i];
            s_start[i] = run;
            run += v;
        }
    }
    __syncthreads();
    if (dbg) dbg[3] = clock64();
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        if (tv[k]) {
            const int pos = s_start[tbin[k]] + trank[k];
            s_xy[pos] = tp[k];
            // row mode: a feature whose y IS its bin (an integer row inside the image: every key point the detector emits) passes
            // struct.cpp:133 `y >= start_y && y <= end_y` for exactly the queries whose row range holds its bin -- the walk needs
            // no comparison and no coordinates.  Any other y (fractional, clamped into an edge bin, NaN) is marked: bit 15 of its
            // index entry (indices stay below HB_NMAX = 4096) sends the walk to the reference's own comparison.
            const bool recheck = (MODE == 1) && !((float)tbin[k] == tp[k].y);
            if (recheck) s_recheck = 1;
            s_idx[pos] = (uint16_t)((tid + k * HB_THREADS) | (recheck ? 0x8000 : 0));
        }
    }

    // candidate ranges of one query (named scalars, not arrays: they must stay in VGPRs).  l_k = 0 for a range that
    // does not exist.
    constexpr int NS = NSP > 0 ? NSP : 1;
    struct Ranges {
        int s0, l0, s1, l1, s2, l2, s3, l3, s4, l4;
        int y0, y1, x0, x1;
    };
    auto ranges = [&](float2 p) -> Ranges {
        Ranges R;
        R.s0 = R.l0 = R.s1 = R.l1 = R.s2 = R.l2 = R.s3 = R.l3 = R.s4 = R.l4 = 0;
        if (MODE == 1) {  // struct.cpp:124-131: rows [int(y)-2, int(y)+2] clipped to [0, rows] are contiguous bins
            R.y0 = max((int)p.y - ROW_RADIUS, 0);
            R.y1 = min(min((int)p.y + ROW_RADIUS, a.img_rows), a.nby - 1);
            R.x0 = R.x1 = 0;
            const bool ok = R.y0 <= R.y1;
            R.s0 = s_start[ok ? R.y0 : 0];
            R.l0 = s_start[ok ? R.y1 + 1 : 0] - R.s0;
        } else {  // struct.cpp:71-83: the cells of one window row are contiguous
            const int hy = (int)floorf(div_cell(p.y)), hx = (int)floorf(div_cell(p.x));
            R.y0 = max(hy - a.csr, 0);
            R.y1 = min(hy + a.csr, a.nby - 1);
            R.x0 = max(hx - a.csr, 0);
            R.x1 = min(hx + a.csr, a.nbx - 1);
#define LVT_RANGE(k, S, L)                                                     \
    if (NSP > k) {                                                             \
        const bool ok = (R.y0 + k <= R.y1) && (R.x0 <= R.x1);                  \
        const int row = ok ? (R.y0 + k) * a.nbx : 0;                           \
        S = s_start[row + (ok ? R.x0 : 0)];                                    \
        L = s_start[row + (ok ? R.x1 + 1 : 0)] - S;                            \
    }
            LVT_RANGE(0, R.s0, R.l0)
            LVT_RANGE(1, R.s1, R.l1)
            LVT_RANGE(2, R.s2, R.l2)
            LVT_RANGE(3, R.s3, R.l3)
            LVT_RANGE(4, R.s4, R.l4)
#undef LVT_RANGE
        }
        return R;
    };
    auto count_of = [&](float2 p) -> int {
        const Ranges R = ranges(p);
        int c = 0;
        if (MODE == 0 && NSP == 0) {
            if (R.x0 <= R.x1)
                for (int by = R.y0; by <= R.y1; by++) c += s_start[by * a.nbx + R.x1 + 1] - s_start[by * a.nbx + R.x0];
        } else
            c = R.l0 + R.l1 + R.l2 + R.l3 + R.l4;
        return c;
    };

    // ---- 3. queries sorted by candidate count, heaviest first (needs the bin starts only: overlaps the scatter)
    int qkey[QPT], qrank[QPT];
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        const int q = tid + k * HB_THREADS;
        qkey[k] = 0, qrank[k] = 0;
        if (q < M) {
            if (MODE == 0 && NSP == 3) {
                // the three candidate ranges travel with the query through both stages, packed into two words (11-bit
                // starts, 6-bit lengths): neither stage recomputes hash cells, window clamps or bin lookups
                const Ranges R = ranges(qp[k]);
                qkey[k] = HB_HIST - 1 - min(R.l0 + R.l1 + R.l2, HB_HIST - 1);
                const bool fits = (N <= 2048) && (R.l0 < 64) && (R.l1 < 64) && (R.l2 < 64);
                s_q[3 * q] = fits ? ((uint32_t)R.s0 | ((uint32_t)R.s1 << 11) | ((uint32_t)R.l0 << 22)) : 0xFFFFFFFFu;
                s_q[3 * q + 1] = (uint32_t)R.s2 | ((uint32_t)R.l1 << 11) | ((uint32_t)R.l2 << 17);
            } else {
                if (MODE == 1) {  // the one range travels with the query (the per-query words of the radius mode are unused here): start | length << 16
                    const Ranges R = ranges(qp[k]);
                    s_q[q] = (uint32_t)R.s0 | ((uint32_t)R.l0 << 16);
                    qkey[k] = HB_HIST - 1 - min(R.l0, HB_HIST - 1);
                } else
                    qkey[k] = HB_HIST - 1 - min(count_of(qp[k]), HB_HIST - 1);
                // stage 4a visits the queries in sorted order: it finds the coordinates in the query's (still unused) mask
                // slot instead of going back to HBM for them
                if (MODE == 0 && NSP > 0) s_mask[q] = make_uint2(__float_as_uint(qp[k].x), __float_as_uint(qp[k].y));
            }
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
        const int q = tid + k * HB_THREADS;
        if (q < M) s_order[s_hist[qkey[k]] + qrank[k]] = (uint16_t)q;
    }
    __syncthreads();
    if (dbg) dbg[5] = clock64();

#if defined(LAB_STOP) && LAB_STOP == 1   // (tools/hamming_lab/lab.py --stop 1: counters of the load / sort phases alone)
    return;
#endif
    __builtin_amdgcn_s_setprio(0);
    // ---- 4. rounds of 512 queries in sorted order; odd rounds reverse the wave order so every wave gets a similar sum
    constexpr int rounds = QPT;
    auto slot_query = [&](int j, int count) -> int {
        const int slot = j * HB_THREADS + ((j & 1) ? (HB_WAVES - 1 - wv) : wv) * 64 + lane;
        return (j < rounds && slot < count) ? (int)s_order[slot] : -1;
    };
    // positions of the flattened index space: v in [c_k, c_{k+1}) lies in range k at LDS position v + o_k
    // (scalars and a macro, not arrays and a lambda: they must stay in VGPRs)
#define LVT_POS_OF(dst, v)                          \
    {                                               \
        int o_ = o0;                                \
        if (NS > 1) o_ = ((v) >= c1) ? o1 : o_;     \
        if (NS > 2) o_ = ((v) >= c2) ? o2 : o_;     \
        if (NS > 3) o_ = ((v) >= c3) ? o3 : o_;     \
        if (NS > 4) o_ = ((v) >= c4) ? o4 : o_;     \
        dst = (v) + o_;                             \
    }
    // stage A: radius test of the candidates [lo_v, hi_v) of the flattened index space, last to first, two per step (the
    // pair shares the packed fp32 subtract / multiply / add); candidate v lands in bit v - lo_v of the mask
#define LVT_RADIUS_BITS(mask, lo_v, hi_v)                                      \
    {                                                                          \
        int v_ = (hi_v)-1;                                                     \
        for (; v_ > (lo_v); v_ -= 2) {                                         \
            int ia_, ib_;                                                      \
            LVT_POS_OF(ia_, v_)                                                \
            LVT_POS_OF(ib_, v_ - 1)                                            \
            const float2 ra_ = s_xy[ia_], rb_ = s_xy[ib_];                     \
            const float dxa_ = ra_.x - p.x, dya_ = ra_.y - p.y;                \
            const float dxb_ = rb_.x - p.x, dyb_ = rb_.y - p.y;                \
            push_bit(mask, dxa_ * dxa_ + dya_ * dya_, a.r2);                   \
            push_bit(mask, dxb_ * dxb_ + dyb_ * dyb_, a.r2);                   \
        }                                                                      \
        if (v_ == (lo_v)) {                                                    \
            int ia_;                                                           \
            LVT_POS_OF(ia_, v_)                                                \
            const float2 ra_ = s_xy[ia_];                                      \
            const float dxa_ = ra_.x - p.x, dya_ = ra_.y - p.y;                \
            push_bit(mask, dxa_ * dxa_ + dya_ * dya_, a.r2);                   \
        }                                                                      \
    }
    // all candidates of the ranges, filter and distance in one pass (row mode, any-csr mode, over-long windows)
    auto walk_all = [&](auto recheck_tag, float2 p, float fy0, float fy1, uint64_t d0, uint64_t d1, uint64_t d2, uint64_t d3, uint32_t &k1, uint32_t &k2,
                        int total, int o0, int c1, int o1, int c2, int o2, int c3, int o3, int c4, int o4) {
        constexpr bool RECHECK = decltype(recheck_tag)::value;  // row mode only: marked entries exist in this problem
        constexpr bool NEED_XY = (MODE != 1) || RECHECK;
        if (total <= 0) return;
        int it;
        LVT_POS_OF(it, 0)
        float2 r = make_float2(0.f, 0.f);
        if (NEED_XY) r = s_xy[it];
        uint32_t id = s_idx[it];
        uint4 a0 = td[2 * min(id & 0x7FFFu, (uint32_t)N - 1)], a1 = td[2 * min(id & 0x7FFFu, (uint32_t)N - 1) + 1];
        for (int v = 0; v < total; v++) {  // software-pipelined by one candidate; the last prefetch reads one past (valid LDS)
            int itn;
            LVT_POS_OF(itn, v + 1)
            float2 rn = make_float2(0.f, 0.f);
            if (NEED_XY) rn = s_xy[itn];
            const uint32_t idn = s_idx[itn];
            const uint4 b0 = td[2 * min(idn & 0x7FFFu, (uint32_t)N - 1)], b1 = td[2 * min(idn & 0x7FFFu, (uint32_t)N - 1) + 1];
            bool ok = true;
            if (MODE == 1) {
                if (RECHECK) ok = !(id & 0x8000u) || ((r.y >= fy0) && (r.y <= fy1));
            } else {
                const float dx = r.x - p.x, dy = r.y - p.y;
                ok = (dx * dx + dy * dy) < a.r2;
            }
            const uint32_t d = hamming256(d0, d1, d2, d3, a0, a1);
            uint32_t key = ((uint32_t)d << 16) | (RECHECK ? (id & 0x7FFFu) : id);
            if (NEED_XY) key = ok ? key : 0xFFFFFFFFu;
            top2_insert(k1, k2, key);
            r = rn, a0 = b0, a1 = b1, id = idn;
        }
    };
    // row mode, no marked entries in the problem: the one range, two candidates per trip, no coordinates, no comparison.  Hand-unrolled: the
    // accumulating popcounts are inline asm, which the loop unroller leaves alone.
    // (merging a sorted pair lo <= hi into the running top-2 k1 <= k2: k1' = min(k1, lo), k2' = min(max(k1, lo), k2, hi).)
    auto row_walk_lean = [&](int s0, int len, uint64_t d0, uint64_t d1, uint64_t d2, uint64_t d3, uint32_t &k1, uint32_t &k2) {
        const uint16_t *pid = s_idx + s0;
        auto pair = [&](const uint4 &a0, const uint4 &a1, uint32_t ia, const uint4 &b0, const uint4 &b1, uint32_t ib) {
            const uint32_t ka = (hamming256(d0, d1, d2, d3, a0, a1) << 16) | ia, kb = (hamming256(d0, d1, d2, d3, b0, b1) << 16) | ib;
            const uint32_t lo = min(ka, kb), hi = max(ka, kb);
            const uint32_t t = max(k1, lo);
            k1 = min(k1, lo);
            k2 = min(t, min(k2, hi));
        };
        // two candidates per trip; the NEXT pair's gathers are in flight while this pair is ranked (global latency, not LDS latency: one pair ahead)
        if (len <= 0) return;
        uint32_t ia = pid[0], ib = pid[min(1, len - 1)];
        uint4 a0 = td[2 * ia], a1 = td[2 * ia + 1], b0 = td[2 * ib], b1 = td[2 * ib + 1];
        int v = 0;
        for (; v + 2 <= len; v += 2) {
            const uint32_t na = pid[min(v + 2, len - 1)], nb = pid[min(v + 3, len - 1)];
            const uint4 c0 = td[2 * na], c1 = td[2 * na + 1], e0 = td[2 * nb], e1 = td[2 * nb + 1];
            pair(a0, a1, ia, b0, b1, ib);
            ia = na, ib = nb, a0 = c0, a1 = c1, b0 = e0, b1 = e1;
        }
        if (v < len) {
            const uint32_t ka = (hamming256(d0, d1, d2, d3, a0, a1) << 16) | ia;
            top2_insert(k1, k2, ka);
        }
    };
    auto match_all = [&](int q, float2 p, uint4 w0, uint4 w1, const Ranges &R) {
        const uint64_t d0 = ((uint64_t)w0.y << 32) | w0.x, d1 = ((uint64_t)w0.w << 32) | w0.z;
        const uint64_t d2 = ((uint64_t)w1.y << 32) | w1.x, d3 = ((uint64_t)w1.w << 32) | w1.z;
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        const float fy0 = (float)R.y0, fy1 = (float)min((int)p.y + ROW_RADIUS, a.img_rows);
        if (MODE == 0 && NSP == 0) {
            if (R.x0 <= R.x1)
                for (int by = R.y0; by <= R.y1; by++) {
                    const int s = s_start[by * a.nbx + R.x0];
                    walk_all(std::false_type{}, p, fy0, fy1, d0, d1, d2, d3, k1, k2, s_start[by * a.nbx + R.x1 + 1] - s, s, 0, 0, 0, 0, 0, 0, 0, 0);
                }
        } else {
            static_assert(NS <= 5, "range registers");
            const int c1 = R.l0, c2 = c1 + R.l1, c3 = c2 + R.l2, c4 = c3 + R.l3;
            if (MODE == 1 && s_recheck)  // (block-uniform)
                walk_all(std::true_type{}, p, fy0, fy1, d0, d1, d2, d3, k1, k2, c4 + R.l4, R.s0, c1, R.s1 - c1, c2, R.s2 - c2, c3, R.s3 - c3, c4, R.s4 - c4);
            else if (MODE == 1)
                row_walk_lean(R.s0, R.l0, d0, d1, d2, d3, k1, k2);
            else
                walk_all(std::false_type{}, p, fy0, fy1, d0, d1, d2, d3, k1, k2, c4 + R.l4, R.s0, c1, R.s1 - c1, c2, R.s2 - c2, c3, R.s3 - c3, c4, R.s4 - c4);
        }
        int4 o;
        o.x = (k1 == 0xFFFFFFFFu) ? -1 : (int)(k1 & 0xFFFFu);
        o.y = (k1 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k1 >> 16);
        o.z = (k2 == 0xFFFFFFFFu) ? -1 : (int)(k2 & 0xFFFFu);
        o.w = (k2 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k2 >> 16);
        out[q] = o;
    };

    constexpr bool TWO_STAGE = (MODE == 0 && NSP > 0);
    constexpr bool PACKED = (MODE == 0 && NSP == 3);
    if (PACKED) {
        // ---- 4a (packed ranges). radius test over every window candidate -> one bit per candidate of the flattened index space
        constexpr int MASK_BITS = 41;  // 32 in word 2, 9 in the free top of word 1
        if (tid < HB_HIST) s_hist[tid] = 0;
        int aq[QPT], akey[QPT], arank[QPT];
        {
            int q = slot_query(0, M);
            float2 p = qxy[max(q, 0)];
#pragma unroll
            for (int j = 0; j < QPT; j++) {
                aq[j] = -1, akey[j] = 0, arank[j] = 0;
                const int qn = slot_query(j + 1, M);
                const float2 np = qxy[max(qn, 0)];
                if (q >= 0) {
                    const uint32_t W0 = s_q[3 * q], W1 = s_q[3 * q + 1];
                    Ranges R;
                    R.s3 = R.l3 = R.s4 = R.l4 = 0;
                    R.y0 = R.y1 = R.x0 = R.x1 = 0;
                    R.s0 = (int)(W0 & 2047u), R.s1 = (int)((W0 >> 11) & 2047u), R.l0 = (int)(W0 >> 22);
                    R.s2 = (int)(W1 & 2047u), R.l1 = (int)((W1 >> 11) & 63u), R.l2 = (int)((W1 >> 17) & 63u);
                    const bool packed = W0 != 0xFFFFFFFFu;
                    const int c1 = R.l0, c2 = c1 + R.l1, total = c2 + R.l2;
                    const int o0 = R.s0, o1 = R.s1 - c1, o2 = R.s2 - c2;
                    const int c3 = 0, c4 = 0, o3 = 0, o4 = 0;
                    (void)c3, (void)c4, (void)o3, (void)o4;
                    if (!packed || total > MASK_BITS) {  // ranges or mask do not fit their slot: matched here and now (rare)
                        match_all(q, p, qd[2 * q], qd[2 * q + 1], packed ? R : ranges(p));
                    } else {
                        uint32_t lo = 0, hi = 0;
                        const int t0 = min(total, 32);
LVT_RADIUS_BITS(lo, 0, t0)
                        LVT_RADIUS_BITS(hi, 32, total)
                        s_q[3 * q + 2] = lo;
                        s_q[3 * q + 1] = (W1 & 0x7FFFFFu) | (hi << 23);
                        aq[j] = q;
                        akey[j] = HB_HIST - 1 - min(__popc(lo) + __popc(hi), HB_HIST - 1);
                    }
                }
                q = qn, p = np;
            }
        }
        // ---- 4b. the queries again, sorted by the number of candidates inside the circle; only descriptors are fetched
        __syncthreads();  // s_hist zeroed, every stage-4a read of s_order done
#pragma unroll
        for (int j = 0; j < QPT; j++)
            if (aq[j] >= 0) arank[j] = atomicAdd(&s_hist[akey[j]], 1);
        __syncthreads();
        if (wv == 0) {
            const int v = s_hist[lane];
            const int incl = wave_incl_scan(v);
            s_hist[lane] = incl - v;
            if (lane == 63) s_scan[0] = incl;
        }
        __syncthreads();
        const int M2 = s_scan[0];
#pragma unroll
        for (int j = 0; j < QPT; j++)
            if (aq[j] >= 0) s_order[s_hist[akey[j]] + arank[j]] = (uint16_t)aq[j];
        __syncthreads();
        if (dbg) dbg[7] = clock64();

        int q = slot_query(0, M2);
        uint4 w0 = qd[2 * max(q, 0)], w1 = qd[2 * max(q, 0) + 1];
        for (int j = 0; j < rounds; j++) {
            const int qn = slot_query(j + 1, M2);
            const uint4 nw0 = qd[2 * max(qn, 0)], nw1 = qd[2 * max(qn, 0) + 1];
            if (q >= 0) {
                const uint64_t d0 = ((uint64_t)w0.y << 32) | w0.x, d1 = ((uint64_t)w0.w << 32) | w0.z;
                const uint64_t d2 = ((uint64_t)w1.y << 32) | w1.x, d3 = ((uint64_t)w1.w << 32) | w1.z;
                uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
                const uint32_t W0 = s_q[3 * q], W1 = s_q[3 * q + 1], W2 = s_q[3 * q + 2];
                const int c1 = (int)(W0 >> 22), c2 = c1 + (int)((W1 >> 11) & 63u);
                const int o0 = (int)(W0 & 2047u), o1 = (int)((W0 >> 11) & 2047u) - c1, o2 = (int)(W1 & 2047u) - c2;
                const int c3 = 0, c4 = 0, o3 = 0, o4 = 0;
                (void)c3, (void)c4, (void)o3, (void)o4;
                auto walk_bits = [&](uint32_t m, int base) {  // set bits of m, software-pipelined by one candidate
                    if (m == 0) return;
                    int it;
                    {
                        const int v = base + __ffs((int)m) - 1;
                        LVT_POS_OF(it, v)
                    }
                    m &= m - 1;
                    uint32_t id = s_idx[it];
                    uint4 a0 = td[2 * id], a1 = td[2 * id + 1];
                    for (;;) {
                        const bool more = m != 0;
                        int itn;
                        {
                            const int v = base + ((__ffs((int)m) - 1) & 31);
                            LVT_POS_OF(itn, v)
                        }
                        m &= m - 1;
                        const uint32_t idn = min((uint32_t)s_idx[itn], (uint32_t)N - 1);  // (past the last bit: any valid line)
                        const uint4 b0 = td[2 * idn], b1 = td[2 * idn + 1];
                        const uint32_t d = hamming256(d0, d1, d2, d3, a0, a1);
                        const uint32_t key = (d << 16) | id;
                        top2_insert(k1, k2, key);
                        if (!more) break;
                        a0 = b0, a1 = b1, id = idn;
                    }
                };
                walk_bits(W2, 0);
                walk_bits(W1 >> 23, 32);
                int4 o;
                o.x = (k1 == 0xFFFFFFFFu) ? -1 : (int)(k1 & 0xFFFFu);
                o.y = (k1 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k1 >> 16);
                o.z = (k2 == 0xFFFFFFFFu) ? -1 : (int)(k2 & 0xFFFFu);
                o.w = (k2 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k2 >> 16);
                out[q] = o;
            }
            q = qn, w0 = nw0, w1 = nw1;
        }
    } else if (MODE == 1 && !s_recheck) {
        // Row mode without marked entries (block-uniform): the range travels with the query (stage 3), the walk needs neither coordinates nor bin starts.
        // (Re-dealing a wavefront's 64 queries among its lanes by range start mod 16 -- 16 ballots, rank k to lane (k % 4) * 16 + k / 4, which halves the
        //  conflicts of random starts in tools/lab/lds_b128_groups.hip -- was built and changed nothing here: queries of one image row share their range,
        //  the sorted order keeps them in one wavefront, and equal addresses broadcast; the walk's ds_read_b128 already run at the dealt pattern's ~10 cycles.)
        int q = slot_query(0, M);
        uint4 w0 = qd[2 * max(q, 0)], w1 = qd[2 * max(q, 0) + 1];
        for (int j = 0; j < rounds; j++) {
            const int qn = slot_query(j + 1, M);
            const uint4 nw0 = qd[2 * max(qn, 0)], nw1 = qd[2 * max(qn, 0) + 1];  // next round's query: in flight during this round
            if (q >= 0) {
                const uint32_t W = s_q[q];
                const uint64_t d0 = ((uint64_t)w0.y << 32) | w0.x, d1 = ((uint64_t)w0.w << 32) | w0.z;
                const uint64_t d2 = ((uint64_t)w1.y << 32) | w1.x, d3 = ((uint64_t)w1.w << 32) | w1.z;
                uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
                row_walk_lean((int)(W & 0xFFFFu), (int)(W >> 16), d0, d1, d2, d3, k1, k2);
                int4 o;
                o.x = (k1 == 0xFFFFFFFFu) ? -1 : (int)(k1 & 0xFFFFu);
                o.y = (k1 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k1 >> 16);
                o.z = (k2 == 0xFFFFFFFFu) ? -1 : (int)(k2 & 0xFFFFu);
                o.w = (k2 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k2 >> 16);
                out[q] = o;
            }
            q = qn, w0 = nw0, w1 = nw1;
        }
    } else if (!TWO_STAGE) {
        int q = slot_query(0, M);
        uint4 w0 = qd[2 * max(q, 0)], w1 = qd[2 * max(q, 0) + 1];
        float2 p = qxy[max(q, 0)];
        for (int j = 0; j < rounds; j++) {
            const int qn = slot_query(j + 1, M);
            const uint4 nw0 = qd[2 * max(qn, 0)], nw1 = qd[2 * max(qn, 0) + 1];  // next round's query: in flight during this round
            const float2 np = qxy[max(qn, 0)];
            if (q >= 0) match_all(q, p, w0, w1, ranges(p));
            q = qn, w0 = nw0, w1 = nw1, p = np;
        }
    } else {
        // ---- 4a. the radius test alone over every window candidate (8 B of LDS and ~12 instructions each): one bit per
        //          candidate of the flattened index space.  Only ~1/3 of a 3x3-cell window lies inside the circle, so
        //          the descriptor work (40 B, ~30 instructions) is kept for stage 4b.
        if (tid < HB_HIST) s_hist[tid] = 0;
        int aq[QPT], akey[QPT], arank[QPT];
        {
            int q = slot_query(0, M);
#pragma unroll
            for (int j = 0; j < QPT; j++) {
                aq[j] = -1, akey[j] = 0, arank[j] = 0;
                {
                    const int qn = slot_query(j + 1, M);
                    if (q >= 0) {
                        const uint2 pw = s_mask[q];  // coordinates stashed by the pre-pass; the slot receives the mask below
                        const float2 p = make_float2(__uint_as_float(pw.x), __uint_as_float(pw.y));
                        const Ranges R = ranges(p);
                        const int c1 = R.l0, c2 = c1 + R.l1, c3 = c2 + R.l2, c4 = c3 + R.l3, total = c4 + R.l4;
                        const int o0 = R.s0, o1 = R.s1 - c1, o2 = R.s2 - c2, o3 = R.s3 - c3, o4 = R.s4 - c4;
                        if (total > 64) {  // does not fit the mask: matched here and now (rare)
                            match_all(q, p, qd[2 * q], qd[2 * q + 1], R);
                        } else {
                            uint32_t lo = 0, hi = 0;
                            const int t0 = min(total, 32);
LVT_RADIUS_BITS(lo, 0, t0)
                            LVT_RADIUS_BITS(hi, 32, total)
                            s_mask[q] = make_uint2(lo, hi);
                            aq[j] = q;
                            akey[j] = HB_HIST - 1 - min(__popc(lo) + __popc(hi), HB_HIST - 1);
                        }
                    }
                    q = qn;
                }
            }
        }
        // ---- 4b. the queries again, now sorted by the number of candidates inside the circle
        __syncthreads();  // s_hist zeroed, every stage-4a read of s_order done
#pragma unroll
        for (int j = 0; j < QPT; j++)
            if (aq[j] >= 0) arank[j] = atomicAdd(&s_hist[akey[j]], 1);
        __syncthreads();
        if (wv == 0) {
            const int v = s_hist[lane];
            const int incl = wave_incl_scan(v);
            s_hist[lane] = incl - v;
            if (lane == 63) s_scan[0] = incl;
        }
        __syncthreads();
        const int M2 = s_scan[0];
#pragma unroll
        for (int j = 0; j < QPT; j++)
            if (aq[j] >= 0) s_order[s_hist[akey[j]] + arank[j]] = (uint16_t)aq[j];
        __syncthreads();
        if (dbg) dbg[7] = clock64();

        int q = slot_query(0, M2);
        uint4 w0 = qd[2 * max(q, 0)], w1 = qd[2 * max(q, 0) + 1];
        float2 p = qxy[max(q, 0)];
        for (int j = 0; j < rounds; j++) {
            const int qn = slot_query(j + 1, M2);
            const uint4 nw0 = qd[2 * max(qn, 0)], nw1 = qd[2 * max(qn, 0) + 1];
            const float2 np = qxy[max(qn, 0)];
            if (q >= 0) {
                const uint64_t d0 = ((uint64_t)w0.y << 32) | w0.x, d1 = ((uint64_t)w0.w << 32) | w0.z;
                const uint64_t d2 = ((uint64_t)w1.y << 32) | w1.x, d3 = ((uint64_t)w1.w << 32) | w1.z;
                uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
                const Ranges R = ranges(p);
                const int c1 = R.l0, c2 = c1 + R.l1, c3 = c2 + R.l2, c4 = c3 + R.l3;
                const int o0 = R.s0, o1 = R.s1 - c1, o2 = R.s2 - c2, o3 = R.s3 - c3, o4 = R.s4 - c4;
                const uint2 mk = s_mask[q];
                auto walk_bits = [&](uint32_t m, int base) {  // set bits of m, software-pipelined by one candidate
                    if (m == 0) return;
                    int it;
                    {
                        const int v = base + __ffs((int)m) - 1;
                        LVT_POS_OF(it, v)
                    }
                    m &= m - 1;
                    uint32_t id = s_idx[it];
                    uint4 a0 = td[2 * id], a1 = td[2 * id + 1];
                    for (;;) {
                        const bool more = m != 0;
                        int itn;
                        {
                            const int v = base + ((__ffs((int)m) - 1) & 31);
                            LVT_POS_OF(itn, v)
                        }
                        m &= m - 1;
                        const uint32_t idn = min((uint32_t)s_idx[itn], (uint32_t)N - 1);  // (past the last bit: any valid line)
                        const uint4 b0 = td[2 * idn], b1 = td[2 * idn + 1];
                        const uint32_t d = hamming256(d0, d1, d2, d3, a0, a1);
                        const uint32_t key = (d << 16) | id;
                        top2_insert(k1, k2, key);
                        if (!more) break;
                        a0 = b0, a1 = b1, id = idn;
                    }
                };
                walk_bits(mk.x, 0);
                walk_bits(mk.y, 32);
                int4 o;
                o.x = (k1 == 0xFFFFFFFFu) ? -1 : (int)(k1 & 0xFFFFu);
                o.y = (k1 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k1 >> 16);
                o.z = (k2 == 0xFFFFFFFFu) ? -1 : (int)(k2 & 0xFFFFu);
                o.w = (k2 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k2 >> 16);
                out[q] = o;
            }
            q = qn, w0 = nw0, w1 = nw1, p = np;
        }
    }
#undef LVT_RADIUS_BITS
#undef LVT_POS_OF
    if (dbg) dbg[6] = clock64();
}

// no train features at all: every query gets the "no neighbour" record
__global__ __launch_bounds__(256) void k_hamming_none(int4 *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_int4(-1, 0x7FFFFFFF, -1, 0x7FFFFFFF);
}

static inline size_t hamming_lds_bytes(int N, int M, int nbins) {
    return (size_t)N * 8 + (size_t)M * 12 + (size_t)(nbins + 1) * 4 + (size_t)((N + 1) & ~1) * 2 + (size_t)((M + 1) & ~1) * 2 + 16;
}

}  // namespace lvt
