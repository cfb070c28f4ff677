// v2: fast path of the batched matcher for the reference's standard geometry (25-px hash cells, radius <= 25 => a 3 x 3 cell
// window; <= 2048 train features, <= 2048 queries, <= 1023 hash cells).  Same algorithm as k_hamming_batched<0, 3, ...> with
//  * the train DESCRIPTORS never in registers: after the counting sort has placed the 10-byte (x, y, index) part of every train
//    feature, the workgroup gathers the 32-B descriptor records straight from HBM into their BIN-ORDER positions in LDS with
//    LDS-DMA loads (global_load_lds_dwordx4: the global address is per lane, the LDS side is lane-linear), while stage A runs;
//  * all loads of a phase in flight at once (inline asm: the compiler serialised the flag byte's compare into the load sequence);
//  * the query coordinates handed from their owner to the sorted slot through the (still empty) descriptor region;
//  * trimmed index arithmetic everywhere (the launch is VALU-issue-bound: 11.4k wave-instructions per problem at 4 cycles).
#include "lvt_dev.h"
#ifndef LAB_STOP
#define LAB_STOP 0
#endif

namespace lvt {

struct HammingArgs {
    const uint64_t *q_desc;  // [B][M][4]
    const float2 *q_xy;      // [B][M]
    const uint64_t *t_desc;  // [B][N][4]
    const float2 *t_xy;      // [B][N]
    const uint8_t *t_flag;   // [B][N]
    int4 *out;               // [B][M] (idx1, d1, idx2, d2)
    int B, M, N;
    float r2;
    int img_rows, img_cols;
    int nbx, nby, csr;
    long long *dbg;
};

constexpr int HB_THREADS = 1024;
constexpr int HB_WAVES = HB_THREADS / 64;
constexpr int HB_HIST = 64;

static_assert(HASH_CELL == 25, "div_cell is specialised to the reference's 25-px hash cell");
__device__ __forceinline__ float div_cell(float y) {
    const float c = 0.04f;
    const float q0 = y * c;
    return __builtin_fmaf(__builtin_fmaf(-25.0f, q0, y), c, q0);
}
__device__ __forceinline__ void push_bit(uint32_t &mask, float d2, float r2) {
    asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(d2), "v"(r2) : "vcc");
}
// a wave-uniform pointer, forced into SGPRs (an "s" operand the compiler happens to hold in VGPRs does not assemble)
template <typename T>
__device__ __forceinline__ const T *hf_uniform(const T *p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const T *>((uintptr_t)(((uint64_t)hi << 32) | lo));
}
// loads whose waits are placed by hand: base in SGPRs, 32-bit byte offset per lane
__device__ __forceinline__ uint64_t hf_load64(const void *base, uint32_t off) {
    uint64_t v;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t hf_load8(const void *base, uint32_t off) {
    uint32_t v;
    asm volatile("global_load_ubyte %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
    return v;
}
__device__ __forceinline__ void hf_settle(uint64_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void hf_settle(uint32_t &v) { asm volatile("" : "+v"(v)); }
// 16 B per lane: global base + per-lane byte offset -> LDS byte address lds_base + 16 * lane
__device__ __forceinline__ void hf_dma16(const void *base, uint32_t off, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds_base) : "memory", "m0");
}
__device__ __forceinline__ uint32_t hf_med3(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
// popcount(x) + acc in one instruction, as ONE dependent chain (the compiler builds four chains and adds them up)
__device__ __forceinline__ int hf_bcnt(uint32_t x, int acc) {
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ void hf_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void hf_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef __attribute__((address_space(3))) uint8_t hf_lds_byte;
__device__ __forceinline__ uint32_t hf_lds_address(void *p) { return (uint32_t)(uintptr_t)(hf_lds_byte *)p; }

static inline size_t hamming_fast_lds_bytes(int N, int M, int nbins) {
    return (size_t)((2 * N + 63) / 64) * 1024 + (size_t)N * 8 + (size_t)M * 12 + (size_t)(nbins + 1) * 4 + (size_t)((N + 1) & ~1) * 2 +
           (size_t)((M + 1) & ~1) * 2 + (16 + HB_HIST) * 4 + 16;
}
static inline size_t hamming_lds_bytes(int N, int M, int nbins) { return hamming_fast_lds_bytes(N, M, nbins); }

template <int QPT, int TPT>  // QPT = ceil(M / 1024) <= 2, TPT = ceil(N / 1024) <= 2
__global__ __launch_bounds__(HB_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_hamming_batched_csr1(HammingArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int N = a.N, M = a.M;
    const int nbins = a.nbx * a.nby;  // <= 1023
    const int region = ((2 * N + 63) >> 6) << 10;
    // carve: descriptor image [bin order][lo, hi] (DMA target; before the DMA starts: the query coordinates in sorted order) |
    //        xy [N] float2, bin order | per-query slot [M] 12 B | start [nbins + 1] | idx [N] u16, bin order | order [M] u16 | scan | hist
    uint4 *s_desc = reinterpret_cast<uint4 *>(smem);
    float2 *s_p = reinterpret_cast<float2 *>(smem);
    float2 *s_xy = reinterpret_cast<float2 *>(smem + region);
    uint32_t *s_q = reinterpret_cast<uint32_t *>(s_xy + N);
    int *s_start = reinterpret_cast<int *>(s_q + 3 * (size_t)M);
    uint16_t *s_idx = reinterpret_cast<uint16_t *>(s_start + nbins + 1);
    uint16_t *s_order = s_idx + ((N + 1) & ~1);
    int *s_scan = reinterpret_cast<int *>(s_order + ((M + 1) & ~1));
    int *s_hist = s_scan + 16;

    int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dbg_b = (int)gridDim.x / 2 + ((a.B > (int)gridDim.x) ? (int)gridDim.x : 0);  // a problem in the steady state
#define LVT_STAMP(i) if (dbg_on && tid == 0) a.dbg[i] = clock64();

    // coordinates and flags of "my" train features, coordinates of my queries: all in flight at once
    uint64_t tpw[TPT], qpw[QPT];
    uint32_t tfw[TPT];
    auto issue_coords = [&](int pb, uint64_t (&tp_)[TPT], uint32_t (&tf_)[TPT], uint64_t (&qp_)[QPT]) {
        const float2 *txy = hf_uniform(a.t_xy + (size_t)pb * N);
        const uint8_t *tf = hf_uniform(a.t_flag + (size_t)pb * N);
        const float2 *qxy_ = hf_uniform(a.q_xy + (size_t)pb * M);
#pragma unroll
        for (int k = 0; k < TPT; k++) {
            const int j = (k < TPT - 1) ? tid + k * HB_THREADS : min(tid + k * HB_THREADS, N - 1);  // only the last round can pass the end
            tp_[k] = hf_load64(txy, (uint32_t)j * 8u);
            tf_[k] = hf_load8(tf, (uint32_t)j);
        }
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            const int q = (k < QPT - 1) ? tid + k * HB_THREADS : min(tid + k * HB_THREADS, M - 1);
            qp_[k] = hf_load64(qxy_, (uint32_t)q * 8u);
        }
    };
    auto settle_coords = [&](uint64_t (&tp_)[TPT], uint32_t (&tf_)[TPT], uint64_t (&qp_)[QPT]) {
#pragma unroll
        for (int k = 0; k < TPT; k++) hf_settle(tp_[k]), hf_settle(tf_[k]);
#pragma unroll
        for (int k = 0; k < QPT; k++) hf_settle(qp_[k]);
    };
    const bool timeline = a.dbg != nullptr && a.dbg[15] == 12345;   // host asked for one (start, end, CU) record per workgroup
    if (timeline && tid == 0) {
        a.dbg[16 + 4 * (size_t)b] = wall_clock64();
        a.dbg[16 + 4 * (size_t)b + 3] = (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) * 65536 +
                                      (__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xFFFF);
    }
    issue_coords(b, tpw, tfw, qpw);
    hf_wait_vm0();
    settle_coords(tpw, tfw, qpw);

  // resident workgroups (two per CU) walk the problems b, b + grid, ...: no dispatch gap between two problems of a slot (measured:
  // 1.45 us of a 14-us slot period), and the next problem's coordinates arrive while stage B runs
  for (;;) {
    asm volatile("" : "+v"(tid));  // everything derived from the thread index is recomputed per problem (hoisted out of the loop it would
                                   // occupy registers through all of it)
    lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float2 *qxy = a.q_xy + (size_t)b * M;
    const uint4 *qd = reinterpret_cast<const uint4 *>(a.q_desc + (size_t)b * M * 4);
    const uint8_t *td = hf_uniform(reinterpret_cast<const uint8_t *>(a.t_desc) + (size_t)b * N * 32);
    int4 *out = a.out + (size_t)b * M;
    const int b_next = b + (int)gridDim.x;
    const bool have_next = b_next < a.B;
    const bool dbg_on = a.dbg != nullptr && b == dbg_b;
    (void)qxy;
    LVT_STAMP(0)
    if (dbg_on && tid == 0) a.dbg[8] = wall_clock64();
    __builtin_amdgcn_s_setprio(3);
    if (tid <= nbins) s_start[tid] = 0;
    if (tid < HB_HIST) s_hist[tid] = 0;
    float2 tp[TPT], qp[QPT];
    bool tv[TPT];
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        tp[k] = make_float2(__uint_as_float((uint32_t)tpw[k]), __uint_as_float((uint32_t)(tpw[k] >> 32)));
        tv[k] = ((k < TPT - 1) || (tid + k * HB_THREADS < N)) && (tfw[k] == 0);
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) qp[k] = make_float2(__uint_as_float((uint32_t)qpw[k]), __uint_as_float((uint32_t)(qpw[k] >> 32)));
    hf_barrier();
    LVT_STAMP(1)

    // ---- 2. counting sort of the unflagged train features into the 25-px hash cells; the counting atomic returns the rank.
    //         (int) truncates where the reference floors: the two differ for negative quotients only, and those clamp to cell 0 either way
    int tbin[TPT], trank[TPT];
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        const int cy = min(max((int)div_cell(tp[k].y), 0), a.nby - 1);
        const int cx = min(max((int)div_cell(tp[k].x), 0), a.nbx - 1);
        tbin[k] = cy * a.nbx + cx;
        trank[k] = 0;
        if (tv[k]) trank[k] = atomicAdd(&s_start[tbin[k]], 1);
    }
    hf_barrier();
    LVT_STAMP(2)
    {  // counts -> exclusive starts, in place (one entry per thread); entry nbins receives the total
        const int v = (tid <= nbins) ? s_start[tid] : 0;
        const int incl = wave_incl_scan(v);
        if (lane == 63) s_scan[wv] = incl;
        hf_barrier();
        const int part = (lane < HB_WAVES) ? s_scan[lane] : 0;
        const int pin = row16_incl_scan(part);
        const int base = __builtin_amdgcn_readlane(pin, wv) - __builtin_amdgcn_readlane(part, wv);
        if (tid <= nbins) s_start[tid] = base + incl - v;
    }
    hf_barrier();
    LVT_STAMP(3)
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        if (tv[k]) {
            const int pos = s_start[tbin[k]] + trank[k];
            s_xy[pos] = tp[k];
            s_idx[pos] = (uint16_t)(tid + k * HB_THREADS);
        }
    }
    const int n_valid = s_start[nbins];
    hf_barrier();
    // gather DMA: unit u (16 B) of the image = half (u & 1) of the descriptor of the train feature at bin-order position u >> 1.
    // The first c_low 1-KB chunks of the region carry the query coordinates to their sorted slots until stage A has picked them up:
    // those chunks are copied there, everything above them now -- the copy has the query sort and stage A to land
    const int n_units = 2 * n_valid;
    const int c_low = (M * 8 + 1023) >> 10;
    auto dma_chunk = [&](int c) {
        const int u = c * 64 + lane;
        const uint32_t id = s_idx[min(u >> 1, n_valid - 1)];
        hf_dma16(td, id * 32u + (uint32_t)(u & 1) * 16u, __builtin_amdgcn_readfirstlane(hf_lds_address(s_desc) + (uint32_t)c * 1024u));
    };
#pragma unroll
    for (int k = 0; k < 2 * TPT; k++) {
        const int c = wv + k * HB_WAVES;  // 64-unit chunk of this wave
        if (c >= c_low && c * 64 < n_units) dma_chunk(c);
    }

    // ---- 3. the three candidate ranges of a query (struct.cpp:71-83: the cells of one window row are contiguous), packed into two
    //         words (11-bit starts, 6-bit lengths); queries counting-sorted by their candidate count, heaviest first
    int qkey[QPT], qrank[QPT];
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        const int q = tid + k * HB_THREADS;
        qkey[k] = 0, qrank[k] = 0;
        if ((k < QPT - 1) || q < M) {
            const int hy = (int)floorf(div_cell(qp[k].y)), hx = (int)floorf(div_cell(qp[k].x));
            const int x0 = max(hx - 1, 0), x1 = min(hx + 1, a.nbx - 1);
            const bool xok = x0 <= x1;
            int s0 = 0, l0 = 0, s1 = 0, l1 = 0, s2 = 0, l2 = 0;
            const int y0 = max(hy - 1, 0), y1 = min(hy + 1, a.nby - 1);
#define LVT_RANGE(k_, S, L)                                   \
    {                                                         \
        const bool ok = xok && (y0 + k_ <= y1);               \
        const int row = ok ? (y0 + k_) * a.nbx : 0;           \
        S = s_start[row + (ok ? x0 : 0)];                     \
        L = s_start[row + (ok ? x1 + 1 : 0)] - S;             \
    }
            LVT_RANGE(0, s0, l0)
            LVT_RANGE(1, s1, l1)
            LVT_RANGE(2, s2, l2)
#undef LVT_RANGE
            qkey[k] = HB_HIST - 1 - min(l0 + l1 + l2, HB_HIST - 1);
            const bool fits = (l0 < 64) && (l1 < 64) && (l2 < 64);
            s_q[3 * q] = fits ? ((uint32_t)s0 | ((uint32_t)s1 << 11) | ((uint32_t)l0 << 22)) : 0xFFFFFFFFu;
            s_q[3 * q + 1] = (uint32_t)s2 | ((uint32_t)l1 << 11) | ((uint32_t)l2 << 17);
            qrank[k] = atomicAdd(&s_hist[qkey[k]], 1);
        }
    }
    hf_barrier();
    LVT_STAMP(4)
    if (wv == 0) {
        const int v = s_hist[lane];
        s_hist[lane] = wave_incl_scan(v) - v;
    }
    hf_barrier();
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        const int q = tid + k * HB_THREADS;
        if ((k < QPT - 1) || q < M) {
            const int slot = s_hist[qkey[k]] + qrank[k];
            s_order[slot] = (uint16_t)q;
            s_p[slot] = qp[k];  // the descriptor region is still empty: it carries the coordinates to the lane that owns the slot
        }
    }
    hf_barrier();
    LVT_STAMP(5)
    __builtin_amdgcn_s_setprio(0);

    // ---- 4a. the slots of this lane (rounds of 1024 queries in sorted order; odd rounds reverse the wave order)
    int aq[QPT];
    float2 ap[QPT];
#pragma unroll
    for (int j = 0; j < QPT; j++) {
        const int slot = j * HB_THREADS + ((j & 1) ? (HB_WAVES - 1 - wv) : wv) * 64 + lane;
        aq[j] = -1;
        ap[j] = make_float2(0.f, 0.f);
        if (slot < M) aq[j] = s_order[slot], ap[j] = s_p[slot];
    }
    if (tid < HB_HIST) s_hist[tid] = 0;
    hf_barrier();  // every slot's coordinates are in registers: the low chunks of the region may receive their descriptors
#pragma unroll
    for (int k = 0; k < 2 * TPT; k++) {
        const int c = wv + k * HB_WAVES;
        if (c < c_low && c * 64 < n_units) dma_chunk(c);
    }

    // positions of the flattened index space: v in [c_k, c_{k+1}) lies in range k at LDS position v + o_k.  The ranges are padded to
    // EVEN lengths (c_1 = l_0 rounded up, ...): candidates 2w and 2w + 1 of the space are then always neighbours in LDS, stage A
    // reads them with one address computation, and the pad slot of an odd range (whatever follows the range in bin order) is struck
    // from the mask afterwards
#define LVT_POS_OF(dst, v)                  \
    {                                       \
        int o_ = o0;                        \
        o_ = ((v) >= c1) ? o1 : o_;         \
        o_ = ((v) >= c2) ? o2 : o_;         \
        dst = (v) + o_;                     \
    }
    // radius test of the pairs (2w, 2w + 1), w in [lo_w, hi_w), last to first: candidate v lands in bit v - 2 lo_w of the mask
#define LVT_RADIUS_PAIRS(mask, lo_w, hi_w)                                     \
    for (int w_ = (hi_w)-1; w_ >= (lo_w); w_--) {                              \
        int ia_;                                                               \
        LVT_POS_OF(ia_, 2 * w_)                                                \
        const float2 ra_ = s_xy[ia_], rb_ = s_xy[ia_ + 1];                     \
        const float dxa_ = ra_.x - p.x, dya_ = ra_.y - p.y;                    \
        const float dxb_ = rb_.x - p.x, dyb_ = rb_.y - p.y;                    \
        push_bit(mask, dxb_ * dxb_ + dyb_ * dyb_, a.r2);                       \
        push_bit(mask, dxa_ * dxa_ + dya_ * dya_, a.r2);                       \
    }
    // the mask: 32 bits in word 2, up to 8 more (the padded total is even, <= 40) in the free top of word 1
    int akey[QPT], arank[QPT], slowq[QPT];
#pragma unroll
    for (int j = 0; j < QPT; j++) {
        akey[j] = 0, arank[j] = 0, slowq[j] = -1;
        const int q = aq[j];
        if (q >= 0) {
            const float2 p = ap[j];
            const uint32_t W0 = s_q[3 * q], W1 = s_q[3 * q + 1];
            const int l0 = (int)(W0 >> 22), l1 = (int)((W1 >> 11) & 63u), l2 = (int)((W1 >> 17) & 63u);
            const int c1 = (l0 + 1) & ~1, c2 = c1 + ((l1 + 1) & ~1), total = c2 + ((l2 + 1) & ~1);
            const int o0 = (int)(W0 & 2047u), o1 = (int)((W0 >> 11) & 2047u) - c1, o2 = (int)(W1 & 2047u) - c2;
            if (W0 == 0xFFFFFFFFu || total > 40) {  // ranges or mask do not fit their slot: matched after the DMA has landed (rare)
                slowq[j] = q;
                aq[j] = -1;
            } else {
                uint32_t lo = 0, hi = 0;
                const int t0 = min(total, 32) >> 1;
                LVT_RADIUS_PAIRS(lo, 0, t0)
                LVT_RADIUS_PAIRS(hi, 16, total >> 1)
                // strike the pad slots: range k owns the bits [c_k, c_k + l_k)
                const uint64_t own = ((1ull << l0) - 1) | (((1ull << l1) - 1) << c1) | (((1ull << l2) - 1) << c2);
                lo &= (uint32_t)own;
                hi &= (uint32_t)(own >> 32);
                s_q[3 * q + 2] = lo;
                s_q[3 * q + 1] = (W1 & 0x7FFFFFu) | (hi << 23);
                akey[j] = HB_HIST - 1 - min(__popc(lo) + __popc(hi), HB_HIST - 1);
            }
        }
    }
    // ---- 4b. the queries again, sorted by the number of candidates inside the circle
    hf_barrier();  // s_hist zeroed, every read of s_order done
#pragma unroll
    for (int j = 0; j < QPT; j++)
        if (aq[j] >= 0) arank[j] = atomicAdd(&s_hist[akey[j]], 1);
    hf_barrier();
    if (wv == 0) {
        const int v = s_hist[lane];
        const int incl = wave_incl_scan(v);
        s_hist[lane] = incl - v;
        if (lane == 63) s_scan[0] = incl;
    }
    hf_barrier();
    const int M2 = s_scan[0];
#pragma unroll
    for (int j = 0; j < QPT; j++)
        if (aq[j] >= 0) s_order[s_hist[akey[j]] + arank[j]] = (uint16_t)aq[j];
    hf_wait_vm0();  // this wave's part of the descriptor image has landed ...
    hf_barrier();   // ... and so has everybody's
    LVT_STAMP(7)

    // ---- 5. stage B: descriptors of the candidates inside the circle; top-2 as packed (distance << 16 | index) keys
    int bq[QPT];
#pragma unroll
    for (int j = 0; j < QPT; j++) {
        const int slot = j * HB_THREADS + ((j & 1) ? (HB_WAVES - 1 - wv) : wv) * 64 + lane;
        bq[j] = (slot < M2) ? (int)s_order[slot] : -1;
    }
    uint4 w0[QPT], w1[QPT];
#pragma unroll
    for (int j = 0; j < QPT; j++) w0[j] = qd[2 * max(bq[j], 0)], w1[j] = qd[2 * max(bq[j], 0) + 1];
    // the next problem's coordinates, requested right behind the query descriptors (results return in order: they delay nothing)
    uint64_t ntpw[TPT], nqpw[QPT];
    uint32_t ntfw[TPT];
#pragma unroll
    for (int k = 0; k < TPT; k++) ntpw[k] = 0, ntfw[k] = 0;
#pragma unroll
    for (int k = 0; k < QPT; k++) nqpw[k] = 0;
    if (have_next) issue_coords(b_next, ntpw, ntfw, nqpw);
#define LVT_STORE(q_, k1_, k2_)                                                        \
    {                                                                                  \
        int4 o;                                                                        \
        o.x = ((k1_) == 0xFFFFFFFFu) ? -1 : (int)((k1_)&0xFFFFu);                      \
        o.y = ((k1_) == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)((k1_) >> 16);                \
        o.z = ((k2_) == 0xFFFFFFFFu) ? -1 : (int)((k2_)&0xFFFFu);                      \
        o.w = ((k2_) == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)((k2_) >> 16);                \
        out[q_] = o;                                                                   \
    }
#define LVT_DIST(acc, a0_, a1_)                                      \
    {                                                                \
        acc = hf_bcnt((a0_).x ^ d0, 0);                              \
        acc = hf_bcnt((a0_).y ^ d1, acc);                            \
        acc = hf_bcnt((a0_).z ^ d2, acc);                            \
        acc = hf_bcnt((a0_).w ^ d3, acc);                            \
        acc = hf_bcnt((a1_).x ^ d4, acc);                            \
        acc = hf_bcnt((a1_).y ^ d5, acc);                            \
        acc = hf_bcnt((a1_).z ^ d6, acc);                            \
        acc = hf_bcnt((a1_).w ^ d7, acc);                            \
    }
#pragma unroll
    for (int j = 0; j < QPT; j++) {  // the rare queries whose window did not fit the mask: every candidate of the three ranges
        const int q = slowq[j];
        if (q >= 0) {
            const float2 p = ap[j];
            const uint4 v0 = qd[2 * q], v1 = qd[2 * q + 1];
            const uint32_t d0 = v0.x, d1 = v0.y, d2 = v0.z, d3 = v0.w, d4 = v1.x, d5 = v1.y, d6 = v1.z, d7 = v1.w;
            const int hy = (int)floorf(div_cell(p.y)), hx = (int)floorf(div_cell(p.x));
            const int x0 = max(hx - 1, 0), x1 = min(hx + 1, a.nbx - 1);
            uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
            if (x0 <= x1)
                for (int by = max(hy - 1, 0); by <= min(hy + 1, a.nby - 1); by++)
                    for (int it = s_start[by * a.nbx + x0]; it < s_start[by * a.nbx + x1 + 1]; it++) {
                        const float2 r = s_xy[it];
                        const float dx = r.x - p.x, dy = r.y - p.y;
                        if (dx * dx + dy * dy < a.r2) {
                            const uint4 a0 = s_desc[2 * it], a1 = s_desc[2 * it + 1];
                            int d;
                            LVT_DIST(d, a0, a1)
                            const uint32_t key = ((uint32_t)d << 16) | s_idx[it];
                            k2 = min(k2, max(k1, key));
                            k1 = min(k1, key);
                        }
                    }
            LVT_STORE(q, k1, k2)
        }
    }
    uint32_t rk1[QPT], rk2[QPT];
#pragma unroll
    for (int j = 0; j < QPT; j++) {
        const int q = bq[j];
        rk1[j] = rk2[j] = 0xFFFFFFFFu;
        if (q >= 0) {
            const uint32_t d0 = w0[j].x, d1 = w0[j].y, d2 = w0[j].z, d3 = w0[j].w, d4 = w1[j].x, d5 = w1[j].y, d6 = w1[j].z, d7 = w1[j].w;
            uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
            const uint32_t W0 = s_q[3 * q], W1 = s_q[3 * q + 1], W2 = s_q[3 * q + 2];
            const int c1 = ((int)(W0 >> 22) + 1) & ~1, c2 = c1 + (((int)((W1 >> 11) & 63u) + 1) & ~1);
            const int o0 = (int)(W0 & 2047u), o1 = (int)((W0 >> 11) & 2047u) - c1, o2 = (int)(W1 & 2047u) - c2;
            // set bits of m (candidates base + bit of the flattened index space); the next candidate's LDS reads are issued before the
            // current one is evaluated
#define LVT_WALK_BITS(m_in, base)                                                  \
    {                                                                              \
        uint32_t m_ = (m_in);                                                      \
        if (m_ != 0) {                                                             \
            int it_;                                                               \
            {                                                                      \
                const int v_ = (base) + __builtin_ctz(m_);                         \
                LVT_POS_OF(it_, v_)                                                \
            }                                                                      \
            m_ &= m_ - 1;                                                          \
            uint4 a0_ = s_desc[2 * it_], a1_ = s_desc[2 * it_ + 1];                \
            uint32_t id_ = s_idx[it_];                                             \
            while (m_ != 0) {                                                      \
                {                                                                  \
                    const int v_ = (base) + __builtin_ctz(m_);                     \
                    LVT_POS_OF(it_, v_)                                            \
                }                                                                  \
                m_ &= m_ - 1;                                                      \
                const uint4 b0_ = s_desc[2 * it_], b1_ = s_desc[2 * it_ + 1];      \
                const uint32_t idn_ = s_idx[it_];                                  \
                int d_;                                                            \
                LVT_DIST(d_, a0_, a1_)                                             \
                const uint32_t key_ = ((uint32_t)d_ << 16) | id_;                  \
                k2 = hf_med3(k1, k2, key_);                         \
                k1 = min(k1, key_);                                                \
                a0_ = b0_, a1_ = b1_, id_ = idn_;                                  \
            }                                                                      \
            int d_;                                                                \
            LVT_DIST(d_, a0_, a1_)                                                 \
            const uint32_t key_ = ((uint32_t)d_ << 16) | id_;                      \
            k2 = hf_med3(k1, k2, key_);                             \
            k1 = min(k1, key_);                                                    \
        }                                                                          \
    }
            LVT_WALK_BITS(W2, 0)
            LVT_WALK_BITS(W1 >> 23, 32)
            rk1[j] = k1, rk2[j] = k2;
        }
    }
    // only loads are outstanding (query descriptors, the next problem's coordinates): once they are here the coordinates may be
    // copied around freely, and the stores that follow are never waited for
    hf_wait_vm0();
    settle_coords(ntpw, ntfw, nqpw);
#pragma unroll
    for (int j = 0; j < QPT; j++)
        if (bq[j] >= 0) LVT_STORE(bq[j], rk1[j], rk2[j])
#undef LVT_WALK_BITS
#undef LVT_DIST
#undef LVT_STORE
#undef LVT_RADIUS_PAIRS
#undef LVT_POS_OF
    LVT_STAMP(6)
    if (dbg_on && tid == 0) a.dbg[9] = wall_clock64();
    if (!have_next) {
        if (timeline && tid == 0) a.dbg[16 + 4 * (size_t)blockIdx.x + 1] = a.dbg[16 + 4 * (size_t)blockIdx.x + 2] = wall_clock64();
        break;
    }
#pragma unroll
    for (int k = 0; k < TPT; k++) tpw[k] = ntpw[k], tfw[k] = ntfw[k];
#pragma unroll
    for (int k = 0; k < QPT; k++) qpw[k] = nqpw[k];
    b = b_next;
    hf_barrier();  // every wave has left stage B: the LDS arrays may be rewritten
  }
#undef LVT_STAMP
}

}  // namespace lvt
