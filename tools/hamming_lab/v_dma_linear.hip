// k_hamming.hip -- batched masked 2-NN Hamming matcher (the matcher of lvt_image_features_struct.cpp:68-148
// + cv::BFMatcher::knnMatch(k=2, mask)) as ONE launch over B independent problems.
//
// One 1024-thread workgroup per problem, two workgroups per CU (79 KB of LDS each at N = 1500, M = 1000; <= 64 VGPRs, so
// the CU holds its maximum of 8 waves per SIMD).  Round 3: the train DESCRIPTORS (41 % of a problem's bytes) never pass
// through registers.  Every wave copies its share of the 32-B records straight from HBM into LDS with LDS-DMA loads
// (global_load_lds_dwordx4, 1 KB per instruction) IN INPUT ORDER, issued behind the problem's coordinate loads; the
// workgroup waits for the coordinates only (a counted vmcnt) and bins, sorts and radius-tests while the descriptors are
// still in flight -- they are first read in stage B.  What is counting-sorted into bin order is the 10-byte (x, y, index)
// part of a train feature; stage B reaches a descriptor through the index it needs for the result key anyway.
//   1. coordinate / flag loads (inline asm: their waits are counted by hand, the compiler must not drain the DMA for
//      them), then the descriptor DMA; all bytes of the problem are in flight at once;
//   2. the unflagged train features are counting-sorted into the reference's 25-px hash cells (tracking mode) or
//      image rows (row mode): the LDS atomicAdd that counts a bin also returns the feature's rank inside it, so after
//      one in-place scan of the counts each feature is stored at start[bin] + rank; the candidates of a query are then
//      <= 2*csr+1 contiguous LDS ranges;
//   3. the queries are counting-sorted by their candidate count (known from the bin starts alone), so the 64 queries
//      a wavefront works on in one round need the same number of steps;
//   4. stage A: the radius test alone over every window candidate -> one bit per candidate of the query's flattened
//      index space; the queries are re-sorted by the number of candidates inside the circle;
//   5. stage B (after the DMA has landed): each lane owns one query per round (descriptor in 4 x u64 VGPR pairs), walks
//      the set bits, evaluates 4 x (xor, popcount) per candidate and keeps the running top-2 as packed
//      (distance << 16 | index) keys, so ties resolve to the lowest index exactly as batchDistance does.
// Between the DMA issue and stage B every barrier is a raw s_barrier behind lgkmcnt(0): __syncthreads() would wait for
// vmcnt(0) and serialise the copy with the sort.
// PERSIST = true: 2 x 256 resident workgroups walk the problems b, b + grid, ...; the next problem's coordinates are
// fetched into registers while stage B runs and its descriptor DMA starts the moment stage B has released the LDS image.
// Algorithmic HBM bytes per problem: 40*(M+N) + N + 16*M; every byte is read or written exactly once.
#include "lvt_dev.h"

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
    int nbx, nby, csr;       // bins: hash cells (mode 0) or rows (mode 1: nbx = 1, nby = rows + 1)
    long long *dbg;          // optional: phase cycle stamps of one workgroup
    int stagger;             // PERSIST: start offset between resident workgroups, in units of 1024 cycles (0: none)
};

constexpr int HB_THREADS = 1024;
constexpr int HB_WAVES = HB_THREADS / 64;
constexpr int HB_TPT = 4;          // train features per thread  => N <= 4096 (LDS permitting)
constexpr int HB_QPT = 4;          // queries per thread         => M <= 4096 (LDS permitting)
constexpr int HB_NMAX = HB_THREADS * HB_TPT;
constexpr int HB_MMAX = HB_THREADS * HB_QPT;
constexpr int HB_HIST = 64;        // query classes by candidate count (>= 63 candidates share the first class)

// The descriptor image in LDS: N records of 32 B in input order = 2N units of 16 B, cut into 16 contiguous per-wave
// spans; a wave copies its span with ceil(span / 64) DMA instructions of 64 units.  The lanes of a span's last
// instruction that reach past its end copy the first units of the NEXT span (the right bytes for those addresses), past
// the end of the image they re-read its last unit into the padding.
__host__ __device__ static inline int hb_units_per_wave(int N) { return (2 * N + HB_WAVES - 1) / HB_WAVES; }
__host__ __device__ static inline int hb_dma_per_wave(int N) { return (hb_units_per_wave(N) + 63) >> 6; }
__host__ __device__ static inline int hb_desc_region_bytes(int N) { return ((HB_WAVES - 1) * hb_units_per_wave(N) + hb_dma_per_wave(N) * 64) * 16; }

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

// ---- memory operations the compiler must not see (it would wait vmcnt(0) at every LDS access behind an LDS-DMA and at
//      the first use of any ordinary load issued beside one; cdna_hip_programming.md, "Pipelining across barriers")
__device__ __forceinline__ uint64_t hb_load64(const void *p) {
    uint64_t v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t hb_load8(const void *p) {
    uint32_t v;
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// the value of an asm load may only be used behind the hb_wait_vm that covers it: this ties the use to that point
__device__ __forceinline__ void hb_settle(uint64_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void hb_settle(uint32_t &v) { asm volatile("" : "+v"(v)); }
// 16 B per lane from global memory to LDS byte address lds_base + 16 * lane (lds_base wave-uniform, in an SGPR)
__device__ __forceinline__ void hb_dma16(const void *g, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_base) : "memory", "m0");
}
// wait until at most n vector-memory operations of this wave are outstanding (they complete in issue order)
__device__ __forceinline__ void hb_wait_vm(int n) {
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}
// workgroup barrier that orders LDS accesses only: an LDS-DMA in flight stays in flight
__device__ __forceinline__ void hb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __attribute__((address_space(3))) uint8_t hb_lds_byte;
__device__ __forceinline__ uint32_t hb_lds_address(void *p) { return (uint32_t)(uintptr_t)(hb_lds_byte *)p; }

// exclusive scan over the 1024 threads with raw barriers; scratch >= 16 ints
__device__ __forceinline__ int hb_block_excl_scan(int v, int *scratch, int *total) {
    const int incl = wave_incl_scan(v);
    const int w = __builtin_amdgcn_readfirstlane(wave_id()), l = lane_id();
    hb_barrier();  // readers of the previous call are done with scratch
    if (l == 63) scratch[w] = incl;
    hb_barrier();
    const int part = (l < HB_WAVES) ? scratch[l] : 0;
    const int pin = row16_incl_scan(part);  // lanes 0..15: inclusive prefix of the wave totals
    const int base = __builtin_amdgcn_readlane(pin, w) - __builtin_amdgcn_readlane(part, w);
    *total = __builtin_amdgcn_readlane(pin, 15);
    return base + incl - v;
}

// NSP = number of candidate ranges a query keeps in registers: 1 (row mode), 3 (csr 1), 5 (csr 2); 0 = any csr, the
// ranges are walked one after the other (no flattening)
template <int MODE, int NSP, int QPT, int TPT, bool PERSIST>  // QPT = ceil(M / 1024) rounds of queries, TPT = ceil(N / 1024) train features per thread
__global__ __launch_bounds__(HB_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_hamming_batched(HammingArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int N = a.N, M = a.M;
    const int nbins = a.nbx * a.nby;
    // carve: desc image (DMA target, input order) | xy [N] float2, bin order | per-query slot [M] 12 B | start [nbins + 1] |
    //        idx [N] u16, bin order | order [M] u16
    uint4 *s_desc = reinterpret_cast<uint4 *>(smem);  // record j = units 2j (low half), 2j + 1 (high half)
    float2 *s_xy = reinterpret_cast<float2 *>(smem + hb_desc_region_bytes(N));
    uint2 *s_mask = reinterpret_cast<uint2 *>(s_xy + N);
    uint32_t *s_q = reinterpret_cast<uint32_t *>(s_mask);  // 12 B per query: packed ranges + candidate mask (see stage 4)
    int *s_start = reinterpret_cast<int *>(s_q + 3 * (size_t)M);
    uint16_t *s_idx = reinterpret_cast<uint16_t *>(s_start + nbins + 1);
    uint16_t *s_order = s_idx + ((N + 1) & ~1);
    __shared__ int s_scan[32];
    __shared__ int s_hist[HB_HIST];

    int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int units = 2 * N, upw = hb_units_per_wave(N), ng = hb_dma_per_wave(N);
    const uint32_t lds_image = hb_lds_address(s_desc);

    auto issue_desc = [&](int b) {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(a.t_desc) + (size_t)b * N * 32;
#pragma unroll
        for (int k = 0; k < 2 * TPT; k++)
            if (k < ng) {
                const int u = min(wv * upw + k * 64 + lane, units - 1);
                hb_dma16(src + (size_t)u * 16, __builtin_amdgcn_readfirstlane(lds_image + (uint32_t)(wv * upw) * 16u + (uint32_t)k * 1024u));  // this wave's span of the image
            }
    };
    uint64_t tpw[TPT], qpw[QPT];
    uint32_t tfw[TPT];
    auto issue_train_coords = [&](int b, uint64_t (&tp_)[TPT], uint32_t (&tf_)[TPT]) {
        const float2 *txy = a.t_xy + (size_t)b * N;
        const uint8_t *tf = a.t_flag + (size_t)b * N;
#pragma unroll
        for (int k = 0; k < TPT; k++) {  // indices clamped: no branches
            const int jc = min(tid + k * HB_THREADS, N - 1);
            tp_[k] = hb_load64(txy + jc);
            tf_[k] = hb_load8(tf + jc);
        }
    };
    auto issue_query_coords = [&](int b, uint64_t (&qp_)[QPT]) {
        const float2 *qxy = a.q_xy + (size_t)b * M;
#pragma unroll
        for (int k = 0; k < QPT; k++) qp_[k] = hb_load64(qxy + min(tid + k * HB_THREADS, M - 1));
    };
    // a train feature that is flagged (or lies behind the end of the set) gets a NaN x: it takes part in nothing (no bin, no radius
    // or band test passes), and the flag needs no register of its own
    auto fold_flags = [&](uint64_t (&tp_)[TPT], uint32_t (&tf_)[TPT]) {
#pragma unroll
        for (int k = 0; k < TPT; k++) {
            hb_settle(tp_[k]);
            hb_settle(tf_[k]);
            const bool valid = (tid + k * HB_THREADS < N) & ((tf_[k] & 0xFFu) == 0);
            if (!valid) tp_[k] = (tp_[k] & 0xFFFFFFFF00000000ull) | 0x7FC00000ull;
        }
    };

    int b = blockIdx.x;
    if (PERSIST && a.stagger > 0) {
        // resident workgroups that start together stay in lock-step (same phase on every CU at the same time: the whole chip asks HBM
        // for its next problem at once, then nobody does): the second workgroup of a CU starts half a problem later, and the CUs are
        // spread over a further fraction
        const int units_ = (int)((blockIdx.x >= gridDim.x / 2) ? 8 : 0) + (int)(blockIdx.x & 7);
        for (int i = 0; i < units_ * a.stagger; i++) __builtin_amdgcn_s_sleep(16);  // 1024 cycles each
    }
    // phase stamps of one workgroup (its first problem); the condition is wave-uniform except for the lane
    const int dbg_b = PERSIST ? (int)gridDim.x / 2 + 8 * (int)gridDim.x : (int)gridDim.x / 2;  // a problem in the steady state of the launch
    bool dbg_on = a.dbg != nullptr && b == dbg_b;
#define LVT_STAMP(i) if (dbg_on && tid == 0) a.dbg[i] = clock64();
    LVT_STAMP(0)
    // ---- 1. everything of the problem is requested from HBM back to back: coordinates and flags first (they are waited for
    //         first and vector-memory results return in order), the descriptors behind them
    issue_train_coords(b, tpw, tfw);
    issue_query_coords(b, qpw);
    issue_desc(b);
    bool first = true;

    do {
        if (PERSIST) {  // everything derived from the thread index is recomputed per problem: hoisted out of this loop it would
                        // occupy (and spill) registers through all of it
            asm volatile("" : "+v"(tid));
            lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        }
        const float2 *qxy = a.q_xy + (size_t)b * M;
        const uint4 *qd = reinterpret_cast<const uint4 *>(a.q_desc + (size_t)b * M * 4);
        int4 *out = a.out + (size_t)b * M;
        const int b_next = b + (int)gridDim.x;
        const bool have_next = PERSIST && b_next < a.B;
        __builtin_amdgcn_s_setprio(3);  // the load / sort phases are latency-bound: let them through ahead of the other
                                        // workgroup's VALU-bound query phases
        for (int i = tid; i <= nbins; i += HB_THREADS) s_start[i] = 0;
        if (tid < HB_HIST) s_hist[tid] = 0;
        if (!PERSIST || first) {
            hb_wait_vm(ng);  // the coordinates are here; the ng DMA instructions behind them stay in flight
            fold_flags(tpw, tfw);
        }  // (PERSIST, later problems: the train coordinates were settled and folded at the head of the previous stage B)
        float2 tp[TPT], qp[QPT];
        bool tv[TPT];
#pragma unroll
        for (int k = 0; k < TPT; k++) {
            tp[k] = make_float2(__uint_as_float((uint32_t)tpw[k]), __uint_as_float((uint32_t)(tpw[k] >> 32)));
            tv[k] = tp[k].x == tp[k].x;
        }
        hb_barrier();
        LVT_STAMP(1)

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
        hb_barrier();
        LVT_STAMP(2)
        {  // counts -> exclusive starts, in place; entry nbins receives the total.  One contiguous chunk per thread.
            const int chunk = (nbins + 1 + HB_THREADS - 1) / HB_THREADS;
            const int i0 = min(tid * chunk, nbins + 1), i1 = min(i0 + chunk, nbins + 1);
            int sum = 0;
            for (int i = i0; i < i1; i++) sum += s_start[i];
            int total;
            int run = hb_block_excl_scan(sum, s_scan, &total);
            for (int i = i0; i < i1; i++) {
                const int v = s_start[i];
                s_start[i] = run;
                run += v;
            }
        }
        hb_barrier();
        LVT_STAMP(3)
#pragma unroll
        for (int k = 0; k < TPT; k++) {
            if (tv[k]) {
                const int pos = s_start[tbin[k]] + trank[k];
                s_xy[pos] = tp[k];
                s_idx[pos] = (uint16_t)(tid + k * HB_THREADS);
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
        if (PERSIST && !first) hb_wait_vm(ng);  // the query coordinates were requested just ahead of the DMA
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            hb_settle(qpw[k]);
            qp[k] = make_float2(__uint_as_float((uint32_t)qpw[k]), __uint_as_float((uint32_t)(qpw[k] >> 32)));
        }
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
                    qkey[k] = HB_HIST - 1 - min(count_of(qp[k]), HB_HIST - 1);
                    // stage 4a visits the queries in sorted order: it finds the coordinates in the query's (still unused) mask
                    // slot instead of going back to HBM for them
                    if (MODE == 0 && NSP > 0) s_mask[q] = make_uint2(__float_as_uint(qp[k].x), __float_as_uint(qp[k].y));
                }
                qrank[k] = atomicAdd(&s_hist[qkey[k]], 1);
            }
        }
        hb_barrier();
        LVT_STAMP(4)
        if (wv == 0) {
            const int v = s_hist[lane];
            s_hist[lane] = wave_incl_scan(v) - v;
        }
        hb_barrier();
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            const int q = tid + k * HB_THREADS;
            if (q < M) s_order[s_hist[qkey[k]] + qrank[k]] = (uint16_t)q;
        }
        hb_barrier();
        LVT_STAMP(5)

        __builtin_amdgcn_s_setprio(0);
        // ---- 4. rounds of 1024 queries in sorted order; odd rounds reverse the wave order so every wave gets a similar sum
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
        // distance of the query (d0..d3) to train feature id, folded into the running top-2
#define LVT_FOLD(a0_, a1_, id_)                                                                                                  \
    {                                                                                                                            \
        const int d_ = __popcll(d0 ^ (((uint64_t)(a0_).y << 32) | (a0_).x)) + __popcll(d1 ^ (((uint64_t)(a0_).w << 32) | (a0_).z)) + \
                       __popcll(d2 ^ (((uint64_t)(a1_).y << 32) | (a1_).x)) + __popcll(d3 ^ (((uint64_t)(a1_).w << 32) | (a1_).z));  \
        const uint32_t key_ = ((uint32_t)d_ << 16) | (uint32_t)(id_);                                                            \
        k2 = min(k2, max(k1, key_));                                                                                             \
        k1 = min(k1, key_);                                                                                                      \
    }
        const uint32_t id_max = (uint32_t)(N - 1);
        // all candidates of the ranges, filter and distance in one pass (row mode, any-csr mode, over-long windows); software
        // pipeline: (coordinates, index) two candidates ahead, descriptor one ahead; prefetches past the end read valid LDS
        auto walk_all = [&](float2 p, float fy0, float fy1, uint64_t d0, uint64_t d1, uint64_t d2, uint64_t d3, uint32_t &k1, uint32_t &k2,
                            int total, int o0, int c1, int o1, int c2, int o2, int c3, int o3, int c4, int o4) {
            if (total <= 0) return;
            int it;
            LVT_POS_OF(it, 0)
            float2 r = s_xy[it];
            uint32_t id = s_idx[it];
            LVT_POS_OF(it, 1)
            float2 rn = s_xy[it];
            uint32_t idn = min((uint32_t)s_idx[it], id_max);
            uint4 a0 = s_desc[2 * id], a1 = s_desc[2 * id + 1];
            for (int v = 0; v < total; v++) {
                int itf;
                LVT_POS_OF(itf, v + 2)
                const float2 rf = s_xy[itf];
                const uint32_t idf = min((uint32_t)s_idx[itf], id_max);
                const uint4 b0 = s_desc[2 * idn], b1 = s_desc[2 * idn + 1];
                bool ok;
                if (MODE == 1) ok = (r.y >= fy0) && (r.y <= fy1);
                else {
                    const float dx = r.x - p.x, dy = r.y - p.y;
                    ok = (dx * dx + dy * dy) < a.r2;
                }
                LVT_FOLD(a0, a1, ok ? id : 0xFFFFFFFFu)  // a candidate outside the mask folds the neutral key
                r = rn, id = idn, a0 = b0, a1 = b1;
                rn = rf, idn = idf;
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
                        walk_all(p, fy0, fy1, d0, d1, d2, d3, k1, k2, s_start[by * a.nbx + R.x1 + 1] - s, s, 0, 0, 0, 0, 0, 0, 0, 0);
                    }
            } else {
                static_assert(NS <= 5, "range registers");
                const int c1 = R.l0, c2 = c1 + R.l1, c3 = c2 + R.l2, c4 = c3 + R.l3;
                walk_all(p, fy0, fy1, d0, d1, d2, d3, k1, k2, c4 + R.l4, R.s0, c1, R.s1 - c1, c2, R.s2 - c2, c3, R.s3 - c3, c4, R.s4 - c4);
            }
            int4 o;
            o.x = (k1 == 0xFFFFFFFFu) ? -1 : (int)(k1 & 0xFFFFu);
            o.y = (k1 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k1 >> 16);
            o.z = (k2 == 0xFFFFFFFFu) ? -1 : (int)(k2 & 0xFFFFu);
            o.w = (k2 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k2 >> 16);
            out[q] = o;
        };
        // set bits of m (candidates base + bit of the flattened index space), same software pipeline as walk_all
#define LVT_WALK_BITS(m_in, base)                                              \
    {                                                                          \
        uint32_t m_ = (m_in);                                                  \
        if (m_ != 0) {                                                         \
            int it_;                                                           \
            {                                                                  \
                const int v_ = (base) + __ffs((int)m_) - 1;                    \
                LVT_POS_OF(it_, v_)                                            \
            }                                                                  \
            m_ &= m_ - 1;                                                      \
            uint32_t id_ = s_idx[it_];                                         \
            bool more_ = m_ != 0;                                              \
            {                                                                  \
                const int v_ = (base) + ((__ffs((int)m_) - 1) & 31);           \
                LVT_POS_OF(it_, v_)                                            \
            }                                                                  \
            m_ &= m_ - 1;                                                      \
            uint32_t idn_ = min((uint32_t)s_idx[it_], id_max);                 \
            uint4 a0_ = s_desc[2 * id_], a1_ = s_desc[2 * id_ + 1];            \
            for (;;) {                                                         \
                const bool moren_ = m_ != 0;                                   \
                {                                                              \
                    const int v_ = (base) + ((__ffs((int)m_) - 1) & 31);       \
                    LVT_POS_OF(it_, v_)                                        \
                }                                                              \
                m_ &= m_ - 1;                                                  \
                const uint32_t idf_ = min((uint32_t)s_idx[it_], id_max);       \
                const uint4 b0_ = s_desc[2 * idn_], b1_ = s_desc[2 * idn_ + 1]; \
                LVT_FOLD(a0_, a1_, id_)                                        \
                if (!more_) break;                                             \
                a0_ = b0_, a1_ = b1_, id_ = idn_, idn_ = idf_, more_ = moren_; \
            }                                                                  \
        }                                                                      \
    }
        // before stage B: this wave's DMA has landed, then every wave's; PERSIST: the next problem's coordinates are requested
        // behind the query descriptors of the first round (results return in order: they delay nothing)
        uint64_t ntpw[TPT];
#pragma unroll
        for (int k = 0; k < TPT; k++) ntpw[k] = 0;

        auto prefetch_next = [&]() {
            if (!PERSIST) return;
            // the next problem's train coordinates arrive right behind the query descriptors this wave is about to wait for anyway;
            // settled here, the values may be copied freely afterwards
            uint32_t ntfw[TPT];
#pragma unroll
            for (int k = 0; k < TPT; k++) ntfw[k] = 0;
            if (have_next) issue_train_coords(b_next, ntpw, ntfw);
            hb_wait_vm(0);
            fold_flags(ntpw, ntfw);
        };
        constexpr bool TWO_STAGE = (MODE == 0 && NSP > 0);
        constexpr bool PACKED = (MODE == 0 && NSP == 3);
        if (PACKED) {
            // ---- 4a (packed ranges). radius test over every window candidate -> one bit per candidate of the flattened index space
            constexpr int MASK_BITS = 41;  // 32 in word 2, 9 in the free top of word 1
            if (tid < HB_HIST) s_hist[tid] = 0;
            int aq[QPT], akey[QPT], arank[QPT], slowq[QPT];
            {
                int q = slot_query(0, M);
                float2 p = qxy[max(q, 0)];
#pragma unroll
                for (int j = 0; j < QPT; j++) {
                    aq[j] = -1, akey[j] = 0, arank[j] = 0, slowq[j] = -1;
                    const int qn = slot_query(j + 1, M);
                    const float2 np = qxy[max(qn, 0)];
                    if (q >= 0) {
                        const uint32_t W0 = s_q[3 * q], W1 = s_q[3 * q + 1];
                        const bool packed = W0 != 0xFFFFFFFFu;
                        const int l0 = (int)(W0 >> 22), l1 = (int)((W1 >> 11) & 63u), l2 = (int)((W1 >> 17) & 63u);
                        const int c1 = l0, c2 = c1 + l1, total = c2 + l2;
                        const int o0 = (int)(W0 & 2047u), o1 = (int)((W0 >> 11) & 2047u) - c1, o2 = (int)(W1 & 2047u) - c2;
                        const int c3 = 0, c4 = 0, o3 = 0, o4 = 0;
                        (void)c3, (void)c4, (void)o3, (void)o4;
                        if (!packed || total > MASK_BITS) {  // ranges or mask do not fit their slot: matched after the DMA has landed (rare)
                            slowq[j] = q;
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
            hb_barrier();  // s_hist zeroed, every stage-4a read of s_order done
#pragma unroll
            for (int j = 0; j < QPT; j++)
                if (aq[j] >= 0) arank[j] = atomicAdd(&s_hist[akey[j]], 1);
            hb_barrier();
            if (wv == 0) {
                const int v = s_hist[lane];
                const int incl = wave_incl_scan(v);
                s_hist[lane] = incl - v;
                if (lane == 63) s_scan[0] = incl;
            }
            hb_barrier();
            const int M2 = s_scan[0];
#pragma unroll
            for (int j = 0; j < QPT; j++)
                if (aq[j] >= 0) s_order[s_hist[akey[j]] + arank[j]] = (uint16_t)aq[j];
            hb_wait_vm(0);  // this wave's part of the descriptor image has landed ...
            hb_barrier();   // ... and so has everybody's
            LVT_STAMP(7)

            int q = slot_query(0, M2);
            uint4 w0 = qd[2 * max(q, 0)], w1 = qd[2 * max(q, 0) + 1];
            prefetch_next();
#pragma unroll
            for (int j = 0; j < QPT; j++)
                if (slowq[j] >= 0) {
                    const float2 p = qxy[slowq[j]];
                    match_all(slowq[j], p, qd[2 * slowq[j]], qd[2 * slowq[j] + 1], ranges(p));
                }
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
                    LVT_WALK_BITS(W2, 0)
                    LVT_WALK_BITS(W1 >> 23, 32)
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
            hb_wait_vm(0);
            hb_barrier();
            LVT_STAMP(7)
            int q = slot_query(0, M);
            uint4 w0 = qd[2 * max(q, 0)], w1 = qd[2 * max(q, 0) + 1];
            float2 p = qxy[max(q, 0)];
            prefetch_next();
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
            int aq[QPT], akey[QPT], arank[QPT], slowq[QPT];
            {
                int q = slot_query(0, M);
#pragma unroll
                for (int j = 0; j < QPT; j++) {
                    aq[j] = -1, akey[j] = 0, arank[j] = 0, slowq[j] = -1;
                    {
                        const int qn = slot_query(j + 1, M);
                        if (q >= 0) {
                            const uint2 pw = s_mask[q];  // coordinates stashed by the pre-pass; the slot receives the mask below
                            const float2 p = make_float2(__uint_as_float(pw.x), __uint_as_float(pw.y));
                            const Ranges R = ranges(p);
                            const int c1 = R.l0, c2 = c1 + R.l1, c3 = c2 + R.l2, c4 = c3 + R.l3, total = c4 + R.l4;
                            const int o0 = R.s0, o1 = R.s1 - c1, o2 = R.s2 - c2, o3 = R.s3 - c3, o4 = R.s4 - c4;
                            if (total > 64) {  // does not fit the mask: matched after the DMA has landed (rare)
                                slowq[j] = q;
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
            hb_barrier();  // s_hist zeroed, every stage-4a read of s_order done
#pragma unroll
            for (int j = 0; j < QPT; j++)
                if (aq[j] >= 0) arank[j] = atomicAdd(&s_hist[akey[j]], 1);
            hb_barrier();
            if (wv == 0) {
                const int v = s_hist[lane];
                const int incl = wave_incl_scan(v);
                s_hist[lane] = incl - v;
                if (lane == 63) s_scan[0] = incl;
            }
            hb_barrier();
            const int M2 = s_scan[0];
#pragma unroll
            for (int j = 0; j < QPT; j++)
                if (aq[j] >= 0) s_order[s_hist[akey[j]] + arank[j]] = (uint16_t)aq[j];
            hb_wait_vm(0);
            hb_barrier();
            LVT_STAMP(7)

            int q = slot_query(0, M2);
            uint4 w0 = qd[2 * max(q, 0)], w1 = qd[2 * max(q, 0) + 1];
            float2 p = qxy[max(q, 0)];
            prefetch_next();
#pragma unroll
            for (int j = 0; j < QPT; j++)
                if (slowq[j] >= 0) {
                    const float2 ps = qxy[slowq[j]];
                    match_all(slowq[j], ps, qd[2 * slowq[j]], qd[2 * slowq[j] + 1], ranges(ps));
                }
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
                    LVT_WALK_BITS(mk.x, 0)
                    LVT_WALK_BITS(mk.y, 32)
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
#undef LVT_WALK_BITS
#undef LVT_FOLD
#undef LVT_RADIUS_BITS
#undef LVT_POS_OF
        LVT_STAMP(6)
        if (!PERSIST) break;
        // ---- the next problem of this workgroup: its train coordinates are in registers (requested at the head of stage B), its
        //      query coordinates are requested now, ahead of the DMA, which may start once every wave has left stage B
#pragma unroll
        for (int k = 0; k < TPT; k++) tpw[k] = ntpw[k];
        if (have_next) issue_query_coords(b_next, qpw);
        hb_barrier();
        if (have_next) issue_desc(b_next);
        b = b_next;
        dbg_on = a.dbg != nullptr && b == dbg_b;
        LVT_STAMP(0)
        first = false;
    } while (b < a.B);
#undef LVT_STAMP
}

// no train features at all: every query gets the "no neighbour" record
__global__ __launch_bounds__(256) void k_hamming_none(int4 *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_int4(-1, 0x7FFFFFFF, -1, 0x7FFFFFFF);
}

static inline size_t hamming_lds_bytes(int N, int M, int nbins) {
    return (size_t)hb_desc_region_bytes(N) + (size_t)N * 8 + (size_t)M * 12 + (size_t)(nbins + 1) * 4 + (size_t)((N + 1) & ~1) * 2 + (size_t)((M + 1) & ~1) * 2 + 16;
}

}  // namespace lvt
