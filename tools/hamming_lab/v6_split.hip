// v6_split.hip (round 5 experiment; lab first -- tools/hamming_lab/lab.py --split S --threads T [--mode 1])
// The batched masked 2-NN Hamming matcher (lvt_image_features_struct.cpp:68-148 + knnMatch(k = 2, mask)) with every problem cut
// SPATIALLY into S sub-problems, one smaller workgroup each, so that 3 - 7 sub-problems are resident per CU where k_hamming_batched holds two
// whole problems (76 KB of LDS each): the HBM round trips and barrier chains of more independent workgroups overlap each other's VALU / LDS phases.
//
// Sub-problem s of a problem = the queries whose (clamped) bin row -- hash-cell row in radius mode, image row in row mode -- lies in
// [lo_s, hi_s) = the s-th of S equal parts of the bin rows, and the train features of rows [lo_s - h, hi_s + h): h = the cell search radius
// (radius mode) or ROW_RADIUS (row mode).  A query's window rows are clipped to the grid and lie inside [row - h, row + h], so its candidate
// set inside the sub-problem is exactly its candidate set in the whole problem: the output is that of k_hamming_batched on every input.
// Price: the halo rows are binned twice (radius mode, KITTI: 1 of 8 cell rows per side -> 9/16 of the train set per half), and every sub-problem
// reads ALL coordinates (20 KB) to find its members; the descriptors -- 80 % of the bytes -- are fetched by members only, in a second, dependent
// round of loads that runs under the query set-up and the radius stage.  The S workgroups of a problem are placed on the same XCD next to each other
// in dispatch order (workgroup w runs on XCD w % 8), so the coordinates and the halo descriptors of the second one are L2 hits.
// Capacities: the LDS image holds ncap train features and mcap queries per sub-problem (expected share + 6 sigma of a uniform scatter); a
// sub-problem that exceeds one takes an exact brute-force path over the problem's global arrays (slow, never wrong).
#pragma once
#include "k_hamming.hip"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace lvt {

struct SplitArgs {
    HammingArgs h;
    int B;
    int ncap, mcap;   // LDS capacities of one sub-problem
    int nbins_max;    // bins of the largest band (+ halo)
};

__device__ __forceinline__ int4 top2_record(uint32_t k1, uint32_t k2) {
    int4 o;
    o.x = (k1 == 0xFFFFFFFFu) ? -1 : (int)(k1 & 0xFFFFu);
    o.y = (k1 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k1 >> 16);
    o.z = (k2 == 0xFFFFFFFFu) ? -1 : (int)(k2 & 0xFFFFu);
    o.w = (k2 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k2 >> 16);
    return o;
}
// a sub-problem that does not fit its LDS image: every member query against every train feature of the problem, the reference's own predicates
template <int MODE>
__device__ __forceinline__ void split_brute_force(const HammingArgs &a, int b, int q, float2 p) {
    const int N = a.N;
    const uint4 *td = reinterpret_cast<const uint4 *>(a.t_desc + (size_t)b * N * 4);
    const float2 *txy = a.t_xy + (size_t)b * N;
    const uint8_t *tf = a.t_flag + (size_t)b * N;
    const uint4 *qd = reinterpret_cast<const uint4 *>(a.q_desc + (size_t)b * a.M * 4);
    const uint4 w0 = qd[2 * q], w1 = qd[2 * q + 1];
    const uint64_t d0 = ((uint64_t)w0.y << 32) | w0.x, d1 = ((uint64_t)w0.w << 32) | w0.z;
    const uint64_t d2 = ((uint64_t)w1.y << 32) | w1.x, d3 = ((uint64_t)w1.w << 32) | w1.z;
    int y0, y1, x0 = 0, x1 = 0;
    float fy0 = 0.f, fy1 = 0.f;
    if (MODE == 1) {
        y0 = max((int)p.y - ROW_RADIUS, 0);
        y1 = min(min((int)p.y + ROW_RADIUS, a.img_rows), a.nby - 1);
        fy0 = (float)y0, fy1 = (float)min((int)p.y + ROW_RADIUS, a.img_rows);
    } else {
        const int hy = (int)floorf(div_cell(p.y)), hx = (int)floorf(div_cell(p.x));
        y0 = max(hy - a.csr, 0), y1 = min(hy + a.csr, a.nby - 1);
        x0 = max(hx - a.csr, 0), x1 = min(hx + a.csr, a.nbx - 1);
    }
    uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
    for (int j = 0; j < N; j++) {
        const float2 t = txy[j];
        if (tf[j]) continue;
        bool ok;
        if (MODE == 1) {
            const int br = min(max((int)floorf(t.y), 0), a.nby - 1);
            ok = (br >= y0) && (br <= y1) && (t.y >= fy0) && (t.y <= fy1);
        } else {
            const int cy = min(max((int)floorf(div_cell(t.y)), 0), a.nby - 1);
            const int cx = min(max((int)floorf(div_cell(t.x)), 0), a.nbx - 1);
            const float dx = t.x - p.x, dy = t.y - p.y;
            ok = (cy >= y0) && (cy <= y1) && (cx >= x0) && (cx <= x1) && ((dx * dx + dy * dy) < a.r2);
        }
        if (!ok) continue;
        const uint4 a0 = td[2 * j], a1 = td[2 * j + 1];
        const uint32_t key = (hamming256(d0, d1, d2, d3, a0, a1) << 16) | (uint32_t)j;
        k2 = min(k2, max(k1, key));
        k1 = min(k1, key);
    }
    a.out[(size_t)b * a.M + q] = top2_record(k1, k2);
}

// MODE 0: radius mode with csr == 1 (three packed candidate ranges per query, two stages); MODE 1: row mode (one range, one stage)
// T threads; S bands; TPT / QPT = train features / queries LOADED per thread (ceil(N / T), ceil(M / T)); DPT = descriptors fetched per thread
// (ceil(ncap / T)); RQ = rounds of member queries (ceil(mcap / T)); WPE = waves per SIMD the register allocation is held to
template <int MODE, int T, int S, int TPT, int QPT, int DPT, int RQ, int WPE>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_hamming_split(SplitArgs sa) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const HammingArgs &a = sa.h;
    const int N = a.N, M = a.M, ncap = sa.ncap, mcap = sa.mcap;
    constexpr int NW = T / 64;
    // block -> (problem, band): the S bands of one problem on one XCD, adjacent in dispatch order
    const int w = blockIdx.x, g = w >> 3;
    const int s = g % S, b = (g / S) * 8 + (w & 7);
    if (b >= sa.B) return;
    const int nrow = a.nby, nbx = a.nbx;
    const int lo = (s * nrow) / S, hi = ((s + 1) * nrow) / S;
    const int halo = (MODE == 1) ? ROW_RADIUS : a.csr;
    const int base = max(lo - halo, 0), top = min(hi + halo, nrow);
    const int nbins = (top - base) * nbx;

    // carve: desc lo [ncap] | desc hi [ncap] | xy [ncap] (radius mode) | per-query words [mcap] x 3 (radius) / 1 (row) | start [nbins_max + 1] |
    //        idx [ncap] u16 | query id [mcap] u16 | order [mcap] u16
    uint4 *s_dlo = reinterpret_cast<uint4 *>(smem);
    uint4 *s_dhi = s_dlo + ncap;
    float2 *s_xy = reinterpret_cast<float2 *>(s_dhi + ncap);
    uint32_t *s_q = reinterpret_cast<uint32_t *>(s_xy + (MODE == 0 ? ncap : 0));
    constexpr int QW = (MODE == 0) ? 3 : 1;
    int *s_start = reinterpret_cast<int *>(s_q + QW * (size_t)mcap);
    uint16_t *s_idx = reinterpret_cast<uint16_t *>(s_start + sa.nbins_max + 1);
    uint16_t *s_qid = s_idx + ncap;
    uint16_t *s_order = s_qid + mcap;
    __shared__ int s_scan[32];
    __shared__ int s_hist[HB_HIST], s_hist2[HB_HIST];
    __shared__ int s_recheck, s_nq;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint4 *td = reinterpret_cast<const uint4 *>(a.t_desc + (size_t)b * N * 4);
    const float2 *txy = a.t_xy + (size_t)b * N;
    const uint8_t *tf = a.t_flag + (size_t)b * N;
    const uint4 *qd = reinterpret_cast<const uint4 *>(a.q_desc + (size_t)b * M * 4);
    const float2 *qxy = a.q_xy + (size_t)b * M;
    int4 *out = a.out + (size_t)b * M;
    long long *dbg = (a.dbg && blockIdx.x == gridDim.x / 2 && tid == 0) ? a.dbg : nullptr;
    if (dbg) dbg[0] = clock64();
    __builtin_amdgcn_s_setprio(3);

    // ---- 1. all coordinates and flags of the problem (20 KB: the second band of the problem finds them in L2)
    float2 tp[TPT], qp[QPT];
    uint8_t tfl[TPT];
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        const int jc = min(tid + k * T, N - 1);
        tp[k] = txy[jc];
        tfl[k] = tf[jc];
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) qp[k] = qxy[min(tid + k * T, M - 1)];
    for (int i = tid; i <= nbins; i += T) s_start[i] = 0;
    if (tid < HB_HIST) s_hist[tid] = 0, s_hist2[tid] = 0;
    if (tid == 0) s_recheck = 0;
    __syncthreads();
    if (dbg) dbg[1] = clock64();

    // ---- 2. members of this band (+ halo), counting-sorted into its bins
    auto row_of = [&](float y) -> int { return (MODE == 1) ? (int)floorf(y) : (int)floorf(div_cell(y)); };
    int tbin[TPT], trank[TPT];
    bool tv[TPT];
#pragma unroll
    for (int k = 0; k < TPT; k++) {
        const int row = min(max(row_of(tp[k].y), 0), nrow - 1);
        const int cx = (MODE == 1) ? 0 : min(max((int)floorf(div_cell(tp[k].x)), 0), nbx - 1);
        tv[k] = (tid + k * T < N) && (tfl[k] == 0) && (row >= base) && (row < top);
        tbin[k] = (row - base) * nbx + cx;
        trank[k] = 0;
        if (tv[k]) trank[k] = atomicAdd(&s_start[tbin[k]], 1);
    }
    __syncthreads();
    {
        const int chunk = (nbins + 1 + T - 1) / T;
        const int i0 = min(tid * chunk, nbins + 1), i1 = min(i0 + chunk, nbins + 1);
        int sum = 0;
        for (int i = i0; i < i1; i++) sum += s_start[i];
        int total_;
        int run = block_excl_scan(sum, s_scan, &total_);
        for (int i = i0; i < i1; i++) {
            const int v = s_start[i];
            s_start[i] = run;
            run += v;
        }
    }
    __syncthreads();
    if (dbg) dbg[2] = clock64();
    const int total = s_start[nbins];
    const bool t_ovf = total > ncap;  // (block-uniform)
    if (!t_ovf) {
#pragma unroll
        for (int k = 0; k < TPT; k++)
            if (tv[k]) {
                const int pos = s_start[tbin[k]] + trank[k];
                if (MODE == 0) s_xy[pos] = tp[k];
                // row mode: see k_hamming.hip -- an in-range integer row needs no comparison in the walk; anything else is marked (bit 15)
                const bool recheck = (MODE == 1) && !((float)(tbin[k] + base) == tp[k].y);
                if (recheck) s_recheck = 1;
                s_idx[pos] = (uint16_t)((tid + k * T) | (recheck ? 0x8000 : 0));
            }
    }

    // ---- 3. member queries: candidate ranges (local bin rows), sorted by candidate count, heaviest first
    int qkey[QPT], qrank[QPT];
    uint32_t qW0[QPT], qW1[QPT];
    bool qv[QPT];
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        const int q = tid + k * T;
        const float2 p = qp[k];
        qkey[k] = 0, qrank[k] = 0, qW0[k] = 0, qW1[k] = 0;
        if (MODE == 1) {
            const int r = (int)p.y;
            const int rc = min(max(r, 0), nrow - 1);
            qv[k] = (q < M) && (rc >= lo) && (rc < hi);
            const int y0 = max(r - ROW_RADIUS, 0), y1 = min(min(r + ROW_RADIUS, a.img_rows), nrow - 1);
            const bool ok = qv[k] && (y0 <= y1) && !t_ovf;
            const int s0 = s_start[ok ? y0 - base : 0];
            const int l0 = s_start[ok ? y1 + 1 - base : 0] - s0;
            qW0[k] = (uint32_t)s0 | ((uint32_t)l0 << 16);
            qkey[k] = HB_HIST - 1 - min(l0, HB_HIST - 1);
        } else {
            const int hy = (int)floorf(div_cell(p.y)), hx = (int)floorf(div_cell(p.x));
            const int rc = min(max(hy, 0), nrow - 1);
            qv[k] = (q < M) && (rc >= lo) && (rc < hi);
            const int y0 = max(hy - 1, 0), y1 = min(hy + 1, nrow - 1);
            const int x0 = max(hx - 1, 0), x1 = min(hx + 1, nbx - 1);
            int rs[3], rl[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const bool ok = qv[k] && (y0 + c <= y1) && (x0 <= x1) && !t_ovf;
                const int row = ok ? (y0 + c - base) * nbx : 0;
                rs[c] = s_start[row + (ok ? x0 : 0)];
                rl[c] = s_start[row + (ok ? x1 + 1 : 0)] - rs[c];
            }
            qkey[k] = HB_HIST - 1 - min(rl[0] + rl[1] + rl[2], HB_HIST - 1);
            const bool fits = (rl[0] < 64) && (rl[1] < 64) && (rl[2] < 64);  // (positions < ncap <= 2048: 11 bits)
            qW0[k] = fits ? ((uint32_t)rs[0] | ((uint32_t)rs[1] << 11) | ((uint32_t)rl[0] << 22)) : 0xFFFFFFFFu;
            qW1[k] = (uint32_t)rs[2] | ((uint32_t)rl[1] << 11) | ((uint32_t)rl[2] << 17);
        }
        if (qv[k]) qrank[k] = atomicAdd(&s_hist[qkey[k]], 1);
    }
    __syncthreads();  // scatter of coordinates / indices visible; histogram complete
    if (dbg) dbg[3] = clock64();
    // the members' descriptors: the second, dependent round of loads.  Vector-memory results return in order, so a wave that waits for a
    // load also waits for everything it issued before it: row mode issues them here (nothing else is loaded before the walk needs them), radius
    // mode BEHIND the query coordinates of its radius stage (below), so that stage runs while they are in flight
    uint4 rlo[DPT], rhi[DPT];
    auto fetch_descriptors = [&]() {
#pragma unroll
        for (int u = 0; u < DPT; u++) {
            const int pos = tid + u * T;
            rlo[u] = rhi[u] = make_uint4(0, 0, 0, 0);
            if (pos < total && !t_ovf) {
                const int j = s_idx[pos] & 0x7FFF;
                rlo[u] = td[2 * j];
                rhi[u] = td[2 * j + 1];
            }
        }
    };
    if (MODE == 1) fetch_descriptors();
    if (wv == 0) {
        const int v = s_hist[lane];
        const int incl = wave_incl_scan(v);
        s_hist[lane] = incl - v;
        if (lane == 63) s_nq = incl;
    }
    __syncthreads();
    const int Mq = s_nq;
    if (t_ovf || Mq > mcap) {  // (block-uniform) does not fit the LDS image: exact, slow
#pragma unroll
        for (int k = 0; k < QPT; k++)
            if (qv[k]) split_brute_force<MODE>(a, b, tid + k * T, qp[k]);
        return;
    }
#pragma unroll
    for (int k = 0; k < QPT; k++)
        if (qv[k]) {
            const int slot = s_hist[qkey[k]] + qrank[k];
            s_qid[slot] = (uint16_t)(tid + k * T);
            s_q[QW * slot] = qW0[k];
            if (MODE == 0) s_q[QW * slot + 1] = qW1[k];
        }
    __syncthreads();
    if (dbg) dbg[4] = clock64();
    __builtin_amdgcn_s_setprio(0);

    // rounds of T member queries in sorted order; odd rounds reverse the wave order
    auto slot_of = [&](int j, int count) -> int {
        const int slot = j * T + ((j & 1) ? (NW - 1 - wv) : wv) * 64 + lane;
        return (j < RQ && slot < count) ? slot : -1;
    };
    auto store_descriptors = [&]() {
#pragma unroll
        for (int u = 0; u < DPT; u++) {
            const int pos = tid + u * T;
            if (pos < total) s_dlo[pos] = rlo[u], s_dhi[pos] = rhi[u];
        }
    };

    if (MODE == 1) {
        store_descriptors();
        __syncthreads();
        if (dbg) dbg[5] = clock64(), dbg[7] = dbg[5];
        const bool recheck = s_recheck != 0;  // (block-uniform)
        int slot = slot_of(0, Mq);
        int q = slot >= 0 ? (int)s_qid[slot] : 0;
        uint4 w0 = qd[2 * q], w1 = qd[2 * q + 1];
        for (int j = 0; j < RQ; j++) {
            const int slotn = slot_of(j + 1, Mq);
            const int qn = slotn >= 0 ? (int)s_qid[slotn] : 0;
            const uint4 nw0 = qd[2 * qn], nw1 = qd[2 * qn + 1];
            if (slot >= 0) {
                const uint64_t d0 = ((uint64_t)w0.y << 32) | w0.x, d1 = ((uint64_t)w0.w << 32) | w0.z;
                const uint64_t d2 = ((uint64_t)w1.y << 32) | w1.x, d3 = ((uint64_t)w1.w << 32) | w1.z;
                const uint32_t W = s_q[slot];
                const int s0 = (int)(W & 0xFFFFu), len = (int)(W >> 16);
                uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
                if (len > 0) {
                    if (!recheck) {  // two candidates per trip, by value (k_hamming.hip: row_walk_lean)
                        const uint4 *plo = s_dlo + s0, *phi = s_dhi + s0;
                        const uint16_t *pid = s_idx + s0;
                        int v = 0;
                        for (; v + 2 <= len; v += 2) {
                            const uint4 a0 = plo[v], a1 = phi[v], b0 = plo[v + 1], b1 = phi[v + 1];
                            const uint32_t ia = pid[v], ib = pid[v + 1];
                            const uint32_t ka = (hamming256(d0, d1, d2, d3, a0, a1) << 16) | ia, kb = (hamming256(d0, d1, d2, d3, b0, b1) << 16) | ib;
                            const uint32_t lo = min(ka, kb), hi = max(ka, kb);
                            const uint32_t t = max(k1, lo);
                            k1 = min(k1, lo);
                            k2 = min(t, min(k2, hi));
                        }
                        if (v < len) {
                            const uint4 a0 = plo[v], a1 = phi[v];
                            const uint32_t ka = (hamming256(d0, d1, d2, d3, a0, a1) << 16) | pid[v];
                            k2 = min(k2, max(k1, ka));
                            k1 = min(k1, ka);
                        }
                    } else {
                        const float2 p = qxy[q];
                        const float fy0 = (float)max((int)p.y - ROW_RADIUS, 0), fy1 = (float)min((int)p.y + ROW_RADIUS, a.img_rows);
                        uint4 a0 = s_dlo[s0], a1 = s_dhi[s0];
                        uint32_t id = s_idx[s0];
                        for (int v = 0; v < len; v++) {
                            const uint4 b0 = s_dlo[s0 + v + 1], b1 = s_dhi[s0 + v + 1];
                            const uint32_t idn = s_idx[s0 + v + 1];
                            bool ok = true;
                            if (id & 0x8000u) {  // struct.cpp:133 on the feature's own coordinates
                                const float ty = txy[id & 0x7FFFu].y;
                                ok = (ty >= fy0) && (ty <= fy1);
                            }
                            uint32_t key = (hamming256(d0, d1, d2, d3, a0, a1) << 16) | (id & 0x7FFFu);
                            key = ok ? key : 0xFFFFFFFFu;
                            k2 = min(k2, max(k1, key));
                            k1 = min(k1, key);
                            a0 = b0, a1 = b1, id = idn;
                        }
                    }
                }
                out[q] = top2_record(k1, k2);
            }
            slot = slotn, q = qn, w0 = nw0, w1 = nw1;
        }
        if (dbg) dbg[6] = clock64();
        return;
    }

    // ---- 4a (radius mode). radius test over every window candidate -> one bit per candidate of the flattened index space
    constexpr int MASK_BITS = 41;  // 32 in word 2, 9 in the free top of word 1
#define LVT_POS_OF(dst, v)                  \
    {                                       \
        int o_ = o0;                        \
        o_ = ((v) >= c1) ? o1 : o_;         \
        o_ = ((v) >= c2) ? o2 : o_;         \
        dst = (v) + o_;                     \
    }
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
    int aslot[RQ], akey[RQ], arank[RQ];
    float2 ap[RQ];
#pragma unroll
    for (int j = 0; j < RQ; j++) {
        aslot[j] = slot_of(j, Mq);
        ap[j] = qxy[aslot[j] >= 0 ? (int)s_qid[aslot[j]] : 0];
    }
    fetch_descriptors();
#pragma unroll
    for (int j = 0; j < RQ; j++) {
        akey[j] = 0, arank[j] = 0;
        const int slot = aslot[j];
        const float2 p = ap[j];
        if (slot >= 0) {
            const uint32_t W0 = s_q[3 * slot], W1 = s_q[3 * slot + 1];
            const int l0 = (int)(W0 >> 22), l1 = (int)((W1 >> 11) & 63u), l2 = (int)((W1 >> 17) & 63u);
            const int c1 = l0, c2 = c1 + l1, tot = c2 + l2;
            const int o0 = (int)(W0 & 2047u), o1 = (int)((W0 >> 11) & 2047u) - c1, o2 = (int)(W1 & 2047u) - c2;
            if (W0 == 0xFFFFFFFFu || tot > MASK_BITS) {  // ranges or mask do not fit their slot: matched in one stage once the descriptors are in LDS (rare)
                s_q[3 * slot] = 0xFFFFFFFFu;
                akey[j] = 0;
            } else {
                uint32_t mlo = 0, mhi = 0;
                const int t0 = min(tot, 32);
                LVT_RADIUS_BITS(mlo, 0, t0)
                LVT_RADIUS_BITS(mhi, 32, tot)
                s_q[3 * slot + 2] = mlo;
                s_q[3 * slot + 1] = (W1 & 0x7FFFFFu) | (mhi << 23);
                akey[j] = HB_HIST - 1 - min(__popc(mlo) + __popc(mhi), HB_HIST - 1);
            }
            arank[j] = atomicAdd(&s_hist2[akey[j]], 1);
        }
    }
    store_descriptors();
    __syncthreads();
    if (wv == 0) {
        const int v = s_hist2[lane];
        s_hist2[lane] = wave_incl_scan(v) - v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RQ; j++)
        if (aslot[j] >= 0) s_order[s_hist2[akey[j]] + arank[j]] = (uint16_t)aslot[j];
    __syncthreads();
    if (dbg) dbg[5] = clock64(), dbg[7] = dbg[5];

    // ---- 4b. the queries again, sorted by the number of candidates inside the circle
    {
        int slot2 = slot_of(0, Mq);
        int slot = slot2 >= 0 ? (int)s_order[slot2] : 0;
        int q = slot2 >= 0 ? (int)s_qid[slot] : 0;
        uint4 w0 = qd[2 * q], w1 = qd[2 * q + 1];
        for (int j = 0; j < RQ; j++) {
            const int slot2n = slot_of(j + 1, Mq);
            const int slotn = slot2n >= 0 ? (int)s_order[slot2n] : 0;
            const int qn = slot2n >= 0 ? (int)s_qid[slotn] : 0;
            const uint4 nw0 = qd[2 * qn], nw1 = qd[2 * qn + 1];
            if (slot2 >= 0) {
                const uint64_t d0 = ((uint64_t)w0.y << 32) | w0.x, d1 = ((uint64_t)w0.w << 32) | w0.z;
                const uint64_t d2 = ((uint64_t)w1.y << 32) | w1.x, d3 = ((uint64_t)w1.w << 32) | w1.z;
                uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
                const uint32_t W0 = s_q[3 * slot], W1 = s_q[3 * slot + 1], W2 = s_q[3 * slot + 2];
                if (W0 == 0xFFFFFFFFu) {  // one stage over the whole window (rare): the ranges again from the bin starts
                    const float2 p = qxy[q];
                    const int hy = (int)floorf(div_cell(p.y)), hx = (int)floorf(div_cell(p.x));
                    const int y0 = max(hy - 1, 0), y1 = min(hy + 1, nrow - 1);
                    const int x0 = max(hx - 1, 0), x1 = min(hx + 1, nbx - 1);
                    if (x0 <= x1)
                        for (int by = y0; by <= y1; by++) {
                            const int r0 = s_start[(by - base) * nbx + x0], r1 = s_start[(by - base) * nbx + x1 + 1];
                            for (int it = r0; it < r1; it++) {
                                const float2 r = s_xy[it];
                                const float dx = r.x - p.x, dy = r.y - p.y;
                                if ((dx * dx + dy * dy) < a.r2) {
                                    const uint32_t key = (hamming256(d0, d1, d2, d3, s_dlo[it], s_dhi[it]) << 16) | (uint32_t)s_idx[it];
                                    k2 = min(k2, max(k1, key));
                                    k1 = min(k1, key);
                                }
                            }
                        }
                } else {
                    const int c1 = (int)(W0 >> 22), c2 = c1 + (int)((W1 >> 11) & 63u);
                    const int o0 = (int)(W0 & 2047u), o1 = (int)((W0 >> 11) & 2047u) - c1, o2 = (int)(W1 & 2047u) - c2;
                    auto walk_bits = [&](uint32_t m, int vbase) {  // set bits of m, software-pipelined by one candidate
                        if (m == 0) return;
                        int it;
                        {
                            const int v = vbase + __ffs((int)m) - 1;
                            LVT_POS_OF(it, v)
                        }
                        m &= m - 1;
                        uint4 a0 = s_dlo[it], a1 = s_dhi[it];
                        uint32_t id = s_idx[it];
                        for (;;) {
                            const bool more = m != 0;
                            int itn;
                            {
                                const int v = vbase + ((__ffs((int)m) - 1) & 31);
                                LVT_POS_OF(itn, v)
                            }
                            m &= m - 1;
                            const uint4 b0 = s_dlo[itn], b1 = s_dhi[itn];
                            const uint32_t idn = s_idx[itn];
                            const uint32_t key = (hamming256(d0, d1, d2, d3, a0, a1) << 16) | id;
                            k2 = min(k2, max(k1, key));
                            k1 = min(k1, key);
                            if (!more) break;
                            a0 = b0, a1 = b1, id = idn;
                        }
                    };
                    walk_bits(W2, 0);
                    walk_bits(W1 >> 23, 32);
                }
                out[q] = top2_record(k1, k2);
            }
            slot2 = slot2n, slot = slotn, q = qn, w0 = nw0, w1 = nw1;
        }
    }
#undef LVT_RADIUS_BITS
#undef LVT_POS_OF
    if (dbg) dbg[6] = clock64();
}

// ---- host side ----------------------------------------------------------------------------------------------------------
struct SplitPlan {
    SplitArgs sa;
    size_t lds;
    int grid, threads;
};
// capacity for a sub-problem: its expected share of a uniform scatter + nsig sigma, rounded DOWN to 8 (it is a margin either way)
static inline int split_cap(int n, double frac, double nsig) {
    const double mean = n * frac, sig = std::sqrt(std::max(n * frac * (1.0 - frac), 1.0));
    const int c = ((int)(mean + nsig * sig) / 8) * 8;
    return std::max(8, std::min(c, ((n + 7) / 8) * 8));
}
template <int MODE>
static inline SplitPlan hamming_split_plan(const HammingArgs &a, int B, int S, int T) {
    SplitPlan p;
    p.sa.h = a, p.sa.B = B;
    const int nrow = a.nby, halo = (MODE == 1) ? ROW_RADIUS : a.csr;
    int rows_t = 0, rows_q = 0;
    for (int s = 0; s < S; s++) {
        const int lo = (s * nrow) / S, hi = ((s + 1) * nrow) / S;
        rows_q = std::max(rows_q, hi - lo);
        rows_t = std::max(rows_t, std::min(hi + halo, nrow) - std::max(lo - halo, 0));
    }
    // (row mode bins rows 0 .. img_rows; features lie in rows 0 .. img_rows - 1: nrow - 1 populated rows)
    const double denom = (MODE == 1) ? (double)(nrow - 1) : (double)a.img_rows / HASH_CELL;
    p.sa.ncap = split_cap(a.N, std::min(1.0, rows_t / denom), 5.0);
    p.sa.mcap = split_cap(a.M, std::min(1.0, rows_q / denom), 6.0);
    if (const char *e = std::getenv("LAB_NCAP")) p.sa.ncap = std::atoi(e);
    if (const char *e = std::getenv("LAB_MCAP")) p.sa.mcap = std::atoi(e);
    p.sa.nbins_max = rows_t * a.nbx;
    const size_t qw = (MODE == 0) ? 12 : 4;
    p.lds = (size_t)p.sa.ncap * 32 + (MODE == 0 ? (size_t)p.sa.ncap * 8 : 0) + (size_t)p.sa.mcap * qw + (size_t)(p.sa.nbins_max + 1) * 4 +
            (size_t)p.sa.ncap * 2 + (size_t)p.sa.mcap * 4 + 16;
    p.grid = ((B + 7) / 8) * 8 * S;
    p.threads = T;
    return p;
}

typedef void (*SplitKernel)(SplitArgs);
// the instances the lab launches: (MODE, T, S) with the per-thread counts of the plan
template <int MODE, int T, int S, int WPE>
static inline SplitKernel hamming_split_pick(const SplitPlan &p) {
    const int N = p.sa.h.N, M = p.sa.h.M;
    const int tpt = (N + T - 1) / T, qpt = (M + T - 1) / T, dpt = (p.sa.ncap + T - 1) / T, rq = (p.sa.mcap + T - 1) / T;
#define LVT_TRY(TPT_, QPT_, DPT_, RQ_) \
    if (tpt == TPT_ && qpt == QPT_ && dpt == DPT_ && rq == RQ_) return k_hamming_split<MODE, T, S, TPT_, QPT_, DPT_, RQ_, WPE>;
    if constexpr (T == 512) {
        LVT_TRY(3, 2, 2, 2)
        LVT_TRY(3, 2, 2, 1)
        LVT_TRY(3, 2, 1, 1)
    }
    if constexpr (T == 256) {
        LVT_TRY(6, 4, 3, 2)
        LVT_TRY(6, 4, 2, 2)
        LVT_TRY(6, 4, 4, 3)
        LVT_TRY(6, 4, 4, 2)
    }
    if constexpr (T == 384) {
        LVT_TRY(4, 3, 2, 1)
        LVT_TRY(4, 3, 3, 2)
        LVT_TRY(4, 3, 2, 2)
    }
    if constexpr (T == 640) {
        LVT_TRY(3, 2, 2, 1)
    }
#undef LVT_TRY
    return nullptr;
}

}  // namespace lvt
