// k_lists.hip -- k_hamming_batched_lists: the pipeline's candidate LISTS from the binned matcher (round 2; rebuilt in round 6).
//
// The greedy accept / mark scans of find_matches and row_match (k_early_mid / k_track_mid / k_triangulate) walk, per query, the
// candidates of lvt_image_features_struct.cpp:68-148 sorted by (Hamming distance, index).  k_candidates / k_early_map build those
// lists with one WAVEFRONT per query scanning ALL train features (64 lanes x N / 64 trips of predicate tests per query).  This kernel
// stages the train set once per workgroup in LDS, counting-sorted into the reference's own 25-px hash cells (tracking) or image rows
// (row matching), exactly as k_hamming_batched does (k_hamming.hip) -- a query's candidates are then <= 5 CONTIGUOUS LDS ranges.
//
// Round 6: A LANE PER CANDIDATE, FOUR QUERIES PER WAVEFRONT (one per DPP row of 16 lanes).  (Rounds 2 - 5 gave every query one lane that walked its ~20 candidates
// serially -- twice, to size an arena segment first -- and ranked them by counting, (n / 4)^2 trips per lane: 89 us for the row lists of a
// 16-sequence batch, the kernel ending with its longest list.)  Now
//   1. every global load of the train set is in flight before the counting sort starts (one round trip, not two);
//   2. one thread per query of a chunk of <= 256 projects the point (map mode: is_point_visible, lvt_local_map.cpp:62-82,152-156, leaving
//      proj / vis / match / counter exactly as k_early_map does) and packs the window's ranges into eight LDS words; the chunk's query
//      descriptors are copied to LDS with coalesced loads;
//   3. a wavefront takes four queries, a row of 16 lanes each: lane s of a row evaluates candidates s, s + 16, ... of its query's flattened ranges --
//      predicate, 256-bit distance, key (distance << 16 | index) -- the keys that pass are compacted into the row's own LDS segment (ballot
//      prefix), and every lane ranks its keys among them with reads of four keys at a time: the rank is the key's place in the list.
//      Neighbouring lanes read neighbouring descriptors: no bank conflicts, no serial chain longer than n / 4 trips.  (A wavefront per
//      query was built first: ~225 wave instructions per query against ~60 here -- its ranking by v_readlane alone cost 160.)
//
// Output contract = k_candidates': cand[q][0 .. min(n, KC)) ascending, ncand[q] = n (n > KC: the resolvers' exact slow path).
// The bins ARE the reference's hash cells (Feat::hcx / hcy), so "candidate in the window's cells" is decided by the range bounds and the
// wavefront only evaluates the radius; row mode evaluates the band test itself (struct.cpp:132-134).
// Capacity: 1536 train features and 1100 bins per LDS image (94 KB of LDS per 512-thread workgroup); beyond that (or for a cell search radius above 2) the kernel does
// NOTHING and raises Seq::lists_fb, and the wave-per-query kernel launched behind it does the work as before.
#include "lvt_dev.h"
#include "lvt_math.h"

namespace lvt {

constexpr int LS_THREADS = 512;
constexpr int LS_WAVES = LS_THREADS / 64;
constexpr int LS_TPT = 3;       // train features per thread
constexpr int LS_NMAX = LS_THREADS * LS_TPT;  // 1536 train features of one LDS image
constexpr int LS_BINS = 1100;   // hash cells / image rows + 1
constexpr int LS_QCH = 256;     // queries of one chunk: descriptors and window words in LDS
constexpr int LS_SEG = 256;     // key words of one wavefront: four segments of 64 (a query per DPP row), or one of KC + 64 (a window of more than 64 features)
constexpr int LS_STARTS = (LS_BINS + 4) & ~3;
constexpr int LS_LDS_BYTES = LS_NMAX * 42 + LS_STARTS * 4 + LS_QCH * 64 + LS_WAVES * LS_SEG * 4 + 256;
static_assert(KC + 64 <= LS_SEG, "the long-window path shares the wavefront's segment words");
typedef unsigned int ls_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ls_bcnt(uint32_t x, uint32_t acc) {  // acc + popcount(x): one instruction
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

template <int MODE, bool BV>
__global__ __launch_bounds__(LS_THREADS) void k_hamming_batched_lists(SeqArg<BV> sa, int par, seq_t seq) {
    const Seq &S = sa.get();
    Ctl &ctl = *S.ctl;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int *fb = S.lists_fb + (MODE == MODE_ROW ? 1 : 0);
    // ---- what there is to do (block-uniform), exactly as the kernel this one stands in for decides it
    int q_end = 0;
    if (MODE == MODE_MAP) {
        const int n_early = (ctl.gate_ok == seq) ? ctl.early_done : 0;  // k_early_map
        if (n_early <= 0) return;
        q_end = min(*S.map_n, n_early);
    } else {
        if (S.prm.sensor != 1) return;
        if (seq && __hip_atomic_load(&S.fb[par].fc->feat_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) return;  // k_candidates<ROW>
        q_end = *S.fb[par].feat[0].n;
    }
    const Feat &T = (MODE == MODE_ROW) ? S.fb[par].feat[1] : S.fb[par].feat[0];
    const int N = *T.n;
    const int ccx = S.prm.hash_ccx, ccy = S.prm.hash_ccy, csr = S.prm.cell_search_radius;
    const int nbins = (MODE == MODE_ROW) ? S.prm.H + 1 : ccx * ccy;
    if (N > LS_NMAX || nbins > LS_BINS || (MODE != MODE_ROW && csr > 2)) {  // not for this kernel: the wave-per-query kernel behind it runs
        if (tid == 0) *fb = 1;
        return;
    }
    if (tid == 0) *fb = 0;

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    ls_u32x4 *s_dlo = reinterpret_cast<ls_u32x4 *>(smem);
    ls_u32x4 *s_dhi = s_dlo + LS_NMAX;
    float2 *s_xy = reinterpret_cast<float2 *>(s_dhi + LS_NMAX);
    uint16_t *s_idx = reinterpret_cast<uint16_t *>(s_xy + LS_NMAX);
    int *s_start = reinterpret_cast<int *>(s_idx + LS_NMAX);
    ls_u32x4 *s_qd = reinterpret_cast<ls_u32x4 *>(s_start + LS_STARTS);  // [LS_QCH][2]: the chunk's query descriptors
    ls_u32x4 *s_qw = s_qd + 2 * LS_QCH;                                  // [LS_QCH][2]: live | x | y | - , five ranges (start | length << 16) (three used words spare)
    uint32_t *s_seg = reinterpret_cast<uint32_t *>(s_qw + 2 * LS_QCH) + wv * LS_SEG;
    __shared__ int s_scan[32];
    __shared__ double w2c[12];

    long long *stamp = (MODE == MODE_ROW && blockIdx.x == 0 && tid == 0) ? ctl.dbg + 26 : nullptr;  // (tools/lists_phases.py)
    if (stamp) stamp[0] = clock64();
    // ---- 1. everything the counting sort needs from HBM / L2, issued back to back (LS_TPT train features per thread at most; indices clamped)
    float tx[LS_TPT], ty[LS_TPT];
    int tbin[LS_TPT], trank[LS_TPT];
    ls_u32x4 tlo[LS_TPT], thi[LS_TPT];
#pragma unroll
    for (int k = 0; k < LS_TPT; k++) {
        const int jc = max(min(tid + k * LS_THREADS, N - 1), 0);
        tx[k] = ty[k] = 0.f, tbin[k] = trank[k] = 0;
        tlo[k] = thi[k] = ls_u32x4{0, 0, 0, 0};
        if (N > 0) {
            tx[k] = T.x[jc], ty[k] = T.y[jc];
            if (MODE != MODE_ROW) tbin[k] = min(max((int)T.hcy[jc], 0), ccy - 1) * ccx + min(max((int)T.hcx[jc], 0), ccx - 1);
            const ls_u32x4 *d = reinterpret_cast<const ls_u32x4 *>(T.desc + (size_t)jc * 4);
            tlo[k] = d[0], thi[k] = d[1];
        }
    }
    if (MODE == MODE_MAP && tid == 0) {  // the prediction, recomputed from the persistent state (k_early_map does the same)
        Pose predicted;
        double mmn[14];
        motion_predict(ctl, ctl.last_pose, predicted, mmn);
        world_to_camera(predicted, w2c);
    }
    for (int i = tid; i <= nbins; i += LS_THREADS) s_start[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LS_TPT; k++) {
        if (MODE == MODE_ROW) tbin[k] = min(max((int)floorf(ty[k]), 0), nbins - 1);
        if (tid + k * LS_THREADS < N) trank[k] = atomicAdd(&s_start[tbin[k]], 1);
    }
    __syncthreads();
    {
        const int chunk = (nbins + 1 + LS_THREADS - 1) / LS_THREADS;
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
#pragma unroll
    for (int k = 0; k < LS_TPT; k++) {
        const int j = tid + k * LS_THREADS;
        if (j < N) {
            const int pos = s_start[tbin[k]] + trank[k];
            s_xy[pos] = make_float2(tx[k], ty[k]);
            s_idx[pos] = (uint16_t)j;
            s_dlo[pos] = tlo[k];
            s_dhi[pos] = thi[k];
        }
    }
    if (stamp) stamp[1] = clock64();

    const uint64_t *qdesc = (MODE == MODE_MAP) ? S.map[*S.map_cur].desc : S.fb[par].feat[0].desc;
    uint32_t *cand = (MODE == MODE_MAP) ? S.cand : S.rcand + (size_t)par * NF_MAX * KC;
    int *ncand = (MODE == MODE_MAP) ? S.ncand : S.rncand + par * NF_MAX;
    const int radius = S.prm.tracking_radius;
    const float r2 = (float)(radius * radius);

    // gridDim.x workgroups per sequence share the queries in EQUAL contiguous parts (whole wavefronts; each stages the train set itself)
    const int part = (((q_end + (int)gridDim.x - 1) / (int)gridDim.x) + 63) & ~63;
    const int q_lo = (int)blockIdx.x * part, q_hi = min(q_end, q_lo + part);
    for (int q0 = q_lo; q0 < q_hi; q0 += LS_QCH) {
        const int nq = min(LS_QCH, q_hi - q0);
        // ---- 2. the chunk's queries: descriptors by coalesced loads, one thread per query for the projection and the window's ranges
        __syncthreads();  // the bin starts are final (first chunk) / every wavefront is done with the previous chunk's words
        if (tid < 2 * nq) s_qd[tid] = reinterpret_cast<const ls_u32x4 *>(qdesc + (size_t)q0 * 4)[tid];
        if (tid < nq) {
            const int q = q0 + tid;
            bool live = true;
            Query Q;
            Q.x = Q.y = Q.r2 = 0.f;
            Q.sy = Q.sx = 0, Q.ey = Q.ex = 0;
            if (MODE == MODE_MAP) {  // candidates_body<MODE_MAP, PROJECT>, one lane per point
                const MapSoA &P = S.map[*S.map_cur];
                const double X[3] = {P.pos[3 * q], P.pos[3 * q + 1], P.pos[3 * q + 2]};
                double u, v;
                if (is_point_visible(X, w2c, S.prm, u, v)) {
                    S.proj[2 * q] = (float)u;
                    S.proj[2 * q + 1] = (float)v;
                    S.vis[q] = 1;
                    S.match[q] = -1;
                    make_query_track(S.prm, (float)u, (float)v, radius, Q);
                } else {
                    S.vis[q] = 0;
                    P.counter[q] += 1;  // lvt_local_map.cpp:154
                    S.match[q] = -2;
                    live = false;
                }
            } else {
                make_query_row(S.prm, S.fb[par].feat[0].x[q], S.fb[par].feat[0].y[q], Q);
            }
            // the window as contiguous LDS ranges: one per hash row of the window (tracking), one in all (row band)
            uint32_t rg[5] = {0, 0, 0, 0, 0};
            int total = 0;
            if (live) {
                if (MODE == MODE_ROW) {
                    const int y0 = min(Q.sy, nbins - 1), y1 = min(Q.ey, nbins - 1);  // rows [sy, ey], both inside [0, H]
                    if (y0 <= y1) {
                        const int s0 = s_start[y0], l0 = s_start[y1 + 1] - s0;
                        rg[0] = (uint32_t)s0 | ((uint32_t)l0 << 16), total = l0;
                    }
                } else if (Q.sx < Q.ex) {
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const int r = Q.sy + k;  // cells [sy, ey) x [sx, ex), already clipped to the grid by make_query_track
                        if (r < Q.ey) {
                            const int s0 = s_start[r * ccx + Q.sx], l0 = s_start[r * ccx + Q.ex] - s0;
                            rg[k] = (uint32_t)s0 | ((uint32_t)l0 << 16), total += l0;
                        }
                    }
                }
            }
            if (total == 0) ncand[q] = 0;  // (not visible, or an empty window: nothing for a wavefront to do)
            s_qw[2 * tid] = ls_u32x4{(uint32_t)total, __float_as_uint(Q.x), __float_as_uint(Q.y), rg[0]};
            s_qw[2 * tid + 1] = ls_u32x4{rg[1], rg[2], rg[3], rg[4]};
        }
        __syncthreads();  // (... and the train set is in place)
        if (stamp) stamp[2] = clock64();
        // ---- 3. FOUR queries per wavefront, one per DPP row of 16 lanes, a lane per candidate and pass of 16 candidates
        const int row = lane >> 4, sl = lane & 15;
        uint32_t *dst_base = cand + (size_t)q0 * KC;
        // a window of more than 64 features (rare): the whole wavefront takes the one query, 64 candidates per pass; the keys that pass are compacted into the
        // wavefront's segment (ballot prefix) and ranked with broadcast reads of four keys at a time
        auto long_window = [&](int ql) {
            const ls_u32x4 W0 = s_qw[2 * ql], W1 = s_qw[2 * ql + 1], d0 = s_qd[2 * ql], d1 = s_qd[2 * ql + 1];
            const int total = __builtin_amdgcn_readfirstlane((int)W0.x);
            const float qx = __uint_as_float(W0.y), qy = __uint_as_float(W0.z);
            const int l0 = (int)(W0.w >> 16), l1 = (int)(W1.x >> 16), l2 = (int)(W1.y >> 16), l3 = (int)(W1.z >> 16);
            const int c1 = l0, c2 = c1 + l1, c3 = c2 + l2, c4 = c3 + l3;
            const int o0 = (int)(W0.w & 0xFFFFu), o1 = (int)(W1.x & 0xFFFFu) - c1, o2 = (int)(W1.y & 0xFFFFu) - c2, o3 = (int)(W1.z & 0xFFFFu) - c3,
                      o4 = (int)(W1.w & 0xFFFFu) - c4;
            const float fsy = (float)max((int)qy - ROW_RADIUS, 0), fey = (float)min((int)qy + ROW_RADIUS, nbins - 1);
            int n = 0;
            for (int vb = 0; vb < total; vb += 64) {
                const int v = vb + lane;
                const bool in = v < total;
                int pos = v + o0;
                if (MODE != MODE_ROW) {
                    pos = (v >= c1) ? v + o1 : pos;
                    pos = (v >= c2) ? v + o2 : pos;
                    pos = (v >= c3) ? v + o3 : pos;
                    pos = (v >= c4) ? v + o4 : pos;
                }
                pos = in ? pos : 0;
                const float2 c = s_xy[pos];
                const uint32_t id = s_idx[pos];
                const ls_u32x4 a0 = s_dlo[pos], a1 = s_dhi[pos];
                bool ok;
                if (MODE == MODE_ROW) ok = c.y >= fsy && c.y <= fey;
                else {
                    const float dx = c.x - qx, dy = c.y - qy;
                    ok = (dx * dx + dy * dy) < r2;
                }
                ok = ok && in;
                uint32_t d = ls_bcnt(d0.x ^ a0.x, 0u);
                d = ls_bcnt(d0.y ^ a0.y, d), d = ls_bcnt(d0.z ^ a0.z, d), d = ls_bcnt(d0.w ^ a0.w, d);
                d = ls_bcnt(d1.x ^ a1.x, d), d = ls_bcnt(d1.y ^ a1.y, d), d = ls_bcnt(d1.z ^ a1.z, d), d = ls_bcnt(d1.w ^ a1.w, d);
                const uint64_t bal = __ballot(ok);
                const int slot = n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (ok && slot < KC + 64) s_seg[slot] = (d << 16) | id;
                n += __popcll(bal);
            }
            if (lane == 0) ncand[q0 + ql] = n;  // (> KC: the resolvers take the exact slow path, as with k_candidates)
            if (n > 0 && n <= KC) {
                if (lane < ((n + 3) & ~3) - n) s_seg[n + lane] = 0xFFFFFFFFu;  // padded to whole vectors of four
                __builtin_amdgcn_wave_barrier();  // (LDS operations of one wavefront complete in order: no hardware barrier needed)
                const uint4 *seg4 = reinterpret_cast<const uint4 *>(s_seg);
                const uint32_t ka = lane < n ? s_seg[lane] : 0xFFFFFFFFu, kb = lane + 64 < n ? s_seg[lane + 64] : 0xFFFFFFFFu;
                int ra = 0, rb = 0;
                for (int o = 0; o < n; o += 4) {
                    const uint4 kv = seg4[o >> 2];
                    ra += (kv.x < ka) + (kv.y < ka) + (kv.z < ka) + (kv.w < ka);
                    rb += (kv.x < kb) + (kv.y < kb) + (kv.z < kb) + (kv.w < kb);
                }
                uint32_t *dst = dst_base + (size_t)ql * KC;
                if (lane < n) dst[ra] = ka;
                if (lane + 64 < n) dst[rb] = kb;
            }
            __builtin_amdgcn_wave_barrier();
        };
        for (int qb = wv * 4; qb < nq; qb += LS_WAVES * 4) {
            const int ql = min(qb + row, nq - 1);
            const bool have = qb + row < nq;
            const ls_u32x4 W0 = s_qw[2 * ql], W1 = s_qw[2 * ql + 1];
            const int total = have ? (int)W0.x : 0;
            const int tmax = max(max(__builtin_amdgcn_readlane(total, 0), __builtin_amdgcn_readlane(total, 16)),
                                 max(__builtin_amdgcn_readlane(total, 32), __builtin_amdgcn_readlane(total, 48)));
            if (tmax == 0) continue;
            if (tmax > 64) {  // (wave-uniform)
                for (int r = 0; r < 4; r++)
                    if (qb + r < nq && __builtin_amdgcn_readfirstlane((int)s_qw[2 * (qb + r)].x) > 0) long_window(qb + r);
                continue;
            }
            const ls_u32x4 d0 = s_qd[2 * ql], d1 = s_qd[2 * ql + 1];
            const float qx = __uint_as_float(W0.y), qy = __uint_as_float(W0.z);
            // flattened index space: v in [c_k, c_{k+1}) is position v + o_k
            const int l0 = (int)(W0.w >> 16), l1 = (int)(W1.x >> 16), l2 = (int)(W1.y >> 16), l3 = (int)(W1.z >> 16);
            const int c1 = l0, c2 = c1 + l1, c3 = c2 + l2, c4 = c3 + l3;
            const int o0 = (int)(W0.w & 0xFFFFu), o1 = (int)(W1.x & 0xFFFFu) - c1, o2 = (int)(W1.y & 0xFFFFu) - c2, o3 = (int)(W1.z & 0xFFFFu) - c3,
                      o4 = (int)(W1.w & 0xFFFFu) - c4;
            // make_query_row's band again (struct.cpp:124-131); nbins - 1 = Params::H (a load from the Seq record here would be a memory round trip per query)
            const float fsy = (float)max((int)qy - ROW_RADIUS, 0), fey = (float)min((int)qy + ROW_RADIUS, nbins - 1);
            uint32_t *seg = s_seg + row * 64;
            const int P = (tmax + 15) >> 4;  // passes of 16 candidates per row
            uint32_t K[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            int n = 0;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                if (p < P) {
                    const int v = p * 16 + sl;
                    const bool in = v < total;
                    int pos = v + o0;
                    if (MODE != MODE_ROW) {
                        pos = (v >= c1) ? v + o1 : pos;
                        pos = (v >= c2) ? v + o2 : pos;
                        pos = (v >= c3) ? v + o3 : pos;
                        pos = (v >= c4) ? v + o4 : pos;
                    }
                    pos = in ? pos : 0;
                    const float2 c = s_xy[pos];
                    const uint32_t id = s_idx[pos];
                    const ls_u32x4 a0 = s_dlo[pos], a1 = s_dhi[pos];
                    bool ok;
                    if (MODE == MODE_ROW) ok = c.y >= fsy && c.y <= fey;  // struct.cpp:132-134
                    else {
                        const float dx = c.x - qx, dy = c.y - qy;
                        ok = (dx * dx + dy * dy) < r2;  // struct.cpp:95-99
                    }
                    ok = ok && in;
                    uint32_t d = ls_bcnt(d0.x ^ a0.x, 0u);
                    d = ls_bcnt(d0.y ^ a0.y, d), d = ls_bcnt(d0.z ^ a0.z, d), d = ls_bcnt(d0.w ^ a0.w, d);
                    d = ls_bcnt(d1.x ^ a1.x, d), d = ls_bcnt(d1.y ^ a1.y, d), d = ls_bcnt(d1.z ^ a1.z, d), d = ls_bcnt(d1.w ^ a1.w, d);
                    const uint32_t key = (d << 16) | id;
                    const uint32_t brow = (uint32_t)(__ballot(ok) >> (row * 16)) & 0xFFFFu;  // this row's candidates that passed
                    if (ok) {
                        K[p] = key;
                        seg[n + __popc(brow & ((1u << sl) - 1u))] = key;
                    }
                    n += __popc(brow);
                }
            }
            if (have && sl == 0) ncand[q0 + ql] = n;  // (<= 64 <= KC)
            if (sl < ((n + 3) & ~3) - n) seg[n + sl] = 0xFFFFFFFFu;  // padded to whole vectors of four
            __builtin_amdgcn_wave_barrier();  // (LDS operations of one wavefront complete in order: no hardware barrier needed)
            // rank = number of smaller keys of the same query (keys are unique: they carry the index): every row reads ITS segment, four keys per read
            const int nmax = max(max(__builtin_amdgcn_readlane(n, 0), __builtin_amdgcn_readlane(n, 16)), max(__builtin_amdgcn_readlane(n, 32), __builtin_amdgcn_readlane(n, 48)));
            const uint4 *seg4 = reinterpret_cast<const uint4 *>(seg);
            int rk[4] = {0, 0, 0, 0};
            for (int o = 0; o < nmax; o += 4) {
                uint4 kv = seg4[o >> 2];
                if (o >= n) kv = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);  // (past this row's keys: whatever an earlier query left there)
#pragma unroll
                for (int p = 0; p < 4; p++)
                    if (p < P) rk[p] += (kv.x < K[p]) + (kv.y < K[p]) + (kv.z < K[p]) + (kv.w < K[p]);
            }
            uint32_t *dst = dst_base + (size_t)ql * KC;
#pragma unroll
            for (int p = 0; p < 4; p++)
                if (p < P && K[p] != 0xFFFFFFFFu) dst[rk[p]] = K[p];
            __builtin_amdgcn_wave_barrier();  // the segments are rewritten by this wavefront's next four queries
        }
        if (stamp) stamp[3] = clock64();
    }
    if (stamp) stamp[4] = stamp[5] = clock64();
}

}  // namespace lvt
