// k_lists.hip -- k_hamming_batched_lists: the pipeline's candidate LISTS from the binned matcher (round 2).
//
// The greedy accept / mark scans of find_matches and row_match (k_early_mid / k_track_mid / k_triangulate) walk, per query, the
// candidates of lvt_image_features_struct.cpp:68-148 sorted by (Hamming distance, index).  k_candidates / k_early_map build those
// lists with one WAVEFRONT per query scanning ALL train features (64 lanes x N / 64 trips of predicate tests per query): in a
// lock-step batch of 16 sequences that is 95 + 103 us of kernels per frame.  This kernel builds the same lists the way
// k_hamming_batched finds its top-2 (k_hamming.hip): ONE 1024-thread workgroup per sequence stages the train set once in LDS,
// counting-sorted into the reference's own 25-px hash cells (tracking) or image rows (row matching); ONE LANE per query then walks
// only its window's contiguous LDS ranges -- ~18 candidates instead of N -- twice: first counting the hits of the exact
// predicate, so that a block scan can hand every query its own segment of an LDS arena, then computing the 256-bit distances
// into that segment as packed (distance << 16 | index) keys, and finally ranking every key among its segment's keys: the rank is its
// place in the list.
//
// Output contract = k_candidates': cand[q][0 .. min(n, KC)) ascending, ncand[q] = n (n > KC: the resolvers' exact slow path);
// map mode also projects the point first (is_point_visible, lvt_local_map.cpp:62-82,152-156) and leaves proj / vis / match /
// counter exactly as k_early_map does.  The bins ARE the reference's hash cells (Feat::hcx / hcy), so "candidate in the window's
// cells" is decided by the range bounds and the walk only evaluates the radius; row mode evaluates the band test itself.
// Capacity: 2048 train features and 4100 bins per LDS image; beyond that (or for a cell search radius above 2) the kernel does
// NOTHING and raises Seq::lists_fb, and the wave-per-query kernel launched behind it does the work as before.
#include "lvt_dev.h"
#include "lvt_math.h"

namespace lvt {

constexpr int LS_THREADS = 1024;
constexpr int LS_NMAX = 2048;   // train features of one LDS image
constexpr int LS_BINS = 4100;   // hash cells / image rows + 1
constexpr int LS_ARENA = 11776; // list entries in flight (one chunk of <= 1024 queries; larger chunks are cut)
constexpr int LS_WORK = 1024;   // ... and behind them the work list of the lists a whole wavefront ranks (one word per thread at most)
constexpr int LS_COOP_MIN = 20; // lists at least this long are ranked by a wavefront: a lane's rank-by-counting is (n / 4)^2 trips and the kernel ends with its longest list
constexpr int LS_STARTS = (LS_BINS + 4) & ~3;  // (the arena behind the bin starts is 16-byte aligned)
constexpr int LS_LDS_BYTES = LS_NMAX * 42 + LS_STARTS * 4 + (LS_ARENA + LS_WORK) * 4 + 256;
typedef unsigned int ls_u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool BV>
__global__ __launch_bounds__(LS_THREADS) void k_hamming_batched_lists(SeqArg<BV> sa, int par, seq_t seq) {
    const Seq &S = sa.get();
    Ctl &ctl = *S.ctl;
    const int tid = threadIdx.x;
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
    uint32_t *s_arena = reinterpret_cast<uint32_t *>(s_start + LS_STARTS);
    uint32_t *s_work = s_arena + LS_ARENA;
    __shared__ int s_work_n;
    __shared__ int s_scan[32];
    __shared__ int s_next;
    __shared__ double w2c[12];

    long long *stamp = (MODE == MODE_ROW && blockIdx.x == 0 && tid == 0) ? ctl.dbg + 26 : nullptr;  // (tools/lists_phases.py)
    if (stamp) stamp[0] = clock64();
    if (MODE == MODE_MAP && tid == 0) {  // the prediction, recomputed from the persistent state (k_early_map does the same)
        Pose predicted;
        double mmn[14];
        motion_predict(ctl, ctl.last_pose, predicted, mmn);
        world_to_camera(predicted, w2c);
    }
    // ---- the train set, counting-sorted into the reference's hash cells / image rows (two features per thread at most)
    if (tid == 0) s_work_n = 0;
    for (int i = tid; i <= nbins; i += LS_THREADS) s_start[i] = 0;
    __syncthreads();
    int tbin[2], trank[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int j = tid + k * LS_THREADS;
        tbin[k] = 0, trank[k] = 0;
        if (j < N) {
            if (MODE == MODE_ROW) tbin[k] = min(max((int)floorf(T.y[j]), 0), nbins - 1);
            else tbin[k] = min(max((int)T.hcy[j], 0), ccy - 1) * ccx + min(max((int)T.hcx[j], 0), ccx - 1);
            trank[k] = atomicAdd(&s_start[tbin[k]], 1);
        }
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
    for (int k = 0; k < 2; k++) {
        const int j = tid + k * LS_THREADS;
        if (j < N) {
            const int pos = s_start[tbin[k]] + trank[k];
            s_xy[pos] = make_float2(T.x[j], T.y[j]);
            s_idx[pos] = (uint16_t)j;
            const ls_u32x4 *d = reinterpret_cast<const ls_u32x4 *>(T.desc + (size_t)j * 4);
            s_dlo[pos] = d[0];
            s_dhi[pos] = d[1];
        }
    }
    __syncthreads();

    if (stamp) stamp[1] = clock64();
    const uint64_t *qdesc = (MODE == MODE_MAP) ? S.map[*S.map_cur].desc : S.fb[par].feat[0].desc;
    uint32_t *cand = (MODE == MODE_MAP) ? S.cand : S.rcand + (size_t)par * NF_MAX * KC;
    int *ncand = (MODE == MODE_MAP) ? S.ncand : S.rncand + par * NF_MAX;
    const int radius = S.prm.tracking_radius;

    // gridDim.x workgroups per sequence share the queries in EQUAL contiguous parts (whole wavefronts; each stages the train set itself): a list is one
    // lane's serial walk, so what a second workgroup buys is issue slots and LDS bandwidth -- 1 000 row lists on one CU are 16 wavefronts on 4 SIMDs
    // (117 us in a batch of 64 sequences), on eight CUs two wavefronts each
    const int part = (((q_end + (int)gridDim.x - 1) / (int)gridDim.x) + 63) & ~63;
    const int q_lo = (int)blockIdx.x * part, q_hi = min(q_end, q_lo + part);
    for (int q0 = q_lo; q0 < q_hi; q0 += LS_THREADS) {
        const int q = q0 + tid;
        bool live = q < q_hi;
        Query Q;
        Q.x = Q.y = Q.r2 = 0.f;
        Q.sy = Q.sx = 0, Q.ey = Q.ex = 0;
        if (live) {
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
                    ncand[q] = 0;
                    live = false;
                }
            } else {
                make_query_row(S.prm, S.fb[par].feat[0].x[q], S.fb[par].feat[0].y[q], Q);
            }
        }
        // the window as contiguous LDS ranges: one per hash row of the window (tracking), one in all (row band)
        int rs[5], rl[5];
#pragma unroll
        for (int k = 0; k < 5; k++) rs[k] = rl[k] = 0;
        if (live) {
            if (MODE == MODE_ROW) {
                const int y0 = min(Q.sy, nbins - 1), y1 = min(Q.ey, nbins - 1);  // rows [sy, ey], both inside [0, H]
                if (y0 <= y1) rs[0] = s_start[y0], rl[0] = s_start[y1 + 1] - rs[0];
            } else if (Q.sx < Q.ex) {
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const int r = Q.sy + k;  // cells [sy, ey) x [sx, ex), already clipped to the grid by make_query_track
                    if (r < Q.ey) rs[k] = s_start[r * ccx + Q.sx], rl[k] = s_start[r * ccx + Q.ex] - rs[k];
                }
            }
        }
        const float fsy = (float)Q.sy, fey = (float)Q.ey;
        // ---- pass 1: how many candidates satisfy the reference's predicate
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 5; k++)
            for (int pos = rs[k]; pos < rs[k] + rl[k]; pos++) {
                const float2 c = s_xy[pos];
                bool ok;
                if (MODE == MODE_ROW) ok = c.y >= fsy && c.y <= fey;  // struct.cpp:132-134
                else {
                    const float dx = c.x - Q.x, dy = c.y - Q.y;
                    ok = (dx * dx + dy * dy) < Q.r2;  // struct.cpp:95-99
                }
                cnt += ok ? 1 : 0;
            }
        if (stamp) stamp[2] = clock64();
        int total;
        const int cnt4 = (cnt + 3) & ~3;  // segments start on 16-byte boundaries: the ranking below reads four keys per LDS access
        const int off = block_excl_scan(cnt4, s_scan, &total);
        uint64_t qd[4] = {0, 0, 0, 0};
        if (live && cnt > 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) qd[k] = qdesc[(size_t)q * 4 + k];
        }
        // ---- pass 2: distances, insertion-sorted into the query's arena segment; chunks whose lists exceed the arena are cut at a query
        int base = 0;
        bool done = !live;
        if (live && cnt == 0) {
            ncand[q] = 0;
            done = true;
        }
        for (;;) {
            const bool mine = !done && (off + cnt4 - base <= LS_ARENA);  // (monotone in tid: the threads that fit are a prefix of those left)
            if (mine) {
                uint32_t *seg = s_arena + (off - base);
                int n = 0;
#pragma unroll
                for (int k = 0; k < 5; k++)
                    for (int pos = rs[k]; pos < rs[k] + rl[k]; pos++) {
                        const float2 c = s_xy[pos];
                        bool ok;
                        if (MODE == MODE_ROW) ok = c.y >= fsy && c.y <= fey;
                        else {
                            const float dx = c.x - Q.x, dy = c.y - Q.y;
                            ok = (dx * dx + dy * dy) < Q.r2;
                        }
                        if (ok) {
                            const ls_u32x4 a0 = s_dlo[pos], a1 = s_dhi[pos];
                            const int d = __popcll(qd[0] ^ (((uint64_t)a0.y << 32) | a0.x)) + __popcll(qd[1] ^ (((uint64_t)a0.w << 32) | a0.z)) +
                                          __popcll(qd[2] ^ (((uint64_t)a1.y << 32) | a1.x)) + __popcll(qd[3] ^ (((uint64_t)a1.w << 32) | a1.z));
                            seg[n++] = ((uint32_t)d << 16) | (uint32_t)s_idx[pos];
                        }
                    }
                // rank = number of smaller keys (keys are unique: they carry the index): n^2 INDEPENDENT LDS reads and compares per lane,
                // where an insertion sort is a chain of dependent read-modify-writes as long as its longest list in the wavefront
                // (measured: 400 us for the row lists of a 16-sequence batch with the insertion sort)
                if (n <= KC && n >= LS_COOP_MIN) {
                    // a long list: a whole wavefront ranks it below (one key per lane, every comparison partner a broadcast read)
                    for (int e = n; e < cnt4; e++) seg[e] = 0xFFFFFFFFu;
                    s_work[atomicAdd(&s_work_n, 1)] = (uint32_t)tid | ((uint32_t)(off - base) << 10) | ((uint32_t)(n - 1) << 24);
                } else if (n <= KC) {
                    for (int e = n; e < cnt4; e++) seg[e] = 0xFFFFFFFFu;  // padding: never smaller than a key
                    const uint4 *seg4 = reinterpret_cast<const uint4 *>(seg);
                    for (int e = 0; e < n; e += 4) {  // four keys against every vector of four: n^2 / 16 LDS reads
                        const uint4 k4 = seg4[e >> 2];
                        int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
                        for (int o = 0; o < cnt4; o += 4) {
                            const uint4 v = seg4[o >> 2];
                            r0 += (v.x < k4.x) + (v.y < k4.x) + (v.z < k4.x) + (v.w < k4.x);
                            r1 += (v.x < k4.y) + (v.y < k4.y) + (v.z < k4.y) + (v.w < k4.y);
                            r2 += (v.x < k4.z) + (v.y < k4.z) + (v.z < k4.z) + (v.w < k4.z);
                            r3 += (v.x < k4.w) + (v.y < k4.w) + (v.z < k4.w) + (v.w < k4.w);
                        }
                        uint32_t *dst = cand + (size_t)q * KC;
                        dst[r0] = k4.x;
                        if (e + 1 < n) dst[r1] = k4.y;
                        if (e + 2 < n) dst[r2] = k4.z;
                        if (e + 3 < n) dst[r3] = k4.w;
                    }
                }
                ncand[q] = n;  // (> KC: the resolvers take the exact slow path, as with k_candidates)
                done = true;
            }
            if (stamp) stamp[3] = clock64();
            // ---- the long lists of this pass, one per wavefront at a time: lane l owns keys l and l + 64 (n <= KC = 128), counts the smaller ones
            //      among all n (four per broadcast read) and stores its keys at their ranks -- ~n / 4 trips where a single lane needs (n / 4)^2
            __syncthreads();
            if (stamp) stamp[4] = clock64();
            {
                const int nwork = s_work_n, lane = tid & 63;
                for (int w = tid >> 6; w < nwork; w += LS_THREADS / 64) {
                    const uint32_t e = s_work[w];
                    const int qw = q0 + (int)(e & 1023u), n = (int)(e >> 24) + 1;
                    const uint32_t *seg = s_arena + ((e >> 10) & 16383u);
                    const uint4 *seg4 = reinterpret_cast<const uint4 *>(seg);
                    const uint32_t ka = lane < n ? seg[lane] : 0xFFFFFFFFu, kb = lane + 64 < n ? seg[lane + 64] : 0xFFFFFFFFu;
                    int ra = 0, rb = 0;
                    for (int o = 0; o < n; o += 4) {  // (the segment is padded with 0xFFFFFFFF up to a multiple of four)
                        const uint4 v = seg4[o >> 2];
                        ra += (v.x < ka) + (v.y < ka) + (v.z < ka) + (v.w < ka);
                        rb += (v.x < kb) + (v.y < kb) + (v.z < kb) + (v.w < kb);
                    }
                    uint32_t *dst = cand + (size_t)qw * KC;
                    if (lane < n) dst[ra] = ka;
                    if (lane + 64 < n) dst[rb] = kb;
                }
            }
            __syncthreads();
            if (stamp) stamp[5] = clock64();
            if (tid == 0) s_work_n = 0;
            if (total <= LS_ARENA) break;  // (block-uniform: everything fitted in one go)
            __syncthreads();
            if (tid == 0) s_next = 0x7FFFFFFF;
            __syncthreads();
            if (!done) atomicMin(&s_next, off);
            __syncthreads();
            const int nb = s_next;
            if (nb == 0x7FFFFFFF) break;
            base = nb;
        }
        __syncthreads();  // the arena is reused by the next chunk
    }
}

}  // namespace lvt
