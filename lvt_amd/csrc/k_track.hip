// k_track.hip -- map<->frame matching, motion-only BA, map maintenance, stereo triangulation (gfx950).
//
//   T = tracking stream (the frame period), E = early stream (frame t+1's share of find_matches, behind k_pnp(t)), F = feature stream
//   k_candidates<ROW|STAGED> : masked Hamming candidate lists, one wavefront per query   (lvt_image_features_struct.cpp:68-148)
//   k_gate / k_gate_buf / k_gate_late / k_feat_done / k_row_done : the polling hand-over between the streams
//   E k_early_map  : projection + candidate lists of the map points that survived the previous frame's clean-up
//   E k_early_mid  : their greedy accept/mark scan (the first part of find_matches' storage-order scan)
//   T k_match_map  : (single sequence: delivers the previous frame's record and waits for the early stream, then) frame prologue
//                    (motion model, state machine; lvt_system.cpp:157-197) + is_point_visible / projection
//                    and candidate lists of the points appended since                        (lvt_local_map.cpp:62-82,152)
//   T k_track_mid  : the rest of the accept/mark scan of find_matches (pass 1, rare pass 2), counters / ages / PnP input,
//                    LOST decision, clean_untracked_points   (lvt_local_map.cpp:146-224,393-413, lvt_system.cpp:267-274)
//   T k_pnp        : g2o Levenberg-Marquardt, 2 passes x optimize(5), on device; projects the staged points with the result
//                                                                               (lvt_pnp_solver.cpp:60-128)
//   T k_triangulate: update_staged_map_points + triangulation policy (lvt_local_map.cpp:355-391, lvt_system.cpp:308-334),
//                    row_match accept/mark scan, linear-LS triangulation + gates, append to map / staged, frame epilogue,
//                    result record              (handler.cpp:302-323, lvt_local_map.cpp:231-353, lvt_system.cpp:185-193)
//
// The Hamming distance evaluation is parallel (k_candidates); only the accept/mark scan is sequential,
// and it walks pre-sorted candidate lists so it touches a few words per query.
#include "lvt_dev.h"
#include "wave_reduce.h"
#include "lvt_math.h"

namespace lvt {

enum : int { MODE_MAP = 0, MODE_STAGED = 1, MODE_ROW = 2 };

// =================================================================================================
// frame prologue (the head of lvt_system::track, lvt_system.cpp:157-167,196-197), evaluated by k_match_map.  The motion
// model's next state goes to a shadow (mm_next) committed by k_track_mid.
// =================================================================================================
__device__ __forceinline__ void frame_prologue(const Seq &S, Ctl &c, int par, const Pose &predicted, const double mm_next[14], bool active,
                                               bool first, bool skipped) {
    const bool absent = skipped && S.fb[par].fc->absent != 0;  // (a pooled handle's idle step: nothing may change, not even the frame counter)
    for (int i = 0; i < N_COUNTS; i++) c.counts[i] = 0;
    c.counts[C_FRAME] = c.frame_number;
    if (!absent) c.frame_number++;
    c.active = active ? 1 : 0;
    c.first_frame = first ? 1 : 0;
    c.do_pass2 = 0;
    c.n_pass1 = c.n_pass2 = 0;
    c.n_matches = 0;
    c.lost_now = 0;
    c.need_tri = 0;
    c.dont_stage = 0;
    c.n_pairs = 0;
    c.mm_pending = 0;
    const FeatCtl &fc = *S.fb[par].fc;
    c.overflow = fc.overflow;
    c.counts[C_RETRY_LEFT] = active ? fc.retry[0] : 0;  // (LOST: the reference returns before it detects anything; the feature
    c.counts[C_RETRY_RIGHT] = active ? fc.retry[1] : 0;  //  stream here has run regardless -- its results are not reported)
    if (!active) {  // LOST: return the last pose forever (lvt_system.cpp:161-166).  A SKIPPED frame (its features never arrived: a gate
                    // timed out, reported through lvt_amd_last_error) also returns the last pose, but the state stays what it was
        pose_to_Rt(c.last_pose, c.out_R, c.out_t);
        c.out_status = skipped ? c.state : 3;
        return;
    }
    if (!first) {  // lvt_system.cpp:197 -> lvt_motion_model.cpp:42-65
        c.predicted = predicted;
        for (int k = 0; k < 14; k++) c.mm_next[k] = mm_next[k];
        c.mm_pending = 1;
    }
}

// staged points are projected with the optimised pose in the tail of k_pnp
__device__ __forceinline__ void project_staged(const Seq &S, const Pose &pose, double *w2c_lds) {
    if (threadIdx.x == 0) world_to_camera(pose, w2c_lds);
    __syncthreads();
    const int M = *S.staged_n;
    const MapSoA &P = S.staged[*S.staged_cur];
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const double X[3] = {P.pos[3 * i], P.pos[3 * i + 1], P.pos[3 * i + 2]};
        double u, v;
        if (is_point_visible(X, w2c_lds, S.prm, u, v)) {
            S.sproj[2 * i] = (float)u;
            S.sproj[2 * i + 1] = (float)v;
            S.svis[i] = 1;
        } else
            S.svis[i] = 0;
    }
}

// map SoA copy helper
__device__ __forceinline__ void copy_point(const MapSoA &src, int i, const MapSoA &dst, int o) {
    dst.pos[3 * o] = src.pos[3 * i];
    dst.pos[3 * o + 1] = src.pos[3 * i + 1];
    dst.pos[3 * o + 2] = src.pos[3 * i + 2];
#pragma unroll
    for (int k = 0; k < 4; k++) dst.desc[(size_t)o * 4 + k] = src.desc[(size_t)i * 4 + k];
    dst.counter[o] = src.counter[i];
    dst.age[o] = src.age[i];
    dst.match_idx[o] = src.match_idx[i];
}

// =================================================================================================
// candidate predicate shared by k_candidates and the exact slow path of the resolvers
// =================================================================================================
struct Query {
    float x, y;
    int sy, ey, sx, ex;   // hash-cell window (tracking)  /  [sy, ey] row band (row matching)
    float r2;
    uint64_t d[4];
};

__device__ __forceinline__ void make_query_track(const Params &p, float x, float y, int radius, Query &q) {
    q.x = x;
    q.y = y;
    const int hy = (int)floorf(y / (float)HASH_CELL), hx = (int)floorf(x / (float)HASH_CELL);
    q.sy = max(hy - p.cell_search_radius, 0);
    q.ey = min(hy + p.cell_search_radius + 1, p.hash_ccy);
    q.sx = max(hx - p.cell_search_radius, 0);
    q.ex = min(hx + p.cell_search_radius + 1, p.hash_ccx);
    q.r2 = (float)(radius * radius);
}
__device__ __forceinline__ void make_query_row(const Params &p, float x, float y, Query &q) {
    q.x = x;
    q.y = y;
    int s = (int)y - ROW_RADIUS;
    if (s < 0) s = 0;
    int e = (int)y + ROW_RADIUS;
    if (e > p.H) e = p.H;
    q.sy = s;
    q.ey = e;
    q.sx = q.ex = 0;
    q.r2 = 0;
}
template <int MODE>
__device__ __forceinline__ bool cand_pred(const Query &q, const Feat &F, int j) {
    if (MODE == MODE_ROW) {
        const float fy = F.y[j];
        return fy >= (float)q.sy && fy <= (float)q.ey;
    } else {
        const int cy = F.hcy[j], cx = F.hcx[j];
        if (cy < q.sy || cy >= q.ey || cx < q.sx || cx >= q.ex) return false;
        const float dx = F.x[j] - q.x, dy = F.y[j] - q.y;
        return (dx * dx + dy * dy) < q.r2;
    }
}

// =================================================================================================
// k_candidates : one wavefront per query; output sorted ascending by (distance << 16 | index)
// =================================================================================================
// LDS of the candidate generator: per-wave list buffer + the train-side predicate data staged once per block
struct CandLds {
    uint32_t *lbuf;          // [waves][KC]
    float *tx, *ty;          // [NF_MAX]
    uint32_t *tc;            // [NF_MAX] packed hash cell (cy << 16 | cx)
    const double *w2c;       // PROJECT: world -> camera of the predicted pose (LDS)
};

// wave `wave0 + k * wave_stride` handles query wave0 + k * wave_stride; all `nthreads` threads of the block stage the
// train data.  MODE_ROW lists are built for EVERY left feature (they depend on the two feature sets only, so the
// kernel runs on the feature stream); the resolver skips the left features tracking has already matched.
// PROJECT (map mode, pass 1): the wavefront first projects its map point with the predicted pose (is_point_visible,
// lvt_local_map.cpp:62-82,152-156) and records the projection for the rest of the chain.
template <int MODE, bool PROJECT = false>
__device__ __forceinline__ void candidates_body(const Seq &S, int pass2, int par, CandLds &C, int wave0, int wave_stride, int nthreads, int q_begin = 0,
                                                int q_end = -1) {
    uint32_t *s_tc = C.tc;
    float *s_tx = C.tx, *s_ty = C.ty;
    const int lane = lane_id(), wv = wave_id();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const Feat &T = (MODE == MODE_ROW) ? S.fb[par].feat[1] : S.fb[par].feat[0];
    const int N = *T.n;
    int M;
    const uint64_t *qdesc;
    uint32_t *cand;
    int *ncand;
    if (MODE == MODE_MAP) {
        M = *S.map_n;
        qdesc = S.map[*S.map_cur].desc;
        cand = S.cand;
        ncand = S.ncand;
    } else if (MODE == MODE_STAGED) {
        M = *S.staged_n;
        qdesc = S.staged[*S.staged_cur].desc;
        cand = S.scand;
        ncand = S.sncand;
    } else {
        M = *S.fb[par].feat[0].n;
        qdesc = S.fb[par].feat[0].desc;
        cand = S.rcand + (size_t)par * NF_MAX * KC;
        ncand = S.rncand + par * NF_MAX;
    }
    if (q_end >= 0) M = min(M, q_end);  // only queries [q_begin, q_end): the split of find_matches across two launches
    wave0 += q_begin;
    if (wave0 - wv >= M) return;  // no query for this block (block-uniform: wave0 - wv is the block's first query)
    // the first query's descriptor (and map point) are requested before the train data is staged: one memory round trip less
    uint64_t qd0[4] = {0, 0, 0, 0};
    double X0[3] = {0, 0, 0};
    if (wave0 < M) {
#pragma unroll
        for (int k = 0; k < 4; k++) qd0[k] = qdesc[(size_t)wave0 * 4 + k];
        if (MODE == MODE_MAP && PROJECT) {
            const MapSoA &P0 = S.map[*S.map_cur];
            X0[0] = P0.pos[3 * wave0], X0[1] = P0.pos[3 * wave0 + 1], X0[2] = P0.pos[3 * wave0 + 2];
        }
    }
    {   // PROJECT: wavefront 0 arrives late (its thread 0 ran the frame prologue), the other waves stage the train data meanwhile
        const int t0 = PROJECT ? (int)threadIdx.x - 64 : (int)threadIdx.x, tn = PROJECT ? nthreads - 64 : nthreads;
        if (t0 >= 0)
            for (int j0 = t0; j0 < N; j0 += 4 * tn) {  // four features per thread and trip: all their loads are in flight together
                float vx[4], vy[4];
                int16_t cy[4], cx[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = min(j0 + k * tn, N - 1);
                    vx[k] = T.x[j], vy[k] = T.y[j];
                    if (MODE != MODE_ROW) cy[k] = T.hcy[j], cx[k] = T.hcx[j];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = j0 + k * tn;
                    if (j < N) {
                        s_tx[j] = vx[k];
                        s_ty[j] = vy[k];
                        if (MODE != MODE_ROW) s_tc[j] = ((uint32_t)(uint16_t)cy[k] << 16) | (uint32_t)(uint16_t)cx[k];
                    }
                }
            }
    }
    __syncthreads();
    const int radius = S.prm.tracking_radius * ((MODE == MODE_MAP && pass2) ? 2 : 1);
    uint32_t *buf = C.lbuf + wv * KC;
    for (int i = wave0; i < M; i += wave_stride) {
        Query q;
        if (MODE == MODE_MAP && PROJECT) {
            const MapSoA &P = S.map[*S.map_cur];
            double X[3] = {X0[0], X0[1], X0[2]};
            if (i != wave0) X[0] = P.pos[3 * i], X[1] = P.pos[3 * i + 1], X[2] = P.pos[3 * i + 2];
            double u, v;
            const bool visible = is_point_visible(X, C.w2c, S.prm, u, v);  // wave-uniform
            if (lane == 0) {
                if (visible) {
                    S.proj[2 * i] = (float)u;
                    S.proj[2 * i + 1] = (float)v;
                    S.vis[i] = 1;
                    S.match[i] = -1;
                } else {
                    S.vis[i] = 0;
                    P.counter[i] += 1;  // lvt_local_map.cpp:154
                    S.match[i] = -2;
                    ncand[i] = 0;
                }
            }
            if (!visible) continue;
            make_query_track(S.prm, (float)u, (float)v, radius, q);
        } else if (MODE == MODE_MAP) {
            if (!S.vis[i]) {
                if (lane == 0) ncand[i] = 0;
                continue;
            }
            make_query_track(S.prm, S.proj[2 * i], S.proj[2 * i + 1], radius, q);
        } else if (MODE == MODE_STAGED) {
            if (!S.svis[i]) {
                if (lane == 0) ncand[i] = 0;
                continue;
            }
            make_query_track(S.prm, S.sproj[2 * i], S.sproj[2 * i + 1], radius, q);
        } else {
            make_query_row(S.prm, S.fb[par].feat[0].x[i], S.fb[par].feat[0].y[i], q);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) q.d[k] = (i == wave0) ? qd0[k] : qdesc[(size_t)i * 4 + k];
        int cnt = 0;
        // the train data of the NEXT 64 features is read while the current ones are tested: with a dependent LDS read, a
        // conditional second one and the ballot in every iteration the 16 iterations ran at ~550 cycles each (LDS latency)
        uint32_t c_n = 0;
        float x_n = 0.f, y_n = 0.f;
        {
            const int j0 = min(lane, NF_MAX - 1);
            y_n = s_ty[j0];
            if (MODE != MODE_ROW) c_n = s_tc[j0], x_n = s_tx[j0];
        }
        for (int base = 0; base < N; base += 64) {
            const int j = base + lane;
            const uint32_t c = c_n;
            const float fx = x_n, fy = y_n;
            {
                const int jn = min(j + 64, NF_MAX - 1);  // (past N: stale or unwritten entries, never used)
                y_n = s_ty[jn];
                if (MODE != MODE_ROW) c_n = s_tc[jn], x_n = s_tx[jn];
            }
            bool ok = false;
            if (j < N) {
                if (MODE == MODE_ROW) {
                    ok = fy >= (float)q.sy && fy <= (float)q.ey;
                } else {
                    const int cy = (int)(int16_t)(c >> 16), cx = (int)(int16_t)(c & 0xFFFFu);
                    if (cy >= q.sy && cy < q.ey && cx >= q.sx && cx < q.ex) {
                        const float dx = fx - q.x, dy = fy - q.y;
                        ok = (dx * dx + dy * dy) < q.r2;
                    }
                }
            }
            // pass 1 only collects the indices of the candidates (ascending): fetching each candidate's descriptor inside this
            // loop made every iteration with a hit wait for its own global round trip (~4 per query)
            const uint64_t m = __ballot(ok);
            if (ok) {
                const int pos = cnt + __popcll(m & lt_mask);
                if (pos < KC) buf[pos] = (uint32_t)j;
            }
            cnt += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (cnt <= KC) {
            // pass 2: all descriptors of the query's candidates in one round trip (KC / 64 per lane), keys = (distance << 16 | index)
            for (int e = lane; e < cnt; e += 64) {
                const uint32_t j = buf[e];
                uint64_t t[4];
#pragma unroll
                for (int k = 0; k < 4; k++) t[k] = T.desc[(size_t)j * 4 + k];
                buf[e] = ((uint32_t)hamming256(q.d, t) << 16) | j;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            // rank sort (keys are unique): each lane places up to KC/64 entries
            for (int e = lane; e < cnt; e += 64) {
                const uint32_t k = buf[e];
                int rank = 0;
                for (int o = 0; o < cnt; o++) rank += (buf[o] < k) ? 1 : 0;
                cand[(size_t)i * KC + rank] = k;
            }
        }
        if (lane == 0) ncand[i] = cnt;  // > KC => resolvers take the exact slow path
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
}

// need_seq (row mode on the early stream): the lists are only built from COMPLETE features (the stream's gate may have given up
// waiting for them; then k_row_done does not publish either and k_triangulate reports the frame as lost)
// only_fb: the binned list kernel (k_lists.hip) ran in front of this one -- build the lists only where it stood down
template <int MODE, bool BV>
__global__ __launch_bounds__(256) void k_candidates(SeqArg<BV> sa, int pass2, int par, seq_t need_seq, int only_fb) {
    const Seq &S = sa.get();
    if (only_fb && !S.lists_fb[MODE == MODE_ROW ? 1 : 0]) return;
    if (MODE != MODE_ROW) {
        const Ctl &ctl = *S.ctl;
        if (!ctl.active || ctl.first_frame) return;
        if (MODE == MODE_STAGED && (ctl.lost_now || S.prm.staged_th <= 0)) return;
    } else {
        if (S.prm.sensor != 1) return;
        if (need_seq && __hip_atomic_load(&S.fb[par].fc->feat_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need_seq) return;
    }
    __shared__ uint32_t lbuf[4 * KC];
    __shared__ float s_tx[NF_MAX], s_ty[NF_MAX];
    __shared__ uint32_t s_tc[NF_MAX];
    CandLds C;
    C.lbuf = lbuf, C.tx = s_tx, C.ty = s_ty, C.tc = s_tc;
    candidates_body<MODE>(S, pass2, par, C, blockIdx.x * 4 + wave_id(), gridDim.x * 4, 256);
}

__device__ __forceinline__ void deliver_record(const Ctl &ctl, Ctl *rec_out, seq_t *done_out, seq_t seq, int nthreads);
__device__ __forceinline__ void gate_late_poll(Ctl &ctl, FeatCtl &fc, seq_t seq);
// k_match_map : frame prologue + projection of the map points + their candidate lists (find_matches pass 1).  Every block
// derives the per-frame facts it needs (active / first frame / predicted pose) from the PERSISTENT part of Ctl, which
// nobody writes during this kernel; block 0 additionally publishes them for the rest of the chain.
template <bool BV>
// gated (single sequence, polling mode: launched with 16 workgroups -- it only has the points appended since the early part to
// list): the wait for the early stream happens HERE instead of in a k_gate_late launch of its own, and workgroup 0 first delivers
// the previous frame's record.  16 spinning workgroups leave 240 CUs untouched for whatever they wait for (k_cells needs 20 CUs
// with free LDS; with 256 workgroups this deadlocked the synchronous mode until the time-out); a lock-step batch, whose grid
// would cover the chip again, keeps the separate gate kernel.
__global__ __launch_bounds__(256) void k_match_map(SeqArg<BV> sa, int par, seq_t seq, int gated, Ctl *prev_rec, seq_t *prev_done) {
    const Seq &S = sa.get();
    Ctl &ctl = *S.ctl;
    __shared__ int s_state;
    if (gated) {
        if (threadIdx.x == 0 && blockIdx.x == 0) ctl.dbg[38] = (long long)wall_clock64();  // (timeline: the wait starts)
        if (blockIdx.x == 0 && prev_rec) deliver_record(ctl, prev_rec, prev_done, seq - 1, 256);
        // ONE decision per frame: workgroup 0 waits for the early stream and the features, decides whether the frame is skipped and counts
        // the time-outs; the other workgroups wait for its verdict (they are co-resident: 16 workgroups) -- each polling on its own, two
        // workgroups could have disagreed about `skip` at the 2-s boundary
        if (threadIdx.x == 0) {
            if (blockIdx.x == 0) {
                gate_late_poll(ctl, *S.fb[par].fc, seq);
                __hip_atomic_store(&ctl.late_gate_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(&ctl.late_gate_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) __builtin_amdgcn_s_sleep(8);
            }
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) ctl.dbg[40] = (long long)wall_clock64();  // (timeline: the frame's work starts)
    __shared__ int s_skip;
    if (threadIdx.x == 0) {
        int sk = ctl.skip;  // state: persistent, not written by this kernel; skip: set by this frame's gate (above / k_gate_late)
        // event ordering runs no gate at all: a frame published WITHOUT features (a pooled seat that had no frame in this step, a buffer given up on) is found
        // here -- the feature stream's k_feat_done, ordered in front of this kernel by the event, wrote the word.  (Behind k_gate_late the flag is set already.)
        if (!gated && __hip_atomic_load(&S.fb[par].fc->skip_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == seq) {
            sk = 1;
            if (blockIdx.x == 0) ctl.skip = 1;  // (every workgroup decides from skip_seq itself: nobody waits for this store; the epilogue reads and clears it)
        }
        s_state = ctl.state, s_skip = sk;
    }
    __syncthreads();
    const int state = s_state;
    const bool skipped = s_skip != 0;
    const bool active = (state != 3) && !skipped, first = (state == 1);
    __shared__ double w2c[12];
    if (threadIdx.x == 0) {
        Pose predicted;
        double mmn[14];
        if (active && !first) {
            motion_predict(ctl, ctl.last_pose, predicted, mmn);
            world_to_camera(predicted, w2c);
        }
        if (blockIdx.x == 0) {
            frame_prologue(S, ctl, par, predicted, mmn, active, first, skipped);
            if (active && !first) ctl.counts[C_MAP_SIZE_AT_MATCH] = *S.map_n;
        }
    }
    if (!active || first) return;
    // no barrier here: the one after the train staging inside candidates_body also publishes w2c
    __shared__ uint32_t lbuf[4 * KC];
    __shared__ float s_tx[NF_MAX], s_ty[NF_MAX];
    __shared__ uint32_t s_tc[NF_MAX];
    CandLds C;
    C.lbuf = lbuf, C.tx = s_tx, C.ty = s_ty, C.tc = s_tc, C.w2c = w2c;
    // the points [0, early_done) were projected and listed by k_early_map while the previous frame finished
    candidates_body<MODE_MAP, true>(S, 0, par, C, blockIdx.x * 4 + wave_id(), gridDim.x * 4, 256, (ctl.early_ran_seq == seq) ? max(ctl.early_done, 0) : 0, -1);
}

// k_early_map : projection + candidate lists of the map points that survived the PREVIOUS frame's clean-up, launched on its own
// stream as soon as that frame's pose is known (after its k_pnp) -- k_triangulate of that frame (staged update, triangulation) only appends points
// behind early_done.  Nothing of the per-frame state is published here (k_match_map does that when the previous frame is
// complete); the prediction is recomputed from the same persistent inputs, so both kernels see the same pose.
// k_gate : one wavefront per sequence at the head of the early stream; returns when the previous frame's k_pnp has published
// its sequence number (or after 20 ms of wall clock: then the early kernels stand down and the late ones do all the work)
// head of the feature stage: returns when the tracking chain of the frame that used this feature buffer last (NPAR frames ago)
// is finished.  Polled: a feature stream parked on an event barrier stalls the queues that share its hardware pipe -- with
// the barrier, SHORTENING the feature chain made the whole pipeline slower -- and the event record costs the tracking stream
// 3-4 us per frame.  (LVT_AMD_ORDERING=events uses the barrier instead of this kernel.)
// the wait of k_gate_buf for one sequence (one thread): the tracking chain of frame `want` has released feature buffer `par`
__device__ __forceinline__ void gate_buf_wait(const Seq &S, seq_t want, int par) {
    Ctl &ctl = *S.ctl;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(&ctl.track_done_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 200000000ull) {  // 2 s: the tracking stream is wedged, or a tool serialises the dispatches and this
            // kernel holds the slot.  The buffer still belongs to an older frame: this frame's feature kernels leave it alone
            // (FeatCtl::poison), k_brief publishes "no features" (skip_seq) and the tracking chain SKIPS the frame -- last pose
            // returned, state kept, reported through lvt_amd_last_error.  Never LOST: that state is sticky and, in the reference,
            // a matter of match counts only.
            S.fb[par].fc->poison = 1;
            atomicAdd(&ctl.gate_fatal, 1);
            __threadfence();
            break;
        }
    }
}
__global__ __launch_bounds__(64) void k_gate_buf(Seq *seqs, seq_t want, int par) {
    if (threadIdx.x != 0) return;
    gate_buf_wait(seq_const(seqs, blockIdx.z), want, par);
}

// last kernel of the feature stage of a batch (one thread per sequence): this buffer's features are complete.  (A single sequence lets
// k_brief's last workgroup publish instead -- one launch less on its longest chain.  With a batch's 2048 workgroups that cost 60 us:
// every workgroup's release fence is an L2 write-back.)
__global__ void k_feat_done(Seq *seqs, int par, seq_t seq) {
    if (threadIdx.x != 0) return;
    FeatCtl &fc = *seq_const(seqs, blockIdx.x).fb[par].fc;
    seq_const(seqs, blockIdx.x).ctl->dbg[47] = (long long)wall_clock64();  // (written from the feature stream: the frame it belongs to may differ)
    if (fc.poison) {  // the frame has no features (k_gate_buf timed out): say so to the gates of the other streams, then release the flag
        fc.skip_seq = seq;
        fc.poison = 0;
    }
    __threadfence();
    atomicExch(&fc.feat_seq, seq);
}

// behind k_candidates<ROW>: this frame's row-match candidate lists are complete (k_triangulate's head polls the word)
__global__ void k_row_done(Seq *seqs, int par, seq_t seq) {
    if (threadIdx.x != 0) return;
    FeatCtl &fc = *seq_const(seqs, blockIdx.x).fb[par].fc;
    if (__hip_atomic_load(&fc.feat_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) return;  // (see k_candidates)
    __threadfence();
    atomicExch(&fc.row_seq, seq);
}

template <bool BV>
__global__ __launch_bounds__(64) void k_gate(SeqArg<BV> sa, int par, seq_t want, seq_t seq, int test_timeout) {
    if (threadIdx.x == 0 && blockIdx.x == 0) sa.get().ctl->dbg[32] = (long long)wall_clock64();
    Ctl &ctl = *sa.get().ctl;
    FeatCtl &fc = *sa.get().fb[par].fc;
    if (threadIdx.x != 0) return;
    bool ok = true;
    if (test_timeout) {  // (tests: behave as if the wait had run into its limit -- LVT_AMD_TEST_GATE_TIMEOUT)
        atomicAdd(&ctl.gate_timeouts, 1);
        ok = false;
    } else {
        // both conditions are polled: a stream parked on an event barrier stalls the other queues of its hardware pipe until
        // a time slice expires (the feature stream did not advance while this stream waited for the NEXT frame's features).
        // After 20 ms of wall clock (100 MHz) on either, the early kernels stand down and the late ones do all the work --
        // that happens when a tool serialises the dispatches of all queues (rocprofv3 --pmc): this kernel then holds the only
        // dispatch slot and what it waits for cannot start.
        const unsigned long long t0 = wall_clock64();
        // (a) this frame's features
        while (__hip_atomic_load(&fc.feat_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 2000000ull) {
                atomicAdd(&ctl.gate_timeouts, 1);
                ok = false;
                break;
            }
        }
        if (ok && __hip_atomic_load(&fc.skip_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == seq) ok = false;  // published, but empty (k_gate_buf)
        // (b) the previous frame's k_pnp
        const unsigned long long t1 = wall_clock64();
        ctl.dbg[35] = (long long)t1;  // (tools/timeline.py: when this frame's features were seen)
        while (ok && __hip_atomic_load(&ctl.pnp_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t1 > 2000000ull) {
                atomicAdd(&ctl.gate_timeouts, 1);
                ok = false;
                break;
            }
        }
    }
    // claim the frame (or record that nothing will be done); if the tracking stream has cancelled it meanwhile, stand down
    const seq_t mine = 4u * seq + (ok ? 1u : 3u);
    seq_t cur = __hip_atomic_load(&ctl.early_state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (cur >= 4u * seq + 1u) {
            ok = false;
            break;
        }
        const seq_t seen = atomicCAS(&ctl.early_state, cur, mine);
        if (seen == cur) break;
        cur = seen;
    }
    ctl.gate_ok = ok ? seq : (seq_t)0;
    ctl.dbg[33] = (long long)wall_clock64();
}

template <bool BV>
__global__ __launch_bounds__(256) void k_early_map(SeqArg<BV> sa, int par, seq_t seq, int only_fb) {
    if (threadIdx.x == 0 && blockIdx.x == 0) sa.get().ctl->dbg[34] = (long long)wall_clock64();
    const Seq &S = sa.get();
    if (only_fb && !S.lists_fb[0]) return;  // (k_hamming_batched_lists<MODE_MAP> has projected and listed these points)
    Ctl &ctl = *S.ctl;
    const int n_early = (ctl.gate_ok == seq) ? ctl.early_done : 0;
    if (n_early <= 0) return;  // first frame, LOST, the previous frame did not reach its pose refinement, or the gate timed out
    __shared__ double w2c[12];
    if (threadIdx.x == 0) {
        Pose predicted;
        double mmn[14];
        motion_predict(ctl, ctl.last_pose, predicted, mmn);
        world_to_camera(predicted, w2c);
    }
    __shared__ uint32_t lbuf[4 * KC];
    __shared__ float s_tx[NF_MAX], s_ty[NF_MAX];
    __shared__ uint32_t s_tc[NF_MAX];
    CandLds C;
    C.lbuf = lbuf, C.tx = s_tx, C.ty = s_ty, C.tc = s_tc, C.w2c = w2c;
    candidates_body<MODE_MAP, true>(S, 0, par, C, blockIdx.x * 4 + wave_id(), gridDim.x * 4, 256, 0, n_early);
}

// =================================================================================================
// exact slow path: best two unmatched candidates of one query by scanning all train features
// =================================================================================================
template <int MODE>
__device__ void wave_top2_slow(const Query &q, const Feat &T, int N, const uint8_t *flag_lds, uint32_t &k1, uint32_t &k2, int &cnt) {
    const int lane = lane_id();
    uint32_t b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;
    int c = 0;
    for (int base = 0; base < N; base += 64) {
        const int j = base + lane;
        if (j < N && !flag_lds[j] && cand_pred<MODE>(q, T, j)) {
            uint64_t t[4];
#pragma unroll
            for (int k = 0; k < 4; k++) t[k] = T.desc[(size_t)j * 4 + k];
            const uint32_t key = ((uint32_t)hamming256(q.d, t) << 16) | (uint32_t)j;
            if (key < b1) {
                b2 = b1;
                b1 = key;
            } else if (key < b2)
                b2 = key;
            c++;
        }
    }
    // wave reduce: smallest and second smallest of all lanes' (b1,b2)
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o1 = __shfl_xor(b1, d, 64), o2 = __shfl_xor(b2, d, 64);
        const int oc = __shfl_xor(c, d, 64);
        const uint32_t n1 = min(b1, o1);
        const uint32_t n2 = min(max(b1, o1), min(b2, o2));
        b1 = n1;
        b2 = n2;
        c += oc;
    }
    k1 = b1;
    k2 = b2;
    cnt = c;
}

// accept test of find_match_index / row_match (lvt_image_features_struct.cpp:105-116,141-144)
__device__ __forceinline__ bool accept_match(int cnt, uint32_t k1, uint32_t k2, float ratio_th, float desc_th) {
    if (cnt > 1) {
        const float d1 = (float)(k1 >> 16), d2 = (float)(k2 >> 16);
        return (d1 / d2) < ratio_th;  // 0/0 = NaN -> false
    }
    if (cnt == 1) return (float)(k1 >> 16) <= desc_th;
    return false;
}

// first two unmatched candidates of a sorted list held as buf[0..n) (LDS or registers via lambda)
// executed by a full wavefront; KC <= 128 => two candidates per lane
__device__ __forceinline__ void wave_first_two(uint32_t c0, uint32_t c1, bool v0, bool v1, uint32_t &k1, uint32_t &k2, int &cnt) {
    const uint64_t m0 = __ballot(v0), m1 = __ballot(v1);
    cnt = __popcll(m0) + __popcll(m1);
    int s1 = -1, s2 = -1;  // slot index 0..127
    uint64_t a = m0, b = m1;
    if (a) {
        s1 = __ffsll((long long)a) - 1;
        a &= a - 1;
    } else if (b) {
        s1 = 64 + __ffsll((long long)b) - 1;
        b &= b - 1;
    }
    if (a) s2 = __ffsll((long long)a) - 1;
    else if (b) s2 = 64 + __ffsll((long long)b) - 1;
    k1 = k2 = 0xFFFFFFFFu;
    if (s1 >= 0) k1 = (s1 < 64) ? __shfl(c0, s1, 64) : __shfl(c1, s1 - 64, 64);
    if (s2 >= 0) k2 = (s2 < 64) ? __shfl(c0, s2, 64) : __shfl(c1, s2 - 64, 64);
}

// =================================================================================================
// The greedy, order-dependent accept/mark scans (find_matches, row_match, update_staged_map_points) as a
// BLOCK-WIDE FIXPOINT ITERATION.
//
// Sequentially, query q takes the best two still-unmarked candidates of its list, applies the ratio / threshold
// test and marks the winner; later queries see that mark.  Here all queries of a "super-chunk" decide at once:
// in iteration t a candidate counts as unavailable when it is marked, or when a query with a SMALLER index
// accepted it in iteration t-1 (tab[(t-1)&1][f] = earliest such query, stamped with t-1 so the tables are never
// cleared).  Query 0 is final after one iteration, query q after at most q+1.  When an iteration changes no
// decision, every decision satisfies D_q = f(marks, {D_j : j < q}) -- the defining recurrence of the sequential
// scan (lvt_local_map.cpp:146-199, lvt_image_features_handler.cpp:302-323, lvt_local_map.cpp:355-391), whose
// solution is unique.  Observed depth: 2-4 iterations.  Candidate lists (already sorted by (distance, index) by
// k_candidates) are packed into LDS once per super-chunk.
// =================================================================================================
constexpr int RES_THREADS = 1024;
constexpr int QCAP = 2 * RES_THREADS;  // queries per super-chunk (two per thread, contiguous)
constexpr int LCAP = 24576;            // packed candidate entries per super-chunk (96 KB of LDS)

struct ResolveLds {
    uint8_t *flag;     // [NF_MAX] matched marks
    uint32_t *tab0;    // [NF_MAX] claims of odd iterations
    uint32_t *tab1;    // [NF_MAX] claims of even iterations
    uint32_t *lists;   // [LCAP]
    int *scan;         // [32]
    int *misc;         // [8]
};

// exact slow path for ONE query whose list overflowed KC; executed by wavefront 0 (all lanes)
template <int MODE>
__device__ __forceinline__ int decide_query_slow(const Seq &S, const Feat &T, int N, const uint8_t *flag, const uint64_t *qdesc, int i,
                                                 float qx, float qy, int radius, float ratio, float desc_th) {
    Query q;
    if (MODE == MODE_ROW) make_query_row(S.prm, qx, qy, q);
    else make_query_track(S.prm, qx, qy, radius, q);
#pragma unroll
    for (int k = 0; k < 4; k++) q.d[k] = qdesc[(size_t)i * 4 + k];
    uint32_t k1, k2;
    int cnt;
    wave_top2_slow<MODE>(q, T, N, flag, k1, k2, cnt);
    return accept_match(cnt, k1, k2, ratio, desc_th) ? (int)(k1 & 0xFFFFu) : -1;
}

// One super-chunk starting at query b0.  nfn(q) = candidate count of query q (0 = nothing to decide).
// Returns the number of queries consumed (>= 1), or -1 when query b0 itself overflows KC (caller: slow path).
// Thread t owns local queries t and t + RES_THREADS; acc[u] = accepted feature (or -1).  Marks: a matched feature
// carries the permanent stamp PERM in BOTH claim tables (one LDS read per candidate tells "marked or claimed");
// L.flag mirrors them for the write-back.
constexpr uint32_t PERM = 0xFFFFFFFFu;

template <class NFn>
__device__ __forceinline__ int resolve_super(int b0, int M, NFn nfn, const uint32_t *cand, ResolveLds &L, uint32_t &iter, float ratio,
                                             float desc_th, int acc[2], int nq[2] = nullptr, const uint32_t *qidx = nullptr, int qid[2] = nullptr,
                                             bool no_marks = false) {
    // no_marks (block-uniform): no feature carries a permanent mark yet -- the first super-chunk of a call that starts from cleared marks -- so the
    // packing below has nothing to drop and skips its 32 table look-ups per query (they were half of the packing's time)
    // qidx != nullptr: b0 / M / the return value count entries of the COMPACTED query list qidx[] (query | candidate count << 24, in
    // query order -- the greedy scan only ever decides queries that have candidates, and their relative order is all it depends on)
    const int tid = threadIdx.x;
    int n[2], off[2], qq[2];
    const long long ts0 = clock64();
    if (tid == 0) L.misc[0] = QCAP;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int lq = tid + u * RES_THREADS, q = b0 + lq;
        qq[u] = q;
        if (qidx) {
            const uint32_t w = (q < M) ? qidx[q] : 0u;
            qq[u] = (int)(w & 0xFFFFFFu);
            n[u] = (int)(w >> 24);  // (capped at 255 > KC: still "overflow")
        } else
            n[u] = (q < M) ? nfn(q) : 0;
        if (n[u] > KC) atomicMin(&L.misc[0], lq);
    }
    if (qid) qid[0] = qq[0], qid[1] = qq[1];
    __syncthreads();
    int limit = min(min(QCAP, M - b0), L.misc[0]);
    if (limit == 0) return -1;
#pragma unroll
    for (int u = 0; u < 2; u++)
        if (tid + u * RES_THREADS >= limit) n[u] = 0;
    // every list starts on a 16-byte boundary of L.lists (lengths rounded up to 4 entries): the walk reads 4 candidates
    // with one ds_read_b128
    const int n4[2] = {(n[0] + 3) & ~3, (n[1] + 3) & ~3};
    int t0, t1;
    off[0] = block_excl_scan(n4[0], L.scan, &t0);
    if (limit > RES_THREADS) {
        off[1] = t0 + block_excl_scan(n4[1], L.scan, &t1);
    } else {
        off[1] = t0;
        t1 = 0;
    }
    if (t0 + t1 > LCAP) {  // keep the longest prefix of queries whose lists fit
        int f0 = (tid < limit && off[0] + n4[0] <= LCAP) ? 1 : 0, f1 = (tid + RES_THREADS < limit && off[1] + n4[1] <= LCAP) ? 1 : 0;
        int c0, c1;
        block_excl_scan(f0, L.scan, &c0);
        block_excl_scan(f1, L.scan, &c1);
        limit = (c0 < min(limit, RES_THREADS)) ? c0 : c0 + c1;   // fits are a prefix in query order
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (tid + u * RES_THREADS >= limit) n[u] = 0;
    }
    const long long ts1 = clock64();
    // pack the lists into LDS: 16-byte loads, 32 entries (8 loads) in flight per query -- one memory round trip for
    // the usual list of < 32 candidates (rows of `cand` are KC words apart, so every row start is 16-byte aligned).
    // Candidates that already carry a permanent mark (matched by an earlier super-chunk, or before this call) are dropped here: they
    // stay unavailable for the rest of the call, and on a large map most entries of the later super-chunks are such (TUM: 8000 queries
    // compete for 1000 features) -- every fixpoint iteration would walk over them again.
    if (nq) nq[0] = n[0], nq[1] = n[1];  // (original list lengths of this thread's queries: the caller need not fetch them again)
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint4 *src = reinterpret_cast<const uint4 *>(cand + (size_t)qq[u] * KC);
        uint32_t *dst = L.lists + off[u];
        int w = 0;
        for (int e0 = 0; e0 < n[u]; e0 += 32) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = (e0 + 4 * k < n[u]) ? src[(e0 >> 2) + k] : make_uint4(0, 0, 0, 0);
            if (no_marks) {
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (e0 + 4 * k < n[u]) *reinterpret_cast<uint4 *>(dst + e0 + 4 * k) = v[k];  // (list starts are 16-byte aligned; the padding is never evaluated)
                w = n[u];
                continue;
            }
            // (groups of four entries past the list's end are skipped: the usual list holds ~13 entries, and 32 table look-ups + 32 predicated
            //  stores per query whatever its length were 13k cycles of every super-chunk on a TUM-shaped map)
            uint32_t mk[32];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (e0 + 4 * k < n[u]) {
                    mk[4 * k] = L.tab0[v[k].x & (uint32_t)(NF_MAX - 1)];
                    mk[4 * k + 1] = L.tab0[v[k].y & (uint32_t)(NF_MAX - 1)];
                    mk[4 * k + 2] = L.tab0[v[k].z & (uint32_t)(NF_MAX - 1)];
                    mk[4 * k + 3] = L.tab0[v[k].w & (uint32_t)(NF_MAX - 1)];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = e0 + 4 * k;
                if (e < n[u]) {
                    if (mk[4 * k] != PERM) dst[w++] = v[k].x;
                    if (e + 1 < n[u] && mk[4 * k + 1] != PERM) dst[w++] = v[k].y;
                    if (e + 2 < n[u] && mk[4 * k + 2] != PERM) dst[w++] = v[k].z;
                    if (e + 3 < n[u] && mk[4 * k + 3] != PERM) dst[w++] = v[k].w;
                }
            }
        }
        n[u] = w;
    }
    if (tid == 0) L.misc[12] = L.misc[13] = QCAP;  // first changed query of an iteration (two slots, by iteration parity)
    __syncthreads();
    int prev[2] = {-2, -2};
    acc[0] = acc[1] = -1;
    int dbg_steps = 0;
    const long long tj0 = clock64();
    const uint32_t it0 = iter;
    // queries [0, fin) of this super-chunk are FINAL: none of them, nor any query before them, changed its decision in the last
    // iteration, so their inputs -- the decisions of earlier queries -- will never change again.  They stop walking; the features
    // they accepted get their permanent marks at once (a permanent mark hides a feature from every query, a claim only from the later
    // ones -- and only later ones are still deciding).
    int fin = 0;
    while (true) {
        iter++;
        const uint32_t *rd = (iter & 1u) ? L.tab0 : L.tab1;  // written during iteration iter-1
        uint32_t *wr = (iter & 1u) ? L.tab1 : L.tab0;
        int first_changed = QCAP;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int lq = tid + u * RES_THREADS;
            if (n[u] == 0 || lq < fin) continue;
            const uint32_t *list = L.lists + off[u];
            const uint32_t mine = ((iter - 1u) << 11) | (uint32_t)(2047 - lq);
            uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
            int cnt = 0;
            // 4 candidates per step: the 4 list reads and then the 4 claim reads are independent, so their LDS latencies
            // overlap; evaluation stays in list order
            for (int e = 0; e < n[u] && cnt < 2; e += 4) {
                uint32_t c[4], v[4];
                {
                    const uint4 c4 = *reinterpret_cast<const uint4 *>(list + e);  // entries past n[u] are padding (never evaluated)
                    c[0] = c4.x, c[1] = c4.y, c[2] = c4.z, c[3] = c4.w;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = rd[c[k] & (uint32_t)(NF_MAX - 1)];  // valid indices are < NF_MAX; padding stays in range
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // unavailable: marked (PERM), or accepted in iteration iter-1 by an earlier query.  ONE comparison: a claim of iteration
                    // iter-1 by query q' reads (iter-1) << 11 | 2047 - q', which exceeds `mine` exactly for q' < lq; claims left over from older
                    // iterations are below (iter-1) << 11, PERM is the largest word there is
                    const bool un = v[k] > mine;
                    if (e + k < n[u] && !un && cnt < 2) {
                        if (cnt == 0) k1 = c[k];
                        else k2 = c[k];
                        cnt++;
                    }
                }
            }
            const bool ok = accept_match(cnt, k1, k2, ratio, desc_th);
            dbg_steps = max(dbg_steps, n[u]);
            acc[u] = ok ? (int)(k1 & 0xFFFFu) : -1;
            if (ok) atomicMax(&wr[acc[u]], (iter << 11) | (uint32_t)(2047 - lq));
            if (acc[u] != prev[u]) first_changed = min(first_changed, lq);
            prev[u] = acc[u];
        }
        int *slot = &L.misc[12 + (iter & 1u)];
        if (first_changed < QCAP) atomicMin(slot, first_changed);
        __syncthreads();
        const int fc = *slot;
        if (fc >= QCAP) break;  // nothing changed: every decision is final
        if (tid == 0) L.misc[12 + ((iter + 1u) & 1u)] = QCAP;  // the next iteration's slot (nobody touches it before the barrier below)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int lq = tid + u * RES_THREADS;
            if (lq >= fin && lq < fc && n[u] > 0 && acc[u] >= 0) {
                L.flag[acc[u]] = 1;
                L.tab0[acc[u]] = PERM;
                L.tab1[acc[u]] = PERM;
            }
        }
        fin = fc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        L.misc[4] = (int)(iter - it0);
        L.misc[5] = (int)(clock64() - tj0);
        L.misc[6] = (int)(ts1 - ts0);
        L.misc[3] = 0;
        L.misc[7] = (int)(tj0 - ts1);
    }
    __syncthreads();
    {   // (bring-up statistic: the longest list of the batch.  One LDS atomic per wavefront -- 1024 of them on one address cost
        //  ~4k cycles of every resolver call)
        int m = dbg_steps;
        for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
        if (lane_id() == 0) atomicMax(&L.misc[3], m);
    }
#pragma unroll
    for (int u = 0; u < 2; u++)
        if (acc[u] >= 0) {
            L.flag[acc[u]] = 1;
            L.tab0[acc[u]] = PERM;
            L.tab1[acc[u]] = PERM;
        }
    __syncthreads();
    return limit;
}

#define RESOLVE_LDS_DECL                                    \
    __shared__ uint8_t r_flag[NF_MAX];                      \
    __shared__ uint32_t r_tab[2 * NF_MAX];                  \
    __shared__ __attribute__((aligned(16))) uint32_t r_lists[LCAP]; \
    __shared__ int r_scan[32];                              \
    __shared__ int r_misc[16];  /* [12], [13]: first changed query of the odd / even fixpoint iterations */                              \
    ResolveLds L;                                           \
    L.flag = r_flag, L.tab0 = r_tab, L.tab1 = r_tab + NF_MAX, L.lists = r_lists, L.scan = r_scan, L.misc = r_misc;

template <int MODE>
// phase 0: all queries; phase 1 (map mode, "early"): queries [0, q_split) only, no per-frame bookkeeping; phase 2 ("late"): queries
// [q_split, M), continuing from the marks and the match count phase 1 left behind
__device__ __forceinline__ void resolve_body(const Seq &S, Ctl &ctl, int pass2, int par, ResolveLds &L, uint32_t *r_tab, int phase = 0, int q_split = 0) {
    const int tid = threadIdx.x;
    const long long tk0 = clock64();
    const Feat &T = (MODE == MODE_ROW) ? S.fb[par].feat[1] : S.fb[par].feat[0];
    const int N = *T.n;
    // Both users start from cleared marks: find_matches works on the marks of a fresh frame (k_gather zeroes them; pass 2
    // clears them again, lvt_local_map.cpp:176) and row_match on those of the right image, which nothing else marks --
    // so the tables are initialised without reading the global flags (one memory round trip less at kernel start).
    for (int j = tid; j < NF_MAX; j += RES_THREADS) {
        const uint8_t f = (phase == 2 && j < N) ? T.flag[j] : 0;  // late part: the marks the early part wrote back
        L.flag[j] = f;
        r_tab[j] = r_tab[NF_MAX + j] = f ? PERM : 0u;
    }
    int M;
    const uint64_t *qdesc;
    const uint32_t *cand;
    const int *ncand;
    if (MODE == MODE_MAP) {
        M = *S.map_n;
        qdesc = S.map[*S.map_cur].desc;
        cand = S.cand;
        ncand = S.ncand;
    } else {
        M = *S.fb[par].feat[0].n;
        qdesc = S.fb[par].feat[0].desc;
        cand = S.rcand + (size_t)par * NF_MAX * KC;
        ncand = S.rncand + par * NF_MAX;
    }
    const float ratio = (MODE == MODE_ROW) ? S.prm.tri_ratio : S.prm.track_ratio;
    const float desc_th = S.prm.desc_th;
    const int radius = S.prm.tracking_radius * ((MODE == MODE_MAP && pass2) ? 2 : 1);
    uint32_t iter = 0;
    int accepted = (phase == 2) ? ctl.early_accepted : 0;  // block-uniform
    if (phase == 1) M = min(M, q_split);
    int b0 = (phase == 2) ? q_split : 0;
    // A map much larger than a super-chunk (TUM-shaped sequences: 9000 points, 800 of them visible with candidates) would walk
    // through five mostly empty super-chunks: the points that have candidates are compacted first, in storage order, and the
    // super-chunks run over that list -- the scan's decisions depend on nothing else.
    const uint32_t *qidx = nullptr;
    if (MODE == MODE_MAP && M - b0 > QCAP) {
        // coalesced chunks of 1024 points (point k 1024 + t belongs to thread t), a wavefront's ballot per chunk, ONE block scan over the (chunk,
        // wavefront) counts (a contiguous run of points per thread reads 64 cache lines per load instruction: 17k cycles on a 9 000-point map)
        constexpr int NWV = RES_THREADS / 64;
        const int lane = tid & 63, wv = tid >> 6;
        const int nch = (M - b0 + RES_THREADS - 1) / RES_THREADS;  // <= 32
        uint32_t *cnts = L.lists;  // (the list area is free until the first super-chunk packs)
        const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        uint32_t hasbits = 0;
        for (int k = 0; k < nch; k++) {
            const int i = b0 + k * RES_THREADS + tid;
            const bool has = (i < M) && (ncand[i] > 0);
            hasbits |= (has ? 1u : 0u) << k;
            const uint64_t bm = __ballot(has);
            if (lane == 0) cnts[k * NWV + wv] = (uint32_t)__popcll(bm);
        }
        __syncthreads();
        int total;
        {
            const int v = (tid < nch * NWV) ? (int)cnts[tid] : 0;
            const int ex = block_excl_scan(v, L.scan, &total);
            if (tid < nch * NWV) cnts[tid] = (uint32_t)ex;
        }
        __syncthreads();
        for (int k = 0; k < nch; k++) {
            const int i = b0 + k * RES_THREADS + tid;
            const bool has = (hasbits >> k) & 1u;
            const uint64_t bm = __ballot(has);
            if (has) S.qidx[cnts[k * NWV + wv] + __popcll(bm & lt)] = (uint32_t)i | ((uint32_t)min(ncand[i], 255) << 24);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        qidx = S.qidx;
        b0 = 0;
        M = total;
    }
    __syncthreads();
    if (qidx) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const long long tk1 = clock64();
    int st_chunks = 0, st_slow = 0, st_iter = 0, st_itmax = 0, st_fix = 0, st_cs = 0, st_pack = 0;  // bring-up statistics (thread 0)
    for (; b0 < M;) {
        int acc[2], nq[2] = {0, 0}, qid[2];
        const uint8_t *lflag = S.fb[par].feat[0].flag;
        const int used = resolve_super(b0, M, [&](int q) { return (MODE == MODE_ROW && lflag[q]) ? 0 : ncand[q]; }, cand, L, iter, ratio, desc_th, acc, nq, qidx, qid,
                                       /* no_marks = */ phase != 2 && iter == 0);
        if (used < 0) st_slow++;
        else if (tid == 0) st_chunks++, st_iter += L.misc[4], st_itmax = max(st_itmax, L.misc[4]), st_fix += L.misc[5], st_cs += L.misc[6], st_pack += L.misc[7];
        if (used < 0) {  // query b0 overflowed KC: exact scan of all train features by wavefront 0
            const int qb = qidx ? (int)(qidx[b0] & 0xFFFFFFu) : b0;
            if (wave_id() == 0) {
                float qx, qy;
                if (MODE == MODE_MAP) qx = S.proj[2 * qb], qy = S.proj[2 * qb + 1];
                else qx = S.fb[par].feat[0].x[qb], qy = S.fb[par].feat[0].y[qb];
                const int idx = decide_query_slow<MODE>(S, T, N, L.flag, qdesc, qb, qx, qy, radius, ratio, desc_th);
                if (lane_id() == 0) {
                    if (MODE == MODE_MAP) S.match[qb] = idx;
                    if (idx >= 0) {
                        L.flag[idx] = 1;
                        L.tab0[idx] = L.tab1[idx] = PERM;
                        if (MODE == MODE_ROW) {
                            S.pair_l[accepted] = qb;
                            S.pair_r[accepted] = idx;
                            S.fb[par].feat[0].flag[qb] = 1;
                        }
                    }
                    L.misc[1] = (idx >= 0) ? 1 : 0;
                }
            }
            __syncthreads();
            accepted += L.misc[1];
            __syncthreads();
            b0 += 1;
            continue;
        }
        // results, in query order (local query = tid + u * RES_THREADS)
        const int a0 = (tid < used && acc[0] >= 0) ? 1 : 0, a1 = (tid + RES_THREADS < used && acc[1] >= 0) ? 1 : 0;
        int tot0, tot1 = 0, ex0 = 0, ex1 = 0;
        if (MODE == MODE_MAP) {  // only the number of matches is needed
            tot0 = __syncthreads_count(a0);
            if (used > RES_THREADS) tot1 = __syncthreads_count(a1);
        } else {
            ex0 = block_excl_scan(a0, L.scan, &tot0);
            if (used > RES_THREADS) ex1 = block_excl_scan(a1, L.scan, &tot1);
        }
        const int tot = tot0 + tot1;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int lq = tid + u * RES_THREADS, q = qid[u];
            if (lq >= used) continue;
            if (MODE == MODE_MAP) {
                if (nq[u] > 0) S.match[q] = acc[u];  // invisible / candidate-less points keep -2 / -1 (k_match_map / k_early_map)
            } else if (acc[u] >= 0) {
                const int slot = accepted + (u ? tot0 + ex1 : ex0);
                S.pair_l[slot] = q;
                S.pair_r[slot] = acc[u];
                S.fb[par].feat[0].flag[q] = 1;  // handler.cpp:319
            }
        }
        accepted += tot;
        b0 += used;
        __syncthreads();
    }
    __syncthreads();
    if (MODE == MODE_MAP) {
        // a failed pass 1 (< 50) is discarded: pass 2 rebuilds the marks from scratch
        for (int j = tid; j < N; j += RES_THREADS) S.fb[par].feat[0].flag[j] = L.flag[j];
        if (tid == 0 && phase == 1) {
            ctl.early_accepted = accepted;
            // bring-up stamps of the resolver (tools/cells_phases.py): the early part is where most of find_matches is resolved
            ctl.dbg[18] = st_iter;   // sums over the super-chunks of this call
            ctl.dbg[19] = st_fix;
            ctl.dbg[28] = L.misc[3];
            ctl.dbg[24] = st_chunks, ctl.dbg[25] = st_slow, ctl.dbg[26] = st_itmax, ctl.dbg[27] = M;
            ctl.dbg[29] = st_cs;
            ctl.dbg[30] = st_pack;
            ctl.dbg[31] = tk1 - tk0;
            ctl.dbg[23] = clock64() - tk0;
        }
        if (tid == 0 && phase != 1) {
            if (!pass2) {
                ctl.n_pass1 = accepted;
                ctl.do_pass2 = (accepted < N_MATCHES_TH) ? 1 : 0;  // lvt_local_map.cpp:173
                L.misc[2] = ctl.do_pass2;
                ctl.counts[C_SECOND_PASS] = ctl.do_pass2;
            } else
                ctl.n_pass2 = accepted;
        }
    } else {
        for (int j = tid; j < N; j += RES_THREADS) S.fb[par].feat[1].flag[j] = L.flag[j];
        if (tid == 0) {
            ctl.n_pairs = accepted;
            ctl.counts[C_N_ROW_MATCHES] = accepted;
            L.misc[2] = accepted;
        }
    }
}

// =================================================================================================
// bookkeeping of find_matches + LOST decision + clean_untracked_points : lvt_local_map.cpp:201-224, lvt_system.cpp:266-274, lvt_local_map.cpp:393-413
// (the clean-up does not depend on the pose, so it runs here, ahead of the pose refinement; the reference runs it right after)
// =================================================================================================
// bookkeeping + LOST decision + clean_untracked_points in ONE pass for maps of up to RES_THREADS points (the map of a
// KITTI sequence holds ~800): every field of a point is loaded once (independent loads: one memory round trip), both
// compactions (PnP input = matched points, surviving map = counter < untracked_th) come from one packed scan, and the
// survivors go straight to the other map buffer.  Returns true when the frame is LOST (block-uniform).
__device__ __forceinline__ bool bookkeep_cull_small(const Seq &S, Ctl &ctl, int par, int *scan) {
    const int tid = threadIdx.x;
    const int cur = *S.map_cur, M = *S.map_n;
    const MapSoA &A = S.map[cur], &B = S.map[cur ^ 1];
    const Feat &F = S.fb[par].feat[0];
    const int th = S.prm.untracked_th;
    const int i = tid;
    int m = -2, cnt = 0, ag = 0;
    double pos[3] = {0, 0, 0};
    uint64_t dsc[4] = {0, 0, 0, 0};
    if (i < M) {
        m = S.match[i];
        cnt = A.counter[i];
        ag = A.age[i];
        pos[0] = A.pos[3 * i], pos[1] = A.pos[3 * i + 1], pos[2] = A.pos[3 * i + 2];
#pragma unroll
        for (int k = 0; k < 4; k++) dsc[k] = A.desc[(size_t)i * 4 + k];
        if (m == -1) cnt += 1;       // lvt_local_map.cpp:201-224
        else if (m >= 0) ag += 1;
    }
    float ox = 0.f, oy = 0.f;
    if (m >= 0) ox = F.x[m], oy = F.y[m];
    const bool keep = (i < M) && (cnt < th);
    int total;
    const int ex = block_excl_scan((m >= 0 ? 1 : 0) | (keep ? 0x10000 : 0), scan, &total);
    const int n_match = total & 0xFFFF, n_keep = total >> 16;
    const int offm = ex & 0xFFFF, offk = ex >> 16;
    if (m >= 0 && offm < NF_MAX) {
        S.pnp_X[3 * offm] = pos[0];
        S.pnp_X[3 * offm + 1] = pos[1];
        S.pnp_X[3 * offm + 2] = pos[2];
        S.pnp_obs[2 * offm] = ox;
        S.pnp_obs[2 * offm + 1] = oy;
        S.pnp_feat[offm] = m;
        S.pnp_level[offm] = 0;
    }
    const bool lost = n_match < S.prm.min_matches;  // lvt_system.cpp:267-272, :199-204
    if (tid == 0) {
        ctl.n_matches = n_match;
        ctl.counts[C_N_MATCHES] = n_match;
        if (lost) {
            ctl.lost_now = 1;
            ctl.state = 3;
            pose_to_Rt(ctl.last_pose, ctl.out_R, ctl.out_t);
            ctl.out_status = 3;
        } else {  // push_back / pop_front
            ctl.last_matches[0] = ctl.last_matches[1];
            ctl.last_matches[1] = ctl.last_matches[2];
            ctl.last_matches[2] = n_match;
        }
    }
    if (lost) {  // the map stays where it is, with the bookkeeping applied
        if (i < M) A.match_idx[i] = m, A.counter[i] = cnt, A.age[i] = ag;
        return true;
    }
    // clean_untracked_points (lvt_local_map.cpp:393-413): stable compaction into the other buffer
    if (keep) {
        B.pos[3 * offk] = pos[0], B.pos[3 * offk + 1] = pos[1], B.pos[3 * offk + 2] = pos[2];
#pragma unroll
        for (int k = 0; k < 4; k++) B.desc[(size_t)offk * 4 + k] = dsc[k];
        B.counter[offk] = cnt;
        B.age[offk] = ag;
        B.match_idx[offk] = m;
    } else if (i < M && m >= 0)
        S.fb[par].feat[0].flag[m] = 0;  // :402-405
    if (tid == 0) {
        ctl.counts[C_N_CULLED] = M - n_keep;
        *S.map_cur = cur ^ 1;
        *S.map_n = n_keep;
    }
    return false;
}

// The same for maps of any size (a TUM-shaped map holds ~9 000 points).  bookkeep_body + cull_body of rounds 1-4 scanned every chunk of 1024 points on
// its own: 2 x ceil(M / 1024) block scans, each behind its own memory round trip (57 us of k_track_mid on such a map).  Here the chunks keep their coalesced
// layout (point k 1024 + t belongs to thread t), pass 1 only COUNTS -- a wavefront's ballots give its matched / surviving points per chunk, no barrier between
// chunks --, ONE block scan over the (chunk, wavefront) counts places both compactions in storage order, and pass 2 reads every field once and writes the PnP
// input and the surviving map, again without a barrier between chunks.  `cnts`: >= 2 * 32 * 16 words of LDS scratch.  Returns true when the frame is LOST.
__device__ __forceinline__ bool bookkeep_cull_large(const Seq &S, Ctl &ctl, int par, int *scan, uint32_t *cnts) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NWV = RES_THREADS / 64;
    const int cur = *S.map_cur, M = *S.map_n;
    const MapSoA &A = S.map[cur], &B = S.map[cur ^ 1];
    const Feat &F = S.fb[par].feat[0];
    const int th = S.prm.untracked_th;
    const int c = (M + RES_THREADS - 1) / RES_THREADS;  // <= MAP_MAX / 1024 = 32 chunks
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t mbits = 0, kbits = 0;  // this thread's point of chunk k: matched / survives
    for (int k = 0; k < c; k++) {
        const int i = k * RES_THREADS + tid;
        int m = -2, cnt = 0;
        if (i < M) m = S.match[i], cnt = A.counter[i];
        const bool mt = m >= 0, kp = (i < M) && ((cnt + (m == -1 ? 1 : 0)) < th);
        mbits |= (mt ? 1u : 0u) << k, kbits |= (kp ? 1u : 0u) << k;
        const uint64_t bm = __ballot(mt), bk = __ballot(kp);
        if (lane == 0) cnts[k * NWV + wv] = (uint32_t)__popcll(bm) | ((uint32_t)__popcll(bk) << 16);
    }
    __syncthreads();
    uint32_t base = 0;
    int total;
    {   // exclusive scan over the (chunk, wavefront) counts in storage order: c * 16 <= 512 entries, one per thread
        const int v = (tid < c * NWV) ? (int)cnts[tid] : 0;
        const int ex = block_excl_scan(v, scan, &total);
        if (tid < c * NWV) cnts[tid] = (uint32_t)ex;
    }
    __syncthreads();
    const int n_match = (int)((unsigned)total & 0xFFFFu), n_keep = (int)((unsigned)total >> 16);  // (MAP_MAX = 32768 survivors reach bit 31)
    const bool lost = n_match < S.prm.min_matches;  // lvt_system.cpp:267-272, :199-204
    if (tid == 0) {
        ctl.n_matches = n_match;
        ctl.counts[C_N_MATCHES] = n_match;
        if (lost) {
            ctl.lost_now = 1;
            ctl.state = 3;
            pose_to_Rt(ctl.last_pose, ctl.out_R, ctl.out_t);
            ctl.out_status = 3;
        } else {  // push_back / pop_front
            ctl.last_matches[0] = ctl.last_matches[1];
            ctl.last_matches[1] = ctl.last_matches[2];
            ctl.last_matches[2] = n_match;
        }
    }
    for (int k = 0; k < c; k++) {
        const int i = k * RES_THREADS + tid;
        const bool mt = (mbits >> k) & 1u, kp = (kbits >> k) & 1u;
        const uint64_t bm = __ballot(mt), bk = __ballot(kp);
        base = cnts[k * NWV + wv];
        const int offm = (int)(base & 0xFFFFu) + __popcll(bm & lt), offk = (int)(base >> 16) + __popcll(bk & lt);
        if (i >= M) continue;
        int m = S.match[i], cnt = A.counter[i], ag = A.age[i];
        if (m == -1) cnt += 1;  // lvt_local_map.cpp:201-224
        else if (m >= 0) ag += 1;
        if (lost) {  // the map stays where it is, with the bookkeeping applied
            A.match_idx[i] = m, A.counter[i] = cnt, A.age[i] = ag;
            continue;
        }
        if (!mt && !kp) continue;
        const double p0 = A.pos[3 * i], p1 = A.pos[3 * i + 1], p2 = A.pos[3 * i + 2];
        if (mt && offm < NF_MAX) {
            S.pnp_X[3 * offm] = p0;
            S.pnp_X[3 * offm + 1] = p1;
            S.pnp_X[3 * offm + 2] = p2;
            S.pnp_obs[2 * offm] = F.x[m];
            S.pnp_obs[2 * offm + 1] = F.y[m];
            S.pnp_feat[offm] = m;
            S.pnp_level[offm] = 0;
        }
        if (kp) {  // clean_untracked_points (lvt_local_map.cpp:393-413): stable compaction into the other buffer
            B.pos[3 * offk] = p0, B.pos[3 * offk + 1] = p1, B.pos[3 * offk + 2] = p2;
#pragma unroll
            for (int q = 0; q < 4; q++) B.desc[(size_t)offk * 4 + q] = A.desc[(size_t)i * 4 + q];
            B.counter[offk] = cnt;
            B.age[offk] = ag;
            B.match_idx[offk] = m;
        } else if (mt)
            S.fb[par].feat[0].flag[m] = 0;  // :402-405
    }
    if (lost) return true;
    __syncthreads();
    if (tid == 0) {
        ctl.counts[C_N_CULLED] = M - n_keep;
        *S.map_cur = cur ^ 1;
        *S.map_n = n_keep;
    }
    return false;
}

// =================================================================================================
// k_track_mid : find_matches pass 1 [+ second pass, if pass 1 found < 50] + bookkeeping + LOST decision + cull
// =================================================================================================
template <bool BV>
__global__ __launch_bounds__(RES_THREADS) void k_track_mid(SeqArg<BV> sa, int par, seq_t seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) sa.get().ctl->dbg[42] = (long long)wall_clock64();
    const Seq &S = sa.get();
    Ctl &ctl = *S.ctl;
    if (!ctl.active) return;
    if (threadIdx.x == 0 && ctl.mm_pending) {  // commit the motion model state shadowed by the prologue
        for (int k = 0; k < 4; k++) ctl.mm_last_q[k] = ctl.mm_next[k], ctl.mm_ang_vel[k] = ctl.mm_next[4 + k];
        for (int k = 0; k < 3; k++) ctl.mm_last_p[k] = ctl.mm_next[8 + k], ctl.mm_lin_vel[k] = ctl.mm_next[11 + k];
        ctl.mm_pending = 0;
    }
    if (ctl.first_frame) return;
    RESOLVE_LDS_DECL
    {   // find_matches pass 1: the points [0, early_done) were resolved by k_early_mid; the greedy scan continues behind them
        const int n_early = (ctl.early_ran_seq == seq) ? max(ctl.early_done, 0) : 0;
        resolve_body<MODE_MAP>(S, ctl, 0, par, L, r_tab, n_early > 0 ? 2 : 0, n_early);
        __syncthreads();
        if (threadIdx.x == 0) ctl.early_done = 0;  // consumed; this frame's k_pnp publishes the next one
    }
    __syncthreads();
    if (L.misc[2]) {  // rare (< 50 matches): doubled-radius candidate lists are built by this one block
        {
            CandLds C;  // carve the (not yet used) list area of the resolver
            C.lbuf = r_lists;                                      // 16 waves x KC words
            C.tx = reinterpret_cast<float *>(r_lists + 16 * KC);   // NF_MAX
            C.ty = C.tx + NF_MAX;
            C.tc = reinterpret_cast<uint32_t *>(C.ty + NF_MAX);
            candidates_body<MODE_MAP>(S, 1, par, C, wave_id(), RES_THREADS / 64, RES_THREADS);
        }
        __syncthreads();
        resolve_body<MODE_MAP>(S, ctl, 1, par, L, r_tab);
        __syncthreads();
    }
    if (*S.map_n <= RES_THREADS) {
        bookkeep_cull_small(S, ctl, par, L.scan);
        return;
    }
    bookkeep_cull_large(S, ctl, par, L.scan, L.lists);
}

// k_early_mid : the greedy resolution of find_matches for the map points [0, early_done) (storage order: their decisions do
// not depend on the points the previous frame is still appending), on the early stream behind k_early_map
template <bool BV>
__global__ __launch_bounds__(RES_THREADS) void k_early_mid(SeqArg<BV> sa, int par, seq_t seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) sa.get().ctl->dbg[36] = (long long)wall_clock64();
    const Seq &S = sa.get();
    Ctl &ctl = *S.ctl;
    const int n_early = (ctl.gate_ok == seq) ? ctl.early_done : 0;
    if (n_early > 0) {
        RESOLVE_LDS_DECL
        resolve_body<MODE_MAP>(S, ctl, 0, par, L, r_tab, 1, n_early);
        __syncthreads();
        if (threadIdx.x == 0) ctl.early_ran_seq = seq;  // the late kernels trust early_done only with this confirmation
    }
    if (threadIdx.x == 0) {  // the tracking stream's gate polls this: the early stream has finished what it claimed
        ctl.dbg[37] = (long long)wall_clock64();
        __threadfence();
        if (n_early > 0 || ctl.gate_ok == seq) atomicCAS(&ctl.early_state, 4u * seq + 1u, 4u * seq + 2u);
    }
}

// the tracking stream's counterpart of k_gate: returns when the early stream has finished this frame.  A barrier packet waiting
// on an event would do the same, but a queue parked on a barrier stalls the other queues of its hardware pipe (measured: the
// feature stream only advanced when the tracking stream's barrier resolved), and the event itself costs ~12 us of latency.
// The frame's result record goes straight into the caller's pinned ring slot (device-visible host memory), then the completion
// flag the host polls (release at system scope).  An asynchronous copy-engine transfer at the end of the chain cost ~20 us of
// the inter-frame critical path; doing this at the end of k_triangulate costs ~5 us of it (the system-scope fence is a PCIe
// round trip) -- so in the asynchronous mode the NEXT frame's k_gate_late delivers it, while it has to wait for the early stream
// anyway (the per-frame part of Ctl is only reset by the k_match_map behind it).
__device__ __forceinline__ void deliver_record(const Ctl &ctl, Ctl *rec_out, seq_t *done_out, seq_t seq, int nthreads) {
    static_assert(sizeof(Ctl) % 8 == 0, "record copy");
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&ctl);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(rec_out);
    for (int i = threadIdx.x; i < (int)(sizeof(Ctl) / 8); i += nthreads) dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done_out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (the last enqueued frame has no successor: the host launches this when it is asked for that frame's result)
__global__ __launch_bounds__(64) void k_deliver(Seq *seqs, Ctl *rec_out, seq_t *done_out, seq_t seq) {
    deliver_record(*seq_const(seqs, blockIdx.z).ctl, rec_out + blockIdx.z, done_out + blockIdx.z, seq, 64);
}

// the tracking stream's wait for the early stream (one thread; see k_gate_late / k_match_map)
__device__ __forceinline__ void gate_late_poll(Ctl &ctl, FeatCtl &fc, seq_t seq) {
    unsigned long long t0 = wall_clock64();
    for (;;) {
        const seq_t cur = __hip_atomic_load(&ctl.early_state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (cur >= 4u * seq + 2u) break;  // finished, stood down or cancelled
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 6000000ull) {  // 60 ms: the early stream's own gate gives up after 2 x 20
            atomicAdd(&ctl.gate_timeouts, 1);
            // not claimed yet: cancel it (an early kernel arriving later does nothing).  Claimed: its kernels are running, wait on.
            if (cur < 4u * seq + 1u && atomicCAS(&ctl.early_state, cur, 4u * seq + 3u) == cur) break;
            t0 = wall_clock64();
        }
    }
    // this stream has no barrier of its own on the frame's features (the early stream normally vouches for them): make sure
    t0 = wall_clock64();
    while (__hip_atomic_load(&fc.feat_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 200000000ull) {  // 2 s: the feature stream is wedged -- nothing valid to track on: the frame is
            ctl.skip = 1;                           // skipped (last pose, state kept), reported through lvt_amd_last_error
            atomicAdd(&ctl.gate_fatal, 1);
            break;
        }
    }
    if (__hip_atomic_load(&fc.skip_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == seq) ctl.skip = 1;  // published without features (k_gate_buf gave up)
}

template <bool BV>
__global__ __launch_bounds__(64) void k_gate_late(SeqArg<BV> sa, int par, seq_t seq, Ctl *prev_rec, seq_t *prev_done) {
    if (threadIdx.x == 0 && blockIdx.x == 0) sa.get().ctl->dbg[38] = (long long)wall_clock64();
    Ctl &ctl = *sa.get().ctl;
    FeatCtl &fc = *sa.get().fb[par].fc;
    if (prev_rec) deliver_record(ctl, prev_rec + blockIdx.z, prev_done + blockIdx.z, seq - 1, 64);  // the previous frame's result
    if (threadIdx.x == 0) gate_late_poll(ctl, fc, seq);
}

// =================================================================================================
// k_pnp : motion-only bundle adjustment, one 256-thread workgroup per sequence
// =================================================================================================
// 1 / sqrt(s) for s > 0 in the normal range: hardware seed (v_rsq_f64, ~26 bits) + two Newton steps in fma form.
// One dependency chain of ~10 instructions where sqrt followed by a division costs ~45 -- thread 0 runs the pose solve
// alone, so these chains ARE its time.  The result is within 1 ulp of the correctly rounded value.
// 1 / s from the hardware seed (v_rcp_f64) and two Newton steps: five instructions where the IEEE division expands to twelve
__device__ __forceinline__ double rcp_nr(double s) {
    double y = __builtin_amdgcn_rcp(s);
#pragma unroll
    for (int it = 0; it < 2; it++) y = __builtin_fma(__builtin_fma(-s, y, 1.0), y, y);
    return y;
}
// log(a) for finite a >= 1 (the Cauchy kernel's 1 + e^2 / delta^2): a = 2^e m with m in [sqrt(1/2), sqrt(2)),
// log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.172: ten odd terms reach 2^-56.  ~30 instructions against the ~55 of
// the library routine (which also handles denormals, zero, negative and non-finite arguments); error < 2 ulp.
__device__ __forceinline__ double log_ge1(double a) {
    int e = __builtin_amdgcn_frexp_exp(a);
    double m = __builtin_amdgcn_frexp_mant(a);  // [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double s = (m - 1.0) * rcp_nr(m + 1.0);
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = __builtin_fma(p, z, 1.0 / 19.0);
    p = __builtin_fma(p, z, 1.0 / 17.0);
    p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, 1.0 / 11.0);
    p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0);
    p = __builtin_fma(p, z, 1.0 / 3.0);
    const double sz = s * z;
    const double ed = (double)e;
    // e ln2_hi is exact (ln2_hi has 11 trailing zero bits), the rest is added smallest first
    return __builtin_fma(ed, 6.93147180369123816490e-01, (s + s) + __builtin_fma(ed, 1.90821492927058770002e-10, (sz + sz) * p));
}
__device__ __forceinline__ double rsqrt_nr(double s) {
    double y = __builtin_amdgcn_rsq(s);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const double e = __builtin_fma(-(s * y), y, 1.0);  // 1 - s y^2
        y = __builtin_fma(0.5 * y, e, y);
    }
    return y;
}

// (H + lambda I) x = b for the 6x6 system packed in `sys` (upper triangle row-major in [0, 21), b in [21, 27)) -- the role of g2o's dense solve
// behind LinearSolverPCG with the exact block-Jacobi preconditioner (A.6).  Thread 0 runs this alone, one wavefront on its SIMD: a dependent fp64
// operation costs its full latency here (32 cycles, profiles/r04_fp64_issue_rate.txt), so the routine is written for the LENGTH OF THE DEPENDENCY
// CHAIN: a right-looking LDL^T (no square root: the pivot's reciprocal is a seed + two Newton steps, 5 dependent operations where 1 / sqrt needs 7),
// fused multiply-adds, the forward substitution folded into the elimination, the back substitution adding its newest term last: ~50 dependent
// operations where the Cholesky form of rounds 1-3 had ~85 (2.5k -> 1.6k cycles per solve).  Same pivots as Cholesky (d_j = the square of L_jj):
// "not positive definite" is detected on the same quantity; the solution differs from the LL^T form by roundings.
__device__ __forceinline__ bool solve6_ldl(const double *sys, double lambda, double *x) {
    double A[6][6], z[6], inv[6];
    {
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int cc = a; cc < 6; cc++) A[cc][a] = sys[k++];
        }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) A[a][a] += lambda, z[a] = sys[21 + a];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const double d = A[j][j];
        ok = ok && (d > 0) && isfinite(d);
        inv[j] = rcp_nr(d);
        double cj[6], lj[6];
#pragma unroll
        for (int i = j + 1; i < 6; i++) cj[i] = A[i][j], lj[i] = cj[i] * inv[j];
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
#pragma unroll
            for (int k2 = j + 1; k2 <= i; k2++) A[i][k2] = __builtin_fma(-lj[i], cj[k2], A[i][k2]);
            z[i] = __builtin_fma(-lj[i], z[j], z[i]);
            A[i][j] = lj[i];
        }
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double v = z[i] * inv[i];
#pragma unroll
        for (int k2 = 5; k2 > i; k2--) v = __builtin_fma(-A[k2][i], x[k2], v);
        x[i] = v;
    }
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] = 0.0;  // the failed solve of the reference leaves the step at zero
    }
    return ok;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

constexpr int PNP_THREADS = 256;  // one wavefront per SIMD.  More wavefronts were measured (768 threads, one edge each instead of three interleaved:
                                   // sweep + reduction 4.7k -> 6.6k cycles): a single wavefront with three independent chains already issues an fp64
                                   // instruction every ~4.5 cycles (profiles/r04_fp64_issue_rate.txt), more of them only add barrier and reduction time
constexpr int PNP_ILP = 3;         // edges per thread in flight in one sweep iteration (KITTI: ~650 edges = one iteration)
constexpr int PNP_NW = PNP_THREADS / 64;
static_assert(PNP_NW == 4, "block_sum's tree is written for four wavefronts");
constexpr int PNP_RED = 16 + PNP_NW * 32 + 2;  // small sums [0, 4) | reduce-scatter partials [16, 16 + 4 * 32) | the inactive edges' sink (2)

__device__ __forceinline__ double tree4(const double *p, int stride) { return (p[0] + p[stride]) + (p[2 * stride] + p[3 * stride]); }  // fixed order: deterministic

// block-wide sum of NV doubles per thread.  NV < 8: result in v[] of every thread.  NV >= 8: result in dst[0..NV-1] (LDS).
// Large NV: every wavefront reduce-scatters its NV values in registers (wave_reduce.h: 29 additions and their lane exchanges for 28 values, the lane
// pair (l, l ^ 1) ends with the total of value wave_rs_index(l)), 32 lanes per wavefront put them into LDS, thread k adds the wavefronts' totals of
// value k.  A wavefront without an edge (`wave_active` false: its values are zero) stores zeros and skips the exchange.
// Two barriers.  red[16, ...) only: block_sum<1> (red[0, 4)) may still be read by a slow wavefront when this one is entered.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *red, double *dst = nullptr, bool wave_active = true) {
    const int w = wave_id(), l = lane_id(), tid = threadIdx.x;
    if constexpr (NV >= 8) {
        static_assert(NV <= 32, "block_sum: one reduce-scatter");
        double *part = red + 16;  // [PNP_NW][32]
        if (wave_active) {
            const double s = wave_reduce_scatter<NV>(v);
            if (!(l & 1)) part[w * 32 + wave_rs_index(l)] = s;
        } else if (l < 32)
            part[w * 32 + l] = 0.0;
        __syncthreads();
        if (tid < NV) dst[tid] = tree4(part + tid, 32);  // v[] is NOT updated (only thread 0 wants the sums)
        __syncthreads();
    } else {
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = wave_sum(v[k]);
        __syncthreads();
        static_assert(NV == 1, "block_sum: the small form carries one value");
        if (l == 0) red[w] = v[0];
        __syncthreads();
        v[0] = tree4(red, 1);
    }
}

// State of the solve.  Everything only thread 0 computes with lives HERE, not in registers: a 6x6 system, its
// right-hand side and the step would otherwise sit in ~110 VGPRs of every lane across the sweeps.
struct PnpShared {
    double r[4], t[3];    // current estimate (during a trial: the trial's), published by thread 0
    double br[4], bt[3];  // push(): the estimate before the trial
    double cam[2][16];    // world -> normalised camera (3x4, SBACam's w2n) + position [12, 15) of two estimates: [camcur] the current one, [camcur ^ 1] a trial's
    int camcur;
    double sys[2][28];    // H (upper triangle), b, chi2: [cur] at the current estimate, [cur ^ 1] receives a trial's
    int cur;
    double dx[6];
    double inv_scale;     // 1 / (dx . (lambda dx + b) + 1e-3): known with the step, so it is formed beside the pose update instead of behind the sweep
    double lambda, ni;
    int cont, ok, accepted, ok2;
    // LM bookkeeping for the tests (the oracle counts the same): trials, rejected trials, passes ended by Terminate; and, stand-alone entry only,
    // one (lambda, chi2 at the estimate, chi2 of the trial, rho) row per trial
    int trials, rejections, terminates, trace_cap;
    double *trace;
};

// thread 0: the matrices every lane's sweep needs from (r, t) -- SBACam's w2n = [R^T | -R^T t] in cam_refresh's operation order -- go to LDS once
// instead of being recomputed by every wavefront
__device__ __forceinline__ void publish_cam(double *dst, const double r[4], const double t[3]) {
    double R[9];
    q_to_R(r, R);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double a = R[i], b = R[3 + i], c = R[6 + i];
        dst[4 * i] = a, dst[4 * i + 1] = b, dst[4 * i + 2] = c;
        dst[4 * i + 3] = -(a * t[0] + b * t[1] + c * t[2]);
    }
    dst[12] = t[0], dst[13] = t[1], dst[14] = t[2];
}

// One sweep over the active edges at the camera w2n / position ct: errors (stored), robust chi2 partial in acc[27], and -- when
// WANT_H -- the edge's contribution to H (upper triangle, acc[0..20]) and b (acc[21..26]) linearised at the same
// estimate (EdgeProjectP2MC::computeError / linearizeOplus / constructQuadraticForm with the Cauchy weight, A.6).
// Returns whether this wavefront had an edge at all.
template <bool WANT_H>
__device__ __forceinline__ bool pnp_sweep(const double (&w)[12], const double ct[3], double fx, double fy, double cx, double cy, double dsqr, double dsqrReci,
                                          const double *__restrict__ X, const float *__restrict__ obs, double *__restrict__ err,
                                          double *__restrict__ sink, const int8_t *__restrict__ level, int n, double (&acc)[28]) {
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = 0.0;
    const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    if (wbase >= n) return false;
    // One edge, straight-line: an inactive slot (past the end, or an edge of level 1) runs the same instructions on a
    // clamped index with its weights selected to zero, so the PNP_ILP edges of one iteration sit in one basic block
    // and their fp64 dependency chains interleave (there is one wavefront per SIMD: nothing else hides the latency).
    // The sweep runs at the fp64 issue rate of a CU, so it is written for operation count: the four IEEE divisions of an edge are
    // reciprocal seeds with two Newton steps, the pixel error is formed from the camera-frame point the Jacobian needs anyway,
    // the logarithm is specialised to arguments >= 1 (400 -> 223 instructions per edge).  Multiply-adds are fused where that
    // saves an instruction slot.  Every value is within a rounding or two of the form the oracle evaluates (deviations).
    auto edge = [&](int i_raw) {
#pragma clang fp contract(fast)
        const int i = min(i_raw, n - 1);
        const int8_t lv = level[i];
        const bool active = (i_raw < n) & (lv == 0);  // (no short circuit: a guarded load would split the block)
        const double x = X[3 * i], y = X[3 * i + 1], z = X[3 * i + 2];
        // pixel = K * (normalised camera point): u = fx pcx / pcz + cx (w2i = K w2n, whose third row is w2n's), so the
        // error needs the camera-frame point only -- the same one the Jacobian is built from
        const double pcx = ((w[0] * x + w[1] * y) + w[2] * z) + w[3];
        const double pcy = ((w[4] * x + w[5] * y) + w[6] * z) + w[7];
        const double pcz = ((w[8] * x + w[9] * y) + w[10] * z) + w[11];
        const double ipcz = rcp_nr(pcz);
        const double e0 = fx * (pcx * ipcz) + (cx - (double)obs[2 * i]), e1 = fy * (pcy * ipcz) + (cy - (double)obs[2 * i + 1]);
        {  // an inactive slot stores into the sink: no conditional store inside the block
            double *ep = active ? err + 2 * i : sink;
            ep[0] = e0;
            ep[1] = e1;
        }
        const double aux = dsqrReci * (e0 * e0 + e1 * e1) + 1.0;
        const double chi = dsqr * log_ge1(aux);
        acc[27] += active ? chi : 0.0;
        if (WANT_H) {
            const double ipz2 = ipcz * ipcz;
            const double ipz2fx = active ? ipz2 * fx : 0.0, ipz2fy = active ? ipz2 * fy : 0.0;
            const double pwt[3] = {x - ct[0], y - ct[1], z - ct[2]};
            double J0[6], J1[6];
            // dp = dRd{x,y,z} * pwt with dRd* = dRid* * w2n[:, :3]  
            const double a0 = (w[0] * pwt[0] + w[1] * pwt[1]) + w[2] * pwt[2];
            const double a1 = (w[4] * pwt[0] + w[5] * pwt[1]) + w[6] * pwt[2];
            const double a2 = (w[8] * pwt[0] + w[9] * pwt[1]) + w[10] * pwt[2];
            {
                const double dp1 = 2.0 * a2, dp2 = -2.0 * a1;  // dRdx (dp0 = 0)
                J0[3] = (-pcx * dp2) * ipz2fx;
                J1[3] = (pcz * dp1 - pcy * dp2) * ipz2fy;
            }
            {
                const double dp0 = -2.0 * a2, dp2 = 2.0 * a0;  // dRdy (dp1 = 0)
                J0[4] = (pcz * dp0 - pcx * dp2) * ipz2fx;
                J1[4] = (-pcy * dp2) * ipz2fy;
            }
            {
                const double dp0 = 2.0 * a1, dp1 = -2.0 * a0;  // dRdz (dp2 = 0)
                J0[5] = (pcz * dp0) * ipz2fx;
                J1[5] = (pcz * dp1) * ipz2fy;
            }
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                const double dp0 = -w[cc], dp1 = -w[4 + cc], dp2 = -w[8 + cc];
                J0[cc] = (pcz * dp0 - pcx * dp2) * ipz2fx;
                J1[cc] = (pcz * dp1 - pcy * dp2) * ipz2fy;
            }
            const double rho1 = active ? rcp_nr(aux) : 0.0;
            const double wr0 = -e0 * rho1, wr1 = -e1 * rho1;
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const double j0r = J0[a] * rho1, j1r = J1[a] * rho1;
#pragma unroll
                for (int cc = a; cc < 6; cc++, k++) acc[k] = __builtin_fma(j0r, J0[cc], __builtin_fma(j1r, J1[cc], acc[k]));
            }
#pragma unroll
            for (int a = 0; a < 6; a++) acc[21 + a] = __builtin_fma(J0[a], wr0, __builtin_fma(J1[a], wr1, acc[21 + a]));
        }
    };
    for (int i = threadIdx.x; i < n; i += PNP_ILP * PNP_THREADS) {
#pragma unroll
        for (int u = 0; u < PNP_ILP; u++) edge(i + u * PNP_THREADS);
    }
    return true;
}

// g2o's OptimizationAlgorithmLevenberg as configured by lvt_pnp_solver.cpp:44-53,60-128, one workgroup.
// solve(i) = computeActiveErrors + buildSystem + trial loop.  The sweep that evaluates a trial's errors also
// accumulates H and b at the trial estimate: when the trial is ACCEPTED, the next solve()'s computeActiveErrors and
// buildSystem would recompute exactly those values (same estimate, same active edges, same arithmetic), so they are
// reused; a rejected last trial or a new pass falls back to a fresh sweep.
__device__ __forceinline__ void pnp_run(const Params &prm, const Pose &prior, const double *X, const float *obs, double *err, double *sink,
                        int8_t *level, int n, PnpShared &sh, double *red, Pose &result, int &inliers, int &solve_calls, int &borderline, long long *dbg = nullptr) {
    const int tid = threadIdx.x;
    long long t_sweep = 0, t_solve = 0, t_dec = 0, t_red = 0, t_all = clock64();
    const double fx = prm.fx, fy = prm.fy, cx = prm.cx, cy = prm.cy;
    const double mono_chi = sqrt(REPROJ_TH2);
    const double dsqr = mono_chi * mono_chi;
    const double dsqrReci = 1.0 / dsqr;
    if (tid == 0) {  // SE3Quat ctor: normalizeRotation()
        double r[4], t[3];
        for (int k = 0; k < 4; k++) r[k] = prior.q[k];
        if (r[0] < 0)
            for (int k = 0; k < 4; k++) r[k] = -r[k];
        q_normalize(r);
        for (int k = 0; k < 4; k++) sh.r[k] = r[k];
        for (int k = 0; k < 3; k++) sh.t[k] = t[k] = prior.p[k];
        publish_cam(sh.cam[0], r, t);
        sh.camcur = 0;
        sh.cur = 0;
        sh.trials = sh.rejections = sh.terminates = 0;
    }
    __syncthreads();
    double w2n[12], ct[3];
    int sw_slot = 0;  // the slot of sh.cam the most recent sweep ran at = the estimate the stored edge errors belong to (after a rejected last trial
                      // that is NOT the estimate the pass ends with: pop() restores the camera, the errors stay those of the trial)
    auto load_cam = [&](int slot) {
        const double *c = sh.cam[slot];
#pragma unroll
        for (int k = 0; k < 12; k++) w2n[k] = c[k];
#pragma unroll
        for (int k = 0; k < 3; k++) ct[k] = c[12 + k];
    };
    load_cam(0);
    int calls = 0;
    double n_border[1] = {0.0};

    bool any_active = n > 0;  // every edge enters at level 0 (k_track_mid's bookkeeping / the stand-alone entry clear the levels); pass 2: the gate below tells
    for (int pass = 0; pass < 2; pass++) {
        bool ok = true;         // uniform across the block (decisions are broadcast through sh)
        bool have_sys = false;  // sh.sys holds chi2 / H / b at the current estimate
        for (int iter = 0; iter < 5 && ok && any_active; iter++) {
            calls++;
            long long c0 = clock64();
            if (!have_sys) {
                double acc[28];
                const bool wa = pnp_sweep<true>(w2n, ct, fx, fy, cx, cy, dsqr, dsqrReci, X, obs, err, sink, level, n, acc);
                sw_slot = sh.camcur;
                block_sum<28>(acc, red, sh.sys[sh.cur], wa);
            }
            t_sweep += clock64() - c0;
            if (tid == 0 && iter == 0) {
                double maxDiag = 0;
                int k = 0;
                for (int a = 0; a < 6; a++) {
                    maxDiag = fmax(fabs(sh.sys[sh.cur][k]), maxDiag);  // diagonal entries of the packed upper triangle
                    k += 6 - a;
                }
                sh.lambda = 1e-5 * maxDiag;
                sh.ni = 2;
            }
            const bool speculate = (iter < 4);  // there is a next solve() in this pass that could reuse the system
            int qmax = 0;
            bool cont;
            do {
                long long c1 = clock64();
                const int trial_slot = sh.camcur ^ 1;
                if (tid == 0) {
                    double cr[4], c0t[3];
                    for (int k = 0; k < 4; k++) cr[k] = sh.r[k], sh.br[k] = cr[k];  // push()
                    for (int k = 0; k < 3; k++) c0t[k] = sh.t[k], sh.bt[k] = c0t[k];
                    double dx[6];
                    const double *sys = sh.sys[sh.cur];
                    const double lambda = sh.lambda;
                    sh.ok2 = solve6_ldl(sys, lambda, dx) ? 1 : 0;
                    for (int a = 0; a < 6; a++) sh.dx[a] = dx[a];
                    {   // the gain ratio's denominator (g2o: computeScale() + 1e-3) needs the step only: formed here, beside the pose update's chain,
                        // as a reciprocal -- the decision behind the sweep is then one subtraction and one product
                        const double s0 = dx[0] * (lambda * dx[0] + sys[21]), s1 = dx[1] * (lambda * dx[1] + sys[22]), s2 = dx[2] * (lambda * dx[2] + sys[23]);
                        const double s3 = dx[3] * (lambda * dx[3] + sys[24]), s4 = dx[4] * (lambda * dx[4] + sys[25]), s5 = dx[5] * (lambda * dx[5] + sys[26]);
                        double scale = 0;
                        scale += s0, scale += s1, scale += s2, scale += s3, scale += s4, scale += s5;  // (the reference's order)
                        scale += 1e-3;
                        sh.inv_scale = rcp_nr(scale);
                    }
                    // SBACam::update
                    double nt[3], qr[4], nr[4];
                    for (int k2 = 0; k2 < 3; k2++) nt[k2] = c0t[k2] + dx[k2];
                    qr[1] = dx[3], qr[2] = dx[4], qr[3] = dx[5];
                    qr[0] = sqrt(1.0 - (dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]));
                    q_mul(cr, qr, nr);
                    {  // normalize(): the product of two unit quaternions has z = 1 + e with |e| ~ 1e-16, and 1 / sqrt(1 + e) = 1 - e / 2 + 3 e^2 / 8 to
                       // the last bit for |e| < 1e-5 (three dependent operations); anything else takes the seed + Newton form
                        const double z = nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2] + nr[3] * nr[3];
                        if (z > 0) {
                            const double e = z - 1.0;
                            const double rn = (fabs(e) < 1e-5) ? __builtin_fma(e, __builtin_fma(0.375, e, -0.5), 1.0) : rsqrt_nr(z);
                            nr[0] *= rn, nr[1] *= rn, nr[2] *= rn, nr[3] *= rn;
                        }
                    }
                    for (int k2 = 0; k2 < 4; k2++) sh.r[k2] = nr[k2];
                    for (int k2 = 0; k2 < 3; k2++) sh.t[k2] = nt[k2];
                    publish_cam(sh.cam[trial_slot], nr, nt);
                }
                t_solve += clock64() - c1;
                __syncthreads();
                load_cam(trial_slot);
                c1 = clock64();
                double tempChi;
                sw_slot = trial_slot;
                {
                    double acc[28];
                    if (speculate) {
                        const bool wa = pnp_sweep<true>(w2n, ct, fx, fy, cx, cy, dsqr, dsqrReci, X, obs, err, sink, level, n, acc);
                        const long long c2 = clock64();
                        block_sum<28>(acc, red, sh.sys[sh.cur ^ 1], wa);  // the trial's system goes to the spare slot
                        t_red += clock64() - c2;
                        tempChi = sh.sys[sh.cur ^ 1][27];
                    } else {
                        const bool wa = pnp_sweep<false>(w2n, ct, fx, fy, cx, cy, dsqr, dsqrReci, X, obs, err, sink, level, n, acc);
                        double ch[1] = {wa ? acc[27] : 0.0};
                        block_sum<1>(ch, red);
                        tempChi = ch[0];
                    }
                }
                t_sweep += clock64() - c1;
                c1 = clock64();
                if (tid == 0) {
                    if (!sh.ok2) tempChi = 1.7976931348623157e308;
                    const double *sys = sh.sys[sh.cur];
                    const double currentChi = sys[27];
                    double lambda = sh.lambda, ni = sh.ni;
                    const double rho = (currentChi - tempChi) * sh.inv_scale;  // (g2o divides by the scale: within a rounding or two)
                    const bool accept = (rho > 0 && isfinite(tempChi));
                    if (sh.trace && sh.trials < sh.trace_cap) {
                        double *row = sh.trace + 4 * sh.trials;
                        row[0] = lambda, row[1] = currentChi, row[2] = tempChi, row[3] = rho;
                    }
                    sh.trials++;
                    sh.rejections += accept ? 0 : 1;
                    if (accept) {
                        const double tr = 2 * rho - 1;
                        double alpha = 1. - tr * tr * tr;  // g2o: pow(2 rho - 1, 3)
                        alpha = fmin(alpha, 2.0 / 3.0);
                        const double scaleFactor = fmax(1.0 / 3.0, alpha);
                        lambda *= scaleFactor;
                        ni = 2;
                        if (speculate) sh.cur ^= 1;  // the trial's system IS the system of the next solve()
                        sh.camcur = trial_slot;      // ... and its camera the current one
                    } else {
                        lambda *= ni;
                        ni *= 2;
                        for (int k = 0; k < 4; k++) sh.r[k] = sh.br[k];  // pop(): edge errors stay those of the rejected trial
                        for (int k = 0; k < 3; k++) sh.t[k] = sh.bt[k];
                    }
                    sh.lambda = lambda, sh.ni = ni;
                    qmax++;
                    const int c = (rho < 0 && qmax < 10) ? 1 : 0;
                    sh.cont = c;
                    sh.ok = (!c && (qmax == 10 || rho == 0)) ? 0 : 1;  // Terminate
                    sh.terminates += sh.ok ? 0 : 1;
                    sh.accepted = accept ? 1 : 0;
                }
                t_dec += clock64() - c1;
                __syncthreads();
                if (!sh.accepted) load_cam(trial_slot ^ 1);  // pop() restored the old estimate; an accepted trial's camera is already loaded
                cont = sh.cont != 0;
                ok = sh.ok != 0;
                have_sys = speculate && (sh.accepted != 0);  // sh.sys = system at the (accepted) current estimate
                __syncthreads();  // sh.cont / sh.camcur consumed before thread 0 publishes again
            } while (cont);
        }
        // chi2 gate on the last computed errors (lvt_pnp_solver.cpp:109-116).  The sweep evaluates an edge's error with refined
        // reciprocals and in the camera-frame form (section 4.6 of DESIGN.md): within ~1e-11 of the value the reference's expression
        // px / pz - u gives.  A decision that close to the threshold is re-taken on the reference's own expression, IEEE divisions, at the
        // estimate the stored errors belong to -- and counted, so that tests see how often (and how near) that happens.
        {
            int kept = 0;
            for (int i = tid; i < n; i += PNP_THREADS) {
                if (level[i] != 0) continue;
                const double e0 = err[2 * i], e1 = err[2 * i + 1];
                const double e2 = e0 * e0 + e1 * e1;
                bool out = e2 > REPROJ_TH2;
                if (fabs(e2 - REPROJ_TH2) < PNP_GATE_MARGIN) {
                    // (rare) the camera of the last sweep; w2i = K w2n formed in the reference's order (cam_refresh)
                    const double *gw = sh.cam[sw_slot];
                    double wi[12];
                    for (int j = 0; j < 4; j++) {
                        wi[j] = fx * gw[j] + cx * gw[8 + j];
                        wi[4 + j] = fy * gw[4 + j] + cy * gw[8 + j];
                        wi[8 + j] = gw[8 + j];
                    }
                    const double x = X[3 * i], y = X[3 * i + 1], z = X[3 * i + 2];
                    const double px = ((wi[0] * x + wi[1] * y) + wi[2] * z) + wi[3];
                    const double py = ((wi[4] * x + wi[5] * y) + wi[6] * z) + wi[7];
                    const double pz = ((wi[8] * x + wi[9] * y) + wi[10] * z) + wi[11];
                    const double r0 = px / pz - (double)obs[2 * i], r1 = py / pz - (double)obs[2 * i + 1];
                    out = (r0 * r0 + r1 * r1) > REPROJ_TH2;
                    n_border[0] += 1.0;
                }
                if (out) level[i] = 1;
                else kept = 1;
            }
            any_active = __syncthreads_or(kept) != 0;  // (also the barrier behind the levels' update)
        }
    }
    {   // inliers and borderline decisions in ONE block-wide sum: both are small integers, inliers + 65536 borderline is exact in a double
        double cnt[1] = {65536.0 * n_border[0]};
        for (int i = tid; i < n; i += PNP_THREADS) cnt[0] += (level[i] == 0) ? 1.0 : 0.0;
        block_sum<1>(cnt, red);
        const long long packed = (long long)cnt[0];
        inliers = (int)(packed & 65535), borderline = (int)(packed >> 16);
    }
    solve_calls = calls;
    for (int k = 0; k < 4; k++) result.q[k] = sh.r[k];
    for (int k = 0; k < 3; k++) result.p[k] = sh.t[k];
    if (dbg && tid == 0) {
        dbg[12] = t_sweep, dbg[13] = t_red, dbg[14] = t_solve, dbg[15] = t_dec, dbg[16] = clock64() - t_all, dbg[17] = calls;
    }
    __syncthreads();
}

// The ten sweeps of one solve read every edge again: for up to PNP_STAGE_MAX edges (KITTI: ~650) the points,
// observations, levels and errors live in LDS for the whole solve instead of costing an L2 round trip per sweep.
constexpr int PNP_STAGE_MAX = 1536;
constexpr int PNP_DYN_BYTES = PNP_STAGE_MAX * (24 + 16 + 8 + 1);

__device__ __forceinline__ void pnp_solve(const Params &prm, const Pose &prior, const double *X, const float *obs, double *err, int8_t *level,
                                          int n, PnpShared &sh, double *red, uint8_t *dyn, Pose &res, int &inliers, int &calls, int &borderline, long long *dbg, bool copy_out) {
    if (n <= PNP_STAGE_MAX) {
        double *sX = reinterpret_cast<double *>(dyn);
        double *sErr = sX + 3 * PNP_STAGE_MAX;
        float *sObs = reinterpret_cast<float *>(sErr + 2 * PNP_STAGE_MAX);
        int8_t *sLvl = reinterpret_cast<int8_t *>(sObs + 2 * PNP_STAGE_MAX);
        for (int i = threadIdx.x; i < 3 * n; i += PNP_THREADS) sX[i] = X[i];
        for (int i = threadIdx.x; i < 2 * n; i += PNP_THREADS) sObs[i] = obs[i];
        for (int i = threadIdx.x; i < n; i += PNP_THREADS) sLvl[i] = level[i];
        __syncthreads();
        pnp_run(prm, prior, sX, sObs, sErr, red + PNP_RED - 2, sLvl, n, sh, red, res, inliers, calls, borderline, dbg);  // (the last two doubles of red: block_sum never touches them)
        if (copy_out) {  // (the stand-alone entry's caller reads errors and levels; nothing behind k_pnp does, and the pose waits for these stores)
            for (int i = threadIdx.x; i < 2 * n; i += PNP_THREADS) err[i] = sErr[i];
            for (int i = threadIdx.x; i < n; i += PNP_THREADS) level[i] = sLvl[i];
        }
    } else
        pnp_run(prm, prior, X, obs, err, err + 2 * (size_t)n, level, n, sh, red, res, inliers, calls, borderline, dbg);  // the caller allocates 2 n + 2 doubles
}

// the pose of a synchronous call, delivered the moment it exists (pinned host memory): lvt_track returns on it while the frame's
// tail (staged update, triangulation: ~30 us that change the map, not the pose) finishes behind the caller's next upload
struct PoseRec {
    double R[9], t[3];
    double q[4], p[3];  // the same pose as the tracker holds it (quaternion w, x, y, z + position): lvt_amd_get_last_pose
    int status, pad;
};
template <bool BV>
__global__ __launch_bounds__(PNP_THREADS) void k_pnp(SeqArg<BV> sa, int par, seq_t seq, PoseRec *pose_out, seq_t *pose_done) {
    if (threadIdx.x == 0 && blockIdx.x == 0) sa.get().ctl->dbg[44] = (long long)wall_clock64();
    const Seq &S = sa.get();
    Ctl &ctl = *S.ctl;
    if (!ctl.active || ctl.first_frame || ctl.lost_now) {
        if (threadIdx.x == 0) {  // nothing for the next frame to start on (early_done is 0), but its gate must not wait
            __threadfence();
            atomicExch(&ctl.pnp_seq, seq);
        }
        return;
    }
    __shared__ PnpShared sh;
    __shared__ double red[PNP_RED];
    extern __shared__ __attribute__((aligned(16))) uint8_t pnp_dyn[];
    Pose res;
    int inliers, calls, borderline;
    if (threadIdx.x == 0) sh.trace = nullptr, sh.trace_cap = 0;
    // err must be defined for every edge before the first gate: all edges are active in pass 1
    pnp_solve(S.prm, ctl.predicted, S.pnp_X, S.pnp_obs, S.pnp_err, S.pnp_level, ctl.n_matches, sh, red, pnp_dyn, res, inliers, calls, borderline, ctl.dbg, false);
    if (threadIdx.x == 0) {
        ctl.optimized = res;
        ctl.last_pose = res;  // lvt_system.cpp:205
        pose_to_Rt(res, ctl.out_R, ctl.out_t);
        ctl.out_status = 2;
        ctl.counts[C_PNP_ITERS] = calls;
        ctl.counts[C_PNP_INLIERS] = inliers;
        ctl.counts[C_PNP_BORDERLINE] = borderline;
        ctl.counts[C_PNP_TRIALS] = sh.trials, ctl.counts[C_PNP_REJECTIONS] = sh.rejections, ctl.counts[C_PNP_TERMINATES] = sh.terminates;
        ctl.early_done = *S.map_n;  // the map after clean_untracked_points: the next frame may start on these points now
        ctl.early_accepted = 0;
        ctl.dbg[45] = (long long)wall_clock64();
        __threadfence();
        atomicExch(&ctl.pnp_seq, seq);  // release: everything the next frame's early kernels read is final
        if (pose_out) {  // synchronous call: hand the pose to the waiting host now (no later kernel of this frame changes it or the state)
            PoseRec &o = pose_out[blockIdx.z];
            for (int k = 0; k < 9; k++) o.R[k] = ctl.out_R[k];
            for (int k = 0; k < 3; k++) o.t[k] = ctl.out_t[k];
            for (int k = 0; k < 4; k++) o.q[k] = res.q[k];
            for (int k = 0; k < 3; k++) o.p[k] = res.p[k];
            o.status = 2;
            __threadfence_system();
            __hip_atomic_store(&pose_done[blockIdx.z], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // update_staged_map_points projects the staged points with the optimised pose (lvt_local_map.cpp:357-368)
    if (S.prm.staged_th > 0) project_staged(S, res, red);
}

// stand-alone entry for differential tests (lvt_amd_pnp)
__global__ __launch_bounds__(PNP_THREADS) void k_pnp_standalone(Params prm, Pose prior, const double *X, const float *obs, double *err,
                                                                int8_t *level, int n, Pose *out, int *info, double *trace, int trace_cap) {
    __shared__ PnpShared sh;
    __shared__ double red[PNP_RED];
    extern __shared__ __attribute__((aligned(16))) uint8_t pnp_dyn[];
    for (int i = threadIdx.x; i < n; i += PNP_THREADS) level[i] = 0;
    if (threadIdx.x == 0) sh.trace = trace, sh.trace_cap = trace_cap;
    __syncthreads();
    Pose res;
    int inliers, calls, borderline;
    pnp_solve(prm, prior, X, obs, err, level, n, sh, red, pnp_dyn, res, inliers, calls, borderline, nullptr, true);
    if (threadIdx.x == 0) {
        *out = res;
        info[0] = calls;
        info[1] = inliers;
        info[2] = borderline;
        info[3] = sh.trials, info[4] = sh.rejections, info[5] = sh.terminates;
    }
}

// =================================================================================================
// staged_body : update_staged_map_points (lvt_local_map.cpp:355-391) then the triangulation policy.
// Matching = the same block-wide fixpoint; the promotion rule "counter == staged_threshold || map_size < 250"
// depends on the RUNNING map size, which has the closed form  map_n0 + #(matched staged points before i)  as long
// as it is below 250 (every matched point is promoted while the map is small), so promotions and their
// destinations come from two prefix sums.
// =================================================================================================
// (runs at the head of k_triangulate: both are single-workgroup stages of 1024 threads, a kernel boundary between them bought nothing)
__device__ __forceinline__ void staged_body(const Seq &S, Ctl &ctl, int par, ResolveLds &L, uint32_t *r_tab) {
    const int tid = threadIdx.x;
    if (!ctl.first_frame && S.prm.staged_th > 0) {
        const Feat &T = S.fb[par].feat[0];
        const int N = *T.n;
        const int scur = *S.staged_cur, SM = *S.staged_n;
        const MapSoA &A = S.staged[scur], &B = S.staged[scur ^ 1];
        const MapSoA &MP = S.map[*S.map_cur];
        if (SM > 0) {  // nothing staged (most frames): nothing to match, promote or compact
        for (int j = tid; j < NF_MAX; j += RES_THREADS) {
            const uint8_t f = (j < N) ? T.flag[j] : 0;
            if (j < N) L.flag[j] = f;
            r_tab[j] = r_tab[NF_MAX + j] = f ? PERM : 0u;
        }
        const int map_n0 = *S.map_n;
        uint32_t iter = 0;
        int matched_before = 0, promoted_before = 0, erased = 0;  // block-uniform running totals
        bool map_ovf = false;
        __syncthreads();
        for (int b0 = 0; b0 < SM;) {
            int acc[2];
            int used = resolve_super(b0, SM, [&](int q) { return S.svis[q] ? S.sncand[q] : 0; }, S.scand, L, iter, S.prm.track_ratio,
                                     S.prm.desc_th, acc);
            if (used < 0) {
                if (wave_id() == 0) {
                    const int idx = decide_query_slow<MODE_STAGED>(S, T, N, L.flag, A.desc, b0, S.sproj[2 * b0], S.sproj[2 * b0 + 1],
                                                                   S.prm.tracking_radius, S.prm.track_ratio, S.prm.desc_th);
                    if (lane_id() == 0) {
                        if (idx >= 0) {
                            L.flag[idx] = 1;
                            L.tab0[idx] = L.tab1[idx] = PERM;
                        }
                        L.misc[1] = idx;
                    }
                }
                __syncthreads();
                acc[0] = (tid == 0) ? L.misc[1] : -1;
                acc[1] = -1;
                used = 1;
                __syncthreads();
            }
            int m[2], c[2], pr[2];
#pragma unroll
            for (int u = 0; u < 2; u++) m[u] = (tid + u * RES_THREADS < used && acc[u] >= 0) ? 1 : 0;
            int mt0, mt1 = 0, me1 = 0;
            const int me0 = block_excl_scan(m[0], L.scan, &mt0);
            if (used > RES_THREADS) me1 = block_excl_scan(m[1], L.scan, &mt1);
            const int mtot = mt0 + mt1;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                pr[u] = 0;
                c[u] = 0;
                if (m[u]) {
                    const int i = b0 + tid + u * RES_THREADS;
                    c[u] = A.counter[i] + 1;
                    const int mb = matched_before + (u ? mt0 + me1 : me0);
                    pr[u] = (c[u] == S.prm.staged_th || map_n0 + mb < N_MAP_POINTS) ? 1 : 0;  // :377
                }
            }
            int pt0, pt1 = 0, pe1 = 0;
            const int pe0 = block_excl_scan(pr[0], L.scan, &pt0);
            if (used > RES_THREADS) pe1 = block_excl_scan(pr[1], L.scan, &pt1);
            const int ptot = pt0 + pt1;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int lq = tid + u * RES_THREADS, i = b0 + lq;
                if (lq >= used) continue;
                bool del = true;
                if (m[u]) {
                    A.counter[i] = c[u];
                    del = false;
                    if (pr[u]) {
                        const int dst = map_n0 + promoted_before + (u ? pt0 + pe1 : pe0);
                        if (dst < MAP_MAX) {
                            copy_point(A, i, MP, dst);
                            MP.counter[dst] = c[u];
                            MP.match_idx[dst] = -1;
                        } else
                            map_ovf = true;
                        del = true;
                    }
                }
                S.sdel[i] = del ? 1 : 0;
            }
            matched_before += mtot;
            promoted_before += ptot;
            erased += used - mtot;
            b0 += used;
            __syncthreads();
        }
        __syncthreads();
        if (map_ovf) atomicOr(&ctl.overflow, OVF_MAP);
        if (tid == 0) {
            *S.map_n = min(map_n0 + promoted_before, MAP_MAX);
            ctl.counts[C_N_STAGED_ERASED] = erased;
            ctl.counts[C_N_STAGED_PROMOTED] = promoted_before;
        }
        for (int j = tid; j < N; j += RES_THREADS) S.fb[par].feat[0].flag[j] = L.flag[j];
        // stable compaction of the staged set
        int n_out = 0;
        for (int base = 0; base < SM; base += RES_THREADS) {
            const int i = base + tid;
            const bool keep = (i < SM) && !S.sdel[i];
            int total;
            const int off = n_out + block_excl_scan(keep ? 1 : 0, L.scan, &total);
            if (keep) copy_point(A, i, B, off);
            n_out += total;
        }
        __syncthreads();
        if (tid == 0) {
            *S.staged_cur = scur ^ 1;
            *S.staged_n = n_out;
        }
        }  // SM > 0
    }
    __syncthreads();
    if (tid == 0) {
        if (ctl.first_frame) {  // lvt_system.cpp:188
            ctl.need_tri = 1;
            ctl.dont_stage = 1;
        } else {  // lvt_system.cpp:308-334
            bool need;
            if (S.prm.tri_policy == 2) need = true;
            else if (S.prm.tri_policy == 3) need = (*S.map_n < 1000);
            else {
                need = true;
                const float ratio = 0.99f;
                for (int i = 2; i > 0; --i)
                    if ((float)ctl.last_matches[i] > ratio * (float)ctl.last_matches[i - 1]) need = false;
            }
            ctl.need_tri = need ? 1 : 0;
            ctl.dont_stage = 0;
        }
    }
}

// =================================================================================================
// k_triangulate : lvt_local_map.cpp:231-353
// =================================================================================================
__device__ bool ls_solve_4x3(double A[4][3], double b[4], double x[3]) {
    for (int k = 0; k < 3; k++) {
        double norm = 0;
        for (int i = k; i < 4; i++) norm += A[i][k] * A[i][k];
        norm = sqrt(norm);
        if (norm < 1e-300) return false;
        const double alpha = (A[k][k] > 0) ? -norm : norm;
        double v[4] = {0, 0, 0, 0};
        for (int i = k; i < 4; i++) v[i] = A[i][k];
        v[k] -= alpha;
        double vnorm2 = 0;
        for (int i = k; i < 4; i++) vnorm2 += v[i] * v[i];
        if (vnorm2 > 0) {
            for (int j = k; j < 3; j++) {
                double dot = 0;
                for (int i = k; i < 4; i++) dot += v[i] * A[i][j];
                const double f = 2.0 * dot / vnorm2;
                for (int i = k; i < 4; i++) A[i][j] -= f * v[i];
            }
            double dot = 0;
            for (int i = k; i < 4; i++) dot += v[i] * b[i];
            const double f = 2.0 * dot / vnorm2;
            for (int i = k; i < 4; i++) b[i] -= f * v[i];
        }
    }
    const double rmax = fmax(fabs(A[0][0]), fmax(fabs(A[1][1]), fabs(A[2][2])));
    for (int k = 0; k < 3; k++)
        if (!(fabs(A[k][k]) > 1e-12 * rmax)) return false;
    x[2] = b[2] / A[2][2];
    x[1] = (b[1] - A[1][2] * x[2]) / A[1][1];
    x[0] = (b[0] - A[0][1] * x[1] - A[0][2] * x[2]) / A[0][0];
    return true;
}

__device__ bool triangulate_pair(const Params &p, const double *cml, const double *cmr, float u1x, float u1y, float u2x, float u2y, double out[3]) {
    const double cx = p.cx, cy = p.cy;
    const double inv_fx = 1.0 / p.fx, inv_fy = 1.0 / p.fy;
    const double a1x = (u1x - cx) * inv_fx, a1y = (u1y - cy) * inv_fy;
    const double a2x = (u2x - cx) * inv_fx, a2y = (u2y - cy) * inv_fy;
    double A[4][3], rhs[4];
    for (int j = 0; j < 3; j++) {
        A[0][j] = a1x * cml[8 + j] - cml[j];
        A[1][j] = a1y * cml[8 + j] - cml[4 + j];
        A[2][j] = a2x * cmr[8 + j] - cmr[j];
        A[3][j] = a2y * cmr[8 + j] - cmr[4 + j];
    }
    rhs[0] = -(a1x * cml[11] - cml[3]);
    rhs[1] = -(a1y * cml[11] - cml[7]);
    rhs[2] = -(a2x * cmr[11] - cmr[3]);
    rhs[3] = -(a2y * cmr[11] - cmr[7]);
    double x[3];
    if (!ls_solve_4x3(A, rhs, x)) return false;
    // the two camera matrices are read from LDS AGAIN for the visibility tests (the pointers are laundered: the compiler would otherwise keep all 24
    // doubles in registers across the least-squares solve, and a 1024-thread workgroup has 128 VGPRs per lane: 19 of them went to scratch)
    asm volatile("" : "+v"(cml), "+v"(cmr));
    double ul, vl, ur, vr;
    if (!is_point_visible(x, cml, p, ul, vl) || !is_point_visible(x, cmr, p, ur, vr)) return false;
    {
        const double ex = ul - u1x, ey = vl - u1y;
        if ((ex * ex + ey * ey) > REPROJ_TH2) return false;
    }
    {
        const double ex = ur - u2x, ey = vr - u2y;
        if ((ex * ex + ey * ey) > REPROJ_TH2) return false;
    }
    out[0] = x[0], out[1] = x[1], out[2] = x[2];
    return true;
}

template <bool BV>
__global__ __launch_bounds__(1024) void k_triangulate(SeqArg<BV> sa, int par, seq_t seq, Ctl *rec_out, seq_t *done_out, int row_gated, int deliver) {
    const Seq &S = sa.get();
    Ctl &ctl = *S.ctl;
    __shared__ double cml[12], cmr[12], R[9];
    __shared__ Pose cam;
    __shared__ int scan[32];
    const int tid = threadIdx.x;
    const long long tq0 = clock64();
    long long tq1 = tq0, tq2 = tq0;
    RESOLVE_LDS_DECL
    if (ctl.active && !ctl.lost_now) {  // update_staged_map_points + the triangulation policy (block-uniform condition)
        staged_body(S, ctl, par, L, r_tab);
        __syncthreads();  // need_tri / dont_stage / map_n / staged_n / the marks: written above, read below by other threads
    }
    tq1 = clock64();
    const bool run = ctl.active && !ctl.lost_now && ctl.need_tri;  // block-uniform
    if (run) {
    int n_pairs = 0;
    if (S.prm.sensor == 1) {  // row_match (lvt_image_features_handler.cpp:299-326): greedy resolution of the lists k_candidates<ROW> built
        // (on the early stream behind k_early_mid: normally finished ~70 us ago.  One workgroup polling holds one CU.  2 s without
        //  the lists: nothing valid to triangulate from -> LOST, reported)
        if (row_gated) {
            // (the lists are normally ~70 us old by now.  5 ms without them -- the early stream's gate stood down, or the stream is
            //  held up -- and this workgroup builds them itself: the features are complete (this frame's gate has seen them), the lists
            //  are a function of the two feature sets only, so a late k_candidates<ROW> writing the same words is harmless.  A
            //  time-out costs time, never the track.)
            __shared__ int s_row_fallback;
            if (tid == 0) {
                FeatCtl &fc = *S.fb[par].fc;
                const unsigned long long t0 = wall_clock64();
                int fb = (row_gated == 2) ? 1 : 0;  // (2: tests force the fallback -- it then runs beside the early stream's list kernel)
                while (!fb && __hip_atomic_load(&fc.row_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) {
                    __builtin_amdgcn_s_sleep(8);
                    if (wall_clock64() - t0 > 500000ull) {
                        atomicAdd(&ctl.gate_timeouts, 1);
                        fb = 1;
                        break;
                    }
                }
                if (fb) ctl.counts[C_ROW_FALLBACK] = 1;
                s_row_fallback = fb;
            }
            __syncthreads();
            if (s_row_fallback) {
                CandLds C;  // carve the (not yet used) list area of the resolver, as k_track_mid's second pass does
                C.lbuf = r_lists;
                C.tx = reinterpret_cast<float *>(r_lists + 16 * KC);
                C.ty = C.tx + NF_MAX;
                C.tc = reinterpret_cast<uint32_t *>(C.ty + NF_MAX);
                candidates_body<MODE_ROW>(S, 0, par, C, wave_id(), RES_THREADS / 64, RES_THREADS);
                __syncthreads();
            }
        }
        resolve_body<MODE_ROW>(S, ctl, 0, par, L, r_tab);
        __syncthreads();
        n_pairs = L.misc[2];
    }
    tq2 = clock64();
    if (tid == 0) {
        if (ctl.first_frame) {
            cam.q[0] = 1, cam.q[1] = cam.q[2] = cam.q[3] = 0;
            cam.p[0] = cam.p[1] = cam.p[2] = 0;
        } else
            cam = ctl.optimized;
        Pose right;
        right_camera_pose(cam, (double)S.prm.baseline, right);
        world_to_camera(cam, cml);
        world_to_camera(right, cmr);
        q_to_R(cam.q, R);
    }
    __syncthreads();
    const Feat &FL = S.fb[par].feat[0], &FR = S.fb[par].feat[1];
    const bool rgbd = (S.prm.sensor == 2);
    const int n_in = rgbd ? *FL.n : n_pairs;
    // destination: lvt_local_map.cpp:345 (decided once, before anything is appended)
    const bool to_map = ctl.dont_stage || S.prm.staged_th == 0 || (*S.map_n < N_MAP_POINTS);
    const MapSoA &D = to_map ? S.map[*S.map_cur] : S.staged[*S.staged_cur];
    const int d_n0 = to_map ? *S.map_n : *S.staged_n;
    const int d_cap = to_map ? MAP_MAX : STAGED_MAX;
    int n_out = 0;
    for (int base = 0; base < n_in; base += 1024) {
        const int i = base + tid;
        bool ok = false;
        double X[3] = {0, 0, 0};
        int li = 0;
        if (i < n_in) {
            if (rgbd) {  // triangulate_rgbd, float then double (lvt_local_map.cpp:231-256)
                li = i;
                const float inv_fx = 1.0f / S.prm.fx, inv_fy = 1.0f / S.prm.fy;
                const float u = FL.x[i], v = FL.y[i], z = FL.depth[i];
                const float x = (u - S.prm.cx) * z * inv_fx, y = (v - S.prm.cy) * z * inv_fy;
                X[0] = ((R[0] * x + R[1] * y) + R[2] * z) + cam.p[0] * 1.0;
                X[1] = ((R[3] * x + R[4] * y) + R[5] * z) + cam.p[1] * 1.0;
                X[2] = ((R[6] * x + R[7] * y) + R[8] * z) + cam.p[2] * 1.0;
                ok = true;
            } else {
                li = S.pair_l[i];
                const int ri = S.pair_r[i];
                ok = triangulate_pair(S.prm, cml, cmr, FL.x[li], FL.y[li], FR.x[ri], FR.y[ri], X);
            }
        }
        int total;
        const int off = d_n0 + n_out + block_excl_scan(ok ? 1 : 0, scan, &total);
        if (ok && off < d_cap) {
            D.pos[3 * off] = X[0];
            D.pos[3 * off + 1] = X[1];
            D.pos[3 * off + 2] = X[2];
#pragma unroll
            for (int k = 0; k < 4; k++) D.desc[(size_t)off * 4 + k] = FL.desc[(size_t)li * 4 + k];
            D.counter[off] = 0;
            D.age[off] = 0;
            D.match_idx[off] = -1;
        }
        n_out += total;
    }
    __syncthreads();
    if (tid == 0) {
        int nn = d_n0 + n_out;
        if (nn > d_cap) {
            atomicOr(&ctl.overflow, to_map ? OVF_MAP : OVF_STAGED);
            nn = d_cap;
        }
        if (to_map) *S.map_n = nn;
        else *S.staged_n = nn;
        ctl.counts[C_TRIANGULATED] = 1;
        ctl.counts[C_N_TRIANGULATED] = n_out;
    }
    }  // run
    __syncthreads();
    // ---- epilogue of the frame (first-frame bookkeeping of lvt_system.cpp:185-193, result record)
    if (tid == 0) {
        if (ctl.active && ctl.first_frame) {
            ctl.state = 2;
            ctl.last_matches[0] = *S.map_n;
            Pose id;
            id.q[0] = 1, id.q[1] = id.q[2] = id.q[3] = 0;
            id.p[0] = id.p[1] = id.p[2] = 0;
            pose_to_Rt(id, ctl.out_R, ctl.out_t);
            ctl.out_status = 2;
        }
        // LOST: no features as far as any caller can see.  (Not for a pooled handle's idle step: the buffer still holds its last real frame's features.)
        if (!ctl.active && !(ctl.skip && S.fb[par].fc->absent)) *S.fb[par].feat[0].n = *S.fb[par].feat[1].n = 0;
        ctl.counts[C_N_LEFT] = *S.fb[par].feat[0].n;
        ctl.counts[C_N_RIGHT] = *S.fb[par].feat[1].n;
        ctl.counts[C_MAP_SIZE] = *S.map_n;
        ctl.counts[C_STAGED_SIZE] = *S.staged_n;
        ctl.overflow |= S.fb[par].fc->overflow;
        ctl.counts[C_OVERFLOW] = ctl.overflow;
        ctl.skip = 0;  // (a skipped frame ends here: every workgroup of k_match_map has read the flag long ago)
        ctl.dbg[46] = (long long)wall_clock64();
        if (run) {  // bring-up stamps of a triangulation frame (tools/cells_phases.py): staged update, row resolution, the rest
            ctl.dbg[20] = tq1 - tq0, ctl.dbg[21] = tq2 - tq1, ctl.dbg[22] = clock64() - tq2;
        }
        __threadfence();
        atomicExch(&ctl.track_done_seq, seq);  // this frame's feature buffer may be refilled (k_gate_buf polls this)
    }
    // ---- the frame's result record (synchronous calls and the events-only ordering: here; otherwise by the next frame's k_gate_late)
    __syncthreads();
    if (deliver) deliver_record(ctl, rec_out + blockIdx.z, done_out + blockIdx.z, seq, 1024);
    if (tid == 0) ctl.dbg[43] = (long long)wall_clock64();  // (tools/timeline.py: lands in the NEXT frame's record)
}

// explicit instantiations used by the host

}  // namespace lvt
