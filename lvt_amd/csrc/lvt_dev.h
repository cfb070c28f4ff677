// lvt_dev.h -- device-resident data layout shared by all kernels of the MI355X tracking path.
//
// One `Seq` per tracked sequence lives in HBM.  Every kernel is launched with gridDim.z = number of
// sequences in the context, so B independent sequences advance in lock-step through ONE launch chain
// (a single lvt_handle is a context with B = 1).  All per-frame control flow of the reference's
// lvt_system::track / perform_tracking (lvt_system.cpp:157-306) is evaluated ON DEVICE from the `Ctl`
// block; the host enqueues the same fixed chain every frame and reads back 1 small result record.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lvt {

// ---- capacities -------------------------------------------------------------------------------
constexpr int NF_MAX = 4096;       // features per image after BRIEF's border filter
constexpr int CELLS_MAX = 64;      // detection grid cells per image
constexpr int CELL_OUT_CAP = 2048; // key points one cell may emit (after ANMS)
constexpr int RAW_CAP = 10240;     // raw (pre-NMS) corners one cell may hold in LDS
constexpr int MAP_MAX = 32768;     // local map points
constexpr int STAGED_MAX = 16384;  // staged points
constexpr int KC = 128;            // candidate-list capacity per query (overflow -> exact slow path)
constexpr int N_COUNTS = 32;

// overflow bits (Ctl::overflow, reported through lvt_amd_get_counts / lvt_amd_last_error)
enum : int { OVF_RAW = 1, OVF_CELL_OUT = 2, OVF_FEATURES = 4, OVF_MAP = 8, OVF_STAGED = 16, OVF_CELL_DIM = 32 };

// counts slots -- identical to the oracle's LVTO_C_* so tests can diff them 1:1
enum : int {
    C_N_LEFT = 0, C_N_RIGHT, C_MAP_SIZE, C_STAGED_SIZE, C_N_MATCHES, C_SECOND_PASS, C_N_ROW_MATCHES,
    C_N_TRIANGULATED, C_TRIANGULATED, C_RETRY_LEFT, C_RETRY_RIGHT, C_PNP_ITERS, C_PNP_INLIERS,
    C_MAP_SIZE_AT_MATCH, C_N_STAGED_ERASED, C_N_STAGED_PROMOTED, C_N_CULLED, C_FRAME, C_OVERFLOW, C_PNP_BORDERLINE,
    C_ROW_FALLBACK,  // (HIP path only) k_triangulate built the row-match lists itself: the early stream's were late (or the test knob is set)
    C_PNP_TRIALS, C_PNP_REJECTIONS, C_PNP_TERMINATES  // LM trials of the frame's pose refinement, rejected ones (rho <= 0: lambda *= ni, pop()), passes ended by Terminate
};

// constants of the reference -- lvt/src/lvt_definitions.h:29-34
constexpr double REPROJ_TH2 = 5.991;
// k_pnp's chi2 gates: an edge whose squared error lies this close to the threshold is counted (C_PNP_BORDERLINE) and decided on an
// error re-evaluated in the reference's operation order with IEEE divisions (the sweep's own arithmetic agrees with it to ~1e-11)
constexpr double PNP_GATE_MARGIN = 1e-8;
constexpr int N_MAP_POINTS = 250;
constexpr int ROW_RADIUS = 2;
constexpr int HASH_CELL = 25;
constexpr int CORNERS_LOW_TH = 200;
constexpr int N_MATCHES_TH = 50;

struct Params {  // lvt_parameters.h:29-64 + derived values
    float fx, fy, cx, cy, baseline;
    float k1, k2, p1, p2, k3;
    float near_plane, far_plane;
    float tri_ratio, track_ratio, desc_th;
    float min_x, max_x, min_y, max_y;  // image bounds (lvt_local_map.cpp:84-123), per instance
    int W, H;
    int min_matches, tracking_radius, cell_size, max_kp_cell, agast_th, agast_th_low;
    int untracked_th, staged_th, tri_policy;
    int cells_x, cells_y, n_cells;          // detection grid (lvt_image_features_handler.cpp:95-114)
    int hash_ccx, hash_ccy, cell_search_radius;  // lvt_image_features_struct.cpp:48-53
    int sensor;                             // 1 stereo, 2 rgbd
    int undistort;                          // |k1| > 1e-5 (handler.cpp:268)
    unsigned cell_magic;                    // ceil(2^32 / cell_size): x / cell_size == __umulhi(x, cell_magic) for x * cell_size < 2^32 (pixel coordinates)
    int big_cell_strips;                    // detection cells taller than 256 px: an oversized cell's NMS runs as row strips on several CUs (k_cells_strip)
};

struct Pose {  // camera-to-world, quaternion (w,x,y,z) + position
    double q[4];
    double p[3];
};

// frame sequence numbers handed between the streams: 64 bits (32 would wrap after 149 h at 8 000 frames/s, 4 * seq after 37 h)
using seq_t = unsigned long long;

struct Ctl {
    // persistent state (lvt_system.h:96-108, lvt_motion_model.h:43-46)
    int state;          // 1 NOT_INITIALIZED, 2 TRACKING, 3 LOST
    int frame_number;
    int last_matches[3];
    Pose last_pose;
    double mm_last_q[4], mm_ang_vel[4], mm_last_p[3], mm_lin_vel[3];
    double mm_next[14];  // motion-model state after this frame's prediction; committed by k_track_mid (the prologue only reads mm_*)
    int mm_pending;
    // cross-frame overlap: k_pnp of frame t publishes the size of the map after clean_untracked_points; the projection,
    // candidate lists and greedy resolution of THOSE points for frame t+1 run on a third stream while frame t still
    // updates its staged points and triangulates (they only append behind early_done)
    int early_done;      // map points [0, early_done) of the NEXT frame are handled by k_early_map / k_early_mid (0: none)
    int early_accepted;  // matches accepted among them
    // hand-over of that work between the streams without a cross-stream event (measured: ~12 us of latency per event against ~3 us
    // for an in-stream boundary): k_pnp publishes its frame's sequence number, a one-wave gate kernel at the head of the early
    // stream polls it (with a wall-clock time-out), k_early_mid confirms that the early part really ran
    seq_t pnp_seq, gate_ok, early_ran_seq;
    // ownership of frame seq's early work, 4 * seq + phase, only ever increasing: 1 = the early stream has claimed it (running),
    // 2 = it has finished with results, 3 = nothing was done (it stood down, or the tracking stream cancelled it after a
    // time-out: an early kernel that arrives later finds the claim taken and does nothing)
    seq_t early_state;
    seq_t track_done_seq;  // last frame whose tracking chain has finished with its feature buffer (polled by k_gate_buf)
    seq_t late_gate_seq;   // k_match_map's folded gate: workgroup 0 ALONE waits, decides (skip / time-outs) and publishes the frame here; the others poll it
    int gate_timeouts;  // a gate gave up waiting and its stream stood down / cancelled (results unaffected)
    int gate_fatal;     // a stream waited 2 s for data it cannot do without: the frame was SKIPPED (last pose returned, state kept)
    int skip;           // set by the tracking stream's gate for the frame it is about to start: its features never arrived; consumed
                        // (and cleared) by the frame prologue.  A time-out never turns into LOST: in the reference LOST depends on
                        // match counts only (lvt_system.cpp:267-272), and it is sticky.
    // per-frame control, written by k_begin / later kernels
    int active;         // 0: LOST at frame start -> every kernel exits
    int first_frame;    // state was NOT_INITIALIZED at frame start
    int do_pass2;       // find_matches second pass (lvt_local_map.cpp:173)
    int n_pass1, n_pass2;
    int n_matches;      // accepted 2D-3D matches
    int lost_now;       // matches < min_num_matches_for_tracking
    int need_tri;       // triangulation policy fired (or first frame)
    int dont_stage;
    int n_pairs;        // row matches
    Pose predicted, optimized;
    int counts[N_COUNTS];
    int overflow;
    long long dbg[48];  // [0..31] phase cycle stamps of single-workgroup kernels; [32..47] wall-clock (100 MHz) start / end of the
                        // kernels on the inter-frame critical path (tools/timeline.py)
    // result record copied to the host
    double out_R[9], out_t[3];
    int out_status;
};

// frames whose feature stage may be in flight / waiting for their tracking chain: three buffers let the feature stream run a
// full frame ahead of the tracking stream (with two, the features of frame t+1 only finish when frame t does)
constexpr int NPAR = 3;

struct FeatCtl {        // per-frame state of the FEATURE stage (own stream, one per feature buffer)
    int ext_corners;    // track_with_external_corners frame
    int n_ext[2];
    int n_detected[2];  // corners before BRIEF (for the <200 retry, handler.cpp:161)
    int retry[2];
    int overflow;
    seq_t row_seq;    // ... of the frame whose row-match candidate lists it holds (k_row_done; polled by k_triangulate)
    unsigned done_blocks;  // workgroups of the feature stage's last kernel that have finished (the last one publishes feat_seq and resets this)
    int absent;         // this step has no frame for the sequence (a pooled handle that did not submit one): poison without a report, the frame is not counted
    int poison;         // k_gate_buf gave up waiting for this buffer's previous user: the feature kernels of this frame must not touch it
    seq_t skip_seq;     // ... and this frame (sequence number) has no features: the tracking chain skips it
    seq_t feat_seq;  // sequence number of the frame whose features this buffer holds, published by k_feat_done (polled by k_gate)
};

struct Feat {  // one image's lvt_image_features_struct (lvt_image_features_struct.h:62-80), SoA
    float *x, *y, *resp;   // keypoint coords as the struct stores them (undistorted for RGB-D)
    float *bx, *by;        // coords BRIEF samples at (== x,y except RGB-D with distortion)
    float *depth;          // RGB-D only
    uint64_t *desc;        // [NF_MAX][4]
    uint8_t *flag;         // matched marks
    int16_t *hcx, *hcy;    // 25-px hash cell of each feature
    int *n;                // feature count (device scalar)
};

struct MapSoA {            // lvt_local_map.h:64-72 as SoA; two copies for stable compaction
    double *pos;           // [cap][3]
    uint64_t *desc;        // [cap][4]
    int *counter, *age, *match_idx;
};

// everything the feature stage of ONE frame produces / consumes; NPAR copies (frame number mod NPAR) so that the
// feature extraction of frame t+1 overlaps the tracking chain of frame t on a second HIP stream
struct FrameBuf {
    const uint8_t *img[2];
    const float *depth_img;     // RGB-D
    int img_pitch, depth_pitch; // bytes / elements
    uint8_t *score[2];          // OAST-9/16 score map (0 = below the lowered threshold / dead band)
    uint16_t *boxsum[2];        // 9x9 box sums
    uint32_t *seg_keys[2];      // [H][tiles_x][64] raw-corner keys (score >= agast_th) of one 64-px tile row, x-ascending
    uint16_t *seg_cnt[2];       // [H][tiles_x]  count | (count left of the cell boundary inside the tile) << 8
    float *cell_kp[2];          // [CELLS_MAX][CELL_OUT_CAP][3] (x, y, response)
    int *cell_n[2];             // [CELLS_MAX]
    const float *ext_xy[2];     // external corners (n_ext x 2, f32) for track_with_external_corners
    const float *ext_xy_own[2]; // the lists the context was created with (ext_xy points at them again whenever a frame brings none of its own)
    Feat feat[2];
    FeatCtl *fc;
};

struct Seq {
    Params prm;
    Ctl *ctl;
    FrameBuf fb[NPAR];
    int plane_pitch;            // elements, multiple of 64
    uint32_t *cell_scratch[2];  // global-memory arrays for cells whose raw corners exceed RAW_CAP (6 words / pixel)
    size_t cell_scratch_off[CELLS_MAX];
    // oversized cells (more raw corners than one workgroup's LDS holds) whose NMS runs as STRIPS row strips: per (cell, strip) the survivors
    // of the strip's core rows (raw-corner keys, raster order) and their count (-1: the strip could not decide exactly -> single-workgroup path)
    uint32_t *strip_kp[2];
    int *strip_n[2];
    int *cell_big[2];           // [CELLS_MAX] 1: this pass's k_cells left the cell to k_cells_strip / k_cells_big
    int *lists_fb;              // [2] k_hamming_batched_lists (map / row) stood down: the wave-per-query kernel behind it builds the lists
    // map + staged, ping-pong
    MapSoA map[2], staged[2];
    int *map_cur, *map_n, *staged_cur, *staged_n;   // device scalars
    // find_matches temporaries
    float *proj;                // [MAP_MAX][2] projected (f32) positions
    int8_t *vis;                // [MAP_MAX]
    int *match;                 // [MAP_MAX]  -2 invisible, -1 none, else feature index
    uint32_t *cand;             // [MAP_MAX][KC] packed (dist << 16 | idx), ascending
    int *ncand;                 // [MAP_MAX]  (> KC => overflow, slow path)
    uint32_t *qidx;             // [MAP_MAX]  resolver scratch: the map points that have candidates, in storage order (index | count << 24)
    // staged temporaries
    float *sproj; int8_t *svis; int *smatch; uint32_t *scand; int *sncand; uint8_t *sdel;
    // pnp input
    double *pnp_X; float *pnp_obs; int *pnp_feat; double *pnp_err; int8_t *pnp_level;
    // row matching / triangulation
    uint32_t *rcand; int *rncand;   // [NPAR buffers][NF_MAX][KC]: built on the feature stream
    int *pair_l, *pair_r;           // [NF_MAX]
    double *tri_X; int8_t *tri_ok;  // [NF_MAX][3]
};

// ---- small device helpers ----------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ int hamming256(const uint64_t a[4], const uint64_t b[4]) {
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

// inclusive wave scan (64 lanes), all lanes active.  DPP row shifts inside each row of 16 lanes, then the two row
// broadcasts (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3): six VALU instructions, no LDS crossbar
// (the __shfl_up form costs six ds_bpermute round trips of ~120 cycles each -- with the barriers around it a block scan
// took 2-5k cycles, and the per-cell / per-map kernels run dozens of them back to back on one CU).
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
// inclusive scan inside each row of 16 lanes only
__device__ __forceinline__ int row16_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    return v;
}

// exclusive block scan over blockDim.x (<= 1024) threads, called by every thread; `scratch` >= 16 ints of LDS; returns the
// exclusive prefix, *total receives the block sum.  Two barriers: every wave reads the <= 16 wave totals itself and scans
// them inside one DPP row, so there is no third "wave 0 publishes the prefixes" round.
__device__ __forceinline__ int block_excl_scan(int v, int *scratch, int *total) {
    const int incl = wave_incl_scan(v);
    const int w = __builtin_amdgcn_readfirstlane(wave_id()), l = lane_id();
    const int nw = (blockDim.x + 63) >> 6;
    __syncthreads();  // readers of the previous call are done with scratch
    if (l == 63) scratch[w] = incl;
    __syncthreads();
    const int part = (l < nw) ? scratch[l] : 0;
    const int pin = row16_incl_scan(part);  // lanes 0..15: inclusive prefix of the wave totals
    const int base = __builtin_amdgcn_readlane(pin, w) - __builtin_amdgcn_readlane(part, w);
    *total = __builtin_amdgcn_readlane(pin, 15);
    return base + incl - v;
}

// A sequence's descriptor reaches the kernels of the early and tracking chains (and k_cells, which touches the feature stage's fixed buffers only) either BY VALUE (single sequence: its fields are
// kernel arguments, fetched with the kernel's own argument load) or as an element of the device array (lock-step batch: one more
// dependent memory hop at every kernel head).  These kernels never read the per-frame mutable fields of Seq (FrameBuf::img ...),
// which only the feature stage writes and reads.
template <bool BYVAL>
struct SeqArg;
// Round 6: the element of the device array is read through the CONSTANT address space.  A pointer loaded from generic / global memory is a FLAT pointer to the
// compiler, and every access through it a flat_load / flat_store: those count on lgkmcnt as well as vmcnt, so each wait for an LDS read also waited for every
// global load and store in flight (the LDS-latency-bound kernels of a batch -- k_cells, the resolvers, k_triangulate, the list kernels -- paid a memory round trip
// per LDS access; the by-value form never did: its pointers come out of the kernel arguments).  A pointer loaded from constant memory is taken to be a global one
// (AMDGPUTargetMachine::getAssumedAddrSpace), the loads of the record's fields become scalar loads.  Valid here: the fields these kernels read are written
// by the host at creation and by kernels that finished before this one started (never during it).
typedef const __attribute__((address_space(4))) Seq *SeqConstPtr;
__device__ __forceinline__ const Seq &seq_const(const Seq *p, unsigned i) { return *(const Seq *)((SeqConstPtr)p + i); }
template <>
struct SeqArg<false> {
    const Seq *p;
    __device__ __forceinline__ const Seq &get() const { return seq_const(p, blockIdx.z); }
};
template <>
struct SeqArg<true> {
    Seq v;
    __device__ __forceinline__ const Seq &get() const { return v; }
};


}  // namespace lvt
