// lvt_host.hip -- host side of the MI355X-native LVT tracking path: context/arena management, the fixed
// per-frame launch chain, the reference-compatible C-ABI (include/lvt_c.h) and the additive entry
// points (include/lvt_amd_ext.h).  The host mirrors lvt_system::create/track/reset
// (reference lvt/src/lvt_system.cpp) but performs NO tracking arithmetic: the whole state machine
// runs on the device, the host enqueues the chain and reads back one result record per frame.
//
// Three HIP streams form a software pipeline: the FEATURE stage (k_score .. k_brief, wide kernels) runs up
// to two frames ahead on stream_f; the TRACKING chain (matching of the newest map points, the serial Levenberg-Marquardt pose
// refinement, map maintenance -- latency-bound, a few workgroups) runs frame after frame on `stream`; the EARLY stream starts the
// next frame's map matching the moment a frame's pose exists and builds the row-match candidate lists.  Feature buffers are
// triple buffered; the streams hand over through sequence numbers polled by one-wave gate kernels (LVT_AMD_ORDERING=events: event
// barriers instead).  Results land in a ring of pinned records with completion flags the host polls, so frames can be enqueued
// asynchronously (lvt_amd_track_device_async / lvt_amd_wait).
#define LVT_EXPORT_FUNCTIONS
#include "../../include/lvt_amd_ext.h"

#include "k_features.hip"
#include "k_track.hip"
#include "k_hamming.hip"
#include "k_lists.hip"
#include "k_rectify.hip"
#include "lvt_odometry.h"

#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <functional>
#include <vector>

namespace lvt {

#define HIPCHK(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            (ctx)->set_error(std::string(#call) + ": " + hipGetErrorString(e__));          \
            throw std::runtime_error((ctx)->err);                                          \
        }                                                                                  \
    } while (0)

constexpr int EXT_MAX = 16384;
// workgroups of k_candidates<ROW> on the early stream: 256 of them (one per CU, as on the feature stream) cost 6 % of the frame
// rate -- they share SIMDs and LDS with the single-workgroup kernels of the tracking chain running beside them; 16 make the kernel
// itself too slow (measured: 24 / 32 / 40 / 48 / 64 / 256 -> 8 180 / 8 330 / 8 320 / 8 270 / 8 240 / 7 840 frames/s)
constexpr int ROW_BLOCKS = 32;
constexpr int MATCH_BLOCKS_GATED = 16;  // k_match_map when it polls for the early stream itself (single sequence): see k_track.hip (8 / 16 / 32: 8 840 / 8 890 / 8 900 frames/s; fewer parked workgroups leave more CUs to other processes on the GPU)
constexpr int ROW_BLOCKS_BATCH = 16;  // per sequence of a lock-step batch (16 sequences: 8 / 16 / 32 / 64 / 256 -> 30.8k / 36.9k / 35.8k / 35.0k / 31.1k frames/s)
constexpr int RING = 8;  // frames that may be in flight / un-collected
// workgroups per sequence of the binned list kernels (k_lists.hip: equal parts of the queries; LVT_AMD_LISTS_WGS=row,map overrides).  Measured in round 4
// (64 sequences, HIP-event time of the row-list launch): 2 / 8 / 16 workgroups per sequence = 113 / 151 / 241 us; lock-step batches of 16 / 32 sequences
// 68.5k / 88.8k, 68.2k / 82.4k, 59.2k / 77.1k frames/s -- every workgroup stages the whole train set again, and from 32 sequences on the chip is full anyway
constexpr int LISTS_WGS_ROW = 2, LISTS_WGS_MAP = 1;
// handles per DEVICE whose k_match_map may poll for the early stream itself: each parks MATCH_BLOCKS_GATED workgroups with 50 KB of LDS;
// 4 x 16 of them still leave most CUs with the 159 KB a k_cells workgroup needs, more handles use the separate one-wave gate kernel
constexpr int MAX_FOLDED_GATES = 4;

// first kernel of the feature stage of a BATCH: publish every sequence's inputs (a single sequence gets them as a kernel
// argument of k_score)
__global__ void k_feat_begin(Seq *seqs, const FrameArgs *fa, int par) {
    if (threadIdx.x != 0) return;
    feat_begin(seqs[blockIdx.x], fa[blockIdx.x], par);
}
// the same with the sequences' inputs in the kernel's own arguments (up to FEAT_PACK sequences): no read of pinned host memory over PCIe at the head of the feature stage
constexpr int FEAT_PACK = 32;
struct FrameArgsPack {
    FrameArgs a[FEAT_PACK];
};
// ... and with k_gate_buf's wait in front (want != 0: the tracking chain of the frame that used this buffer last must have released it): one launch at the head of a
// batch's feature stage instead of two
__global__ void k_feat_begin_pack(Seq *seqs, FrameArgsPack pk, int par, seq_t want) {
    if (threadIdx.x != 0) return;
    if (want) gate_buf_wait(seq_const(seqs, blockIdx.x), want, par);
    const FrameArgs f = pk.a[blockIdx.x];
    feat_begin(seqs[blockIdx.x], f, par);
}

// tightly packed host-layout images (stride == cols) -> pitched device images.
// Host images enter through a pinned staging buffer -- or, when the caller's buffer is page-locked itself, in place -- that this
// kernel reads over PCIe and writes as the pitched planes k_score wants: blockIdx.y = image.  One launch replaces two pageable H2D
// copies + two re-pitch kernels (0.32 ms per stereo pair -> see DESIGN.md section 6).  The source may start at ANY byte address and
// hold any byte count (1241 x 376 = 466 616 is 8 mod 16): bytes up to the first 16-byte boundary and behind the last whole vector
// travel one by one, everything between as aligned 16-byte loads -- no load reaches outside [src, src + n).
__global__ __launch_bounds__(256) void k_stage_in(const uint8_t *src0, const uint8_t *src1, uint8_t *dst0, uint8_t *dst1, int W, int H, int pitch) {
    stage_in_body(blockIdx.y ? src1 : src0, blockIdx.y ? dst1 : dst0, W, H, pitch, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}
// the (unpitched) fp32 depth image: 16-byte loads between the first 16-byte boundary of the source and its last whole vector, single
// floats on either side; the destination is written float by float (it is 16-byte aligned, the source need not be)
__global__ __launch_bounds__(256) void k_stage_copy(const float *src, float *dst, size_t n) {
    const size_t head = min((size_t)(((16 - ((uintptr_t)src & 15)) & 15) / 4), n);
    const size_t nv = (n - head) / 4, tail0 = head + nv * 4;
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t v = t0; v < nv; v += stride) {
        const float4 q = *reinterpret_cast<const float4 *>(src + head + v * 4);
        float *d = dst + head + v * 4;
        d[0] = q.x, d[1] = q.y, d[2] = q.z, d[3] = q.w;
    }
    if (t0 < head) dst[t0] = src[t0];
    if (t0 < n - tail0) dst[tail0 + t0] = src[tail0 + t0];
}

// handles alive in this process: a handle whose k_match_map polls for its early stream parks 32 workgroups (50 KB of LDS each)
// on the GPU; with many handles on one device that would leave no CU with enough free LDS for the 159-KB k_cells workgroups
// they are waiting for -- beyond 4 handles the separate one-wave gate kernel is used again (enqueue_frame)
// Counted PER DEVICE: handles on different GPUs of one process share no hardware queue and no CU.
constexpr int MAX_DEVICES = 64;
static std::atomic<int> g_live_contexts[MAX_DEVICES];

// An lvt_handle is either a Context (a sequence, or a lock-step batch, with its own launch chain) or a PoolSlot (lvt_pool.h: one sequence of a
// lock-step batch SHARED by the pooled handles of a device); the first word tells them apart.
constexpr uint64_t CTX_MAGIC = 0x4C56545F43545831ull, SLOT_MAGIC = 0x4C56545F534C5431ull;

struct Context {
    uint64_t magic = CTX_MAGIC;
    double host_enq_us = 0, host_wait_us = 0;  // host time spent enqueueing / blocking (LVT_AMD_HOST_TIMING=1 prints it at destroy)
    long host_enq_n = 0, host_wait_n = 0;
    int gate_timeouts_seen = 0, gate_fatal_seen = 0;
    // a gate that ran into its time limit means that the streams do not run side by side (a tool serialises the dispatches, or the streams share a
    // hardware queue): the handle then moves to event ordering by itself, once, at the next frame it enqueues (unless LVT_AMD_ORDERING=polling insists)
    bool want_events = false, ordering_forced = false;
    hipEvent_t ev_switch = nullptr, ev_switch_e = nullptr;
    long test_gate_timeout = 0, switched_at = -1;  // LVT_AMD_TEST_GATE_TIMEOUT=n: the early gate of frame n reports a time-out
    long long planes_in_place = 0, planes_staged = 0;  // host-buffer entry points: image / depth planes read in place (page-locked caller buffers) / copied through the staging buffer
    int device = 0;            // the HIP device that owns every allocation, stream and event of this handle (recorded at creation)
    int B = 1;                 // sequences advanced in lock-step by one launch chain
    int launch_seqs = 0;       // > 0: the NEXT step is launched for the first launch_seqs sequences only (a pool with empty upper seats)
    int ring_seqs[8] = {};     // ... and what every un-collected step was launched for
    int sensor = 1;
    Params prm{};
    hipStream_t stream = nullptr;    // tracking chain
    hipStream_t stream_f = nullptr;  // feature stage
    hipStream_t stream_e = nullptr;  // early part of find_matches of the next frame (behind the previous frame's k_pnp)
    bool own_stream = false;
    std::vector<void *> allocs;
    Seq *d_seqs = nullptr;
    std::vector<Seq> h_seqs;   // host mirror (device pointers inside)
    std::vector<Ctl *> d_ctl;
    Ctl *h_ctl = nullptr;          // pinned, RING x B records
    Ctl *h_ctl_dev = nullptr;      // the same memory as the device sees it (k_triangulate writes each frame's record there)
    // completion flags, pinned + coherent, RING x B: k_triangulate stores the frame's sequence number behind its record and the
    // host polls it -- an event record on the tracking stream costs that stream 3-4 us per frame (measured), and nothing else
    // on the device needs the event in the normal mode
    seq_t *h_done = nullptr, *h_done_dev = nullptr;
    // synchronous calls: k_pnp hands the pose over as soon as it exists (PoseRec + flag, same pinned scheme), the call returns on it
    // and the frame's tail finishes behind the caller's next upload; the frame is collected (full record) by the next entry point
    PoseRec *h_pose = nullptr, *h_pose_dev = nullptr;
    seq_t *h_pose_done = nullptr, *h_pose_done_dev = nullptr;
    bool binned_lists = false; // lock-step batches: k_hamming_batched_lists builds the early map lists and the row lists (LVT_AMD_BINNED_LISTS=0 / 1)
    bool early_pose = true;    // LVT_AMD_SYNC_TAIL=wait: synchronous calls return only when the whole frame is done
    bool early_pending = false;  // the last synchronous call returned on its early pose: its frame is still un-collected (and is nobody's to wait for)
    FrameArgs *h_fargs = nullptr;  // pinned, RING x B
    hipEvent_t ev_done[RING] = {};  // the only events of the normal mode: the streams hand over through polling gates (k_gate*)
    // LVT_AMD_ORDERING=events: the streams are ordered by event barriers only and the early stream is not used (the tracking chain
    // does all the matching).  Slower (~6 300 instead of ~7 000 frames/s), but free of kernels that wait for another queue's
    // kernels -- required under tools that serialise the dispatches of all queues (rocprofv3 --pmc): a polling gate then holds
    // the only dispatch slot while what it waits for cannot start.
    bool events_only = false;
    CellOrder cell_order;   // detection cells by decreasing area: the order a batch's k_cells workgroups start in
    // A batch's k_score is thousands of small workgroups: while its grid is being dispatched, every CU slot that frees up goes to the next of them, and the
    // tracking / early chains' workgroups (56 - 133 KB of LDS each) wait until it is through (rocprofv3 timeline: k_match_map 57 - 64 us beside it, 12 alone).
    // In TWO launches the chip drains once in the middle and they get in: +4 .. 8 % for 8 - 24 KITTI-shaped sequences, -6 % from 28 on, where the feature
    // stage is the longer chain and the drain is pure loss.  Which side a batch is on shows in its own early gate (Ctl::dbg[32 / 35 / 33]: start, features
    // seen, pose seen): a gate that mostly waits for the previous pose says the feature stream is ahead.  LVT_AMD_SCORE_PIECES=1 / 2 fixes the choice.
    int brief_from_image = 0;  // LVT_AMD_BRIEF_FROM_IMAGE=1: no box-sum plane; k_brief_img builds the 9 x 9 sums of a key point's patch in LDS (k_features.hip)
    int score_pieces = 1;
    bool score_pieces_auto = true;
    double gate_balance_us = 0;  // running mean of (wait for the pose) - (wait for the features)
    // raw corners a k_cells workgroup holds in LDS.  A batch with more cell workgroups than CUs (16 KITTI-shaped sequences: 320 on 256) runs them with room for
    // RAW_CAP_SMALL corners -- 80 KB of LDS instead of 159, TWO workgroups per CU (k_cells is held to 64 VGPRs for that: at 66 it ran one per CU whatever its LDS) --
    // and a cell with more raw corners than that takes the exact global-memory path, as one beyond RAW_CAP does (LVT_AMD_CELLS_RAW_CAP overrides)
    int cells_raw_cap = RAW_CAP;
    unsigned cell_token = 0;  // one per k_cells launch, never reset: what a split cell's helper workgroups publish their survivors under
    int cell_split = 2;     // workgroups a single sequence's tall detection cell is cut into (cells_work_split; LVT_AMD_CELL_SPLIT=0 | 2 | 3)
    int lists_wgs_row = LISTS_WGS_ROW, lists_wgs_map = LISTS_WGS_MAP;
    int match_blocks_batch = 32;   // workgroups per sequence of a batch's k_match_map (it lists the points appended since the early part: none on most frames,
                                   // and every workgroup's thread 0 recomputes the prediction before it can leave): 256 -> 32 = +6 % frames/s at 16 sequences
                                   // (LVT_AMD_MATCH_BLOCKS_BATCH overrides)
    bool force_row_fallback = false;  // LVT_AMD_TEST_ROW_FALLBACK=1 (tests): k_triangulate does not wait for the early stream's row lists, it builds them itself
                                      // WHILE the early stream's kernel writes the same words -- the situation its 5-ms time-out leads to
    hipEvent_t ev_feat[NPAR] = {};
    hipEvent_t ev_depth = nullptr;  // RGB-D host-buffer calls: the depth plane is pulled on the early stream beside the feature kernels; k_gather waits for it
    bool depth_wait = false;
    std::function<void()> before_gather;  // host work to run once the detection kernels are enqueued (the RGB-D depth upload)
    // owned staging for the host-buffer entry points, per feature buffer
    uint8_t *d_img[NPAR][2] = {};
    uint8_t *h_stage[NPAR] = {}, *h_stage_dev[NPAR] = {};  // pinned staging of host images (lvt_track): [left | right or depth]
    size_t stage_img = 0;                                    // bytes reserved per 8-bit image (16-B multiple)
    float *d_depth[NPAR] = {};
    // asynchronous host-buffer calls (lvt_amd_track_async / lvt_amd_track_rgbd_async): one image set and one staging buffer per RING slot -- frame t's
    // slot was last used by frame t - RING, which make_room() has collected, so neither the pull (a write) nor the CPU copy into the staging
    // buffer needs a device-side hand-over.  The pull kernel runs at the head of the frame's feature stage, on the feature stream: a stream of its own
    // (4 400 against 8 100 frames/s), the null stream or other priorities (6 100) and the copy engine (5 600) were measured and removed
    // (profiles/r04_async_host.md; the variants are in the history at 7f39175).
    uint8_t *d_img_ring[RING][2] = {};
    float *d_depth_ring[RING] = {};
    uint8_t *h_stage_ring[RING] = {}, *h_stage_ring_dev[RING] = {};
    size_t stage_ring_bytes = 0;
    long long async_frames = 0;
    // A stereo frame handed to lvt_amd_track_async is HELD until the next one arrives (or somebody waits for it): the held frame's k_cells launch then carries
    // the workgroups that pull the NEXT frame's images (k_features.hip, NextPull), and a frame whose images came that way starts without a pull of its own.
    struct PendingFrame {
        bool valid = false, pulled = false;
        FrameArgs f{};
        const uint8_t *src[2] = {nullptr, nullptr};
        uint8_t *dst[2] = {nullptr, nullptr};
    } pend;
    NextPull next_pull{};   // what this enqueue_frame's k_cells pulls (src[0] == nullptr: nothing)
    bool fuse_pull = true;  // LVT_AMD_FUSED_PULL=0: every frame pulls its own images at the head of its feature stage
    long long fused_pulls = 0;
    int feature_cus = 0;    // CUs the feature stream is confined to (0: all; lock-step batches, see create_context)
    float *d_ext[NPAR][2] = {};
    int pitch = 0;
    long enq = 0, done = 0;    // frames enqueued / collected
    long delivered = 0;        // frames whose record delivery has been enqueued (k_triangulate, the next frame's k_gate_late, or k_deliver)
    bool sync_call = false;    // the frame being enqueued is collected right away: k_triangulate delivers its record itself
    int last_slot = 0, last_par = 0;
    std::string err;
    // optional per-kernel timing with HIP events on the launch streams (lvt_amd_profile_*)
    bool prof = false;
    static constexpr int PROF_SLOTS = 24;
    hipEvent_t ev[PROF_SLOTS][2] = {};
    bool ev_used[PROF_SLOTS] = {};
    double prof_ms[PROF_SLOTS] = {};
    long prof_calls[PROF_SLOTS] = {};

    void set_error(const std::string &s) { err = s; }

    template <typename T>
    T *dalloc_uncached(size_t n) {  // device memory no cache holds (MTYPE UC): what one workgroup stores, another of the same launch may read without fences
        void *p = nullptr;
        size_t bytes = ((n * sizeof(T) + 255) / 256) * 256;
        if (bytes == 0) bytes = 256;
        HIPCHK(this, hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached));
        HIPCHK(this, hipMemset(p, 0, bytes));
        allocs.push_back(p);
        return static_cast<T *>(p);
    }
    template <typename T>
    T *dalloc(size_t n) {
        void *p = nullptr;
        size_t bytes = ((n * sizeof(T) + 255) / 256) * 256;
        if (bytes == 0) bytes = 256;
        HIPCHK(this, hipMalloc(&p, bytes));
        HIPCHK(this, hipMemset(p, 0, bytes));
        allocs.push_back(p);
        return static_cast<T *>(p);
    }

    bool counted = false;
    void count_in(int dev) {
        device = dev;
        g_live_contexts[dev % MAX_DEVICES].fetch_add(1);
        counted = true;
    }
    int live_on_device() const { return g_live_contexts[device % MAX_DEVICES].load(); }
    Context() {}
    ~Context() {
        if (counted) g_live_contexts[device % MAX_DEVICES].fetch_sub(1);
        if (stream) (void)hipStreamSynchronize(stream);
        if (stream_f) (void)hipStreamSynchronize(stream_f);
        if (stream_e) (void)hipStreamSynchronize(stream_e);
        for (void *p : allocs) (void)hipFree(p);
        for (auto &x : h_stage_ring) if (x) (void)hipHostFree(x);
        for (auto &e : ev)
            for (auto &x : e)
                if (x) (void)hipEventDestroy(x);
        for (auto &x : ev_done) if (x) (void)hipEventDestroy(x);
        for (auto &x : ev_feat) if (x) (void)hipEventDestroy(x);
        if (ev_depth) (void)hipEventDestroy(ev_depth);
        if (ev_switch) (void)hipEventDestroy(ev_switch);
        if (ev_switch_e) (void)hipEventDestroy(ev_switch_e);
        for (auto &x : h_stage) if (x) (void)hipHostFree(x);
        if (h_ctl) (void)hipHostFree(h_ctl);
        if (h_done) (void)hipHostFree(h_done);
        if (h_pose) (void)hipHostFree(h_pose);
        if (h_pose_done) (void)hipHostFree(h_pose_done);
        if (h_fargs) (void)hipHostFree(h_fargs);
        if (stream && own_stream) (void)hipStreamDestroy(stream);
        if (stream_f) (void)hipStreamDestroy(stream_f);
        if (stream_e) (void)hipStreamDestroy(stream_e);
    }
};

// A handle belongs to the device that was current when it was created (or the one named to lvt_amd_create_on_device).  Every entry
// point makes that device current for the duration of the call and restores the caller's: a thread-per-GPU process can
// drive its handles from any thread without calling hipSetDevice itself.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const Context *c) {
        int cur = -1;
        if (c && hipGetDevice(&cur) == hipSuccess && cur != c->device && hipSetDevice(c->device) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// ---- parameters ---------------------------------------------------------------------------------
static bool derive_params(const lvt_amd_params &in, int sensor, Params &p) {
    if (in.img_width <= 0 || in.img_height <= 0 || in.detection_cell_size <= 0 || in.max_keypoints_per_cell <= 0 ||
        in.tracking_radius <= 0 || in.agast_threshold <= 0)
        return false;  // the reference asserts these (lvt_image_features_handler.cpp:88-93)
    std::memset(&p, 0, sizeof(p));
    p.fx = in.fx, p.fy = in.fy, p.cx = in.cx, p.cy = in.cy, p.baseline = in.baseline;
    p.k1 = in.k1, p.k2 = in.k2, p.p1 = in.p1, p.p2 = in.p2, p.k3 = in.k3;
    p.near_plane = in.near_plane_distance, p.far_plane = in.far_plane_distance;
    p.tri_ratio = in.triangulation_ratio_test_threshold;
    p.track_ratio = in.tracking_ratio_test_threshold;
    p.desc_th = in.descriptor_matching_threshold;
    p.W = in.img_width, p.H = in.img_height;
    p.min_matches = in.min_num_matches_for_tracking;
    p.tracking_radius = in.tracking_radius;
    p.cell_size = in.detection_cell_size;
    p.max_kp_cell = in.max_keypoints_per_cell;
    p.agast_th = in.agast_threshold;
    p.agast_th_low = (int)((double)in.agast_threshold * 0.5 + 0.5);  // handler.cpp:165
    p.untracked_th = in.untracked_threshold;
    p.staged_th = in.staged_threshold;
    p.tri_policy = in.triangulation_policy;
    p.sensor = sensor;
    // detection grid -- handler.cpp:95-114
    p.cells_y = 1 + ((p.H - 1) / p.cell_size);
    p.cells_x = 1 + ((p.W - 1) / p.cell_size);
    p.n_cells = p.cells_x * p.cells_y;
    if (p.n_cells > CELLS_MAX) return false;
    // matching hash grid -- struct.cpp:47-53
    const float kc = (float)HASH_CELL;
    p.hash_ccx = (int)std::ceil(p.W / kc);
    p.hash_ccy = (int)std::ceil(p.H / kc);
    p.cell_search_radius = (p.tracking_radius == HASH_CELL) ? 1 : (int)std::ceil((float)p.tracking_radius / kc);
    // image bounds -- lvt_local_map.cpp:84-123 (per instance)
    if (std::fabs(p.k1) < 1e-5) {
        p.min_x = 0.0f, p.max_x = (float)p.W, p.min_y = 0.0f, p.max_y = (float)p.H;
    } else {
        float x[4], y[4];
        undistort_point(p, 0.0f, 0.0f, x[0], y[0]);
        undistort_point(p, (float)p.W, 0.0f, x[1], y[1]);
        undistort_point(p, 0.0f, (float)p.H, x[2], y[2]);
        undistort_point(p, (float)p.W, (float)p.H, x[3], y[3]);
        p.min_x = std::min(x[0], x[2]);
        p.max_x = std::max(x[1], x[3]);
        p.min_y = std::min(y[0], y[1]);
        p.max_y = std::max(y[2], y[3]);
    }
    p.undistort = (std::fabs(p.k1) > 1e-5) ? 1 : 0;
    p.cell_magic = (unsigned)((0x100000000ull + (unsigned long long)p.cell_size - 1) / (unsigned long long)p.cell_size);
    p.big_cell_strips = (p.cell_size > 256) ? 1 : 0;  // (TUM's single 2000-px cell; the 250-px cells of KITTI / EuRoC keep the two-launch feature chain)
    return true;
}

static void default_params(lvt_amd_params *p) {  // lvt_parameters.cpp:29-52
    std::memset(p, 0, sizeof(*p));
    p->fx = p->fy = p->cx = p->cy = 0.5f;
    p->near_plane_distance = 0.1f;
    p->far_plane_distance = 500.0f;
    p->triangulation_ratio_test_threshold = 0.60f;
    p->tracking_ratio_test_threshold = 0.80f;
    p->descriptor_matching_threshold = 30.0f;
    p->min_num_matches_for_tracking = 10;
    p->tracking_radius = 25;
    p->agast_threshold = 25;
    p->untracked_threshold = 10;
    p->staged_threshold = 2;
    p->detection_cell_size = 250;
    p->max_keypoints_per_cell = 150;
    p->triangulation_policy = 1;
}

// minimal reader for the reference's flat `%YAML:1.0` configs (lvt_parameters.cpp:54-93): every field
// is overwritten; a key that is missing reads as 0 (cv::FileNode default), reals round to ints.
static bool params_from_file(const char *fname, lvt_amd_params *p) {
    FILE *f = fname ? std::fopen(fname, "r") : nullptr;
    if (!f) return false;
    std::vector<std::pair<std::string, double>> kv;
    char line[1024];
    while (std::fgets(line, sizeof(line), f)) {
        char *hash = std::strchr(line, '#');
        if (hash) *hash = 0;
        char *colon = std::strchr(line, ':');
        if (!colon || line[0] == '%') continue;
        std::string key(line, colon - line);
        size_t a = key.find_first_not_of(" \t"), b = key.find_last_not_of(" \t");
        if (a == std::string::npos) continue;
        key = key.substr(a, b - a + 1);
        char *end = nullptr;
        double v = std::strtod(colon + 1, &end);
        if (end == colon + 1) continue;
        kv.emplace_back(key, v);
    }
    std::fclose(f);
    auto get = [&](const char *k) -> double {
        for (auto &e : kv)
            if (e.first == k) return e.second;
        return 0.0;
    };
    auto geti = [&](const char *k) -> int { return (int)std::nearbyint(get(k)); };
    p->fx = (float)get("fx"), p->fy = (float)get("fy"), p->cx = (float)get("cx"), p->cy = (float)get("cy");
    p->k1 = (float)get("k1"), p->k2 = (float)get("k2"), p->p1 = (float)get("p1"), p->p2 = (float)get("p2"), p->k3 = (float)get("k3");
    p->baseline = (float)get("baseline");
    p->img_width = geti("img_width"), p->img_height = geti("img_height");
    p->near_plane_distance = (float)get("near_plane_distance");
    p->far_plane_distance = (float)get("far_plane_distance");
    p->triangulation_ratio_test_threshold = (float)get("triangulation_ratio_test_threshold");
    p->tracking_ratio_test_threshold = (float)get("tracking_ratio_test_threshold");
    p->min_num_matches_for_tracking = geti("min_num_matches_for_tracking");
    p->tracking_radius = geti("tracking_radius");
    p->agast_threshold = geti("agast_threshold");
    p->untracked_threshold = geti("untracked_threshold");
    p->staged_threshold = geti("staged_threshold");
    p->descriptor_matching_threshold = (float)get("descriptor_matching_threshold");
    p->detection_cell_size = geti("detection_cell_size");
    p->max_keypoints_per_cell = geti("max_keypoints_per_cell");
    p->triangulation_policy = geti("triangulation_policy");
    return true;
}

// ---- context creation ---------------------------------------------------------------------------
static void alloc_feat(Context *c, Feat &F) {
    F.x = c->dalloc<float>(NF_MAX), F.y = c->dalloc<float>(NF_MAX), F.resp = c->dalloc<float>(NF_MAX);
    F.bx = c->dalloc<float>(NF_MAX), F.by = c->dalloc<float>(NF_MAX), F.depth = c->dalloc<float>(NF_MAX);
    F.desc = c->dalloc<uint64_t>((size_t)NF_MAX * 4);
    F.flag = c->dalloc<uint8_t>(NF_MAX);
    F.hcx = c->dalloc<int16_t>(NF_MAX), F.hcy = c->dalloc<int16_t>(NF_MAX);
    F.n = c->dalloc<int>(1);
}
static void alloc_points(Context *c, MapSoA &P, int cap) {
    P.pos = c->dalloc<double>((size_t)cap * 3);
    P.desc = c->dalloc<uint64_t>((size_t)cap * 4);
    P.counter = c->dalloc<int>(cap), P.age = c->dalloc<int>(cap), P.match_idx = c->dalloc<int>(cap);
}

static void drain(Context *c);

static void reset_state(Context *c, int only = -1) {  // lvt_system::reset (lvt_system.cpp:44-68); only >= 0: that sequence of a batch alone
    drain(c);
    if (only < 0) c->gate_timeouts_seen = 0, c->gate_fatal_seen = 0;
    for (int s = 0; s < c->B; s++) {
        if (only >= 0 && s != only) continue;
        Ctl z;
        std::memset(&z, 0, sizeof(z));
        z.state = 1;
        z.last_matches[0] = z.last_matches[1] = z.last_matches[2] = 0x7FFFFFFF;
        z.last_pose.q[0] = 1.0;
        z.mm_last_q[0] = 1.0;
        z.mm_ang_vel[0] = 1.0;
        z.optimized.q[0] = z.predicted.q[0] = 1.0;
        z.out_R[0] = z.out_R[4] = z.out_R[8] = 1.0;
        z.out_status = 1;
        z.pnp_seq = (seq_t)c->enq;  // the next frame's gate waits for this value: nothing is pending
        z.early_state = 4u * (seq_t)c->enq + 3u;
        z.track_done_seq = (seq_t)c->enq;
        z.late_gate_seq = (seq_t)c->enq;
        z.gate_timeouts = c->gate_timeouts_seen, z.gate_fatal = c->gate_fatal_seen;  // (the host compares these running counters with what it has seen)
        HIPCHK(c, hipMemcpyAsync(c->d_ctl[s], &z, sizeof(Ctl), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(c->h_seqs[s].map_n, 0, sizeof(int), c->stream));
        HIPCHK(c, hipMemsetAsync(c->h_seqs[s].map_cur, 0, sizeof(int), c->stream));
        HIPCHK(c, hipMemsetAsync(c->h_seqs[s].staged_n, 0, sizeof(int), c->stream));
        HIPCHK(c, hipMemsetAsync(c->h_seqs[s].staged_cur, 0, sizeof(int), c->stream));
        for (int r = 0; r < RING; r++) c->h_ctl[r * c->B + s] = z;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
}

static Context *create_context(const lvt_amd_params &in, int sensor, int B, int device = -1) {
    if (sensor != 1 && sensor != 2) return nullptr;
    Params prm;
    if (!derive_params(in, sensor, prm)) return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return nullptr;  // no CPU fallback, by design
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return nullptr;
    if (device < 0) device = cur;
    if (device >= ndev) return nullptr;
    Context *c = new Context();
    c->device = device;
    DeviceGuard guard(c);
    try {
        c->count_in(device);
        c->B = B;
        c->sensor = sensor;
        c->prm = prm;
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        {   // A lock-step batch's feature stream is confined to 7/8 of the device's CUs.  k_cells' workgroups (80 KB of LDS each, two per CU, 50 - 90 us)
            // otherwise hold EVERY CU's LDS while they run, and the single-workgroup kernels of the tracking / early chains (50 - 135 KB) -- whose loop
            // pnp -> early map -> match -> resolve -> pnp is a batch's period up to ~28 sequences -- queue behind them (k_track_mid: 36 us in the
            // pipeline, 12 alone).  Measured, frames/s with all CUs -> with 224 of 256: 8 sequences 43.5k -> 45.7k, 16: 70.1k -> 74.9k, 24: 82.4k -> 86.6k;
            // 32 sequences are bound by the feature stream itself (k_score is VALU-bound) and lose: 92.6k -> 88.6k, so larger batches keep every CU.
            // LVT_AMD_FEATURE_CUS=n overrides (0: all CUs).  (A CU-masked stream cannot be created non-blocking: it synchronises with the NULL
            // stream like any default-flag stream -- this library never uses the NULL stream.)
            int ncu = 0;
            if (B >= 2 && B <= 28) {
                int total = 0;
                if (hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && total >= 64) ncu = total / 8 * 7;
            }
            if (const char *e = std::getenv("LVT_AMD_FEATURE_CUS")) ncu = (B >= 2) ? std::atoi(e) : 0;
            if (ncu > 0 && ncu < 1024) {
                uint32_t mask[32] = {};
                for (int i = 0; i < ncu; i++) mask[i >> 5] |= 1u << (i & 31);
                HIPCHK(c, hipExtStreamCreateWithCUMask(&c->stream_f, 32, mask));
                c->feature_cus = ncu;
            } else
                HIPCHK(c, hipStreamCreateWithFlags(&c->stream_f, hipStreamNonBlocking));
        }
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream_e, hipStreamNonBlocking));
        c->own_stream = true;
        for (auto &e : c->ev_done) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : c->ev_feat) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_depth, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_switch, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_switch_e, hipEventDisableTiming));
        if (const char *e = std::getenv("LVT_AMD_TEST_GATE_TIMEOUT")) c->test_gate_timeout = std::atol(e);
        {
            // "events": barrier-only ordering; "polling": the polling gates + early stream; unset: polling for the first live
            // handle of the process, events for the ones created beside it -- the gates of several independent handles share
            // the process's few hardware queues and hold up each other's kernels (4 handles: 5 300 frames/s aggregate with
            // polling gates, 9 200 with events); several sequences on one GPU belong in one lock-step batch anyway
            const char *o = std::getenv("LVT_AMD_ORDERING");
            if (o && std::strcmp(o, "events") == 0) c->events_only = true;
            else if (o && std::strcmp(o, "polling") == 0) c->events_only = false, c->ordering_forced = true;
            else c->events_only = B == 1 && c->live_on_device() > 1;
        }
        c->pitch = ((prm.W + 63) / 64) * 64;
        HIPCHK(c, hipHostMalloc((void **)&c->h_ctl, sizeof(Ctl) * B * RING, hipHostMallocCoherent));
        HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_ctl_dev, c->h_ctl, 0));
        HIPCHK(c, hipHostMalloc((void **)&c->h_done, sizeof(seq_t) * B * RING, hipHostMallocCoherent));
        HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_done_dev, c->h_done, 0));
        std::memset(c->h_done, 0, sizeof(seq_t) * B * RING);
        HIPCHK(c, hipHostMalloc((void **)&c->h_pose, sizeof(PoseRec) * B * RING, hipHostMallocCoherent));
        HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_pose_dev, c->h_pose, 0));
        HIPCHK(c, hipHostMalloc((void **)&c->h_pose_done, sizeof(seq_t) * B * RING, hipHostMallocCoherent));
        HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_pose_done_dev, c->h_pose_done, 0));
        std::memset(c->h_pose_done, 0, sizeof(seq_t) * B * RING);
        if (const char *e = std::getenv("LVT_AMD_SYNC_TAIL")) c->early_pose = std::strcmp(e, "wait") != 0;
        c->binned_lists = B > 1;
        if (const char *e = std::getenv("LVT_AMD_BINNED_LISTS")) c->binned_lists = std::atoi(e) != 0;
        if (const char *e = std::getenv("LVT_AMD_TEST_ROW_FALLBACK")) c->force_row_fallback = std::atoi(e) != 0;
        {   // cells by decreasing area (stable): see k_cells
            int idx[CELLS_MAX], area[CELLS_MAX];
            for (int cc = 0; cc < prm.n_cells; cc++) {
                const int cx = cc % prm.cells_x, cy = cc / prm.cells_x;
                idx[cc] = cc;
                area[cc] = std::min(prm.cell_size, prm.W - cx * prm.cell_size) * std::min(prm.cell_size, prm.H - cy * prm.cell_size);
            }
            std::stable_sort(idx, idx + prm.n_cells, [&](int a, int b) { return area[a] > area[b]; });
            for (int cc = 0; cc < CELLS_MAX; cc++) c->cell_order.v[cc] = (uint8_t)(cc < prm.n_cells ? idx[cc] : 0);
        }
        if (const char *e = std::getenv("LVT_AMD_BRIEF_FROM_IMAGE")) c->brief_from_image = std::atoi(e) ? 1 : 0;
        if (const char *e = std::getenv("LVT_AMD_SCORE_PIECES")) c->score_pieces = std::max(1, std::min(8, std::atoi(e))), c->score_pieces_auto = false;
        {
            int n_cu = 256;
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
            if (!prm.big_cell_strips && prm.n_cells * 2 * B > n_cu) c->cells_raw_cap = RAW_CAP_SMALL;
        }
        if (const char *e = std::getenv("LVT_AMD_CELL_SPLIT")) c->cell_split = std::max(0, std::min(SPLIT_MAX, std::atoi(e)));
        if (const char *e = std::getenv("LVT_AMD_CELLS_RAW_CAP")) c->cells_raw_cap = std::max(RAW_CAP_SMALL, std::min(RAW_CAP, std::atoi(e) & ~1));
        if (const char *e = std::getenv("LVT_AMD_FUSED_PULL")) c->fuse_pull = std::atoi(e) != 0;
        // list kernels: a workgroup of 8 wavefronts works its queries off 32 at a time (four per wavefront), every workgroup stages the train set itself (94 KB of LDS)
        c->lists_wgs_row = c->lists_wgs_map = (B <= 16) ? 4 : 2;  // (measured: 16 sequences 67 / 81 / 81 / 80 k frames/s with 1 / 2 / 4 / 8, 64 sequences 108 / 106 / 102 / 101 -- two there: one workgroup per sequence is 1.6 % faster in frames/s with a kernel twice as long)
        if (const char *e = std::getenv("LVT_AMD_LISTS_WGS")) {
            int r = 0, m = 0;
            const int got = std::sscanf(e, "%d,%d", &r, &m);
            if (got >= 1) c->lists_wgs_row = std::max(1, std::min(64, r));
            if (got >= 2) c->lists_wgs_map = std::max(1, std::min(64, m));
        }
        if (const char *e = std::getenv("LVT_AMD_MATCH_BLOCKS_BATCH")) c->match_blocks_batch = std::max(1, std::min(256, std::atoi(e)));
        HIPCHK(c, hipHostMalloc((void **)&c->h_fargs, sizeof(FrameArgs) * B * RING, hipHostMallocDefault));
        c->h_seqs.resize(B);
        c->d_ctl.resize(B);
        const size_t plane = (size_t)c->pitch * prm.H;
        for (int s = 0; s < B; s++) {
            Seq &S = c->h_seqs[s];
            std::memset(&S, 0, sizeof(S));
            S.prm = prm;
            S.ctl = c->d_ctl[s] = c->dalloc<Ctl>(1);
            S.plane_pitch = c->pitch;
            {
                size_t off = 0;
                for (int cc = 0; cc < prm.n_cells; cc++) {
                    const int cx = cc % prm.cells_x, cy = cc / prm.cells_x;
                    const int cw = std::min(prm.cell_size, prm.W - cx * prm.cell_size), ch = std::min(prm.cell_size, prm.H - cy * prm.cell_size);
                    S.cell_scratch_off[cc] = off;
                    off += (size_t)cw * ch * 6;
                }
            }
            for (int e = 0; e < 2; e++) S.cell_scratch[e] = c->dalloc<uint32_t>((size_t)prm.W * prm.H * 6 + 64);
            for (int e = 0; e < 2; e++) {
                S.strip_kp[e] = prm.big_cell_strips ? c->dalloc<uint32_t>((size_t)prm.n_cells * STRIPS * RAW_CAP) : c->dalloc_uncached<uint32_t>((size_t)prm.n_cells * SPLIT_MAX * SPLIT_KP_CAP);
                S.strip_n[e] = c->dalloc_uncached<int>((size_t)CELLS_MAX * STRIPS);
                S.cell_big[e] = c->dalloc<int>(CELLS_MAX);
            }
            S.lists_fb = c->dalloc<int>(2);
            for (int par = 0; par < NPAR; par++) {
                FrameBuf &FB = S.fb[par];
                FB.fc = c->dalloc<FeatCtl>(1);
                for (int e = 0; e < 2; e++) {
                    FB.score[e] = c->dalloc<uint8_t>(plane + 64);
                    FB.boxsum[e] = c->dalloc<uint16_t>(plane + 64);
                    {
                        const size_t tiles_x = (size_t)(prm.W + TS_W - 1) / TS_W;
                        FB.seg_keys[e] = c->dalloc<uint32_t>((size_t)prm.H * tiles_x * TS_W);
                        FB.seg_cnt[e] = c->dalloc<uint16_t>((size_t)prm.H * tiles_x);
                    }
                    FB.cell_kp[e] = c->dalloc<float>((size_t)CELLS_MAX * CELL_OUT_CAP * 3);
                    FB.cell_n[e] = c->dalloc<int>(CELLS_MAX);
                    alloc_feat(c, FB.feat[e]);
                    if (s == 0) {
                        c->d_ext[par][e] = c->dalloc<float>((size_t)EXT_MAX * 2);
                        c->d_img[par][e] = c->dalloc<uint8_t>(plane + 64);
                    }
                    FB.ext_xy[e] = FB.ext_xy_own[e] = c->d_ext[par][e];
                }
                if (s == 0 && sensor == 2) c->d_depth[par] = c->dalloc<float>((size_t)prm.W * prm.H + 4);
            }
            for (int k = 0; k < 2; k++) {
                alloc_points(c, S.map[k], MAP_MAX);
                alloc_points(c, S.staged[k], STAGED_MAX);
            }
            S.map_cur = c->dalloc<int>(1), S.map_n = c->dalloc<int>(1);
            S.staged_cur = c->dalloc<int>(1), S.staged_n = c->dalloc<int>(1);
            S.proj = c->dalloc<float>((size_t)MAP_MAX * 2);
            S.vis = c->dalloc<int8_t>(MAP_MAX);
            S.match = c->dalloc<int>(MAP_MAX);
            S.cand = c->dalloc<uint32_t>((size_t)MAP_MAX * KC);
            S.ncand = c->dalloc<int>(MAP_MAX);
            S.qidx = c->dalloc<uint32_t>(MAP_MAX);
            S.sproj = c->dalloc<float>((size_t)STAGED_MAX * 2);
            S.svis = c->dalloc<int8_t>(STAGED_MAX);
            S.smatch = c->dalloc<int>(STAGED_MAX);
            S.scand = c->dalloc<uint32_t>((size_t)STAGED_MAX * KC);
            S.sncand = c->dalloc<int>(STAGED_MAX);
            S.sdel = c->dalloc<uint8_t>(STAGED_MAX);
            S.pnp_X = c->dalloc<double>((size_t)NF_MAX * 3);
            S.pnp_obs = c->dalloc<float>((size_t)NF_MAX * 2);
            S.pnp_feat = c->dalloc<int>(NF_MAX);
            S.pnp_err = c->dalloc<double>((size_t)NF_MAX * 2 + 2);  // + the sink slot of inactive sweep lanes (pnp_solve)
            S.pnp_level = c->dalloc<int8_t>(NF_MAX);
            S.rcand = c->dalloc<uint32_t>((size_t)NPAR * NF_MAX * KC);  // per frame buffer
            S.rncand = c->dalloc<int>(NPAR * NF_MAX);
            S.pair_l = c->dalloc<int>(NF_MAX), S.pair_r = c->dalloc<int>(NF_MAX);
            S.tri_X = c->dalloc<double>((size_t)NF_MAX * 3);
            S.tri_ok = c->dalloc<int8_t>(NF_MAX);
        }
        c->d_seqs = c->dalloc<Seq>(B);
        HIPCHK(c, hipMemcpy(c->d_seqs, c->h_seqs.data(), sizeof(Seq) * B, hipMemcpyHostToDevice));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cells<true>), hipFuncAttributeMaxDynamicSharedMemorySize, CELLS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cells<false>), hipFuncAttributeMaxDynamicSharedMemorySize, CELLS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hamming_batched_lists<MODE_MAP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hamming_batched_lists<MODE_MAP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hamming_batched_lists<MODE_ROW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hamming_batched_lists<MODE_ROW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cells_strip), hipFuncAttributeMaxDynamicSharedMemorySize, CELLS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gather<true>), hipFuncAttributeMaxDynamicSharedMemorySize, CELLS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gather<false>), hipFuncAttributeMaxDynamicSharedMemorySize, CELLS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cells_big), hipFuncAttributeMaxDynamicSharedMemorySize, CELLS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cells_radii), hipFuncAttributeMaxDynamicSharedMemorySize, CELLS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cells_select), hipFuncAttributeMaxDynamicSharedMemorySize, CELLS_LDS_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pnp<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PNP_DYN_BYTES));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pnp<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PNP_DYN_BYTES));
        reset_state(c);
    } catch (...) {
        delete c;
        return nullptr;
    }
    return c;
}

// ---- the per-frame launch chain -------------------------------------------------------------------
static const char *kProfNames[Context::PROF_SLOTS] = {
    "k_feat_begin", "k_gate [early stream: waits for the previous k_pnp]", "k_score", "k_cells(pass0)", "", "k_gather(+ the rare retry pass)", "k_brief", "k_gate_late [waits for the early stream]",
    "k_match_map(wait for the early stream + begin + new points)", "k_early_map [early stream]", "k_early_mid [early stream]", "k_hamming_batched_lists(map) [early stream]", "k_track_mid(resolve+pass2+bookkeep+cull)", "k_pnp(+project staged)", "",
    "", "k_candidates(staged)", "", "k_candidates(row) [early stream]", "k_hamming_batched_lists(row)", "k_triangulate(staged update+row resolve+triangulate+finalize)", "",
    "", ""};

#define LAUNCH(slot, st, kern, grid, block, lds, ...)                              \
    do {                                                                           \
        if (c->prof) (void)hipEventRecord(c->ev[slot][0], st);                     \
        hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);               \
        if (c->prof) {                                                             \
            (void)hipEventRecord(c->ev[slot][1], st);                              \
            c->ev_used[slot] = true;                                               \
        }                                                                          \
    } while (0)

// kernels of the early / tracking chains: a single sequence travels BY VALUE in the kernel arguments (k_track.hip, SeqArg)
#define LAUNCH_S(slot, st, kern, grid, block, lds, ...)                                                      \
    do {                                                                                                     \
        if (B == 1) LAUNCH(slot, st, kern<true>, grid, block, lds, SeqArg<true>{c->h_seqs[0]}, __VA_ARGS__); \
        else LAUNCH(slot, st, kern<false>, grid, block, lds, SeqArg<false>{S}, __VA_ARGS__);                  \
    } while (0)
#define LAUNCH_SM(slot, st, kern, MODE_, grid, block, lds, ...)                                                        \
    do {                                                                                                               \
        if (B == 1) LAUNCH(slot, st, (kern<MODE_, true>), grid, block, lds, SeqArg<true>{c->h_seqs[0]}, __VA_ARGS__);   \
        else LAUNCH(slot, st, (kern<MODE_, false>), grid, block, lds, SeqArg<false>{S}, __VA_ARGS__);                   \
    } while (0)

static void collect_oldest(Context *c);

// frame inputs must already be in h_fargs[slot] (and any upload enqueued on stream_f)
struct HostTimer {
    double *acc;
    std::chrono::steady_clock::time_point t0;
    explicit HostTimer(double *a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { *acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
};
static void enqueue_frame(Context *c) {
    HostTimer ht(&c->host_enq_us);
    c->host_enq_n++;
    const int B = c->B;
    // sequences this step's kernels are launched for: all of them -- or, for a pool whose upper seats are empty, the seats in use (lvt_pool.h);
    // B stays the batch's size: strides of the pinned rings, single-sequence / batch code paths
    const int Bz = (c->launch_seqs > 0 && c->launch_seqs < B) ? c->launch_seqs : B;
    c->ring_seqs[(int)(c->enq % RING)] = Bz;
    const Params &p = c->prm;
    Seq *S = c->d_seqs;
    const int slot = (int)(c->enq % RING), par = (int)(c->enq % NPAR);
    if (c->want_events && !c->events_only) {
        // polling -> events, between two frames.  The frames already enqueued were ordered by the gates: (i) the last of them expects THIS frame's
        // gate to deliver its record -- deliver it now; (ii) their ev_done events were never recorded, so this frame's feature stage waits for
        // everything the tracking stream holds instead (the frames after it are behind it on the same stream; from frame + NPAR on the events exist).
        if (c->enq > 0 && c->delivered < c->enq) {
            const int pslot = (int)((c->enq - 1) % RING);
            hipLaunchKernelGGL(k_deliver, dim3(1, 1, Bz), dim3(64), 0, c->stream, S, c->h_ctl_dev + (size_t)pslot * B, c->h_done_dev + (size_t)pslot * B, (seq_t)c->enq);
            c->delivered = c->enq;
        }
        // the early stream may still hold kernels of the pre-switch frames (k_gate, k_early_map, k_candidates<ROW>, k_row_done: they write the row lists
        // and FeatCtl::row_seq of buffers the event-ordered frames reuse) -- exactly when the streams did not run side by side.  Both other streams wait
        // for it; its gates time out by themselves, so this cannot deadlock.
        (void)hipEventRecord(c->ev_switch_e, c->stream_e);
        (void)hipStreamWaitEvent(c->stream, c->ev_switch_e, 0);
        (void)hipStreamWaitEvent(c->stream_f, c->ev_switch_e, 0);
        (void)hipEventRecord(c->ev_switch, c->stream);
        (void)hipStreamWaitEvent(c->stream_f, c->ev_switch, 0);
        c->events_only = true;
        c->want_events = false;
        c->switched_at = (long)c->enq;
    }
    const FrameArgs *fa = c->h_fargs + (size_t)slot * B;
    const int ext = (B == 1) ? fa[0].ext_corners : 0;  // (a pool's step may mix seats with and without external corners: k_cells is launched, cell_begin skips the seats that bring their own)
    hipStream_t sf = c->stream_f, st = c->stream;
    for (int i = 0; i < Context::PROF_SLOTS; i++) c->ev_used[i] = false;
    // ---- feature stage (stream_f): may start as soon as the tracking chain of frame enq-NPAR released this buffer
    const bool evo = c->events_only;
    const bool feat_pack = B > 1 && Bz <= FEAT_PACK && !std::getenv("LVT_AMD_NO_FEAT_PACK");  // (k_feat_begin_pack: the buffer gate rides in it)
    if (c->enq >= NPAR) {
        if (!evo && !feat_pack)
            hipLaunchKernelGGL(k_gate_buf, dim3(1, 1, Bz), dim3(64), 0, sf, S, (seq_t)(c->enq + 1 - NPAR), par);  // polls; see k_gate_buf
        else if (c->switched_at < 0 || (long)c->enq - NPAR >= c->switched_at)  // (frames from before a switch of the ordering: covered by ev_switch)
            (void)hipStreamWaitEvent(sf, c->ev_done[(int)((c->enq - NPAR) % RING)], 0);
    }
    if (B == 1) {  // the frame's inputs travel as a kernel argument: no separate "begin" launch, no device read of pinned host memory
        LAUNCH(2, sf, k_score<true>, dim3((p.W + TS_W - 1) / TS_W, (p.H + TS_H - 1) / TS_H, 2), dim3(256), 0, S, c->h_fargs[slot], par, 0, c->brief_from_image ? 0 : 1);
    } else {
        if (feat_pack) {
            FrameArgsPack pk;
            std::memset(&pk, 0, sizeof(pk));
            for (int s = 0; s < Bz; s++) pk.a[s] = fa[s];
            LAUNCH(0, sf, k_feat_begin_pack, dim3(Bz), dim3(64), 0, S, pk, par, (seq_t)((c->enq >= NPAR && !evo) ? c->enq + 1 - NPAR : 0));
        } else
            LAUNCH(0, sf, k_feat_begin, dim3(Bz), dim3(64), 0, S, fa, par);
        {   // the batch's images in score_pieces launches: see Context::score_pieces
            const int P = std::max(1, std::min(c->score_pieces, Bz));
            for (int q = 0; q < P; q++) {
                const int z0 = 2 * (int)((long)Bz * q / P), z1 = 2 * (int)((long)Bz * (q + 1) / P);
                if (q == 0) LAUNCH(2, sf, k_score<false>, dim3((p.W + TS_W - 1) / TS_W, (p.H + TS_H - 1) / TS_H, z1 - z0), dim3(256), 0, S, FrameArgs{}, par, z0, c->brief_from_image ? 0 : 1);
                else hipLaunchKernelGGL(k_score<false>, dim3((p.W + TS_W - 1) / TS_W, (p.H + TS_H - 1) / TS_H, z1 - z0), dim3(256), 0, sf, S, FrameArgs{}, par, z0, c->brief_from_image ? 0 : 1);
            }
        }
    }
    if (!ext) {
        {
            const int pass = 0;  // (the <200-corner retry pass runs inside k_gather: it is almost never taken)
            // (a single sequence: + the workgroups that pull the next asynchronous host frame, one 16-byte vector per thread)
            const int pull_wgs = (B == 1 && c->next_pull.src[0]) ? std::min(64, (int)(((size_t)p.W * p.H / 16 + 1023) / 1024)) : 0;
            // a single sequence's tall cells run as cell_split co-operating workgroups (cells_work_split); grid: helpers, main workgroups, pull, padded to 8
            const int ns = (B == 1 && !p.big_cell_strips) ? c->cell_split : 0;
            const int gx = (ns >= 2) ? ((((p.n_cells + 7) & ~7) * ns + pull_wgs + 7) & ~7) : p.n_cells + pull_wgs;
            LAUNCH_S(3, sf, k_cells, (B == 1 ? dim3(gx, 2, 1) : dim3(p.n_cells * 2 * Bz, 1, 1)), dim3(1024), cells_lds_bytes(c->cells_raw_cap), pass, par, c->cell_order, 2 * Bz, c->cells_raw_cap, c->next_pull, ns, (++c->cell_token ? c->cell_token : ++c->cell_token));
            if (p.big_cell_strips) {  // oversized cells: NMS as row strips on several CUs, then ANMS of the merged survivors in three launches
                hipLaunchKernelGGL(k_cells_strip, dim3(p.n_cells * STRIPS, 2, Bz), dim3(1024), CELLS_LDS_BYTES, sf, S, pass, par);
                hipLaunchKernelGGL(k_cells_big, dim3(p.n_cells, 2, Bz), dim3(1024), CELLS_LDS_BYTES, sf, S, pass, par);
                hipLaunchKernelGGL(k_cells_radii, dim3(p.n_cells * RADII_WGS, 2, Bz), dim3(1024), CELLS_LDS_BYTES, sf, S, pass, par);
                hipLaunchKernelGGL(k_cells_select, dim3(p.n_cells, 2, Bz), dim3(1024), CELLS_LDS_BYTES, sf, S, pass, par);
                if (c->prof) (void)hipEventRecord(c->ev[3][1], sf);  // (the slot's time covers the five launches)
            }
        }
    }
    if (c->before_gather) {  // (the GPU is busy with k_score / k_cells from here on: host-side copies behind this point are free)
        c->before_gather();
        c->before_gather = nullptr;
    }
    if (c->depth_wait) {
        (void)hipStreamWaitEvent(sf, c->ev_depth, 0);
        c->depth_wait = false;
    }
    LAUNCH_S(5, sf, k_gather, dim3(1, 2, Bz), dim3(1024), CELLS_LDS_BYTES, (const Seq *)S, par);
    const bool brief_publishes = !evo && B == 1;  // (single sequence: k_brief's last workgroup publishes feat_seq; see k_feat_done)
    if (c->brief_from_image) LAUNCH_S(6, sf, k_brief_img, dim3(64, 2, Bz), dim3(256), 0, (const Seq *)S, par, brief_publishes ? (seq_t)(c->enq + 1) : (seq_t)0);
    else LAUNCH_S(6, sf, k_brief, dim3(64, 2, Bz), dim3(256), 0, (const Seq *)S, par, brief_publishes ? (seq_t)(c->enq + 1) : (seq_t)0);
    if (!evo && !brief_publishes) hipLaunchKernelGGL(k_feat_done, dim3(Bz), dim3(64), 0, sf, S, par, (seq_t)(c->enq + 1));
    const int bl = c->binned_lists ? 1 : 0;
    if (evo && bl && c->sensor == 1) LAUNCH_SM(19, sf, k_hamming_batched_lists, MODE_ROW, dim3(c->lists_wgs_row, 1, Bz), dim3(LS_THREADS), LS_LDS_BYTES, par, (seq_t)0);
    if (evo) LAUNCH_SM(18, sf, k_candidates, MODE_ROW, dim3(256, 1, Bz), dim3(256), 0, 0, par, (seq_t)0, bl);  // (normal mode: on the early stream, below)
    if (evo) hipLaunchKernelGGL(k_feat_done, dim3(Bz), dim3(64), 0, sf, S, par, (seq_t)(c->enq + 1));
    if (evo) (void)hipEventRecord(c->ev_feat[par], sf);
    // ---- early stream: projection, candidate lists and greedy resolution of the map points that survived the previous frame,
    //      as soon as that frame's pose exists (its k_pnp) -- its k_triangulate only appends behind them
    const seq_t seq = (seq_t)(c->enq + 1);  // this frame's sequence number; the previous frame's is enq (0: none)
    if (!evo) {
        hipStream_t se = c->stream_e;
        LAUNCH_S(1, se, k_gate, dim3(1, 1, Bz), dim3(64), 0, par, (seq_t)c->enq, seq, (c->test_gate_timeout && (long)seq == c->test_gate_timeout) ? 1 : 0);  // polls the previous k_pnp and this frame's features
        if (bl) LAUNCH_SM(11, se, k_hamming_batched_lists, MODE_MAP, dim3(c->lists_wgs_map, 1, Bz), dim3(LS_THREADS), LS_LDS_BYTES, par, seq);
        LAUNCH_S(9, se, k_early_map, dim3((bl && B > 1) ? 32 : 256, 1, Bz), dim3(256), 0, par, seq, bl);  // (behind the binned list kernel it is the fall-back only)
        LAUNCH_S(10, se, k_early_mid, dim3(1, 1, Bz), dim3(RES_THREADS), 0, par, seq);
        if (c->sensor == 1) {
            // row-match candidate lists of THIS frame (needed by its k_triangulate, ~70 us from here): they need the two feature
            // sets only, and the feature stream is the longest chain -- here, behind the early part, they lengthen neither it nor
            // the hand-over to the tracking stream.  Few workgroups: the tracking chain's single-workgroup kernels run meanwhile.
            if (bl) LAUNCH_SM(19, se, k_hamming_batched_lists, MODE_ROW, dim3(c->lists_wgs_row, 1, Bz), dim3(LS_THREADS), LS_LDS_BYTES, par, seq);
            LAUNCH_SM(18, se, k_candidates, MODE_ROW, dim3(B == 1 ? ROW_BLOCKS : ROW_BLOCKS_BATCH, 1, Bz), dim3(256), 0, 0, par, seq, bl);
            hipLaunchKernelGGL(k_row_done, dim3(Bz), dim3(64), 0, se, S, par, seq);
        }
    }
    // ---- tracking chain (stream): strictly ordered frame after frame
    // (no barrier on the features here: k_gate_late returns only after the early stream's gate has seen them complete, and
    //  every barrier / event packet costs this stream 3-4 us per frame)
    {
        // the previous frame's record, if nobody has been asked to deliver it yet
        Ctl *prec = nullptr;
        seq_t *pdone = nullptr;
        if (!evo && c->delivered < c->enq) {
            const int pslot = (int)((c->enq - 1) % RING);
            prec = c->h_ctl_dev + (size_t)pslot * B, pdone = c->h_done_dev + (size_t)pslot * B;
            c->delivered = c->enq;
        }
        if (evo) {
            (void)hipStreamWaitEvent(st, c->ev_feat[par], 0);  // (the early stream never claims the frame: early_ran_seq stays behind)
            LAUNCH_S(8, st, k_match_map, dim3(256, 1, Bz), dim3(256), 0, par, seq, 0, (Ctl *)nullptr, (seq_t *)nullptr);
        } else if (B == 1 && c->live_on_device() <= MAX_FOLDED_GATES) {
            // one launch: delivers the record, waits for the early stream (polled, no barrier packet), lists the points appended since
            LAUNCH_S(8, st, k_match_map, dim3(MATCH_BLOCKS_GATED, 1, 1), dim3(256), 0, par, seq, 1, prec, pdone);
        } else {
            LAUNCH_S(7, st, k_gate_late, dim3(1, 1, Bz), dim3(64), 0, par, seq, prec, pdone);
            LAUNCH_S(8, st, k_match_map, dim3(B > 1 ? c->match_blocks_batch : 256, 1, Bz), dim3(256), 0, par, seq, 0, (Ctl *)nullptr, (seq_t *)nullptr);
        }
    }
    LAUNCH_S(12, st, k_track_mid, dim3(1, 1, Bz), dim3(RES_THREADS), 0, par, seq);
    {
        const bool ep = c->sync_call && c->early_pose && !c->prof && B == 1;
        LAUNCH_S(13, st, k_pnp, dim3(1, 1, Bz), dim3(PNP_THREADS), PNP_DYN_BYTES, par, seq, ep ? c->h_pose_dev + (size_t)slot * B : (PoseRec *)nullptr,
                 ep ? c->h_pose_done_dev + (size_t)slot * B : (seq_t *)nullptr);
    }
    if (p.staged_th > 0)  // (a configuration without staging -- EuRoC, TUM -- never has staged points to list)
        LAUNCH_SM(16, st, k_candidates, MODE_STAGED, dim3(64, 1, Bz), dim3(256), 0, 0, par, (seq_t)0, 0);
    LAUNCH_S(20, st, k_triangulate, dim3(1, 1, Bz), dim3(1024), 0, par, seq, c->h_ctl_dev + (size_t)slot * B,
           c->h_done_dev + (size_t)slot * B, evo ? 0 : (c->force_row_fallback ? 2 : 1), (evo || c->sync_call) ? 1 : 0);
    if (evo || c->sync_call) c->delivered = c->enq + 1;
    if (evo) (void)hipEventRecord(c->ev_done[slot], st);  // (events-only ordering: the feature stream's barrier needs it)
    c->enq++;
}

// wait for the oldest un-collected frame; its record becomes "last"
static void collect_oldest(Context *c) {
    HostTimer ht(&c->host_wait_us);
    c->host_wait_n++;
    if (c->done >= c->enq) return;
    const int slot = (int)(c->done % RING);
    const int Bz = (c->ring_seqs[slot] > 0 && c->ring_seqs[slot] <= c->B) ? c->ring_seqs[slot] : c->B;  // sequences the step was launched for
    if (c->delivered < c->done + 1) {  // the last enqueued frame of an asynchronous run: nobody has been asked to deliver it yet
        hipLaunchKernelGGL(k_deliver, dim3(1, 1, Bz), dim3(64), 0, c->stream, c->d_seqs, c->h_ctl_dev + (size_t)slot * c->B,
                           c->h_done_dev + (size_t)slot * c->B, (seq_t)(c->done + 1));
        c->delivered = c->done + 1;
    }
    {   // the frame is complete when every sequence's flag carries its number (the flags follow the records: system-scope
        // release on the device, acquire here)
        const seq_t want = (seq_t)(c->done + 1);
        unsigned spins = 0;
        for (int s = 0; s < Bz; s++) {
            volatile seq_t *f = c->h_done + (size_t)slot * c->B + s;
            while (__atomic_load_n(f, __ATOMIC_ACQUIRE) != want) {
                __builtin_ia32_pause();
                if ((++spins & 0xFFFFF) == 0) {  // every ~10 ms: a dead stream must not hang the caller
                    const hipError_t q = hipStreamQuery(c->stream);
                    if (q != hipSuccess && q != hipErrorNotReady) {
                        c->set_error(std::string("tracking stream: ") + hipGetErrorString(q));
                        s = c->B;
                        break;
                    }
                    if (q == hipSuccess && __atomic_load_n(f, __ATOMIC_ACQUIRE) != want) {  // stream empty but no flag: cannot happen
                        c->set_error("tracking stream finished without a result record");
                        s = c->B;
                        break;
                    }
                }
            }
        }
        if (c->prof) HIPCHK(c, hipStreamSynchronize(c->stream));  // the per-kernel timing events are read below
    }
    c->last_slot = slot;
    c->last_par = (int)(c->done % NPAR);
    c->done++;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) c->set_error(std::string("kernel chain: ") + hipGetErrorString(e));
    if (c->prof && c->done == c->enq) {  // profiling is meaningful when frames are collected one at a time
        for (int i = 0; i < Context::PROF_SLOTS; i++)
            if (c->ev_used[i]) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, c->ev[i][0], c->ev[i][1]) == hipSuccess) {
                    c->prof_ms[i] += ms;
                    c->prof_calls[i]++;
                }
            }
    }
    for (int s = 0; s < Bz; s++)
        if (c->h_ctl[(size_t)slot * c->B + s].overflow) {
            char buf[128];
            std::snprintf(buf, sizeof(buf), "capacity overflow mask 0x%x in sequence %d", c->h_ctl[(size_t)slot * c->B + s].overflow, s);
            c->set_error(buf);
        }
    if (c->B > 1 && c->score_pieces_auto && !c->events_only) {  // see Context::score_pieces
        const Ctl &r0 = c->h_ctl[(size_t)slot * c->B];
        const long long t0 = r0.dbg[32], t1 = r0.dbg[35], t2 = r0.dbg[33];
        if (t0 > 0 && t0 <= t1 && t1 <= t2 && t2 - t0 < 1000000) {  // (a consistent triple of one gate, < 10 ms)
            const double d = 0.01 * (double)((t2 - t1) - (t1 - t0));
            c->gate_balance_us = 0.8 * c->gate_balance_us + 0.2 * d;
            if (c->gate_balance_us > 15.0) c->score_pieces = 2;
            else if (c->gate_balance_us < -15.0) c->score_pieces = 1;
        }
    }
    for (int s = 0; s < Bz; s++)
    {
        const Ctl &r = c->h_ctl[(size_t)slot * c->B + s];
        if (r.gate_fatal != c->gate_fatal_seen) {
            c->gate_fatal_seen = r.gate_fatal;
            c->set_error("a stream waited 2 s for another one (features / buffer hand-over): a frame was SKIPPED (its pose is the previous frame's, the "
                         "tracking state is unchanged); under a tool that serialises kernel dispatches (rocprofv3 --pmc) set LVT_AMD_ORDERING=events"
                         + std::string(c->ordering_forced || c->events_only ? "" : " -- the handle moves to event ordering from the next frame on"));
            if (!c->ordering_forced && !c->events_only) c->want_events = true;
        } else if (r.gate_timeouts != c->gate_timeouts_seen) {
            c->gate_timeouts_seen = r.gate_timeouts;
            // not a wrong result (the frame was tracked without the early stream / with row lists built on the tracking stream), but
            // milliseconds were lost: the streams do not run concurrently (shared hardware queue, or a tool that serialises the dispatches)
            c->set_error(std::string("a stream gate timed out (results unaffected; the streams do not run concurrently: LVT_AMD_ORDERING=events avoids the waits)")
                         + (c->ordering_forced || c->events_only ? "" : " -- the handle moves to event ordering from the next frame on"));
            if (!c->ordering_forced && !c->events_only) c->want_events = true;
        }
    }
}
// the held asynchronous host frame (Context::pend) enters the launch chain; np: the images its k_cells launch pulls for the frame behind it
static void flush_pending(Context *c, const NextPull *np = nullptr) {
    if (!c->pend.valid) return;
    Context::PendingFrame &q = c->pend;
    q.valid = false;
    if (!q.pulled) hipLaunchKernelGGL(k_stage_in, dim3(128, 2), dim3(256), 0, c->stream_f, q.src[0], q.src[1], q.dst[0], q.dst[1], c->prm.W, c->prm.H, c->pitch);
    c->h_fargs[(size_t)(c->enq % RING) * c->B] = q.f;
    if (np) c->next_pull = *np, c->fused_pulls++;
    enqueue_frame(c);
    c->next_pull = NextPull{};
}
static void drain(Context *c) {
    flush_pending(c);
    while (c->done < c->enq) collect_oldest(c);
    c->early_pending = false;
}
// synchronous calls: wait for the frame just enqueued -- its pose (k_pnp) or, when the frame has none of its own (first frame, LOST,
// skipped), its full record.  Returns true when R / t were taken from the early pose; the frame then stays un-collected until the next
// entry point drains (by then its tail has long finished).
static bool wait_pose_or_frame(Context *c, double R[3][3], double t[3]) {
    HostTimer ht(&c->host_wait_us);
    const int slot = (int)((c->enq - 1) % RING);
    const seq_t want = (seq_t)c->enq;
    volatile seq_t *fp = c->h_pose_done + (size_t)slot * c->B, *fd = c->h_done + (size_t)slot * c->B;
    if (!c->early_pose || c->prof || c->B != 1) {
        drain(c);
        return false;
    }
    unsigned spins = 0;
    for (;;) {
        if (__atomic_load_n(fp, __ATOMIC_ACQUIRE) == want) {
            const PoseRec &p = c->h_pose[(size_t)slot * c->B];
            if (R)
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) R[i][j] = p.R[3 * i + j];
            if (t)
                for (int i = 0; i < 3; i++) t[i] = p.t[i];
            c->early_pending = true;
            return true;
        }
        if (__atomic_load_n(fd, __ATOMIC_ACQUIRE) == want) break;
        __builtin_ia32_pause();
        if ((++spins & 0xFFFFF) == 0) {  // a dead stream must not hang the caller: collect_oldest diagnoses it
            const hipError_t q = hipStreamQuery(c->stream);
            if (q != hipErrorNotReady) break;
        }
    }
    drain(c);
    return false;
}
static void make_room(Context *c) {  // at most RING-1 frames un-collected before a new one is enqueued
    flush_pending(c);
    while (c->enq - c->done >= RING - 1) collect_oldest(c);
}
static const Ctl &last_ctl(Context *c, int s = 0) { return c->h_ctl[(size_t)c->last_slot * c->B + s]; }

static void result_out(Context *c, int s, double R[3][3], double t[3]) {
    const Ctl &h = last_ctl(c, s);
    if (R)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R[i][j] = h.out_R[3 * i + j];
    if (t)
        for (int i = 0; i < 3; i++) t[i] = h.out_t[i];
}

static bool size_ok(Context *c, int rows, int cols) { return rows == c->prm.H && cols == c->prm.W; }

template <typename T>
static void d2h(Context *c, T *dst, const T *src, size_t n) {
    if (n && dst) HIPCHK(c, hipMemcpy(dst, src, sizeof(T) * n, hipMemcpyDeviceToHost));
}
static int read_scalar(Context *c, const int *d) {
    int v = 0;
    HIPCHK(c, hipMemcpy(&v, d, sizeof(int), hipMemcpyDeviceToHost));
    return v;
}

}  // namespace lvt

#include "lvt_pool.h"

namespace lvt {
// what the introspection entry points read: a sequence of a context, the feature-buffer parity and the record of its last collected frame.
// A pooled handle's view holds the pool's lock (no step can start underneath the device reads) after every step in flight has been collected.
struct View {
    Context *c = nullptr;
    int s = 0, par = 0;
    const Ctl *rec = nullptr;
    std::unique_lock<std::mutex> lk;
    DeviceGuard *guard = nullptr;
    ~View() { delete guard; }
};
static bool view_of(lvt_handle h, View &v) {
    if (is_slot(h)) {
        PoolSlot *S = static_cast<PoolSlot *>(h);
        slot_drain(S);
        v.lk = std::unique_lock<std::mutex>(S->pool->mu);
        pool_quiesce(S->pool, v.lk);
        v.c = S->pool->ctx, v.s = S->slot, v.par = S->last.par, v.rec = &S->last.rec;
    } else if (is_ctx(h)) {
        Context *c = static_cast<Context *>(h);
        v.guard = new DeviceGuard(c);
        drain(c);
        v.c = c, v.s = 0, v.par = c->last_par, v.rec = &last_ctl(c);
        return true;
    } else
        return false;
    v.guard = new DeviceGuard(v.c);
    return true;
}
// ---- automatic seats (round 6) -------------------------------------------------------------------------------------------------------------
// The reference lets a process hold several lvt_systems (lvt_c.cpp:33-62); here independent handles with a launch chain each share the process's hardware
// queues badly (8 of them deliver LESS than one), while seats of one lock-step chain scale (lvt_pool.h).  So the REFERENCE's create call, lvt_create, hands
// out a small forwarding record (lvt_amd_create / lvt_amd_create_on_device -- the extension API with its per-stage read-back, most of which a seat does not
// offer -- keep handing out what the caller asked for): a stereo handle starts on a chain of its own, and when a SECOND handle with the same parameters is created on the device while the first has not been
// used yet (nothing but create / get_device / get_ordering / last_error was called on it), both -- and every later one -- become seats of the device's
// pool.  A handle that has already tracked keeps its kind (moving a tracker's state between chains is not attempted), so a process that creates its handles
// first and tracks afterwards -- one thread per sequence, the usual shape -- gets the pool without an environment variable; a lone handle never pays for it.
// LVT_AMD_AUTO_POOL=0 turns this off (every handle solo, as before round 6); LVT_AMD_POOL=1 still forces seats from the first handle on.
constexpr uint64_t AUTO_MAGIC = 0x4C56544155544F31ull;
struct AutoHandle {
    uint64_t magic = AUTO_MAGIC;
    lvt_amd_params prm;
    int device = 0;
    std::mutex mu;
    lvt_handle impl = nullptr;  // Context * or PoolSlot *
    bool started = false;       // the tracker behind the handle has been used: it keeps its kind
};
static std::mutex g_auto_mu;
static std::vector<AutoHandle *> g_auto;
static inline bool is_auto(lvt_handle h) { return h && *static_cast<const uint64_t *>(h) == AUTO_MAGIC; }
static inline lvt_handle resolve_handle(lvt_handle h, bool starts = true) {
    if (!is_auto(h)) return h;
    AutoHandle *A = static_cast<AutoHandle *>(h);
    std::lock_guard<std::mutex> g(A->mu);
    if (starts) A->started = true;
    return A->impl;
}
static bool auto_pool_enabled() {
    const char *e = std::getenv("LVT_AMD_AUTO_POOL");
    return !e || std::atoi(e) != 0;
}
static void destroy_impl(lvt_handle h) {
    if (is_slot(h)) {
        pool_leave(static_cast<PoolSlot *>(h));
    } else if (h) {
        DeviceGuard guard(static_cast<Context *>(h));
        delete static_cast<Context *>(h);
    }
}
// a stereo handle of the default create calls (device < 0: the calling thread's current device)
static lvt_handle auto_create(const lvt_amd_params &p, int device) {
    int cur = 0;
    if (device < 0 && hipGetDevice(&cur) == hipSuccess) device = cur;
    std::lock_guard<std::mutex> G(g_auto_mu);
    bool pool = false;
    for (AutoHandle *X : g_auto) {
        if (X->device != device || !same_params(X->prm, p)) continue;
        std::lock_guard<std::mutex> g(X->mu);
        if (is_slot(X->impl)) {
            pool = true;
        } else if (!X->started) {  // an unused solo handle: it takes a seat (its chain is given back)
            if (PoolSlot *S = pool_join(p, 1, device)) {
                destroy_impl(X->impl);
                X->impl = S;
                pool = true;
            }
        }
    }
    AutoHandle *A = new AutoHandle();
    A->prm = p, A->device = device;
    if (pool) A->impl = static_cast<lvt_handle>(pool_join(p, 1, device));  // (no seat left: a chain of its own)
    if (!A->impl) A->impl = static_cast<lvt_handle>(create_context(p, 1, 1, device));
    if (!A->impl) {
        delete A;
        return nullptr;
    }
    g_auto.push_back(A);
    return static_cast<lvt_handle>(A);
}
static void auto_destroy(AutoHandle *A) {
    {
        std::lock_guard<std::mutex> G(g_auto_mu);
        for (size_t i = 0; i < g_auto.size(); i++)
            if (g_auto[i] == A) {
                g_auto.erase(g_auto.begin() + (long)i);
                break;
            }
    }
    lvt_handle impl;
    {
        std::lock_guard<std::mutex> g(A->mu);
        impl = A->impl, A->impl = nullptr, A->started = true;
    }
    destroy_impl(impl);
    A->magic = 0;
    delete A;
}
// LVT_AMD_POOL=1: every handle the create calls make is pooled.  (A mixed mode -- the first handle of a device on its own launch chain, the later ones
// pooled -- was measured and removed: 2 / 4 / 8 handles 6.8k / 13.1k / 25.3k frames/s against 12.0k / 22.8k / 43.6k all pooled: the solo handle's
// polling gates and the pool's chain hold each other up on the shared hardware queues.)
static bool pool_wanted(int) {
    const char *e = std::getenv("LVT_AMD_POOL");
    return e && std::atoi(e) != 0;
}
}  // namespace lvt

using namespace lvt;

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

LVT_API void lvt_amd_default_params(lvt_amd_params *p) { default_params(p); }
LVT_API int lvt_amd_params_from_file(const char *f, lvt_amd_params *p) { return params_from_file(f, p) ? 1 : 0; }

LVT_API lvt_handle lvt_amd_create_pooled(const lvt_amd_params *p, int sensor_type, int device) {
    try {
        return static_cast<lvt_handle>(pool_join(*p, sensor_type, device));
    } catch (...) {
    }
    return nullptr;
}

LVT_API lvt_handle lvt_amd_create(const lvt_amd_params *p, int sensor_type) {
    try {
        if (pool_wanted(-1))
            if (PoolSlot *S = pool_join(*p, sensor_type, -1)) return static_cast<lvt_handle>(S);
        return static_cast<lvt_handle>(create_context(*p, sensor_type, 1));
    } catch (...) {
    }
    return nullptr;
}

LVT_API lvt_handle lvt_amd_create_on_device(const lvt_amd_params *p, int sensor_type, int device) {
    try {
        if (pool_wanted(device))
            if (PoolSlot *S = pool_join(*p, sensor_type, device)) return static_cast<lvt_handle>(S);
        return static_cast<lvt_handle>(create_context(*p, sensor_type, 1, device));
    } catch (...) {
    }
    return nullptr;
}
LVT_API int lvt_amd_get_device(lvt_handle h) {
    if (is_auto(h)) return static_cast<AutoHandle *>(h)->device;  // (fixed at creation)
    if (is_slot(h)) return static_cast<PoolSlot *>(h)->pool->device;
    return h ? static_cast<Context *>(h)->device : -1;
}

LVT_API lvt_handle lvt_create(const char *config_file_name, int sensor_type) {
    try {
        lvt_amd_params p;
        default_params(&p);
        if (params_from_file(config_file_name, &p) && (sensor_type == 1 || sensor_type == 2)) {
            if (pool_wanted(-1))
                if (PoolSlot *S = pool_join(p, sensor_type, -1)) return static_cast<lvt_handle>(S);
            if (sensor_type == 1 && auto_pool_enabled()) return auto_create(p, -1);
            return static_cast<lvt_handle>(create_context(p, sensor_type, 1));
        }
    } catch (...) {
    }
    return nullptr;
}

LVT_API void lvt_destroy(lvt_handle h) {
    if (is_auto(h)) {
        try {
            auto_destroy(static_cast<AutoHandle *>(h));
        } catch (...) {
        }
        return;
    }
    if (is_slot(h)) {
        try {
            pool_leave(static_cast<PoolSlot *>(h));
        } catch (...) {
        }
        return;
    }
    if (h && std::getenv("LVT_AMD_HOST_TIMING")) {
        const Context *c = static_cast<Context *>(h);
        if (c->host_enq_n > 0)
            std::fprintf(stderr, "lvt_amd host timing: %ld frames enqueued, %.1f us each; %ld collected, %.1f us each (blocking)\n", c->host_enq_n,
                         c->host_enq_us / c->host_enq_n, c->host_wait_n, c->host_wait_us / std::max(c->host_wait_n, 1L));
    }
    try {
        DeviceGuard guard(static_cast<Context *>(h));
        delete static_cast<Context *>(h);
    } catch (...) {
    }
}

LVT_API void lvt_amd_reset(lvt_handle h) {
    h = resolve_handle(h);
    if (is_slot(h)) {
        try {
            View v;
            if (view_of(h, v)) reset_state(v.c, v.s);
            PoolSlot *S = static_cast<PoolSlot *>(h);
            std::memset(&S->last.rec, 0, sizeof(Ctl));
            S->last.rec.state = S->last.rec.out_status = 1;
            S->last.rec.out_R[0] = S->last.rec.out_R[4] = S->last.rec.out_R[8] = 1.0;
            S->last.rec.last_pose.q[0] = S->last.rec.predicted.q[0] = 1.0;
        } catch (...) {
        }
        return;
    }
    try {
        DeviceGuard guard(static_cast<Context *>(h));
        reset_state(static_cast<Context *>(h));
    } catch (...) {
    }
}

LVT_API void lvt_amd_set_stream(lvt_handle h, void *hip_stream) {
    h = resolve_handle(h);
    if (!is_ctx(h)) return;  // (a pooled handle runs on the pool's streams)
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        drain(c);
        if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
        c->stream = static_cast<hipStream_t>(hip_stream);
        c->own_stream = false;
    } catch (...) {
    }
}

LVT_API const char *lvt_amd_last_error(lvt_handle h) {
    if (is_auto(h)) {  // answered under the record's lock: a handle that has not been used yet may be given a seat by another thread's lvt_create at any time
        AutoHandle *A = static_cast<AutoHandle *>(h);
        std::lock_guard<std::mutex> g(A->mu);
        static thread_local std::string acopy;
        acopy = lvt_amd_last_error(A->impl);
        return acopy.c_str();
    }
    if (!h) return "no handle (creation failed: bad parameters, or no HIP device -- there is no CPU fallback)";
    if (is_slot(h)) {
        PoolSlot *S = static_cast<PoolSlot *>(h);
        std::lock_guard<std::mutex> g(S->pool->mu);
        static thread_local std::string copy;
        copy = S->err;
        return copy.c_str();
    }
    Context *c = static_cast<Context *>(h);
    if (c->early_pending) {  // the last synchronous call returned on its pose: the frame's own reports (capacity overflow, a gate that
                             // timed out, a skipped frame) arrive with its full record -- collect it before answering
        DeviceGuard guard(c);
        try {
            drain(c);
        } catch (...) {
        }
    }
    return c->err.c_str();
}

LVT_API void lvt_amd_get_host_stats(lvt_handle h, long long out[8]) {
    h = resolve_handle(h);
    Context *c = static_cast<Context *>(h);
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (!c) return;
    if (is_slot(h)) {  // out[0] frames this handle deposited, out[1] lock-step steps the pool launched, out[2] slot frames folded into them, out[3] live handles of the pool
        PoolSlot *S = static_cast<PoolSlot *>(h);
        std::lock_guard<std::mutex> g(S->pool->mu);
        out[0] = S->submitted, out[1] = S->pool->steps, out[2] = S->pool->slot_frames, out[3] = S->pool->live, out[7] = 2;
        return;
    }
    out[0] = (long long)c->enq + (c->pend.valid ? 1 : 0), out[1] = (long long)c->done, out[2] = c->planes_in_place, out[3] = c->planes_staged;
    out[4] = c->async_frames, out[5] = c->fused_pulls, out[6] = c->score_pieces, out[7] = c->events_only ? 1 : 0;
}

LVT_API void lvt_amd_profile_enable(lvt_handle h, int enable) {
    h = resolve_handle(h);
    if (!is_ctx(h)) return;
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        drain(c);
        if (enable) {
            for (auto &e : c->ev)
                for (auto &x : e)
                    if (!x) HIPCHK(c, hipEventCreate(&x));
            for (int i = 0; i < Context::PROF_SLOTS; i++) c->prof_ms[i] = 0, c->prof_calls[i] = 0;
        }
        c->prof = enable != 0;
    } catch (...) {
    }
}
LVT_API int lvt_amd_profile_read(lvt_handle h, int slot, char *name, int name_cap, double *total_ms, long *calls) {
    h = resolve_handle(h);
    if (!is_ctx(h)) return 0;
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    if (slot < 0 || slot >= Context::PROF_SLOTS || !kProfNames[slot][0]) return 0;
    std::snprintf(name, name_cap, "%s", kProfNames[slot]);
    *total_ms = c->prof_ms[slot];
    *calls = c->prof_calls[slot];
    return 1;
}

LVT_API void lvt_amd_track_device_async(lvt_handle h, const void *d_left, const void *d_right, int n_rows, int n_cols, int pitch_bytes) {
    h = resolve_handle(h);
    if (is_slot(h)) {
        try {
            (void)slot_submit(static_cast<PoolSlot *>(h), static_cast<const uint8_t *>(d_left), static_cast<const uint8_t *>(d_right), n_rows, n_cols, pitch_bytes, false);
        } catch (...) {
        }
        return;
    }
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        if (!size_ok(c, n_rows, n_cols) || (pitch_bytes & 15)) {
            c->set_error("lvt_amd_track_device: image size / pitch mismatch");
            return;
        }
        if (c->early_pending) drain(c);  // (a synchronous call's frame whose pose has already been returned: not this caller's to collect)
        make_room(c);
        FrameArgs &f = c->h_fargs[(size_t)(c->enq % RING) * c->B];
        f.img[0] = static_cast<const uint8_t *>(d_left);
        f.img[1] = static_cast<const uint8_t *>(d_right);
        f.depth = nullptr;
        f.img_pitch = pitch_bytes;
        f.depth_pitch = 0;
        f.ext_corners = 0;
        f.absent = 0;
        f.n_ext[0] = f.n_ext[1] = 0;
        enqueue_frame(c);
    } catch (...) {
    }
}

// ---- lock-step batch: B independent sequences advance through ONE launch chain (gridDim.z = B) ------------
LVT_API lvt_handle lvt_amd_batch_create(const lvt_amd_params *p, int sensor_type, int n_sequences) {
    try {
        if (n_sequences < 1 || n_sequences > 256) return nullptr;
        return static_cast<lvt_handle>(create_context(*p, sensor_type, n_sequences));
    } catch (...) {
    }
    return nullptr;
}
LVT_API int lvt_amd_batch_size(lvt_handle h) { return static_cast<Context *>(h)->B; }

LVT_API void lvt_amd_batch_track_device_async(lvt_handle h, const void *const *d_left, const void *const *d_right, int n_rows, int n_cols,
                                              int pitch_bytes) {
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        if (!size_ok(c, n_rows, n_cols) || (pitch_bytes & 15)) {
            c->set_error("lvt_amd_batch_track_device: image size / pitch mismatch");
            return;
        }
        make_room(c);
        for (int s = 0; s < c->B; s++) {
            FrameArgs &f = c->h_fargs[(size_t)(c->enq % RING) * c->B + s];
            f.img[0] = static_cast<const uint8_t *>(d_left[s]);
            f.img[1] = static_cast<const uint8_t *>(d_right[s]);
            f.depth = nullptr;
            f.img_pitch = pitch_bytes;
            f.depth_pitch = 0;
            f.ext_corners = 0;
        f.absent = 0;
            f.n_ext[0] = f.n_ext[1] = 0;
        }
        enqueue_frame(c);
    } catch (...) {
    }
}
// poses of the oldest un-collected batch frame: R = B x 9 doubles (row-major 3x3 each), t = B x 3, status = B ints
LVT_API void lvt_amd_batch_wait(lvt_handle h, double *R, double *t, int *status) {
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        if (c->done < c->enq) collect_oldest(c);
        for (int s = 0; s < c->B; s++) {
            const Ctl &k = last_ctl(c, s);
            if (R) for (int i = 0; i < 9; i++) R[9 * s + i] = k.out_R[i];
            if (t) for (int i = 0; i < 3; i++) t[3 * s + i] = k.out_t[i];
            if (status) status[s] = k.state;
        }
    } catch (...) {
    }
}
LVT_API void lvt_amd_batch_get_counts(lvt_handle h, int seq, int out[LVT_AMD_C__COUNT]) {
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        drain(c);
        if (seq < 0 || seq >= c->B) return;
        for (int i = 0; i < LVT_AMD_C__COUNT; i++) out[i] = last_ctl(c, seq).counts[i];
    } catch (...) {
    }
}

LVT_API void lvt_amd_wait(lvt_handle h, double R[3][3], double t[3]) {
    h = resolve_handle(h);
    if (is_slot(h)) {
        try {
            PoolSlot *S = static_cast<PoolSlot *>(h);
            (void)slot_collect(S);
            slot_result(S, R, t);
        } catch (...) {
        }
        return;
    }
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        if (c->early_pending) drain(c);
        if (c->enq - c->done <= 1) flush_pending(c);  // (a held host frame: kept back only while the device has other frames to work on)
        if (c->done < c->enq) collect_oldest(c);  // FIFO: the oldest frame not yet collected
        result_out(c, 0, R, t);
    } catch (...) {
    }
}

LVT_API int lvt_amd_wait_status(lvt_handle h, double R[3][3], double t[3]) {  // lvt_amd_wait + the tracking state AFTER that frame (1 / 2 / 3; -1: error)
    h = resolve_handle(h);
    if (is_slot(h)) {
        try {
            PoolSlot *S = static_cast<PoolSlot *>(h);
            (void)slot_collect(S);
            slot_result(S, R, t);
            return S->last.rec.state;
        } catch (...) {
        }
        return -1;
    }
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        if (c->early_pending) drain(c);
        if (c->enq - c->done <= 1) flush_pending(c);
        if (c->done < c->enq) collect_oldest(c);
        result_out(c, 0, R, t);
        return last_ctl(c).state;
    } catch (...) {
    }
    return -1;
}

// lvt_amd_wait_status with the pose as the tracker holds it (quaternion w x y z + position) instead of R, t: what lvt_system::track returns
LVT_API int lvt_amd_wait_pose(lvt_handle h, double q_wxyz[4], double p[3]) {
    h = resolve_handle(h);
    double R[3][3], t[3];
    const int st = lvt_amd_wait_status(h, R, t);
    if (st < 0) return st;
    const Ctl *rec = nullptr;
    if (is_slot(h)) rec = &static_cast<PoolSlot *>(h)->last.rec;
    else if (is_ctx(h)) rec = &last_ctl(static_cast<Context *>(h));
    if (!rec) return -1;
    for (int k = 0; k < 4; k++) q_wxyz[k] = rec->last_pose.q[k];
    for (int k = 0; k < 3; k++) p[k] = rec->last_pose.p[k];
    return st;
}

LVT_API void lvt_amd_track_device(lvt_handle h, const void *d_left, const void *d_right, int n_rows, int n_cols, int pitch_bytes,
                                  double R[3][3], double t[3]) {
    h = resolve_handle(h);
    if (is_slot(h)) {
        try {
            PoolSlot *S = static_cast<PoolSlot *>(h);
            slot_drain(S);
            if (slot_submit(S, static_cast<const uint8_t *>(d_left), static_cast<const uint8_t *>(d_right), n_rows, n_cols, pitch_bytes, false) != 0) return;
            (void)slot_collect(S);
            slot_result(S, R, t);
        } catch (...) {
        }
        return;
    }
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    if (!size_ok(c, n_rows, n_cols) || (pitch_bytes & 15)) {
        c->set_error("lvt_amd_track_device: image size / pitch mismatch");
        return;  // outputs untouched, like the reference on an exception
    }
    try {
        drain(c);
        c->sync_call = true;  // collected right away: k_triangulate delivers the record itself
        lvt_amd_track_device_async(h, d_left, d_right, n_rows, n_cols, pitch_bytes);
        c->sync_call = false;
        if (!wait_pose_or_frame(c, R, t)) result_out(c, 0, R, t);
    } catch (...) {
    }
}

static void upload_and_track(Context *c, const unsigned char *left, const void *second, bool rgbd, int n_rows, int n_cols, int ext,
                             const float *cl, int ncl, const float *cr, int ncr, double R[3][3], double t[3]) {
    if (!size_ok(c, n_rows, n_cols)) {
        c->set_error("lvt_track: image size differs from the configured img_width/img_height");
        return;
    }
    // (the previous synchronous call may have returned on its early pose: its frame's tail -- staged update, triangulation -- runs
    //  while this call copies the new images into the staging buffer of THIS frame's feature buffer; the drain comes after)
    flush_pending(c);  // a held lvt_amd_track_async frame goes out FIRST: it takes a frame number, and this frame's buffers (images, corner lists) are chosen by number
    const int par = (int)(c->enq % NPAR);
    hipStream_t sf = c->stream_f;
    const size_t nbytes = (size_t)n_rows * n_cols;
    if (!c->h_stage[par]) {  // first host-buffer call: [image | image] or [image | depth f32], each part padded to 16 B
        c->stage_img = (nbytes + 15) & ~(size_t)15;
        const size_t second_bytes = rgbd ? ((sizeof(float) * nbytes + 15) & ~(size_t)15) : c->stage_img;
        HIPCHK(c, hipHostMalloc((void **)&c->h_stage[par], c->stage_img + second_bytes, hipHostMallocDefault));
        HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_stage_dev[par], c->h_stage[par], 0));
    }
    // the borrowed buffers are copied into pinned memory by the CPU and the GPU pulls them from there -- unless the caller's
    // buffer already IS pinned host memory (hipHostMalloc / hipHostRegister, 16-byte aligned): then the GPU reads it in place
    // (this call only returns after the frame has been tracked, so the buffer outlives every read)
    // (k_stage_in / k_stage_copy never load outside [p, p + bytes): any base address and byte count qualify -- 1241 x 376 bytes are
    //  8 mod 16 -- except a depth image that is not even float-aligned)
    auto device_view = [](const void *p, size_t align) -> const uint8_t * {
        hipPointerAttribute_t a;
        if (((uintptr_t)p & (align - 1)) == 0 && hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost && a.devicePointer)
            return static_cast<const uint8_t *>(a.devicePointer);
        (void)hipGetLastError();  // (an ordinary malloc'ed pointer is "invalid value" to the query: not an error of this call)
        return nullptr;
    };
    const uint8_t *s0 = device_view(left, 1), *s1 = device_view(second, rgbd ? sizeof(float) : 1);
    c->planes_in_place += (s0 != nullptr) + (s1 != nullptr);
    c->planes_staged += (s0 == nullptr) + (s1 == nullptr);
    // The pulls are launched BEFORE the previous frame is collected: this frame's image planes (parity `par`) and staging buffer were last used NPAR
    // frames ago, and what may still be running -- the tail of the previous synchronous call, which returned on its early pose -- reads neither.
    // Two pageable images: the left one is pulled while the CPU still copies the right one.
    const bool split = !rgbd && !s0 && !s1;
    if (!s0) {
        std::memcpy(c->h_stage[par], left, nbytes);
        s0 = c->h_stage_dev[par];
    }
    if (split) hipLaunchKernelGGL(k_stage_in, dim3(128, 1), dim3(256), 0, sf, s0, s0, c->d_img[par][0], c->d_img[par][0], n_cols, n_rows, c->pitch);
    if (!s1 && !rgbd) {
        std::memcpy(c->h_stage[par] + c->stage_img, second, nbytes);
        s1 = c->h_stage_dev[par] + c->stage_img;
    }
    if (split) hipLaunchKernelGGL(k_stage_in, dim3(128, 1), dim3(256), 0, sf, s1, s1, c->d_img[par][1], c->d_img[par][1], n_cols, n_rows, c->pitch);
    else hipLaunchKernelGGL(k_stage_in, dim3(128, rgbd ? 1 : 2), dim3(256), 0, sf, s0, s1, c->d_img[par][0], c->d_img[par][1], n_cols, n_rows, c->pitch);
    drain(c);
    FrameArgs &f = c->h_fargs[(size_t)(c->enq % RING) * c->B];
    f.img[0] = c->d_img[par][0];
    f.img[1] = c->d_img[par][1];
    f.img_pitch = c->pitch;
    f.depth = nullptr;
    f.depth_pitch = 0;
    if (rgbd) {
        // The depth plane (1.2 MB: ~60 us of CPU copy into the staging buffer when the caller's buffer is pageable, ~45 us of PCIe pull)
        // is not needed before k_gather's depth filter.  Both happen once the detection kernels are enqueued -- the copy on the
        // host while the GPU runs them, the pull on the early stream (idle until this frame's features exist) beside them -- and
        // the feature stream waits for the pull in front of k_gather.
        f.depth = c->d_depth[par];
        f.depth_pitch = n_cols;
        c->before_gather = [c, s1, second, nbytes, par, sf]() {
            const uint8_t *src = s1;
            if (!src) {
                std::memcpy(c->h_stage[par] + c->stage_img, second, sizeof(float) * nbytes);
                src = c->h_stage_dev[par] + c->stage_img;
            }
            hipStream_t sd = c->events_only ? sf : c->stream_e;
            hipLaunchKernelGGL(k_stage_copy, dim3(256), dim3(256), 0, sd, reinterpret_cast<const float *>(src), reinterpret_cast<float *>(c->d_depth[par]), nbytes);
            if (sd != sf) {
                (void)hipEventRecord(c->ev_depth, sd);
                c->depth_wait = true;
            }
        };
    }
    f.ext_corners = ext;
    f.ext_xy[0] = f.ext_xy[1] = nullptr;  // (the context's own lists, d_ext[par])
    f.absent = 0;
    f.n_ext[0] = ncl;
    f.n_ext[1] = ncr;
    if (ext) {
        if (ncl) HIPCHK(c, hipMemcpyAsync(c->d_ext[par][0], cl, sizeof(float) * 2 * (size_t)ncl, hipMemcpyHostToDevice, sf));
        if (ncr) HIPCHK(c, hipMemcpyAsync(c->d_ext[par][1], cr, sizeof(float) * 2 * (size_t)ncr, hipMemcpyHostToDevice, sf));
    }
    c->sync_call = true;  // collected right away: k_triangulate delivers the record itself
    enqueue_frame(c);
    c->sync_call = false;
    if (!wait_pose_or_frame(c, R, t)) result_out(c, 0, R, t);
}

// Asynchronous counterpart of upload_and_track (lvt_amd_track_async / lvt_amd_track_rgbd_async): borrowed HOST images, the frame is enqueued and the
// call returns; the pose comes out of the same FIFO as lvt_amd_track_device_async's (lvt_amd_wait / lvt_amd_wait_status).  Pageable buffers are copied
// into this slot's pinned staging buffer during the call; page-locked ones are pulled where they lie (and must stay valid until the frame is collected).
// Stereo: the frame is HELD (Context::pend) until the next one arrives -- its k_cells launch then carries the workgroups that pull that next frame's
// images (24 us of PCIe reads beside 66 us of corner cells: on no stream's chain; SURVEY 8e "pinned H2D staging double-buffered") -- or until
// lvt_amd_wait* finds the device about to run dry / any other entry point needs the launch chain.  A frame nobody pulled for pulls its own images at the
// head of its feature stage (k_stage_in), as every RGB-D frame does.
// Returns 0 when the frame was enqueued, -1 when it was rejected (nothing enqueued; lvt_amd_last_error says why).
static int upload_async(Context *c, const unsigned char *left, const void *second, bool rgbd, int n_rows, int n_cols) {
    if (c->B != 1 || (rgbd ? c->sensor != 2 : c->sensor != 1)) {
        c->set_error("lvt_amd_track_async: wrong sensor type for this entry point (or a batch handle)");
        return -1;
    }
    if (!left || !second || !size_ok(c, n_rows, n_cols)) {
        c->set_error("lvt_amd_track_async: image size differs from the configured img_width/img_height (the frame was NOT enqueued)");
        return -1;
    }
    if (c->early_pending) drain(c);
    if (rgbd) flush_pending(c);
    const int held = c->pend.valid ? 1 : 0;  // (the held frame owns slot enq % RING)
    while (c->enq + held - c->done >= RING - 1) collect_oldest(c);
    const int slot = (int)((c->enq + held) % RING);
    const size_t nbytes = (size_t)n_rows * n_cols, plane = (size_t)c->pitch * c->prm.H;
    if (!c->d_img_ring[0][0]) {  // first asynchronous host-buffer call
        for (int r = 0; r < RING; r++) {
            for (int e = 0; e < (rgbd ? 1 : 2); e++) c->d_img_ring[r][e] = c->dalloc<uint8_t>(plane + 64);
            if (rgbd) c->d_depth_ring[r] = c->dalloc<float>(nbytes + 4);
        }
    }
    auto device_view = [](const void *p, size_t align) -> const uint8_t * {
        hipPointerAttribute_t a;
        if (((uintptr_t)p & (align - 1)) == 0 && hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost && a.devicePointer)
            return static_cast<const uint8_t *>(a.devicePointer);
        (void)hipGetLastError();
        return nullptr;
    };
    const uint8_t *s0 = device_view(left, 1), *s1 = device_view(second, rgbd ? sizeof(float) : 1);
    c->planes_in_place += (s0 != nullptr) + (s1 != nullptr);
    c->planes_staged += (s0 == nullptr) + (s1 == nullptr);
    const size_t img_b = (nbytes + 15) & ~(size_t)15, second_b = rgbd ? ((sizeof(float) * nbytes + 15) & ~(size_t)15) : img_b;
    if ((!s0 || !s1) && !c->h_stage_ring[slot]) {
        c->stage_ring_bytes = img_b + second_b;
        HIPCHK(c, hipHostMalloc((void **)&c->h_stage_ring[slot], c->stage_ring_bytes, hipHostMallocDefault));
        HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_stage_ring_dev[slot], c->h_stage_ring[slot], 0));
    }
    hipStream_t sp = c->stream_f;
    uint8_t *d0 = c->d_img_ring[slot][0], *d1 = c->d_img_ring[slot][1];
    if (!rgbd) {
        if (!s0) {
            std::memcpy(c->h_stage_ring[slot], left, nbytes);
            s0 = c->h_stage_ring_dev[slot];
        }
        if (!s1) {
            std::memcpy(c->h_stage_ring[slot] + img_b, second, nbytes);
            s1 = c->h_stage_ring_dev[slot] + img_b;
        }
        // the frame held so far goes out now, and its k_cells launch pulls THIS frame's images beside its cells
        const bool carried = c->pend.valid && c->fuse_pull;
        if (c->pend.valid) {
            const NextPull np{{s0, s1}, {d0, d1}, n_cols, n_rows, c->pitch};
            flush_pending(c, carried ? &np : nullptr);
        }
        Context::PendingFrame &q = c->pend;
        q.valid = true, q.pulled = carried;
        q.src[0] = s0, q.src[1] = s1, q.dst[0] = d0, q.dst[1] = d1;
        q.f = FrameArgs{};
        q.f.img[0] = d0, q.f.img[1] = d1;
        q.f.img_pitch = c->pitch;
        c->async_frames++;
        if (!c->fuse_pull) flush_pending(c);
        return 0;
    } else {
        if (!s0) {
            std::memcpy(c->h_stage_ring[slot], left, nbytes);
            s0 = c->h_stage_ring_dev[slot];
        }
        hipLaunchKernelGGL(k_stage_in, dim3(128, 1), dim3(256), 0, sp, s0, s0, d0, d0, n_cols, n_rows, c->pitch);
        if (!s1) {
            std::memcpy(c->h_stage_ring[slot] + img_b, second, sizeof(float) * nbytes);
            s1 = c->h_stage_ring_dev[slot] + img_b;
        }
        hipLaunchKernelGGL(k_stage_copy, dim3(256), dim3(256), 0, sp, reinterpret_cast<const float *>(s1), c->d_depth_ring[slot], nbytes);
    }
    FrameArgs &f = c->h_fargs[(size_t)slot * c->B];
    f.img[0] = d0;
    f.img[1] = rgbd ? d0 : d1;
    f.img_pitch = c->pitch;
    f.depth = rgbd ? c->d_depth_ring[slot] : nullptr;
    f.depth_pitch = rgbd ? n_cols : 0;
    f.ext_corners = 0;
    f.absent = 0;
    f.n_ext[0] = f.n_ext[1] = 0;
    c->async_frames++;
    enqueue_frame(c);
    return 0;
}

LVT_API int lvt_amd_track_async(lvt_handle h, const unsigned char *left, const unsigned char *right, int n_rows, int n_cols) {
    h = resolve_handle(h);
    if (is_slot(h)) {
        try {
            return slot_submit(static_cast<PoolSlot *>(h), left, right, n_rows, n_cols, 0, true);
        } catch (...) {
        }
        return -1;
    }
    Context *c = static_cast<Context *>(h);
    if (!c) return -1;
    DeviceGuard guard(c);
    try {
        return upload_async(c, left, right, false, n_rows, n_cols);
    } catch (...) {
    }
    return -1;
}
LVT_API int lvt_amd_track_rgbd_async(lvt_handle h, const unsigned char *gray, const float *depth, int n_rows, int n_cols) {
    h = resolve_handle(h);
    Context *c = static_cast<Context *>(h);
    if (!c || !is_ctx(h)) return -1;
    DeviceGuard guard(c);
    try {
        return upload_async(c, gray, depth, true, n_rows, n_cols);
    } catch (...) {
    }
    return -1;
}

LVT_API void lvt_track(lvt_handle h, unsigned char *left, unsigned char *right, int n_rows, int n_cols, double R[3][3], double t[3]) {
    h = resolve_handle(h);
    if (is_slot(h)) {
        try {
            PoolSlot *S = static_cast<PoolSlot *>(h);
            slot_drain(S);
            if (slot_submit(S, left, right, n_rows, n_cols, 0, true) != 0) return;  // outputs untouched, like the reference on an exception
            (void)slot_collect(S);
            slot_result(S, R, t);
        } catch (...) {
        }
        return;
    }
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        if (c->sensor != 1) return;  // the reference cannot run RGB-D through this entry either
        upload_and_track(c, left, right, false, n_rows, n_cols, 0, nullptr, 0, nullptr, 0, R, t);
    } catch (...) {
    }
}

LVT_API void lvt_amd_track_rgbd(lvt_handle h, const unsigned char *gray, const float *depth, int n_rows, int n_cols, double R[3][3],
                                double t[3]) {
    h = resolve_handle(h);
    if (!is_ctx(h)) return;
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        if (c->sensor != 2) return;
        upload_and_track(c, gray, depth, true, n_rows, n_cols, 0, nullptr, 0, nullptr, 0, R, t);
    } catch (...) {
    }
}

LVT_API void lvt_track_with_external_corners(lvt_handle h, unsigned char *left, unsigned char *right, int n_rows, int n_cols,
                                             double corners_left[][2], int n_corners_left, double corners_right[][2],
                                             int n_corners_right, double R[3][3], double t[3]) {
    h = resolve_handle(h);
    if (!is_ctx(h) && !is_slot(h)) return;
    try {
        if (n_corners_left < 0 || n_corners_right < 0) return;
        char cut[160] = "";
        if (n_corners_left > EXT_MAX || n_corners_right > EXT_MAX)  // (the reference takes any number; here the lists are cut -- never silently)
            std::snprintf(cut, sizeof(cut), "lvt_track_with_external_corners: %d / %d corners, only the first %d of a list are used", n_corners_left,
                          n_corners_right, EXT_MAX);
        const int ncl = std::min(n_corners_left, EXT_MAX), ncr = std::min(n_corners_right, EXT_MAX);
        std::vector<float> cl(2 * (size_t)ncl + 2), cr(2 * (size_t)ncr + 2);
        for (int i = 0; i < ncl; i++) {  // doubles narrowed to float (lvt_c.cpp:104-117)
            cl[2 * i] = (float)corners_left[i][0];
            cl[2 * i + 1] = (float)corners_left[i][1];
        }
        for (int i = 0; i < ncr; i++) {
            cr[2 * i] = (float)corners_right[i][0];
            cr[2 * i + 1] = (float)corners_right[i][1];
        }
        if (is_slot(h)) {  // a pooled handle: the lists are deposited with the frame and ride its lock-step step (lvt_pool.h)
            PoolSlot *S = static_cast<PoolSlot *>(h);
            slot_drain(S);
            const float *const lists[2] = {cl.data(), cr.data()};
            const int counts[2] = {ncl, ncr};
            if (slot_submit(S, left, right, n_rows, n_cols, 0, true, lists, counts) != 0) return;
            if (cut[0]) {
                std::lock_guard<std::mutex> g(S->pool->mu);
                S->err = cut;
            }
            (void)slot_collect(S);
            slot_result(S, R, t);
            return;
        }
        Context *c = static_cast<Context *>(h);
        DeviceGuard guard(c);
        if (c->sensor != 1) return;
        if (cut[0]) c->set_error(cut);
        upload_and_track(c, left, right, false, n_rows, n_cols, 1, cl.data(), ncl, cr.data(), ncr, R, t);
    } catch (...) {
    }
}

LVT_API int lvt_get_status(lvt_handle h) {
    h = resolve_handle(h);
    if (is_slot(h)) {
        try {
            PoolSlot *S = static_cast<PoolSlot *>(h);
            slot_drain(S);
            return S->last.rec.state;
        } catch (...) {
        }
        return -1;
    }
    try {
        Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
        drain(c);
        return last_ctl(c).state;
    } catch (...) {
    }
    return -1;
}

// ---- introspection (of the most recently COLLECTED frame; drains the pipeline first) -----------------
LVT_API void lvt_amd_get_counts(lvt_handle h, int out[LVT_AMD_C__COUNT]) {
    h = resolve_handle(h);
    try {
        View v;
        if (!view_of(h, v)) return;
        for (int i = 0; i < LVT_AMD_C__COUNT; i++) out[i] = v.rec->counts[i];
    } catch (...) {
    }
}

LVT_API int lvt_amd_get_features(lvt_handle h, int eye, float *xy, float *resp, uint8_t *desc, int cap) {
    h = resolve_handle(h);
    try {
        View v;
        if (!view_of(h, v)) return -1;
        Context *c = v.c;
        const Feat &F = c->h_seqs[v.s].fb[v.par].feat[eye ? 1 : 0];
        const int n = read_scalar(c, F.n), m = std::min(n, cap);
        std::vector<float> x(m), y(m);
        d2h(c, x.data(), F.x, m);
        d2h(c, y.data(), F.y, m);
        if (xy)
            for (int i = 0; i < m; i++) xy[2 * i] = x[i], xy[2 * i + 1] = y[i];
        d2h(c, resp, F.resp, m);
        d2h(c, desc, reinterpret_cast<const uint8_t *>(F.desc), (size_t)m * 32);
        return n;
    } catch (...) {
    }
    return -1;
}

LVT_API int lvt_amd_get_matches(lvt_handle h, int *feat_idx, double *xyz, int cap) {
    h = resolve_handle(h);
    try {
        View v;
        if (!view_of(h, v)) return -1;
        Context *c = v.c;
        const int n = v.rec->n_matches, m = std::min(n, cap);
        d2h(c, feat_idx, c->h_seqs[v.s].pnp_feat, m);
        d2h(c, xyz, c->h_seqs[v.s].pnp_X, (size_t)m * 3);
        return n;
    } catch (...) {
    }
    return -1;
}

LVT_API int lvt_amd_get_row_matches(lvt_handle h, int *pairs, int cap) {
    h = resolve_handle(h);
    try {
        View v;
        if (!view_of(h, v)) return -1;
        Context *c = v.c;
        const int n = v.rec->counts[C_N_ROW_MATCHES], m = std::min(n, cap);
        std::vector<int> l(m), r(m);
        d2h(c, l.data(), c->h_seqs[v.s].pair_l, m);
        d2h(c, r.data(), c->h_seqs[v.s].pair_r, m);
        for (int i = 0; i < m; i++) pairs[2 * i] = l[i], pairs[2 * i + 1] = r[i];
        return n;
    } catch (...) {
    }
    return -1;
}

static int get_points(Context *c, const MapSoA *bufs, const int *d_cur, const int *d_n, double *xyz, int *counter, int *age, uint8_t *desc,
                      int cap) {
    const int cur = read_scalar(c, d_cur), n = read_scalar(c, d_n), m = std::min(n, cap);
    const MapSoA &P = bufs[cur];
    d2h(c, xyz, P.pos, (size_t)m * 3);
    d2h(c, counter, P.counter, m);
    d2h(c, age, P.age, m);
    d2h(c, desc, reinterpret_cast<const uint8_t *>(P.desc), (size_t)m * 32);
    return n;
}
LVT_API int lvt_amd_get_map(lvt_handle h, double *xyz, int *counter, int *age, uint8_t *desc, int cap) {
    h = resolve_handle(h);
    try {
        View v;
        if (!view_of(h, v)) return -1;
        Context *c = v.c;
        const Seq &S = c->h_seqs[v.s];
        return get_points(c, S.map, S.map_cur, S.map_n, xyz, counter, age, desc, cap);
    } catch (...) {
    }
    return -1;
}
LVT_API int lvt_amd_get_staged(lvt_handle h, double *xyz, int *counter, uint8_t *desc, int cap) {
    h = resolve_handle(h);
    try {
        View v;
        if (!view_of(h, v)) return -1;
        Context *c = v.c;
        const Seq &S = c->h_seqs[v.s];
        return get_points(c, S.staged, S.staged_cur, S.staged_n, xyz, counter, nullptr, desc, cap);
    } catch (...) {
    }
    return -1;
}
// pose + state of the frame the last tracking call returned, WITHOUT waiting for that frame's tail: a synchronous call that returned on
// the pose k_pnp handed over reads it from the same pinned record (quaternion as the tracker holds it, not re-derived from R)
LVT_API int lvt_amd_get_last_pose(lvt_handle h, double q[4], double p[3]) {
    h = resolve_handle(h);
    if (is_slot(h)) {
        try {
            View v;
            if (!view_of(h, v)) return -1;
            for (int k = 0; k < 4; k++) q[k] = v.rec->last_pose.q[k];
            for (int k = 0; k < 3; k++) p[k] = v.rec->last_pose.p[k];
            return v.rec->state;
        } catch (...) {
        }
        return -1;
    }
    Context *c = static_cast<Context *>(h);
    if (!c) return -1;
    if (c->early_pending && c->enq > 0) {
        const PoseRec &r = c->h_pose[(size_t)((c->enq - 1) % RING) * c->B];
        for (int k = 0; k < 4; k++) q[k] = r.q[k];
        for (int k = 0; k < 3; k++) p[k] = r.p[k];
        return r.status;
    }
    DeviceGuard guard(c);
    try {
        drain(c);
        for (int k = 0; k < 4; k++) q[k] = last_ctl(c).last_pose.q[k];
        for (int k = 0; k < 3; k++) p[k] = last_ctl(c).last_pose.p[k];
        return last_ctl(c).state;
    } catch (...) {
    }
    return -1;
}
LVT_API void lvt_amd_get_pose(lvt_handle h, double q[4], double p[3]) {
    h = resolve_handle(h);
    try {
        View v;
        if (!view_of(h, v)) return;
        for (int k = 0; k < 4; k++) q[k] = v.rec->last_pose.q[k];
        for (int k = 0; k < 3; k++) p[k] = v.rec->last_pose.p[k];
    } catch (...) {
    }
}
LVT_API void lvt_amd_get_predicted_pose(lvt_handle h, double q[4], double p[3]) {
    h = resolve_handle(h);
    try {
        View v;
        if (!view_of(h, v)) return;
        for (int k = 0; k < 4; k++) q[k] = v.rec->predicted.q[k];
        for (int k = 0; k < 3; k++) p[k] = v.rec->predicted.p[k];
    } catch (...) {
    }
}
LVT_API void lvt_amd_get_debug(lvt_handle h, long long out[32]) {
    h = resolve_handle(h);
    if (!is_ctx(h)) return;
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        drain(c);
        for (int i = 0; i < 32; i++) out[i] = last_ctl(c).dbg[i];
    } catch (...) {
    }
}
LVT_API int lvt_amd_get_ordering(lvt_handle h) {  // 0: polling gates + early stream, 1: event barriers only, 2: a pooled handle (a seat of the device's shared lock-step chain)
    if (is_auto(h)) {  // (under the record's lock, see lvt_amd_last_error)
        AutoHandle *A = static_cast<AutoHandle *>(h);
        std::lock_guard<std::mutex> g(A->mu);
        return lvt_amd_get_ordering(A->impl);
    }
    if (is_slot(h)) return 2;
    return h && static_cast<Context *>(h)->events_only ? 1 : 0;
}
LVT_API void lvt_amd_get_timeline(lvt_handle h, long long out[16]) {  // of the frame collected last; a frame whose synchronous call
    h = resolve_handle(h);
    if (!is_ctx(h)) return;
    Context *c = static_cast<Context *>(h);                            // returned on its pose is collected first, nothing else is drained
    DeviceGuard guard(c);
    try {
        if (c->early_pending) drain(c);
    } catch (...) {
    }
    for (int i = 0; i < 16; i++) out[i] = last_ctl(c).dbg[32 + i];
}
LVT_API int lvt_amd_get_plane(lvt_handle h, int eye, int what, void *dst, int cap_bytes, int *pitch_out) {
    h = resolve_handle(h);
    if (!is_ctx(h)) return -1;
    Context *c = static_cast<Context *>(h);
    DeviceGuard guard(c);
    try {
        drain(c);
        const FrameBuf &FB = c->h_seqs[0].fb[c->last_par];
        const size_t n = (size_t)c->pitch * c->prm.H;
        const size_t bytes = n * (what == 0 ? 1 : 2);
        if ((size_t)cap_bytes < bytes) return -1;
        if (what == 0) HIPCHK(c, hipMemcpy(dst, FB.score[eye ? 1 : 0], bytes, hipMemcpyDeviceToHost));
        else HIPCHK(c, hipMemcpy(dst, FB.boxsum[eye ? 1 : 0], bytes, hipMemcpyDeviceToHost));
        if (pitch_out) *pitch_out = c->pitch;
        return (int)bytes;
    } catch (...) {
    }
    return -1;
}

// ---- stage entry: motion-only BA on caller data ---------------------------------------------------
// err_out (2n doubles, may be NULL): the edge errors the second chi2 gate saw; level_out (n ints, may be NULL): 1 = demoted by a gate;
// *borderline (may be NULL): gate decisions taken within PNP_GATE_MARGIN of the threshold
// trace (trace_cap rows x 4 doubles, may be NULL): one row per LM trial {lambda, chi2 at the estimate, chi2 of the trial, rho}; stats (may be NULL):
// {trials, rejected trials, passes ended by Terminate}
LVT_API int lvt_amd_pnp_trace(const lvt_amd_params *pin, const double q_in[4], const double p_in[3], const double *pts, const float *obs, int n,
                              double q_out[4], double p_out[3], int *n_solve_calls, double *err_out, int *level_out, int *borderline,
                              double *trace, int trace_cap, int stats[3]) {
    Params prm;
    lvt_amd_params tmp = *pin;
    if (tmp.img_width <= 0) tmp.img_width = 64;
    if (tmp.img_height <= 0) tmp.img_height = 64;
    if (!derive_params(tmp, 1, prm)) return -1;
    double *dX = nullptr, *dErr = nullptr;
    float *dObs = nullptr;
    int8_t *dLevel = nullptr;
    Pose *dOut = nullptr;
    int *dInfo = nullptr;
    double *dTrace = nullptr;
    if (!trace || trace_cap < 0) trace_cap = 0;
    int rc = -1;
    const size_t nn = (size_t)std::max(n, 1);
    if (hipMalloc((void **)&dX, nn * 24) == hipSuccess && hipMalloc((void **)&dErr, nn * 16 + 16) == hipSuccess &&
        hipMalloc((void **)&dObs, nn * 8) == hipSuccess && hipMalloc((void **)&dLevel, nn) == hipSuccess &&
        hipMalloc((void **)&dOut, sizeof(Pose)) == hipSuccess && hipMalloc((void **)&dInfo, 32) == hipSuccess &&
        hipMalloc((void **)&dTrace, (size_t)std::max(trace_cap, 1) * 32) == hipSuccess) {
        (void)hipMemcpy(dX, pts, (size_t)n * 24, hipMemcpyHostToDevice);
        (void)hipMemcpy(dObs, obs, (size_t)n * 8, hipMemcpyHostToDevice);
        Pose prior;
        for (int k = 0; k < 4; k++) prior.q[k] = q_in[k];
        for (int k = 0; k < 3; k++) prior.p[k] = p_in[k];
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pnp_standalone), hipFuncAttributeMaxDynamicSharedMemorySize, PNP_DYN_BYTES);
        hipLaunchKernelGGL(k_pnp_standalone, dim3(1), dim3(PNP_THREADS), PNP_DYN_BYTES, 0, prm, prior, dX, dObs, dErr, dLevel, n, dOut, dInfo,
                           trace_cap ? dTrace : (double *)nullptr, trace_cap);
        Pose out;
        int info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpy(&out, dOut, sizeof(Pose), hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(info, dInfo, 32, hipMemcpyDeviceToHost) == hipSuccess) {
            if (stats) stats[0] = info[3], stats[1] = info[4], stats[2] = info[5];
            if (trace_cap && hipMemcpy(trace, dTrace, (size_t)std::min(trace_cap, std::max(info[3], 0)) * 32, hipMemcpyDeviceToHost) != hipSuccess) rc = -2;
            for (int k = 0; k < 4; k++) q_out[k] = out.q[k];
            for (int k = 0; k < 3; k++) p_out[k] = out.p[k];
            if (n_solve_calls) *n_solve_calls = info[0];
            if (borderline) *borderline = info[2];
            rc = (rc == -2) ? -1 : info[1];
            if (err_out && hipMemcpy(err_out, dErr, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
            if (level_out) {
                std::vector<int8_t> lv((size_t)std::max(n, 1));
                if (hipMemcpy(lv.data(), dLevel, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
                for (int i = 0; i < n; i++) level_out[i] = lv[i];
            }
        }
    }
    (void)hipFree(dX), (void)hipFree(dErr), (void)hipFree(dObs), (void)hipFree(dLevel), (void)hipFree(dOut), (void)hipFree(dInfo), (void)hipFree(dTrace);
    return rc;
}
LVT_API int lvt_amd_pnp_detail(const lvt_amd_params *pin, const double q_in[4], const double p_in[3], const double *pts, const float *obs, int n,
                               double q_out[4], double p_out[3], int *n_solve_calls, double *err_out, int *level_out, int *borderline) {
    return lvt_amd_pnp_trace(pin, q_in, p_in, pts, obs, n, q_out, p_out, n_solve_calls, err_out, level_out, borderline, nullptr, 0, nullptr);
}
LVT_API int lvt_amd_pnp(const lvt_amd_params *pin, const double q_in[4], const double p_in[3], const double *pts, const float *obs, int n,
                        double q_out[4], double p_out[3], int *n_solve_calls) {
    return lvt_amd_pnp_detail(pin, q_in, p_in, pts, obs, n, q_out, p_out, n_solve_calls, nullptr, nullptr, nullptr);
}


// ---- stage entry: batched masked 2-NN Hamming matcher on device-resident problems --------------------
LVT_API float lvt_amd_hamming_match_batched_n(const void *q_desc, const void *q_xy, const void *t_desc, const void *t_xy, const void *t_flag,
                                              int B, int M, int N, float r2, int mode, int img_rows, int img_cols, void *out,
                                              void *hip_stream, int launches) {
    if (launches < 1) launches = 1;
    if (B > 0 && M > 0 && M <= HB_MMAX && N == 0) {  // an empty train set: knnMatch returns nothing for every query
        hipLaunchKernelGGL(k_hamming_none, dim3((unsigned)(((size_t)B * M + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                           static_cast<int4 *>(out), (size_t)B * M);
        return hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)) == hipSuccess ? 0.0f : -1.0f;
    }
    if (B <= 0 || M <= 0 || N <= 0 || N > HB_NMAX || M > HB_MMAX) {
        std::fprintf(stderr, "lvt_amd_hamming_match_batched: bad sizes B=%d M=%d N=%d (N <= %d, M <= %d)\n", B, M, N, HB_NMAX, HB_MMAX);
        return -1.0f;
    }
    HammingArgs a;
    a.q_desc = static_cast<const uint64_t *>(q_desc);
    a.q_xy = static_cast<const float2 *>(q_xy);
    a.t_desc = static_cast<const uint64_t *>(t_desc);
    a.t_xy = static_cast<const float2 *>(t_xy);
    a.t_flag = static_cast<const uint8_t *>(t_flag);
    a.out = static_cast<int4 *>(out);
    a.M = M, a.N = N, a.r2 = r2, a.img_rows = img_rows, a.img_cols = img_cols;
    a.dbg = nullptr;
    long long *d_dbg = nullptr;
    if (std::getenv("LVT_AMD_HAMMING_DEBUG") && hipMalloc((void **)&d_dbg, 64) == hipSuccess && hipMemset(d_dbg, 0, 64) == hipSuccess) a.dbg = d_dbg;
    if (mode == 1) {
        a.nbx = 1, a.nby = img_rows + 1, a.csr = 0;
    } else {
        a.nbx = (int)std::ceil(img_cols / (float)HASH_CELL), a.nby = (int)std::ceil(img_rows / (float)HASH_CELL);
        a.csr = std::max(1, (int)std::ceil(std::sqrt(r2) / (float)HASH_CELL));
    }
    const size_t lds = hamming_lds_bytes(N, M, a.nbx * a.nby);
    if (lds > 160 * 1024) {
        std::fprintf(stderr, "lvt_amd_hamming_match_batched: %zu bytes of LDS needed\n", lds);
        return -1.0f;
    }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    hipEvent_t e0, e1;
    {
        const hipError_t c0 = hipEventCreate(&e0), c1 = hipEventCreate(&e1);
        if (c0 != hipSuccess || c1 != hipSuccess) {
            std::fprintf(stderr, "lvt_amd_hamming_match_batched: hipEventCreate failed: %s / %s\n", hipGetErrorString(c0), hipGetErrorString(c1));
            return -1.0f;
        }
    }
    float ms = -1.0f;
    // the variant fixes how many candidate ranges a query keeps in registers (k_hamming.hip)
    void (*kern)(HammingArgs) = nullptr;
    {
        const int qpt = (M + HB_THREADS - 1) / HB_THREADS;
        const int tpt = (N + HB_THREADS - 1) / HB_THREADS;
#define LVT_PICK_T(MODE_, NSP_, Q_)                                                                                               \
    ((tpt == 1) ? k_hamming_batched<MODE_, NSP_, Q_, 1> : (tpt == 2) ? k_hamming_batched<MODE_, NSP_, Q_, 2> :                   \
     (tpt == 3) ? k_hamming_batched<MODE_, NSP_, Q_, 3> : k_hamming_batched<MODE_, NSP_, Q_, 4>)
#define LVT_PICK(MODE_, NSP_)                                                                                                     \
    kern = (qpt == 1) ? LVT_PICK_T(MODE_, NSP_, 1) : (qpt == 2) ? LVT_PICK_T(MODE_, NSP_, 2) : (qpt == 3) ? LVT_PICK_T(MODE_, NSP_, 3) \
                                                                                                          : LVT_PICK_T(MODE_, NSP_, 4)
        if (mode == 1) LVT_PICK(1, 1);
        else if (a.csr == 1) LVT_PICK(0, 3);
        else if (a.csr == 2) LVT_PICK(0, 5);
        else LVT_PICK(0, 0);
#undef LVT_PICK_T
#undef LVT_PICK
    }
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipEventRecord(e0, st);
    for (int l = 0; l < launches; l++) hipLaunchKernelGGL(kern, dim3(B), dim3(HB_THREADS), lds, st, a);
    const hipError_t elaunch = hipGetLastError();
    (void)hipEventRecord(e1, st);
    const hipError_t es = hipEventSynchronize(e1);
    hipError_t ee = hipErrorUnknown;
    if (es == hipSuccess && elaunch == hipSuccess) ee = hipEventElapsedTime(&ms, e0, e1);
    if (ee != hipSuccess || ms < 0)
        std::fprintf(stderr, "lvt_amd_hamming_match_batched: attr=%s launch=%s sync=%s elapsed=%s ms=%f (B=%d M=%d N=%d lds=%zu)\n",
                     hipGetErrorString(ea), hipGetErrorString(elaunch), hipGetErrorString(es), hipGetErrorString(ee), ms, B, M, N, lds);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (d_dbg) {
        long long hd[8] = {};
        (void)hipMemcpy(hd, d_dbg, 64, hipMemcpyDeviceToHost);
        if (hd[7] == 0) hd[7] = hd[5];
        std::fprintf(stderr, "hamming phases (cycles): load+zero %lld count %lld scan %lld scatter+qcount %lld qsort %lld radius+qsort2 %lld descriptors %lld total %lld\n",
                     hd[1] - hd[0], hd[2] - hd[1], hd[3] - hd[2], hd[4] - hd[3], hd[5] - hd[4], hd[7] - hd[5], hd[6] - hd[7], hd[6] - hd[0]);
        (void)hipFree(d_dbg);
    }
    return (ee != hipSuccess || ms < 0) ? -1.0f : ms * 1000.0f / (float)launches;
}

LVT_API float lvt_amd_hamming_match_batched(const void *q_desc, const void *q_xy, const void *t_desc, const void *t_xy, const void *t_flag,
                                            int B, int M, int N, float r2, int mode, int img_rows, int img_cols, void *out,
                                            void *hip_stream) {
    return lvt_amd_hamming_match_batched_n(q_desc, q_xy, t_desc, t_xy, t_flag, B, M, N, r2, mode, img_rows, img_cols, out, hip_stream, 1);
}


// ---- EuRoC pre-step: rectification -------------------------------------------------------------------------------------
struct Rectifier {
    int w = 0, h = 0, pitch = 0;
    float *d_map1 = nullptr, *d_map2 = nullptr;
    uint8_t *d_src = nullptr, *d_dst = nullptr;  // staging for the host-buffer entry point
};

LVT_API lvt_amd_rectifier lvt_amd_rectifier_create(const double K[9], const double D[5], const double R[9], const double Pnew[9], int width,
                                                   int height) {
    if (!K || !D || !R || !Pnew || width <= 0 || height <= 0) return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return nullptr;  // no CPU fallback
    Rectifier *r = new (std::nothrow) Rectifier;
    if (!r) return nullptr;
    r->w = width, r->h = height, r->pitch = ((width + 63) / 64) * 64;
    const size_t n = (size_t)width * height;
    if (hipMalloc((void **)&r->d_map1, n * 4) != hipSuccess || hipMalloc((void **)&r->d_map2, n * 4) != hipSuccess ||
        hipMalloc((void **)&r->d_src, n) != hipSuccess || hipMalloc((void **)&r->d_dst, (size_t)r->pitch * height) != hipSuccess) {
        lvt_amd_rectifier_destroy(r);
        return nullptr;
    }
    RectifyArgs a;
    {   // iR = (Pnew * R)^-1: 3x3 product in k-ascending order, closed-form inverse (cv::gemm / cv::invert for 3x3 doubles)
        double AR[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += Pnew[3 * i + k] * R[3 * k + j];
                AR[3 * i + j] = s;
            }
        const double *S = AR;
        double ir[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
        if (d != 0.) {
            d = 1. / d;
            ir[0] = (S[4] * S[8] - S[5] * S[7]) * d, ir[1] = (S[2] * S[7] - S[1] * S[8]) * d, ir[2] = (S[1] * S[5] - S[2] * S[4]) * d;
            ir[3] = (S[5] * S[6] - S[3] * S[8]) * d, ir[4] = (S[0] * S[8] - S[2] * S[6]) * d, ir[5] = (S[2] * S[3] - S[0] * S[5]) * d;
            ir[6] = (S[3] * S[7] - S[4] * S[6]) * d, ir[7] = (S[1] * S[6] - S[0] * S[7]) * d, ir[8] = (S[0] * S[4] - S[1] * S[3]) * d;
        }
        for (int k = 0; k < 9; k++) a.ir[k] = ir[k];
    }
    a.fx = K[0], a.fy = K[4], a.u0 = K[2], a.v0 = K[5];
    a.k1 = D[0], a.k2 = D[1], a.p1 = D[2], a.p2 = D[3], a.k3 = D[4];
    a.w = width, a.h = height;
    hipLaunchKernelGGL(k_rectify_map, dim3((height + 63) / 64), dim3(64), 0, 0, a, r->d_map1, r->d_map2);
    if (hipDeviceSynchronize() != hipSuccess) {
        lvt_amd_rectifier_destroy(r);
        return nullptr;
    }
    return r;
}

LVT_API void lvt_amd_rectifier_destroy(lvt_amd_rectifier h) {
    Rectifier *r = static_cast<Rectifier *>(h);
    if (!r) return;
    (void)hipFree(r->d_map1), (void)hipFree(r->d_map2), (void)hipFree(r->d_src), (void)hipFree(r->d_dst);
    delete r;
}

LVT_API int lvt_amd_rectify_device(lvt_amd_rectifier h, const void *d_src, int src_pitch, void *d_dst, int dst_pitch, void *hip_stream) {
    Rectifier *r = static_cast<Rectifier *>(h);
    if (!r || !d_src || !d_dst || src_pitch < r->w || dst_pitch < r->w || (dst_pitch & 3)) return -1;
    hipLaunchKernelGGL(k_rectify, dim3((dst_pitch / 4 + 255) / 256, r->h), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       static_cast<const uint8_t *>(d_src), r->w, r->h, src_pitch, r->d_map1, r->d_map2, r->w, r->h, static_cast<uint8_t *>(d_dst), dst_pitch);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

LVT_API int lvt_amd_rectify(lvt_amd_rectifier h, const unsigned char *src, unsigned char *dst) {
    Rectifier *r = static_cast<Rectifier *>(h);
    if (!r || !src || !dst) return -1;
    const size_t n = (size_t)r->w * r->h;
    if (hipMemcpy(r->d_src, src, n, hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (lvt_amd_rectify_device(r, r->d_src, r->w, r->d_dst, r->pitch, nullptr) != 0) return -1;
    return hipMemcpy2D(dst, (size_t)r->w, r->d_dst, (size_t)r->pitch, (size_t)r->w, (size_t)r->h, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

LVT_API int lvt_amd_rectifier_get_maps(lvt_amd_rectifier h, float *map1, float *map2) {
    Rectifier *r = static_cast<Rectifier *>(h);
    if (!r || !map1 || !map2) return -1;
    const size_t n = (size_t)r->w * r->h * 4;
    return (hipMemcpy(map1, r->d_map1, n, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(map2, r->d_map2, n, hipMemcpyDeviceToHost) == hipSuccess) ? 0 : -1;
}

// ---- odometry accumulator (lvt_ros.cpp:215-311 without ROS) -----------------------------------------------
LVT_API lvt_amd_odometry lvt_amd_odometry_create(lvt_handle h, const double base_to_sensor[12], int reset_pose_on_lost) {
    h = resolve_handle(h);
    try {
        Odometry *o = new Odometry();
        o->tracker = h;
        o->reset_pose_on_lost = reset_pose_on_lost != 0;
        if (base_to_sensor)
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) o->base_to_sensor.R[3 * i + j] = base_to_sensor[4 * i + j];
                o->base_to_sensor.p[i] = base_to_sensor[4 * i + 3];
            }
        return o;
    } catch (...) {
        return nullptr;
    }
}
LVT_API void lvt_amd_odometry_destroy(lvt_amd_odometry o) { delete static_cast<Odometry *>(o); }
LVT_API void lvt_amd_odometry_reset(lvt_amd_odometry op) {  // reset_vo (lvt_ros.cpp:184-198)
    Odometry *o = static_cast<Odometry *>(op);
    if (!o) return;
    if (o->tracker) lvt_amd_reset(o->tracker);
    o->restart_deltas();
    o->base_to_odom = Rigid();
}
LVT_API int lvt_amd_odometry_push_pose(lvt_amd_odometry op, const double R[3][3], const double t[3], int status, double stamp_sec,
                                       double pose_out[7], double twist_out[6]) {
    Odometry *o = static_cast<Odometry *>(op);
    if (!o || !R || !t) return -1;
    if (o->have_time && o->last_time > stamp_sec) return 0;  // older than the last published frame: ignored (lvt_ros.cpp:224-229)
    if (status == 3) {  // LOST: reset the tracker, optionally the accumulated pose; nothing is published (lvt_ros.cpp:241-253)
        if (o->tracker) lvt_amd_reset(o->tracker);
        if (o->reset_pose_on_lost) {
            o->base_to_odom = Rigid();
            o->restart_deltas();
        }
        return 0;
    }
    return o->push(&R[0][0], t, stamp_sec, pose_out, twist_out) ? 1 : 0;
}
LVT_API int lvt_amd_odometry_update(lvt_amd_odometry op, unsigned char *left, unsigned char *right, int n_rows, int n_cols, double stamp_sec,
                                    double pose_out[7], double twist_out[6]) {
    Odometry *o = static_cast<Odometry *>(op);
    if (!o || !o->tracker || !left || !right) return -1;
    if (o->have_time && o->last_time > stamp_sec) return 0;  // (checked before tracking, like the node)
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, t[3] = {0, 0, 0};
    lvt_track(o->tracker, left, right, n_rows, n_cols, R, t);
    return lvt_amd_odometry_push_pose(op, R, t, lvt_get_status(o->tracker), stamp_sec, pose_out, twist_out);
}

}  // extern "C"
