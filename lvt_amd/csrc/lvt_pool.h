// lvt_pool.h -- pooled handles: independent lvt handles of one device folded into ONE lock-step launch chain (included by lvt_host.hip).
//
// The ABI's unit is the handle (lvt_create, lvt/src/lvt_c.h:57-62); the reference's handles are independent objects (lvt_local_map.cpp:60 is
// their only shared state).  Independent handles with a launch chain EACH share the process's few hardware queues badly: 2 / 4 / 8 handles reach
// 0.95 x / 1.47 x / 0.65 x of ONE handle's frame rate (profiles/r03_independent_handles.md) while a lock-step batch of 8 sequences reaches 4.3 x.
// A pooled handle (lvt_amd_create_pooled, or LVT_AMD_POOL=1 for every handle lvt_create / lvt_amd_create make) is a SLOT of one shared batch
// context per device:
//   * callers deposit frames into their slot's queue from any thread (asynchronous entry points: lvt_amd_track_device_async / lvt_amd_track_async
//     + lvt_amd_wait; the synchronous ones deposit and wait);
//   * ONE submission thread per device folds the head frame of every slot that has one into the next lock-step step -- the batch's launch chain with
//     gridDim.z = slots -- and marks the other slots ABSENT: for them every kernel of the step is a no-op (FrameArgs::absent: the feature kernels
//     leave the buffer alone, the tracking chain's prologue finds the frame skipped and does not count it), so a slot's results do not depend on
//     what its neighbours do -- poses are identical to a solo handle's (tests/test_gpu_parity.py::test_pooled_handles_*);
//   * up to POOL_DEPTH steps are in flight; a completed step's records are handed to the slots' result queues.
// A lone synchronous caller pays for the two thread hand-overs and for a chain launched eight sequences wide; that is why pooling is opt-in.
#pragma once
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

namespace lvt {

constexpr int POOL_SLOTS = 16;  // sequences of the shared batch context (a step is launched for the seats in use: empty upper seats cost memory only)
constexpr int POOL_DEPTH = 3;   // lock-step steps in flight (bench.py's batch leg: 3 beats 2 and 4)
constexpr int POOL_SLOT_QUEUE = 4;  // frames one slot may have deposited + in flight (a fifth deposit waits); collected, un-read results do not count --
                                    // but a slot keeps at most RING - 1 frames un-read in all, like a solo handle (make_room)

struct Pool;
struct SlotFrame {   // a collected frame of one slot
    Ctl rec;
    int par = 0;
};
struct PoolSlot {
    uint64_t magic = SLOT_MAGIC;
    Pool *pool = nullptr;
    int slot = -1;
    struct Pending {
        const uint8_t *img[2];   // device images (pitch % 16 == 0) -- or, host = true, two planes of this slot's staging buffer `stage`
        int pitch;
        bool host;
        int stage;
        bool ext = false;        // lvt_track_with_external_corners: the lists lie in this slot's pinned h_ext[stage] ([left | right], EXT_MAX corners each)
        int n_ext[2] = {0, 0};
    };
    std::deque<Pending> pending;       // deposited, not yet in a step
    int in_flight = 0;                 // ... inside enqueued steps
    std::deque<SlotFrame> results;     // collected steps' records, FIFO, not yet handed to the caller
    SlotFrame last;                    // the frame handed out last (introspection reads it)
    long submitted = 0;
    std::string err;
    uint8_t *h_stage[POOL_SLOT_QUEUE] = {}, *h_stage_dev[POOL_SLOT_QUEUE] = {};   // pinned staging of host images [left | right]
    uint8_t *d_img[POOL_SLOT_QUEUE][2] = {};                                        // their pitched device planes
    float *h_ext[POOL_SLOT_QUEUE] = {};                                             // pinned external-corner lists (first use of lvt_track_with_external_corners)
    float *d_ext[POOL_SLOT_QUEUE][2] = {};
    size_t stage_img = 0;
};

struct Pool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    Context *ctx = nullptr;
    int device = 0, sensor = 1;
    lvt_amd_params params{};
    PoolSlot *slots[POOL_SLOTS] = {};
    int live = 0;
    std::deque<uint32_t> inflight;   // present-mask of every enqueued, un-collected step (FIFO)
    bool busy = false;               // the submission thread is inside HIP calls on ctx (lock released)
    bool stop = false;
    int quiet = 0;                   // callers waiting for exclusive use of ctx (pool_quiesce): no new step is launched while there are any
    std::thread worker;
    long steps = 0, slot_frames = 0;  // statistics: steps launched, slot frames folded into them
    std::string err_seen;
};

static std::mutex g_pool_mu;
static Pool *g_pools[MAX_DEVICES] = {};

static inline bool is_slot(lvt_handle h) { return h && *static_cast<const uint64_t *>(h) == SLOT_MAGIC; }
static inline bool is_ctx(lvt_handle h) { return h && *static_cast<const uint64_t *>(h) == CTX_MAGIC; }

// ---- the submission thread ------------------------------------------------------------------------------------------------------------
static void pool_enqueue_step(Pool *P, std::unique_lock<std::mutex> &lk) {
    Context *c = P->ctx;
    // (called with the lock held) fold the head frame of every slot that has one into the step
    uint32_t mask = 0;
    PoolSlot::Pending take[POOL_SLOTS];
    for (int s = 0; s < POOL_SLOTS; s++) {
        PoolSlot *S = P->slots[s];
        if (S && !S->pending.empty()) {
            take[s] = S->pending.front();
            S->pending.pop_front();
            S->in_flight++;
            mask |= 1u << s;
        }
    }
    int hi = 0;  // the step is launched for the seats up to the highest one in use (a seat that is taken later starts from reset hand-over words)
    for (int s = 0; s < POOL_SLOTS; s++)
        if (P->slots[s]) hi = s + 1;
    c->launch_seqs = hi;
    P->inflight.push_back(mask);
    P->steps++;
    P->slot_frames += __builtin_popcount(mask);
    P->busy = true;
    lk.unlock();
    const unsigned long long enq_before = c->enq;
    bool failed = false;
    try {
        make_room(c);  // (POOL_DEPTH < RING - 1: never blocks)
        const int par_slot = (int)(c->enq % RING);
        for (int s = 0; s < POOL_SLOTS; s++) {
            FrameArgs &f = c->h_fargs[(size_t)par_slot * c->B + s];
            std::memset(&f, 0, sizeof(f));
            if (!(mask & (1u << s))) {
                f.absent = 1;
                continue;
            }
            const PoolSlot::Pending &p = take[s];
            if (p.host) {  // pull this slot's staged images over PCIe at the head of the step's feature stage
                PoolSlot *S = P->slots[s];
                const int rows = c->prm.H, cols = c->prm.W;
                hipLaunchKernelGGL(k_stage_in, dim3(128, 2), dim3(256), 0, c->stream_f, S->h_stage_dev[p.stage], S->h_stage_dev[p.stage] + S->stage_img,
                                   S->d_img[p.stage][0], S->d_img[p.stage][1], cols, rows, c->pitch);
                f.img[0] = S->d_img[p.stage][0], f.img[1] = S->d_img[p.stage][1];
                f.img_pitch = c->pitch;
            } else {
                f.img[0] = p.img[0], f.img[1] = p.img[1];
                f.img_pitch = p.pitch;
            }
            if (p.ext) {  // the corner lists travel with the frame: copied from the slot's pinned list into its device lists in front of the feature stage
                PoolSlot *S = P->slots[s];
                f.ext_corners = 1;
                for (int e = 0; e < 2; e++) {
                    f.n_ext[e] = p.n_ext[e];
                    f.ext_xy[e] = S->d_ext[p.stage][e];
                    if (p.n_ext[e])
                        HIPCHK(c, hipMemcpyAsync(S->d_ext[p.stage][e], S->h_ext[p.stage] + (size_t)e * 2 * EXT_MAX, sizeof(float) * 2 * (size_t)p.n_ext[e],
                                                 hipMemcpyHostToDevice, c->stream_f));
                }
            }
        }
        enqueue_frame(c);
    } catch (...) {
        failed = true;
    }
    lk.lock();
    P->busy = false;
    // (enqueue_frame advances c->enq with its LAST statement: an exception always leaves c->enq == enq_before -- a partly launched chain included; the
    //  kernels already enqueued for it find the same frame number again with the next step and redo the buffers, the step's frames are handed back here.
    //  R / t of such a record are the previous frame's: only `state` = -1 and the handle's error text describe the failed frame.)
    if (failed && c->enq == enq_before) {
        // the step never reached the launch chain: nothing will complete for it.  Its frames are handed back as FAILED records (state -1: what
        // lvt_amd_wait_status returns for an error) with the reason on every handle that had a frame in it -- never the previous step's record.
        P->inflight.pop_back();
        P->steps--;
        P->slot_frames -= __builtin_popcount(mask);
        for (int s = 0; s < POOL_SLOTS; s++) {
            PoolSlot *S = P->slots[s];
            if (!S || !(mask & (1u << s))) continue;
            SlotFrame fr = S->last;
            fr.rec.state = -1;
            S->results.push_back(fr);
            S->in_flight--;
            S->err = c->err.empty() ? std::string("a pooled step could not be enqueued (the frame was NOT tracked)") : c->err;
        }
    }
    P->cv_done.notify_all();
}

static void pool_collect_step(Pool *P, std::unique_lock<std::mutex> &lk) {
    Context *c = P->ctx;
    const uint32_t mask = P->inflight.front();
    P->busy = true;
    lk.unlock();
    int par = 0;
    try {
        collect_oldest(c);
        par = c->last_par;
    } catch (...) {
    }
    lk.lock();
    P->busy = false;
    P->inflight.pop_front();
    for (int s = 0; s < POOL_SLOTS; s++) {
        PoolSlot *S = P->slots[s];
        if (!S || !(mask & (1u << s))) continue;
        SlotFrame fr;
        fr.rec = c->h_ctl[(size_t)c->last_slot * c->B + s];
        fr.par = par;
        S->results.push_back(fr);
        S->in_flight--;
        if (fr.rec.overflow) {
            char buf[96];
            std::snprintf(buf, sizeof(buf), "capacity overflow mask 0x%x", fr.rec.overflow);
            S->err = buf;
        }
    }
    if (c->err != P->err_seen) {  // a NEW report of the shared chain -- a gate time-out, a HIP error -- belongs to every slot (overflows are per slot, above)
        P->err_seen = c->err;
        if (c->err.compare(0, 17, "capacity overflow") != 0)
            for (int s = 0; s < POOL_SLOTS; s++)
                if (P->slots[s]) P->slots[s]->err = c->err;
    }
    P->cv_done.notify_all();
}

static void pool_worker(Pool *P) {
    (void)hipSetDevice(P->device);
    std::unique_lock<std::mutex> lk(P->mu);
    while (!P->stop) {
        int with_frames = 0, idle_live = 0;
        for (int s = 0; s < POOL_SLOTS; s++)
            if (P->slots[s]) {
                if (!P->slots[s]->pending.empty()) with_frames++;
                else if (P->slots[s]->in_flight == 0 && P->slots[s]->results.empty()) idle_live++;
            }
        const bool may_enqueue = with_frames > 0 && (int)P->inflight.size() < POOL_DEPTH && P->quiet == 0;
        if (!may_enqueue && P->inflight.empty()) {
            if (with_frames > 0) P->cv_work.wait_for(lk, std::chrono::microseconds(200));  // (held back by a caller's exclusive section: look again soon)
            else P->cv_work.wait(lk);
            continue;
        }
        if (may_enqueue) {
            // a lone step of synchronous callers: the others arrive within microseconds of each other (they all returned from the same step) --
            // give them 40 us before a step goes out with seats empty.  Never while steps are in flight (asynchronous callers keep their queues filled).
            if (P->inflight.empty() && idle_live > 0 && with_frames < P->live) {
                P->cv_work.wait_for(lk, std::chrono::microseconds(40));
                if (P->stop) break;
            }
            pool_enqueue_step(P, lk);
            continue;
        }
        pool_collect_step(P, lk);
    }
}

// exclusive use of the pool's context from a caller thread (introspection, reset, destroy): no step in flight, the submission thread parked
// (frames the other handles have deposited meanwhile stay in their queues: the thread launches nothing while `quiet` is raised, and cannot take the
//  lock before the caller's exclusive section ends)
static void pool_quiesce(Pool *P, std::unique_lock<std::mutex> &lk) {
    P->quiet++;
    P->cv_done.wait(lk, [&] { return !P->busy && P->inflight.empty(); });
    P->quiet--;
}

static bool same_params(const lvt_amd_params &a, const lvt_amd_params &b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

// a new slot of the device's pool (creating pool, batch context and submission thread with the first one); nullptr: no seat, other parameters,
// or no device
static PoolSlot *pool_join(const lvt_amd_params &prm, int sensor, int device) {
    if (sensor != 1) return nullptr;  // (stereo only: the RGB-D entry point hands over a depth plane the pool's staging does not carry)
    int ndev = 0, cur = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hipGetDevice(&cur) != hipSuccess) return nullptr;
    if (device < 0) device = cur;
    if (device >= ndev || device >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> g(g_pool_mu);
    Pool *P = g_pools[device];
    if (!P) {
        P = new Pool();
        P->device = device, P->sensor = sensor, P->params = prm;
        P->ctx = create_context(prm, sensor, POOL_SLOTS, device);
        if (!P->ctx) {
            delete P;
            return nullptr;
        }
        P->worker = std::thread(pool_worker, P);
        g_pools[device] = P;
    } else if (!same_params(P->params, prm) || P->sensor != sensor)
        return nullptr;
    std::unique_lock<std::mutex> lk(P->mu);
    int s = 0;
    while (s < POOL_SLOTS && P->slots[s]) s++;
    if (s == POOL_SLOTS) return nullptr;
    pool_quiesce(P, lk);
    {
        DeviceGuard guard(P->ctx);
        try {
            reset_state(P->ctx, s);  // a seat that was used before starts like a new handle
        } catch (...) {
            return nullptr;
        }
    }
    PoolSlot *S = new PoolSlot();
    S->pool = P, S->slot = s;
    std::memset(&S->last.rec, 0, sizeof(Ctl));
    S->last.rec.state = 1, S->last.rec.out_status = 1;
    S->last.rec.out_R[0] = S->last.rec.out_R[4] = S->last.rec.out_R[8] = 1.0;
    S->last.rec.last_pose.q[0] = S->last.rec.predicted.q[0] = 1.0;
    P->slots[s] = S;
    P->live++;
    return S;
}

static void pool_leave(PoolSlot *S) {
    Pool *P = S->pool;
    std::unique_lock<std::mutex> g(g_pool_mu);
    std::unique_lock<std::mutex> lk(P->mu);
    // the seat's own frames are through and the thread is not inside a launch that might look at the seat: enough to leave (the other seats go on)
    P->cv_done.wait(lk, [&] { return S->pending.empty() && S->in_flight == 0 && !P->busy; });
    P->slots[S->slot] = nullptr;
    P->live--;
    {
        DeviceGuard guard(P->ctx);
        for (int k = 0; k < POOL_SLOT_QUEUE; k++) {
            if (S->h_stage[k]) (void)hipHostFree(S->h_stage[k]);
            if (S->h_ext[k]) (void)hipHostFree(S->h_ext[k]);
            for (int e = 0; e < 2; e++) {
                if (S->d_img[k][e]) (void)hipFree(S->d_img[k][e]);
                if (S->d_ext[k][e]) (void)hipFree(S->d_ext[k][e]);
            }
        }
    }
    const bool last = P->live == 0;
    if (last) {
        P->stop = true;
        P->cv_work.notify_all();
    }
    lk.unlock();
    if (last) {
        P->worker.join();
        {
            DeviceGuard guard(P->ctx);
            delete P->ctx;
        }
        g_pools[P->device] = nullptr;
        delete P;
    }
    delete S;
}

// ---- a slot's tracking calls ---------------------------------------------------------------------------------------------------------------
// corners: nullptr, or {left list, right list} as x y floats with their counts (lvt_track_with_external_corners; host frames only)
static int slot_submit(PoolSlot *S, const uint8_t *l, const uint8_t *r, int rows, int cols, int pitch, bool host, const float *const corners[2] = nullptr,
                       const int n_corners[2] = nullptr) {
    Pool *P = S->pool;
    Context *c = P->ctx;
    if (rows != c->prm.H || cols != c->prm.W || (!host && (pitch & 15)) || !l || !r) {
        std::lock_guard<std::mutex> g(P->mu);
        S->err = "image size / pitch mismatch (the frame was NOT enqueued)";
        return -1;
    }
    std::unique_lock<std::mutex> lk(P->mu);
    // A staging slot is free again once its frame's step has been collected, so only frames deposited or in flight count here.  Collected results the
    // caller has not read yet do NOT: a single-threaded caller that enqueues a fifth frame before its first lvt_amd_wait must not wait for itself.  Like a
    // solo handle (make_room collects its oldest frame), a slot that runs further ahead than RING - 1 frames loses its oldest un-read result.
    P->cv_done.wait(lk, [&] { return (int)S->pending.size() + S->in_flight < POOL_SLOT_QUEUE; });
    while (!S->results.empty() && (int)S->results.size() + (int)S->pending.size() + S->in_flight + 1 > RING - 1) {
        S->last = S->results.front();
        S->results.pop_front();
    }
    PoolSlot::Pending p;
    p.img[0] = l, p.img[1] = r, p.pitch = pitch, p.host = host, p.stage = (int)(S->submitted % POOL_SLOT_QUEUE);
    S->submitted++;  // (the staging slot is reserved before the lock is dropped for the copy: a second thread depositing on this handle takes the next one)
    if (host) {
        const size_t nbytes = (size_t)rows * cols;
        if (!S->d_img[p.stage][1]) {  // (first use of this staging slot; HIP allocations under the pool's lock: the submission thread is not inside the chain's calls for long)
            DeviceGuard guard(c);
            S->stage_img = (nbytes + 15) & ~(size_t)15;
            if (hipHostMalloc((void **)&S->h_stage[p.stage], 2 * S->stage_img, hipHostMallocDefault) != hipSuccess ||
                hipHostGetDevicePointer((void **)&S->h_stage_dev[p.stage], S->h_stage[p.stage], 0) != hipSuccess ||
                hipMalloc((void **)&S->d_img[p.stage][0], (size_t)c->pitch * rows + 64) != hipSuccess ||
                hipMalloc((void **)&S->d_img[p.stage][1], (size_t)c->pitch * rows + 64) != hipSuccess) {
                // nothing half-built stays behind: the next deposit on this staging slot starts over
                if (S->h_stage[p.stage]) (void)hipHostFree(S->h_stage[p.stage]);
                if (S->d_img[p.stage][0]) (void)hipFree(S->d_img[p.stage][0]);
                if (S->d_img[p.stage][1]) (void)hipFree(S->d_img[p.stage][1]);
                S->h_stage[p.stage] = S->h_stage_dev[p.stage] = nullptr;
                S->d_img[p.stage][0] = S->d_img[p.stage][1] = nullptr;
                (void)hipGetLastError();
                S->err = "out of memory for a pooled handle's staging buffers (the frame was NOT enqueued)";
                S->submitted--;  // (still under the lock: the staging rotation must not run ahead of the frames really queued)
                return -1;
            }
        }
        if (corners) {
            if (!S->d_ext[p.stage][1]) {
                DeviceGuard guard(c);
                if (hipHostMalloc((void **)&S->h_ext[p.stage], sizeof(float) * 4 * (size_t)EXT_MAX, hipHostMallocDefault) != hipSuccess ||
                    hipMalloc((void **)&S->d_ext[p.stage][0], sizeof(float) * 2 * (size_t)EXT_MAX) != hipSuccess ||
                    hipMalloc((void **)&S->d_ext[p.stage][1], sizeof(float) * 2 * (size_t)EXT_MAX) != hipSuccess) {
                    if (S->h_ext[p.stage]) (void)hipHostFree(S->h_ext[p.stage]);
                    if (S->d_ext[p.stage][0]) (void)hipFree(S->d_ext[p.stage][0]);
                    if (S->d_ext[p.stage][1]) (void)hipFree(S->d_ext[p.stage][1]);
                    S->h_ext[p.stage] = nullptr;
                    S->d_ext[p.stage][0] = S->d_ext[p.stage][1] = nullptr;
                    (void)hipGetLastError();
                    S->err = "out of memory for a pooled handle's corner lists (the frame was NOT enqueued)";
                    S->submitted--;
                    return -1;
                }
            }
            p.ext = true;
            p.n_ext[0] = n_corners[0], p.n_ext[1] = n_corners[1];
        }
        uint8_t *dst = S->h_stage[p.stage];
        float *cdst = S->h_ext[p.stage];
        const size_t simg = S->stage_img;
        lk.unlock();  // (the copy of 0.9 MB does not hold the pool up: this staging slot is ours -- at most POOL_SLOT_QUEUE frames of a slot exist)
        std::memcpy(dst, l, nbytes);
        std::memcpy(dst + simg, r, nbytes);
        if (corners)
            for (int e = 0; e < 2; e++)
                if (n_corners[e]) std::memcpy(cdst + (size_t)e * 2 * EXT_MAX, corners[e], sizeof(float) * 2 * (size_t)n_corners[e]);
        lk.lock();
    }
    S->pending.push_back(p);
    P->cv_work.notify_one();
    return 0;
}

// the oldest un-collected frame of the slot becomes its "last" frame; false: nothing outstanding
static bool slot_collect(PoolSlot *S) {
    Pool *P = S->pool;
    std::unique_lock<std::mutex> lk(P->mu);
    if (S->results.empty() && S->pending.empty() && S->in_flight == 0) return false;
    P->cv_done.wait(lk, [&] { return !S->results.empty(); });
    S->last = S->results.front();
    S->results.pop_front();
    P->cv_done.notify_all();  // (a depositor of this slot may be waiting for the seat)
    return true;
}
static void slot_drain(PoolSlot *S) {
    while (slot_collect(S)) {
    }
}
static void slot_result(PoolSlot *S, double R[3][3], double t[3]) {
    const Ctl &h = S->last.rec;
    if (R)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R[i][j] = h.out_R[3 * i + j];
    if (t)
        for (int i = 0; i < 3; i++) t[i] = h.out_t[i];
}

}  // namespace lvt
