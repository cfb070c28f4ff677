/*
 * lvt_amd_ext.h -- ADDITIVE entry points of the MI355X-native library (nothing here is required of an
 * existing lvt_c caller).  They expose (a) the reference's C++-only API surface over the C-ABI
 * (lvt_system::create with an in-memory lvt_parameters, reset, RGB-D track), (b) zero-copy tracking on
 * images already resident in HBM, (c) read-back of per-frame intermediate results for stage-by-stage
 * parity tests, (d) the batched Hamming matcher micro-benchmark, (e) the EuRoC rectification pre-step and the odometry
 * accumulator either side of the path.
 *
 * Behaviour worth knowing:
 *  - lvt_track / lvt_amd_wait block by spinning on a completion flag in pinned host memory (one busy core, no interrupt latency).
 *  - Host images handed to lvt_track are copied into a pinned staging buffer; page-locked, 16-byte aligned caller buffers
 *    (hipHostMalloc / hipHostRegister) are read in place.
 *  - Environment, read by lvt_create: LVT_AMD_ORDERING=events orders the library's three streams with event barriers instead of
 *    polling gate kernels (needed under tools that serialise kernel dispatches, e.g. rocprofv3 --pmc; ~15 % slower).
 *    Unset, the first live handle of a process polls and single-sequence handles created beside it use events
 *    (lvt_amd_get_ordering); LVT_AMD_ORDERING=polling forces the gates.  A polling handle whose gate runs into its time limit (the streams
 *    do not run side by side) reports it through lvt_amd_last_error and orders with events from the next frame on, unless polling was forced.
 *  - lvt_amd_last_error reports capacity overflows, gate time-outs and "a stream waited 2 s" failures (which set LOST).
 */
#ifndef LVT_AMD_EXT_H__
#define LVT_AMD_EXT_H__

#include "lvt_c.h"
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors struct lvt_parameters (reference lvt/src/lvt_parameters.h:29-64), hot-path fields only */
typedef struct lvt_amd_params {
    float fx, fy, cx, cy;
    float baseline;
    int img_width, img_height;
    float k1, k2, p1, p2, k3;
    float near_plane_distance, far_plane_distance;
    float triangulation_ratio_test_threshold;
    float tracking_ratio_test_threshold;
    float descriptor_matching_threshold;
    int min_num_matches_for_tracking;
    int tracking_radius;
    int detection_cell_size;
    int max_keypoints_per_cell;
    int agast_threshold;
    int untracked_threshold;
    int staged_threshold;
    int triangulation_policy;
} lvt_amd_params;

/* reference lvt_parameters.cpp:29-52 */
LVT_API void lvt_amd_default_params(lvt_amd_params *p);
/* reference lvt_parameters.cpp:54-93; returns 1 on success, 0 if the file cannot be opened */
LVT_API int lvt_amd_params_from_file(const char *config_file_name, lvt_amd_params *p);
/* reference lvt_system::create (lvt_system.cpp:70-127): the path the example binaries use, where the
 * intrinsics are filled in after the YAML is read (kitti_example.cpp:98-106) */
LVT_API lvt_handle lvt_amd_create(const lvt_amd_params *p, int sensor_type);
/* A handle OWNS the HIP device that was current when it was created (lvt_create, lvt_amd_create, lvt_amd_batch_create) or the one
 * named here: every entry point makes that device current for the call and restores the caller's, so one process can drive one
 * handle per GPU from any of its threads.  lvt_amd_get_device: the owning device, -1 for a NULL handle. */
LVT_API lvt_handle lvt_amd_create_on_device(const lvt_amd_params *p, int sensor_type, int device);
LVT_API int lvt_amd_get_device(lvt_handle h);
/* POOLED handles: independent handles of one device folded into ONE lock-step launch chain by a per-device submission thread (lvt_pool.h).  Independent
 * handles with a launch chain each share the process's hardware queues badly -- 2 / 4 / 8 of them reach 0.95 x / 1.47 x / 0.65 x of one handle's frame
 * rate; as seats of a shared chain they reach what a lock-step batch of that size does.  A pooled handle takes ALL FIVE reference entry points (lvt_track,
 * lvt_track_with_external_corners -- the corner lists ride the frame's step --, lvt_get_status, lvt_create with LVT_AMD_POOL=1, lvt_destroy),
 * lvt_amd_track_device[_async], lvt_amd_track_async, lvt_amd_wait[_status], lvt_amd_reset and the per-frame introspection calls, from any thread (one
 * thread per handle at a time, as for every handle); its results are those of a solo handle fed the same frames.  Stereo only; every handle of a pool has
 * the same parameters; at most 16 per device; 4 frames deposited or in flight per handle (a fifth deposit waits for the oldest to complete -- never for the
 * caller's own lvt_amd_wait: results not read yet do not count; beyond 7 un-read frames the oldest result is dropped, as on a solo handle).  A step that
 * cannot be enqueued hands its frames back with lvt_amd_wait_status() == -1 and the reason in lvt_amd_last_error.  Returns NULL when there is no seat or the
 * parameters differ from the pool's.
 * Streams: a lock-step batch of 2 - 28 sequences -- hence every pool -- runs its feature stage on a CU-masked stream (7/8 of the CUs, spread over the XCDs
 * by the driver's round-robin; LVT_AMD_FEATURE_CUS=0 turns the mask off).  HIP can only create such a stream with default flags: work a caller puts on the NULL
 * stream synchronises with it (correct, but serialised) -- callers that overlap their own GPU work with tracking should use non-default streams.
 * LVT_AMD_POOL=1 makes lvt_create / lvt_amd_create / lvt_amd_create_on_device hand out pooled handles (falling back to solo ones);
 * lvt_amd_get_ordering() == 2 says a handle is pooled.  A lone synchronous caller pays two thread hand-overs and the batch code path (the chain is launched as
 * wide as the highest seat in use, and the first seat of a device allocates the 16-sequence batch context -- about sixteen handles' worth of device memory):
 * pooling is for processes that track several sequences at once.
 * AUTOMATIC SEATS (round 6): the reference's own create call, lvt_create, decides by itself.  A stereo handle starts on a launch chain of its own; when a second
 * lvt_create with the same parameters arrives on the device while the first handle has not been used yet (nothing but lvt_amd_get_device / _get_ordering /
 * _last_error was called on it), both -- and every later one -- become seats of the device's pool.  A handle that has tracked keeps its kind (a tracker's
 * state is not moved between chains); RGB-D handles and handles beyond the 16 seats stay solo.  So a process that creates its handles first and tracks afterwards,
 * one thread per sequence, gets the pool without an environment variable (2 / 4 / 8 / 16 handles: 1.25 / 2.27 / 4.33 / 6.74 x of one handle's frame rate instead of
 * 0.94 / - / 0.71 / -), and a lone handle never pays for it.  lvt_amd_create / lvt_amd_create_on_device hand out exactly what they are asked for (their per-stage
 * read-back is mostly a solo handle's).  LVT_AMD_AUTO_POOL=0 turns the automatic seats off. */
LVT_API lvt_handle lvt_amd_create_pooled(const lvt_amd_params *p, int sensor_type, int device /* -1: the current device */);
/* reference lvt_system::reset (lvt_system.cpp:44-68) */
LVT_API void lvt_amd_reset(lvt_handle h);
/* reference lvt_system::track, RGB-D branch (lvt_system.cpp:177-183): gray u8 + depth f32 (metres),
 * both tightly packed host buffers.  (Unreachable through the reference's own C-ABI) */
LVT_API void lvt_amd_track_rgbd(lvt_handle h, const unsigned char *gray, const float *depth, int n_rows,
                                int n_cols, double R[3][3], double t[3]);
/* lvt_track on images already resident in HBM (device pointers, row pitch in bytes, pitch % 16 == 0,
 * pointers 16-byte aligned).  No host<->device image traffic. */
LVT_API void lvt_amd_track_device(lvt_handle h, const void *d_left, const void *d_right, int n_rows,
                                  int n_cols, int pitch_bytes, double R[3][3], double t[3]);
/* asynchronous form: enqueue the frame on the handle's stream and return; the pose is fetched with
 * lvt_amd_wait().  Lets one host thread keep several sequences/GPUs busy. */
LVT_API void lvt_amd_track_device_async(lvt_handle h, const void *d_left, const void *d_right, int n_rows,
                                        int n_cols, int pitch_bytes);
/* asynchronous lvt_track / lvt_amd_track_rgbd on HOST buffers (the reference's own boundary hands over borrowed host images, lvt_c.cpp:64-89; its
 * callers decode frame t+1 while frame t tracks, kitti_example.cpp:113-138): the frame is enqueued and the call returns, the pose comes out of the same
 * FIFO (lvt_amd_wait / lvt_amd_wait_status), at most 6 frames may be un-collected (a further call first blocks on the oldest one).
 *  - pageable buffers are copied into a pinned staging ring during the call: they are the caller's again when it returns;
 *  - page-locked buffers (hipHostMalloc / hipHostRegister) are pulled over PCIe WHERE THEY LIE: they must stay valid and unchanged until that frame
 *    has been collected (lvt_amd_get_host_stats out[2] / out[3] count the planes that went either way).
 * A stereo frame is HELD until the next one arrives or somebody waits for it (lvt_amd_wait* release it when the device would otherwise run dry, every
 * other entry point releases it first): the held frame's corner-cell launch then carries the workgroups that pull the NEXT frame's images over PCIe
 * beside its cells, and the pull is on no stream's chain (out[5] of lvt_amd_get_host_stats counts the frames whose images came that way;
 * LVT_AMD_FUSED_PULL=0: every frame pulls its own images at the head of its feature stage, as RGB-D frames and held frames nobody followed do).
 * Returns 0 when the frame was enqueued; -1 when it was rejected -- wrong image size, NULL buffer, wrong sensor type, a batch handle -- and then
 * NOTHING was enqueued (frames in flight are unaffected, lvt_amd_last_error says why). */
LVT_API int lvt_amd_track_async(lvt_handle h, const unsigned char *left, const unsigned char *right, int n_rows, int n_cols);
LVT_API int lvt_amd_track_rgbd_async(lvt_handle h, const unsigned char *gray, const float *depth, int n_rows, int n_cols);
LVT_API void lvt_amd_wait(lvt_handle h, double R[3][3], double t[3]);
/* the same, returning the tracking state after THAT frame (1 not initialised, 2 tracking, 3 lost; -1 error) -- lvt_get_status would
 * drain the whole pipeline first */
LVT_API int lvt_amd_wait_status(lvt_handle h, double R[3][3], double t[3]);
/* the same with the pose as the tracker holds it (quaternion w x y z + position: what lvt_system::track returns) */
LVT_API int lvt_amd_wait_pose(lvt_handle h, double q_wxyz[4], double p[3]);
/* ---- lock-step batch of independent sequences on ONE GPU ----------------------------------------------
 * B sequences (e.g. several KITTI drives) advance frame by frame through a single launch chain (every kernel is
 * launched with gridDim.z = B); the latency-bound serial kernels of the path (pose refinement, greedy resolvers)
 * then occupy B compute units instead of one.  Sequences stay fully independent (no cross-sequence data). */
LVT_API lvt_handle lvt_amd_batch_create(const lvt_amd_params *p, int sensor_type, int n_sequences);
LVT_API int lvt_amd_batch_size(lvt_handle h);
/* d_left / d_right: host arrays of B device pointers (one stereo pair per sequence, same size & pitch) */
LVT_API void lvt_amd_batch_track_device_async(lvt_handle h, const void *const *d_left, const void *const *d_right,
                                              int n_rows, int n_cols, int pitch_bytes);
/* oldest un-collected frame: R = B x 9 doubles, t = B x 3 doubles, status = B ints (any may be NULL) */
LVT_API void lvt_amd_batch_wait(lvt_handle h, double *R, double *t, int *status);
LVT_API void lvt_amd_batch_get_counts(lvt_handle h, int seq, int out[32]);

/* run all work of this handle on an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream) */
LVT_API void lvt_amd_set_stream(lvt_handle h, void *hip_stream);
/* last HIP error string seen by this handle ("" if none); overflow / capacity diagnostics too.
 * BLOCKING: when the last synchronous call returned on its early pose (the frame's tail -- staged update, triangulation -- was still running), this
 * call first collects that frame: its own reports (capacity overflow, a gate that timed out, a skipped frame) arrive with its full record.  A caller
 * that checks the string after EVERY lvt_track therefore gives up the overlap of that tail with its next upload (~30 us per frame); check it every N
 * frames, or use lvt_amd_get_last_pose for pose + state, which never waits.  Not thread-safe against a concurrent call on the same handle (like every
 * entry point); the returned pointer is valid until the next call on the handle (pooled handles: until the calling thread's next call). */
LVT_API const char *lvt_amd_last_error(lvt_handle h);

/* per-kernel timing with HIP events recorded on the handle's stream around every launch of the frame chain.
 * enable=1 resets the accumulators.  lvt_amd_profile_read returns 0 past the last slot. */
/* host-side counters of a handle: out[0] frames enqueued, out[1] frames collected, out[2] image / depth planes of host-buffer calls
   (lvt_track, lvt_amd_track_rgbd, ...) that were read IN PLACE because the caller's buffer is page-locked, out[3] planes copied through the
   library's staging buffer; out[4] frames that came through lvt_amd_track_async / _rgbd_async, out[5] of them: frames whose images were pulled by the launch of the frame before them,
   out[6] launches a lock-step batch's k_score is currently sent in (1 / 2: chosen from the batch's own gate stamps unless LVT_AMD_SCORE_PIECES fixes it),
   out[7] 1 when the handle orders its streams with events */
LVT_API void lvt_amd_get_host_stats(lvt_handle h, long long out[8]);

LVT_API void lvt_amd_profile_enable(lvt_handle h, int enable);
LVT_API int lvt_amd_profile_read(lvt_handle h, int slot, char *name, int name_cap, double *total_ms, long *calls);

/* ---- per-frame introspection (counter slots; the test oracle reports the same ones) ---- */
enum {
    LVT_AMD_C_N_LEFT = 0, LVT_AMD_C_N_RIGHT, LVT_AMD_C_MAP_SIZE, LVT_AMD_C_STAGED_SIZE, LVT_AMD_C_N_MATCHES,
    LVT_AMD_C_SECOND_PASS, LVT_AMD_C_N_ROW_MATCHES, LVT_AMD_C_N_TRIANGULATED, LVT_AMD_C_TRIANGULATED,
    LVT_AMD_C_RETRY_LEFT, LVT_AMD_C_RETRY_RIGHT, LVT_AMD_C_PNP_ITERS, LVT_AMD_C_PNP_INLIERS,
    LVT_AMD_C_MAP_SIZE_AT_MATCH, LVT_AMD_C_N_STAGED_ERASED, LVT_AMD_C_N_STAGED_PROMOTED, LVT_AMD_C_N_CULLED,
    LVT_AMD_C_FRAME, LVT_AMD_C_OVERFLOW /* bitmask of capacity overflows, 0 = none */,
    LVT_AMD_C_PNP_BORDERLINE /* chi2-gate decisions of this frame's pose refinement within 1e-8 of the 5.991 threshold */,
    LVT_AMD_C_ROW_FALLBACK /* 1: the tracking stream built this frame's row-match candidate lists itself (the early stream's were late) */,
    LVT_AMD_C_PNP_TRIALS /* LM trials of this frame's pose refinement */, LVT_AMD_C_PNP_REJECTIONS /* ... rejected ones */,
    LVT_AMD_C_PNP_TERMINATES /* passes ended by g2o's Terminate */,
    LVT_AMD_C__COUNT = 32
};
LVT_API void lvt_amd_get_counts(lvt_handle h, int out[LVT_AMD_C__COUNT]);
LVT_API int lvt_amd_get_features(lvt_handle h, int eye, float *xy, float *resp, uint8_t *desc, int cap);
LVT_API int lvt_amd_get_matches(lvt_handle h, int *feat_idx, double *xyz, int cap);
LVT_API int lvt_amd_get_row_matches(lvt_handle h, int *pairs, int cap);
LVT_API int lvt_amd_get_map(lvt_handle h, double *xyz, int *counter, int *age, uint8_t *desc, int cap);
LVT_API int lvt_amd_get_staged(lvt_handle h, double *xyz, int *counter, uint8_t *desc, int cap);
LVT_API void lvt_amd_get_pose(lvt_handle h, double q_wxyz[4], double p[3]);
/* pose (as the tracker holds it) and state (1 / 2 / 3, -1 on error) of the frame the last tracking call returned; unlike lvt_amd_get_pose and
   lvt_get_status it does not wait for that frame's tail (staged update, triangulation) when the call returned on the early pose */
LVT_API int lvt_amd_get_last_pose(lvt_handle h, double q_wxyz[4], double p[3]);
LVT_API void lvt_amd_get_predicted_pose(lvt_handle h, double q_wxyz[4], double p[3]);
/* bring-up profiling: cycle stamps written by detection cell 0 of the left image */
LVT_API void lvt_amd_get_debug(lvt_handle h, long long out[32]);
/* wall-clock (100 MHz) start / end stamps of the kernels on the inter-frame critical path, for the frame lvt_amd_wait returned
 * last (tools/timeline.py); does not drain the pipeline */
LVT_API void lvt_amd_get_timeline(lvt_handle h, long long out[16]);
/* how the handle's streams hand over: 0 = polling gates + early stream (the fast path of a single handle), 1 = event barriers
 * only.  LVT_AMD_ORDERING=polling / events fixes it at creation; unset, the first live handle of a process polls and
 * single-sequence handles created beside it use events (independent handles share the process's hardware queues, and a
 * polling gate holds up whatever another handle queued behind it) */
LVT_API int lvt_amd_get_ordering(lvt_handle h);
/* raw per-pixel intermediates of the last frame: what = 0 score map (u8, rows x pitch),
 * 1 box-sum map (u16, rows x pitch).  returns bytes written, pitch via *pitch_out (in elements). */
LVT_API int lvt_amd_get_plane(lvt_handle h, int eye, int what, void *dst, int cap_bytes, int *pitch_out);

/* ---- stage entry points on caller-provided data (differential tests vs the oracle) ---- */
/* motion-only BA alone (reference lvt_pnp_solver.cpp:60-128): pts n x 3 f64, obs n x 2 f32 (host) */
LVT_API int lvt_amd_pnp(const lvt_amd_params *p, const double q_in[4], const double p_in[3], const double *pts,
                        const float *obs, int n, double q_out[4], double p_out[3], int *n_solve_calls);
/* the same with the chi2 gates laid open (each may be NULL): err_out = the 2n edge errors the second gate saw, level_out[i] = 1 for an
 * edge a gate demoted (lvt_pnp_solver.cpp:109-116), *borderline = decisions taken within 1e-8 of the 5.991 threshold (those are
 * re-evaluated in the reference's operation order before they are taken; LVT_AMD_C_PNP_BORDERLINE reports the same per frame) */
LVT_API int lvt_amd_pnp_detail(const lvt_amd_params *p, const double q_in[4], const double p_in[3], const double *pts,
                               const float *obs, int n, double q_out[4], double p_out[3], int *n_solve_calls,
                               double *err_out, int *level_out, int *borderline);
/* ... and the Levenberg-Marquardt bookkeeping laid open (g2o's OptimizationAlgorithmLevenberg as configured at lvt_pnp_solver.cpp:44-53, 105-117): trace =
 * trace_cap rows x 4 doubles, one row per LM trial {lambda, chi2 at the estimate, chi2 of the trial, rho} (may be NULL); stats = {trials, rejected
 * trials (rho <= 0 or a non-finite chi2: lambda *= ni, the estimate is popped, the edge errors stay the trial's), passes ended by Terminate (10 trials
 * in one iteration, or rho == 0)}.  LVT_AMD_C_PNP_TRIALS / _REJECTIONS / _TERMINATES report the same per frame. */
LVT_API int lvt_amd_pnp_trace(const lvt_amd_params *p, const double q_in[4], const double p_in[3], const double *pts,
                              const float *obs, int n, double q_out[4], double p_out[3], int *n_solve_calls,
                              double *err_out, int *level_out, int *borderline, double *trace, int trace_cap, int stats[3]);
/* batched masked 2-NN Hamming (reference lvt_image_features_struct.cpp:68-120 + cv::BFMatcher knnMatch k=2):
 * B independent problems; per problem M queries (desc 32 B, xy f32) against N train (desc, xy, flag u8);
 * candidate iff !flag && dx*dx+dy*dy < r2 (f32, strict) [mode 0] or |band| row test [mode 1].
 * All pointers are DEVICE pointers; out: B x M x 4 int32 (idx1,d1,idx2,d2; -1/INT_MAX when absent).
 * Returns the kernel's elapsed time in microseconds (HIP events on `stream`), <0 on error. */
LVT_API float lvt_amd_hamming_match_batched(const void *q_desc, const void *q_xy, const void *t_desc,
                                            const void *t_xy, const void *t_flag, int B, int M, int N,
                                            float r2, int mode, int img_rows, int img_cols, void *out,
                                            void *hip_stream);
/* same, `launches` identical launches back to back between the two events; returns the AVERAGE per launch (a single
 * launch's event pair also times ~5 us of launch latency -- this is the form bench.py's roofline uses) */
LVT_API float lvt_amd_hamming_match_batched_n(const void *q_desc, const void *q_xy, const void *t_desc,
                                              const void *t_xy, const void *t_flag, int B, int M, int N,
                                              float r2, int mode, int img_rows, int img_cols, void *out,
                                              void *hip_stream, int launches);

/* ---- EuRoC pre-step: stereo rectification (reference examples/euroc/euroc_example.cpp:95-107,142-143 = cv::initUndistortRectifyMap
 * with CV_32FC1 maps + cv::remap INTER_LINEAR, BORDER_CONSTANT 0).  K, R, Pnew row-major 3x3, D = k1 k2 p1 p2 k3.
 * The maps are built once on the device; lvt_amd_rectify_device rectifies one 8-bit image already in HBM (dst_pitch % 4 == 0,
 * returns 0 on success), lvt_amd_rectify does the same for tightly packed host buffers. ---- */
typedef void *lvt_amd_rectifier;
LVT_API lvt_amd_rectifier lvt_amd_rectifier_create(const double K[9], const double D[5], const double R[9],
                                                   const double Pnew[9], int width, int height);
LVT_API void lvt_amd_rectifier_destroy(lvt_amd_rectifier r);
LVT_API int lvt_amd_rectify_device(lvt_amd_rectifier r, const void *d_src, int src_pitch, void *d_dst, int dst_pitch,
                                   void *hip_stream);
LVT_API int lvt_amd_rectify(lvt_amd_rectifier r, const unsigned char *src, unsigned char *dst);
LVT_API int lvt_amd_rectifier_get_maps(lvt_amd_rectifier r, float *map1, float *map2);

/* ---- odometry accumulator: the consumer right behind the path -----------------------------------------------
 * What the reference's ROS node does with every pose (lvt/src/lvt_ros.cpp:86-92 constructor, :215-311 on_stereo_image), without
 * ROS: poses are re-expressed in an x-forward / z-up frame (rot_fix = Rz(-pi/2) Rx(-pi/2)), the frame-to-frame delta is moved
 * into the base frame (base_to_sensor, identity by default) and accumulated into base_to_odom; a frame with an older time stamp
 * is ignored; when tracking is LOST the odometry resets the tracker (lvt_system::reset) and -- if reset_pose_on_lost -- its own
 * accumulated pose, and publishes nothing for that frame.
 *   pose_out  : position x y z + orientation qx qy qz qw of the base in the odom frame
 *   twist_out : linear xyz, angular xyz (delta / time since the previous published frame; zeros for the first frame or dt = 0)
 * lvt_amd_odometry_push_pose is the pure host-side step (a pose that was tracked elsewhere; `h` may be NULL at creation);
 * lvt_amd_odometry_update = lvt_track + lvt_get_status + that step (+ the reset on LOST).
 * Return value of both: 1 = published, 0 = nothing published (LOST and reset, or stale time stamp), -1 = bad arguments. */
typedef void *lvt_amd_odometry;
LVT_API lvt_amd_odometry lvt_amd_odometry_create(lvt_handle h, const double base_to_sensor[12] /* 3x4 row-major or NULL */, int reset_pose_on_lost);
LVT_API void lvt_amd_odometry_destroy(lvt_amd_odometry o);
LVT_API void lvt_amd_odometry_reset(lvt_amd_odometry o); /* the node's reset_vo service (lvt_ros.cpp:184-198): tracker reset, deltas restart, accumulated pose back to identity */
LVT_API int lvt_amd_odometry_push_pose(lvt_amd_odometry o, const double R[3][3], const double t[3], int status, double stamp_sec,
                                       double pose_out[7], double twist_out[6]);
LVT_API int lvt_amd_odometry_update(lvt_amd_odometry o, unsigned char *left, unsigned char *right, int n_rows, int n_cols, double stamp_sec,
                                    double pose_out[7], double twist_out[6]);

#ifdef __cplusplus
}
#endif
#endif
