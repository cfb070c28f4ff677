/*
 * lvt_oracle.h -- C-ABI of the CPU ORACLE for the LVT tracking hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a from-scratch scalar restatement of the reference's
 * per-frame tracking path, used (a) as the parity checker in tests/ and __graft_entry__.smoke()
 * and (b) as the `cpu_baseline` leg of bench.py.  The product (lvt_amd/, liblvt_c.so) never
 * includes, links, imports or executes anything in this directory.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or expected trajectories
 *  and cannot be compiled here (OpenCV, opencv_contrib, Eigen, g2o are
 * absent), so this oracle is pinned only by (i) hand-derived known-answer tests per primitive,
 * (ii) ground-truth motion of the synthetic sequences, (iii) fixtures it generated itself.
 * Third-party semantics it restates (not under /root/reference): OpenCV >=3.1 AGAST (OAST_9_16 +
 * AGAST NMS), opencv_contrib xfeatures2d BRIEF-32, cv::BFMatcher(NORM_HAMMING).knnMatch(k=2, mask),
 * Eigen3 JacobiSVD least squares / Quaternion ops, g2o tag 20170730_git (SparseOptimizer +
 * OptimizationAlgorithmLevenberg + BlockSolver_6_3 + LinearSolverPCG + EdgeProjectP2MC/VertexCam/SBACam
 * + RobustKernelCauchy) -- see SURVEY.md Appendix A.
 */
#ifndef LVT_ORACLE_H__
#define LVT_ORACLE_H__

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors struct lvt_parameters, reference lvt/src/lvt_parameters.h:29-64 (hot-path fields only) */
typedef struct lvto_params {
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
} lvto_params;

typedef void *lvto_handle;

/* ---- whole-path entry points (mirror lvt_c.h / lvt_system.h of the reference) ---- */
void lvto_default_params(lvto_params *p);                       /* lvt_parameters.cpp:29-52 */
lvto_handle lvto_create(const lvto_params *p, int sensor_type); /* lvt_system.cpp:70-127 */
void lvto_destroy(lvto_handle h);
void lvto_reset(lvto_handle h);                                 /* lvt_system.cpp:44-68 */
void lvto_set_threads(lvto_handle h, int n);                    /* 2 = reference behaviour (handler.cpp:204-206) */
void lvto_track(lvto_handle h, const uint8_t *left, const uint8_t *right, int rows, int cols,
                double R[9], double t[3]);                      /* lvt_system.cpp:157-207 */
void lvto_track_rgbd(lvto_handle h, const uint8_t *gray, const float *depth, int rows, int cols,
                     double R[9], double t[3]);
void lvto_track_with_external_corners(lvto_handle h, const uint8_t *left, const uint8_t *right,
                                      int rows, int cols, const double *cl, int ncl,
                                      const double *cr, int ncr, double R[9], double t[3]);
int lvto_get_status(lvto_handle h);

/* ---- introspection of the last frame (for stage-by-stage parity) ---- */
enum {
    LVTO_C_N_LEFT = 0,       /* features in left struct */
    LVTO_C_N_RIGHT,          /* features in right struct */
    LVTO_C_MAP_SIZE,         /* map size at end of frame */
    LVTO_C_STAGED_SIZE,      /* staged size at end of frame */
    LVTO_C_N_MATCHES,        /* find_matches output count */
    LVTO_C_SECOND_PASS,      /* 1 if the doubled-radius pass ran */
    LVTO_C_N_ROW_MATCHES,    /* row_match pairs this frame (0 if no triangulation) */
    LVTO_C_N_TRIANGULATED,   /* new points this frame */
    LVTO_C_TRIANGULATED,     /* 1 if update_with_new_triangulation ran */
    LVTO_C_RETRY_LEFT,       /* 1 if low-corner retry ran for left image */
    LVTO_C_RETRY_RIGHT,
    LVTO_C_PNP_ITERS,        /* total LM solve() calls this frame */
    LVTO_C_PNP_INLIERS,      /* edges surviving both chi2 gates */
    LVTO_C_MAP_SIZE_AT_MATCH,/* map size when find_matches ran */
    LVTO_C_N_STAGED_ERASED,
    LVTO_C_N_STAGED_PROMOTED,
    LVTO_C_N_CULLED,
    LVTO_C_FRAME,
    LVTO_C_OVERFLOW_UNUSED,  /* (slot 18 is the HIP path's capacity-overflow mask; always 0 here) */
    LVTO_C_PNP_BORDERLINE,   /* chi2-gate decisions (both passes) within 1e-8 of the 5.991 threshold */
    LVTO_C_ROW_FALLBACK_UNUSED, /* (slot 20 is the HIP path's "row lists built on the tracking stream" flag; always 0 here) */
    LVTO_C_PNP_TRIALS,       /* LM trials of both passes (solve + update + chi2 each) */
    LVTO_C_PNP_REJECTIONS,   /* ... of which rejected (rho <= 0 or non-finite chi2: lambda *= ni, pop()) */
    LVTO_C_PNP_TERMINATES,   /* passes ended by Terminate (qmax == 10 or rho == 0) */
    LVTO_C__COUNT = 32
};
void lvto_get_counts(lvto_handle h, int out[LVTO_C__COUNT]);
/* eye 0 = left, 1 = right.  xy: N x 2 f32, resp: N f32, desc: N x 32 u8.  returns N. */
int lvto_get_features(lvto_handle h, int eye, float *xy, float *resp, uint8_t *desc, int cap);
/* find_matches output, in map order: feat_idx[i], map position xyz[3i..]; returns count */
int lvto_get_matches(lvto_handle h, int *feat_idx, double *xyz, int cap);
/* row_match pairs (left idx, right idx) of this frame's triangulation; returns count */
int lvto_get_row_matches(lvto_handle h, int *pairs, int cap);
/* map at end of frame */
int lvto_get_map(lvto_handle h, double *xyz, int *counter, int *age, uint8_t *desc, int cap);
int lvto_get_staged(lvto_handle h, double *xyz, int *counter, uint8_t *desc, int cap);
void lvto_get_pose(lvto_handle h, double q_wxyz[4], double p[3]);
void lvto_get_predicted_pose(lvto_handle h, double q_wxyz[4], double p[3]);

/* ---- primitive / stage functions (known-answer + differential tests) ---- */
/* OAST-9/16 score of every pixel of a ROI treated as an isolated image:
 * out[y*cols+x] = max{b<=254 : corner(b)} or -1 when not even corner(0); border pixels -1. */
void lvto_agast_score_map(const uint8_t *img, int rows, int cols, int stride, int16_t *out);
/* cv::AgastFeatureDetector(threshold).detect on one ROI; returns n, fills xyr (n x 3 f32) */
int lvto_agast_detect(const uint8_t *img, int rows, int cols, int stride, int threshold,
                      int nonmax, float *xyr, int cap);
/* LVT's ANMS (handler.cpp:34-83) in place on xyr (n x 3): returns new n; adds (tx,ty) */
int lvto_anms(float *xyr, int n, int num_to_keep, float tx, float ty);
/* grid detection + retry (handler.cpp:131-169) on a whole image; returns n */
int lvto_detect_grid(const uint8_t *img, int rows, int cols, const lvto_params *p, float *xyr,
                     int cap, int *retry_used);
/* BRIEF-32 incl. border filter (A.3): xy in (n x 2), kept indices out, desc (n_kept x 32) */
int lvto_brief(const uint8_t *img, int rows, int cols, const float *xy, int n, int *kept,
               uint8_t *desc);
/* full per-image feature extraction (handler.cpp:156-176): returns N, fills xy/resp/desc */
int lvto_compute_features(const uint8_t *img, int rows, int cols, const lvto_params *p, float *xy,
                          float *resp, uint8_t *desc, int cap, int *retry_used);
/* masked 2-NN Hamming, BFMatcher semantics (A.4): out[4] = idx1,d1,idx2,d2 (-1/INT_MAX if absent) */
void lvto_hamming_top2(const uint8_t *query, const uint8_t *train, int n, const uint8_t *mask,
                       int out[4]);
/* motion-only BA (pnp_solver.cpp:60-128 + A.6).  pose in/out: q_wxyz, p.  pts: n x 3 f64, obs: n x 2 f32
 * trace (optional, cap rows x 4): per LM trial {lambda, chi_cur, chi_tmp, rho}; solve_calls (optional): g2o solve() calls of both passes */
int lvto_pnp(const lvto_params *p, const double q_in[4], const double p_in[3], const double *pts,
             const float *obs, int n, double q_out[4], double p_out[3], int *inlier_marks,
             double *trace, int trace_cap, int *solve_calls);
/* the gates of the last lvto_pnp / frame on this thread: copies the 2n edge errors they saw, *min_margin = the closest
   |e^2 - 5.991| of any decision; returns the number of decisions within 1e-8 of the threshold */
int lvto_pnp_last_gate(double *err_out, int n, double *min_margin);
/* ... and its LM bookkeeping: out = {trials, rejected trials, passes ended by Terminate} */
void lvto_pnp_last_stats(int out[3]);
/* linear-LS stereo triangulation of one pair incl. gates (local_map.cpp:276-319); returns 1 if kept */
int lvto_triangulate_one(const lvto_params *p, const double q[4], const double pos[3], float ulx,
                         float uly, float urx, float ury, double out_xyz[3]);
/* motion model (motion_model.cpp:42-65): state = {last_q[4], ang_vel[4], last_p[3], lin_vel[3]} */
void lvto_motion_predict(double state[14], const double q[4], const double p[3], double q_out[4],
                         double p_out[3]);
/* std::sort by response desc exactly as handler.cpp:38-41 (libstdc++ introsort order) */
void lvto_sort_by_response(float *xyr, int n);
/* EuRoC pre-step (euroc_example.cpp:95-107,142-143 of the reference; upstream semantics restated, unverifiable offline):
 * cv::initUndistortRectifyMap(K, D(k1 k2 p1 p2 k3), R, Pnew(3x3), size, CV_32FC1) -> map1 (x), map2 (y), w*h floats each */
void lvto_init_undistort_rectify_map(const double K[9], const double D[5], const double R[9], const double P[9], int w,
                                     int h, float *map1, float *map2);
/* cv::remap(src 8UC1, dst, map1, map2, INTER_LINEAR, BORDER_CONSTANT 0): 5-bit fixed-point coordinates, 15-bit weights */
void lvto_remap_bilinear(const uint8_t *src, int src_w, int src_h, int src_step, const float *map1, const float *map2,
                         int dst_w, int dst_h, uint8_t *dst);

#ifdef __cplusplus
}
#endif
#endif
