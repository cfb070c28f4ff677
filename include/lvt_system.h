/*
 * lvt_system.h -- header-compatible C++ layer over the MI355X-native tracking path (SURVEY 8b "C++ API", 8f row 4).
 *
 * The reference's consumers -- examples/kitti, examples/euroc, examples/tum_rgbd and the ROS node -- talk to
 * `lvt_system`, `lvt_parameters` and `lvt_pose` (reference lvt/src/lvt_system.h:41-109, lvt_parameters.h:29-64,
 * lvt_pose.h:51-75), not to the C-ABI.  This header offers the same names, members and call shapes on top of
 * liblvt_c.so, so that such a caller compiles against it unchanged:
 *
 *     lvt_parameters params;  params.init_from_file("vo_config.yaml");  params.fx = ...;
 *     lvt_system *vo = lvt_system::create(params, lvt_system::eSensor_STEREO);
 *     lvt_pose pose = vo->track(left, right);
 *     pose.get_position();  pose.get_orientation_matrix();  pose.get_orientation_quaternion();
 *     if (vo->get_state() == lvt_system::eState_LOST) ...;   lvt_system::destroy(vo);
 *
 * Images: `lvt_image_view` (pointer, rows, cols, step in bytes -- any 8-bit gray or 32-bit float plane) is always
 * there; when OpenCV's core header is on the include path the `cv::Mat` / `cv::Point2f` overloads of the reference
 * exist as well.  Algebra types: Eigen's when <Eigen/Dense> is on the include path (the reference's typedefs, same
 * accessors), a small stand-in with the accessors the reference's callers use (x() y() z() w(), operator(), toRotationMatrix())
 * otherwise.  Neither library is needed to build or to run.
 *
 * Header-only; link with -llvt_c.  Like the reference, a handle is not thread-safe: one call at a time.
 */
#ifndef LVT_VISUAL_ODOMETRY_SYSTEM_H__
#define LVT_VISUAL_ODOMETRY_SYSTEM_H__

#include "lvt_amd_ext.h"
#include "lvt_c.h"

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>) && !defined(LVT_SYSTEM_NO_OPENCV)
#include <opencv2/core/core.hpp>
#define LVT_SYSTEM_HAVE_OPENCV 1
#endif
#if __has_include(<Eigen/Dense>) && !defined(LVT_SYSTEM_NO_EIGEN)
#include <Eigen/Dense>
#include <Eigen/StdVector>
#define LVT_SYSTEM_HAVE_EIGEN 1
#endif
#endif

/* ---- lvt_parameters: reference lvt_parameters.h:29-64 (same fields, same defaults -- lvt_parameters.cpp:29-52) ---- */
struct lvt_parameters {
    lvt_parameters() {
        lvt_amd_params p;
        lvt_amd_default_params(&p);
        from_pod(p);
        enable_logging = true;  /* accepted and ignored: the log / the viewer are outside the hot path */
        enable_visualization = false;
        viewer_camera_size = 0.6f;
        viewer_point_size = 5;
    }
    /* every key of the reference's flat %YAML:1.0 files; a missing key reads as 0, exactly as there (lvt_parameters.cpp:54-93) */
    bool init_from_file(const char *config_file_name) {
        lvt_amd_params p;
        if (!lvt_amd_params_from_file(config_file_name, &p)) return false;
        from_pod(p);
        return true;
    }

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
    bool enable_logging;
    bool enable_visualization;
    enum { etriangulation_policy_decreasing_matches = 1, etriangulation_policy_always_triangulate, etriangulation_policy_map_size };
    int triangulation_policy;
    float viewer_camera_size;
    int viewer_point_size;

    lvt_amd_params to_pod() const {
        lvt_amd_params p;
        std::memset(&p, 0, sizeof(p));
        p.fx = fx, p.fy = fy, p.cx = cx, p.cy = cy, p.baseline = baseline;
        p.img_width = img_width, p.img_height = img_height;
        p.k1 = k1, p.k2 = k2, p.p1 = p1, p.p2 = p2, p.k3 = k3;
        p.near_plane_distance = near_plane_distance, p.far_plane_distance = far_plane_distance;
        p.triangulation_ratio_test_threshold = triangulation_ratio_test_threshold;
        p.tracking_ratio_test_threshold = tracking_ratio_test_threshold;
        p.descriptor_matching_threshold = descriptor_matching_threshold;
        p.min_num_matches_for_tracking = min_num_matches_for_tracking;
        p.tracking_radius = tracking_radius;
        p.detection_cell_size = detection_cell_size;
        p.max_keypoints_per_cell = max_keypoints_per_cell;
        p.agast_threshold = agast_threshold;
        p.untracked_threshold = untracked_threshold;
        p.staged_threshold = staged_threshold;
        p.triangulation_policy = triangulation_policy;
        return p;
    }

  private:
    void from_pod(const lvt_amd_params &p) {
        fx = p.fx, fy = p.fy, cx = p.cx, cy = p.cy, baseline = p.baseline;
        img_width = p.img_width, img_height = p.img_height;
        k1 = p.k1, k2 = p.k2, p1 = p.p1, p2 = p.p2, k3 = p.k3;
        near_plane_distance = p.near_plane_distance, far_plane_distance = p.far_plane_distance;
        triangulation_ratio_test_threshold = p.triangulation_ratio_test_threshold;
        tracking_ratio_test_threshold = p.tracking_ratio_test_threshold;
        descriptor_matching_threshold = p.descriptor_matching_threshold;
        min_num_matches_for_tracking = p.min_num_matches_for_tracking;
        tracking_radius = p.tracking_radius;
        detection_cell_size = p.detection_cell_size;
        max_keypoints_per_cell = p.max_keypoints_per_cell;
        agast_threshold = p.agast_threshold;
        untracked_threshold = p.untracked_threshold;
        staged_threshold = p.staged_threshold;
        triangulation_policy = p.triangulation_policy;
    }
};

/* ---- algebra types: reference lvt_pose.h:34-49 ---- */
#if defined(LVT_SYSTEM_HAVE_EIGEN)
typedef Eigen::Matrix<double, 3, 1> lvt_vector3;
typedef Eigen::Matrix<double, 3, 3> lvt_matrix33;
typedef Eigen::Matrix<double, 4, 4> lvt_matrix44;
typedef Eigen::Quaternion<double> lvt_quaternion;
#define LVT_SYSTEM_ALIGNED_NEW EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#else
struct lvt_vector3 {
    double v[3];
    lvt_vector3() : v{0, 0, 0} {}
    lvt_vector3(double a, double b, double c) : v{a, b, c} {}
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
    double &operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    void setZero() { v[0] = v[1] = v[2] = 0; }
};
struct lvt_matrix33 {
    double m[9]; /* row-major */
    lvt_matrix33() : m{1, 0, 0, 0, 1, 0, 0, 0, 1} {}
    double &operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }
};
struct lvt_matrix44 {
    double m[16];
    lvt_matrix44() : m{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1} {}
    double &operator()(int r, int c) { return m[4 * r + c]; }
    double operator()(int r, int c) const { return m[4 * r + c]; }
};
struct lvt_quaternion {
    double qw, qx, qy, qz;
    lvt_quaternion() : qw(1), qx(0), qy(0), qz(0) {}
    lvt_quaternion(double w_, double x_, double y_, double z_) : qw(w_), qx(x_), qy(y_), qz(z_) {} /* Eigen's (w, x, y, z) order */
    double w() const { return qw; }
    double x() const { return qx; }
    double y() const { return qy; }
    double z() const { return qz; }
    void setIdentity() { qw = 1, qx = qy = qz = 0; }
    lvt_matrix33 toRotationMatrix() const { /* Eigen::QuaternionBase::toRotationMatrix, coefficient for coefficient */
        lvt_matrix33 R;
        const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
        const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
        R(0, 0) = 1 - (tyy + tzz), R(0, 1) = txy - twz, R(0, 2) = txz + twy;
        R(1, 0) = txy + twz, R(1, 1) = 1 - (txx + tzz), R(1, 2) = tyz - twx;
        R(2, 0) = txz - twy, R(2, 1) = tyz + twx, R(2, 2) = 1 - (txx + tyy);
        return R;
    }
};
#define LVT_SYSTEM_ALIGNED_NEW
#endif

/* ---- lvt_pose: reference lvt_pose.h:51-75 (camera-to-world: orientation + position of the left camera) ---- */
class lvt_pose {
  public:
    LVT_SYSTEM_ALIGNED_NEW
    lvt_pose() {
        m_position.setZero();
        m_orientation.setIdentity();
    }
    lvt_pose(const lvt_vector3 &position, const lvt_quaternion &orientation) { set_parameters(position, orientation); }
    void set_parameters(const lvt_vector3 &position, const lvt_quaternion &orientation) {
        m_orientation = orientation;
        m_position = position;
    }
    lvt_vector3 get_position() const { return m_position; }
    lvt_matrix33 get_orientation_matrix() const { return m_orientation.toRotationMatrix(); }
    lvt_quaternion get_orientation_quaternion() const { return m_orientation; }

  private:
    lvt_quaternion m_orientation;
    lvt_vector3 m_position;
};
#if defined(LVT_SYSTEM_HAVE_EIGEN)
typedef std::vector<lvt_pose, Eigen::aligned_allocator<lvt_pose>> lvt_pose_array;
#else
typedef std::vector<lvt_pose> lvt_pose_array;
#endif

/* a borrowed image plane: 8-bit gray (elem_size 1) or 32-bit float depth in metres (elem_size 4); rows `step` bytes apart */
struct lvt_image_view {
    const void *data;
    int rows, cols;
    size_t step;
    int elem_size;
    lvt_image_view() : data(nullptr), rows(0), cols(0), step(0), elem_size(1) {}
    lvt_image_view(const unsigned char *p, int rows_, int cols_, size_t step_ = 0) : data(p), rows(rows_), cols(cols_), step(step_ ? step_ : (size_t)cols_), elem_size(1) {}
    lvt_image_view(const float *p, int rows_, int cols_, size_t step_ = 0) : data(p), rows(rows_), cols(cols_), step(step_ ? step_ : sizeof(float) * (size_t)cols_), elem_size(4) {}
#if defined(LVT_SYSTEM_HAVE_OPENCV)
    lvt_image_view(const cv::Mat &m) : data(m.data), rows(m.rows), cols(m.cols), step(m.step[0]), elem_size((int)m.elemSize()) {} /* CV_8UC1 / CV_32FC1 */
#endif
};

/* ---- lvt_system: reference lvt_system.h:41-109 ---- */
class lvt_system {
  public:
    LVT_SYSTEM_ALIGNED_NEW
    enum eState { eState_NOT_INITIALIZED = 1, eState_TRACKING, eState_LOST };
    enum eSensor { eSensor_STEREO = 1, eSensor_RGBD };

    /* lvt_system.cpp:70-127.  NULL on bad parameters or when no usable HIP device exists: there is no CPU fallback. */
    static lvt_system *create(const lvt_parameters &params, eSensor sensor_type) {
        const lvt_amd_params pod = params.to_pod();
        lvt_handle h = lvt_amd_create(&pod, (int)sensor_type);
        if (!h) return nullptr;
        lvt_system *s = new lvt_system();
        s->m_handle = h;
        s->m_sensor = sensor_type;
        s->m_params = params;
        return s;
    }
    static void destroy(lvt_system *s) {
        if (!s) return;
        lvt_destroy(s->m_handle);
        delete s;
    }
    void reset() { lvt_amd_reset(m_handle); m_state = -1; } /* lvt_system.cpp:44-68 */

    /* stereo: two rectified 8-bit gray images; RGB-D: gray + 32-bit float depth in metres (lvt_system.cpp:157-207).
     * Returns the camera-to-world pose of the left camera in the first frame's left-camera frame; after LOST, the last pose. */
    lvt_pose track(const lvt_image_view &img1, const lvt_image_view &img2) {
        double R[3][3], t[3];
        if (img2.rows != img1.rows || img2.cols != img1.cols || img1.elem_size != 1 || img2.elem_size != (m_sensor == eSensor_STEREO ? 1 : 4)) {
            /* the C-ABI is told ONE size for both planes: a second plane of another size (or element type) would be read out of bounds.
               The reference would throw inside OpenCV here and lvt_c.cpp:85-87 swallows that: same outcome, the last pose */
            return current_pose();
        }
        const unsigned char *a = static_cast<const unsigned char *>(packed(img1, m_buf1));
        if (m_sensor == eSensor_STEREO) lvt_track(m_handle, const_cast<unsigned char *>(a), const_cast<unsigned char *>(static_cast<const unsigned char *>(packed(img2, m_buf2))), img1.rows, img1.cols, R, t);
        else lvt_amd_track_rgbd(m_handle, a, static_cast<const float *>(packed(img2, m_buf2)), img1.rows, img1.cols, R, t);
        return current_pose();
    }
    /* ADDITIVE (not in the reference): the asynchronous form of track().  track_async() enqueues the frame and returns -- pageable images are copied
     * during the call, page-locked ones must stay valid until wait_pose() has returned that frame -- so the caller can decode frame t + 1 while frame
     * t tracks (kitti_example.cpp:113-138 does both in turn); wait_pose() returns the poses in the order of the track_async() calls.  false: the
     * frame was rejected (size / element type), nothing was enqueued. */
    bool track_async(const lvt_image_view &img1, const lvt_image_view &img2) {
        if (img2.rows != img1.rows || img2.cols != img1.cols || img1.elem_size != 1 || img2.elem_size != (m_sensor == eSensor_STEREO ? 1 : 4)) return false;
        const unsigned char *a = static_cast<const unsigned char *>(packed(img1, m_buf1));
        if (m_sensor == eSensor_STEREO)
            return lvt_amd_track_async(m_handle, a, static_cast<const unsigned char *>(packed(img2, m_buf2)), img1.rows, img1.cols) == 0;
        return lvt_amd_track_rgbd_async(m_handle, a, static_cast<const float *>(packed(img2, m_buf2)), img1.rows, img1.cols) == 0;
    }
    lvt_pose wait_pose() {
        double q[4] = {1, 0, 0, 0}, p[3] = {0, 0, 0};
        m_state = lvt_amd_wait_pose(m_handle, q, p);
        return lvt_pose(lvt_vector3(p[0], p[1], p[2]), lvt_quaternion(q[0], q[1], q[2], q[3]));
    }
    /* lvt_system.cpp:209-250: detection skipped, BRIEF at the given corners; Point = anything with float members x, y */
    template <class Point>
    lvt_pose track_with_external_corners(const lvt_image_view &left_image, const lvt_image_view &right_image, std::vector<Point> &corners_locations_left,
                                         std::vector<Point> &corners_locations_right) {
        std::vector<double> cl(2 * corners_locations_left.size() + 2), cr(2 * corners_locations_right.size() + 2);
        for (size_t i = 0; i < corners_locations_left.size(); i++) cl[2 * i] = corners_locations_left[i].x, cl[2 * i + 1] = corners_locations_left[i].y;
        for (size_t i = 0; i < corners_locations_right.size(); i++) cr[2 * i] = corners_locations_right[i].x, cr[2 * i + 1] = corners_locations_right[i].y;
        double R[3][3], t[3];
        lvt_track_with_external_corners(m_handle, const_cast<unsigned char *>(static_cast<const unsigned char *>(packed(left_image, m_buf1))),
                                        const_cast<unsigned char *>(static_cast<const unsigned char *>(packed(right_image, m_buf2))), left_image.rows, left_image.cols,
                                        reinterpret_cast<double(*)[2]>(cl.data()), (int)corners_locations_left.size(), reinterpret_cast<double(*)[2]>(cr.data()),
                                        (int)corners_locations_right.size(), R, t);
        return current_pose();
    }
#if defined(LVT_SYSTEM_HAVE_OPENCV)
    lvt_pose track(cv::Mat img1, cv::Mat img2) { return track(lvt_image_view(img1), lvt_image_view(img2)); }
    lvt_pose track_with_external_corners(cv::Mat left_image, cv::Mat right_image, std::vector<cv::Point2f> &corners_locations_left,
                                         std::vector<cv::Point2f> &corners_locations_right) {
        return track_with_external_corners<cv::Point2f>(lvt_image_view(left_image), lvt_image_view(right_image), corners_locations_left, corners_locations_right);
    }
#endif

    inline lvt_system::eSensor get_sensor_type() const { return m_sensor; }
    /* the state of the frame track() returned last -- delivered with its pose, no further wait (lvt_get_status collects the whole frame) */
    inline lvt_system::eState get_state() const { return (eState)(m_state > 0 ? m_state : lvt_get_status(m_handle)); }
    inline bool should_quit() const { return false; } /* (set by the reference's Pangolin viewer only; there is none here) */
    /* additive: the text of the last problem the GPU path reported ("" when none) -- the reference has no error channel */
    inline const char *last_error() const { return lvt_amd_last_error(m_handle); }
    inline lvt_handle native_handle() const { return m_handle; }

    lvt_system(const lvt_system &) = delete;
    lvt_system &operator=(const lvt_system &) = delete;

  private:
    lvt_system() : m_handle(nullptr), m_sensor(eSensor_STEREO) {}
    ~lvt_system() {}

    /* the C-ABI takes tightly packed planes (lvt_c.h): a view with padding between its rows is packed into a scratch buffer */
    static const void *packed(const lvt_image_view &v, std::vector<unsigned char> &buf) {
        const size_t row = (size_t)v.cols * (size_t)v.elem_size;
        if (v.step == row || v.rows <= 1) return v.data;
        buf.resize(row * (size_t)v.rows);
        for (int y = 0; y < v.rows; y++) std::memcpy(buf.data() + row * (size_t)y, static_cast<const unsigned char *>(v.data) + v.step * (size_t)y, row);
        return buf.data();
    }
    /* the pose as the tracker holds it (the quaternion g2o's SBACam hands back, lvt_pnp_solver.cpp:120-122), not one re-derived from R */
    lvt_pose current_pose() {
        double q[4] = {1, 0, 0, 0}, p[3] = {0, 0, 0};
        m_state = lvt_amd_get_last_pose(m_handle, q, p); /* (does not wait for the frame's tail: the next track() overlaps it) */
        return lvt_pose(lvt_vector3(p[0], p[1], p[2]), lvt_quaternion(q[0], q[1], q[2], q[3]));
    }

    lvt_handle m_handle;
    int m_state = -1;
    eSensor m_sensor;
    lvt_parameters m_params;
    std::vector<unsigned char> m_buf1, m_buf2;
};

#endif /* LVT_VISUAL_ODOMETRY_SYSTEM_H__ */
