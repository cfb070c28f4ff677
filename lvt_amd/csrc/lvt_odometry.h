// lvt_odometry.h -- odometry accumulator behind the tracker (host-only; include/lvt_amd_ext.h, "odometry accumulator").
// Follows what the reference's ROS node does with each pose (lvt/src/lvt_ros.cpp:86-92, :184-198, :215-311) with plain 3x3 /
// 3-vector arithmetic in place of Eigen and tf2: rot_fix re-expresses the camera pose in an x-forward / z-up frame, the delta
// between consecutive fixed poses is conjugated into the base frame and appended to base_to_odom.
#pragma once
#include <cmath>
#include <cstring>

namespace lvt {

struct Rigid {  // x -> R x + p
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double p[3] = {0, 0, 0};
};
static inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    std::memcpy(C, T, sizeof(T));
}
static inline void mat3_vec(const double A[9], const double v[3], double o[3]) {
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
    o[0] = t[0], o[1] = t[1], o[2] = t[2];
}
static inline Rigid rigid_mul(const Rigid &a, const Rigid &b) {  // a after b
    Rigid c;
    mat3_mul(a.R, b.R, c.R);
    mat3_vec(a.R, b.p, c.p);
    for (int i = 0; i < 3; i++) c.p[i] += a.p[i];
    return c;
}
static inline Rigid rigid_inv(const Rigid &a) {
    Rigid c;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c.R[3 * i + j] = a.R[3 * j + i];
    double t[3];
    mat3_vec(c.R, a.p, t);
    for (int i = 0; i < 3; i++) c.p[i] = -t[i];
    return c;
}
// rotation matrix -> unit quaternion (x, y, z, w), the branch on the trace / largest diagonal element used by Eigen and tf2
static inline void mat3_to_quat(const double R[9], double q[4]) {
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        const double s = std::sqrt(tr + 1.0), r = 0.5 / s;
        q[3] = 0.5 * s;
        q[0] = (R[7] - R[5]) * r, q[1] = (R[2] - R[6]) * r, q[2] = (R[3] - R[1]) * r;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        const double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0), r = 0.5 / s;
        q[i] = 0.5 * s;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * r;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * r;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * r;
    }
}

struct Odometry {
    void *tracker = nullptr;  // lvt_handle or NULL (push_pose only)
    bool reset_pose_on_lost = true;
    double rot_fix[9];
    Rigid base_to_odom, base_to_sensor;
    double last_R[9], last_p[3] = {0, 0, 0};
    bool have_time = false;
    double last_time = 0;

    Odometry() {
        // Rz(a) Rx(a), a = -1.57079632679 (the reference's literal, not exactly -pi/2): camera z-forward / x-right -> x-forward / z-up
        const double a = -1.57079632679, c = std::cos(a), s = std::sin(a);
        const double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1}, Rx[9] = {1, 0, 0, 0, c, -s, 0, s, c};
        mat3_mul(Rz, Rx, rot_fix);
        std::memcpy(last_R, rot_fix, sizeof(rot_fix));
    }
    void restart_deltas() {
        std::memcpy(last_R, rot_fix, sizeof(rot_fix));
        last_p[0] = last_p[1] = last_p[2] = 0;
    }
    // returns true when a pose was published
    bool push(const double R[9], const double t[3], double stamp, double pose_out[7], double twist_out[6]) {
        double cur_R[9], cur_p[3];
        mat3_mul(rot_fix, R, cur_R);
        mat3_vec(rot_fix, t, cur_p);
        Rigid d;  // delta in the (fixed) sensor frame: rotation cur * last^T, translation cur - last
        double lastT[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) lastT[3 * i + j] = last_R[3 * j + i];
        mat3_mul(cur_R, lastT, d.R);
        for (int i = 0; i < 3; i++) d.p[i] = cur_p[i] - last_p[i];
        const Rigid db = rigid_mul(rigid_mul(base_to_sensor, d), rigid_inv(base_to_sensor));
        base_to_odom = rigid_mul(base_to_odom, db);
        if (pose_out) {
            for (int i = 0; i < 3; i++) pose_out[i] = base_to_odom.p[i];
            mat3_to_quat(base_to_odom.R, pose_out + 3);
        }
        if (twist_out) {
            for (int i = 0; i < 6; i++) twist_out[i] = 0;
            const double dt = stamp - last_time;
            if (have_time && dt != 0) {
                for (int i = 0; i < 3; i++) twist_out[i] = db.p[i] / dt;
                double q[4];
                mat3_to_quat(db.R, q);
                // angle / axis of the delta rotation as tf2 reports them: angle = 2 acos(w), axis = xyz / sqrt(1 - w^2), (1, 0, 0) for a null rotation
                const double w = std::fmin(1.0, std::fmax(-1.0, q[3]));
                const double angle = 2.0 * std::acos(w), s2 = 1.0 - w * w;
                double ax[3] = {1, 0, 0};
                if (s2 >= 10.0 * 2.220446049250313e-16) {
                    const double is = 1.0 / std::sqrt(s2);
                    ax[0] = q[0] * is, ax[1] = q[1] * is, ax[2] = q[2] * is;
                }
                for (int i = 0; i < 3; i++) twist_out[3 + i] = ax[i] * angle / dt;
            }
        }
        have_time = true;
        last_time = stamp;
        std::memcpy(last_R, cur_R, sizeof(cur_R));
        last_p[0] = cur_p[0], last_p[1] = cur_p[1], last_p[2] = cur_p[2];
        return true;
    }
};

}  // namespace lvt
