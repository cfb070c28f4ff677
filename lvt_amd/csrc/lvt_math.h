// lvt_math.h -- fp64 pose math used on device (and by the host for nothing else): quaternion ops with
// Eigen's coefficient formulas, world<->camera transform (lvt_pose.cpp:28-43), the visibility gate
// (lvt_local_map.cpp:62-82), the constant-velocity motion model (lvt_motion_model.cpp:42-65) and
// cv::undistortPoints' fixed-point iteration.  Compiled with -ffp-contract=off.
#pragma once
#include "lvt_dev.h"

namespace lvt {

__host__ __device__ inline void q_mul(const double a[4], const double b[4], double r[4]) {
    const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    const double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    r[0] = w, r[1] = x, r[2] = y, r[3] = z;
}
__host__ __device__ inline void q_normalize(double q[4]) {
    const double z = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (z > 0) {
        const double n = sqrt(z);
        q[0] /= n, q[1] /= n, q[2] /= n, q[3] /= n;
    }
}
__host__ __device__ inline void q_to_R(const double q[4], double R[9]) {  // Eigen toRotationMatrix
    const double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
    const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
    const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
    const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
    R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
    R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
    R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
}
__host__ __device__ inline void pose_to_Rt(const Pose &p, double R[9], double t[3]) {
    q_to_R(p.q, R);
    t[0] = p.p[0], t[1] = p.p[1], t[2] = p.p[2];
}
// [R^T | -R^T p]  -- lvt_pose.cpp:36-43.  w[12] row-major 3x4
__host__ __device__ inline void world_to_camera(const Pose &pose, double w[12]) {
    double R[9];
    q_to_R(pose.q, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) w[4 * i + j] = R[3 * j + i];
        w[4 * i + 3] = (-R[0 + i]) * pose.p[0] + (-R[3 + i]) * pose.p[1] + (-R[6 + i]) * pose.p[2];
    }
}
// lvt_pose.cpp:28-34
__host__ __device__ inline void right_camera_pose(const Pose &l, double baseline, Pose &r) {
    double R[9];
    q_to_R(l.q, R);
    for (int k = 0; k < 4; k++) r.q[k] = l.q[k];
    r.p[0] = (R[0] * baseline + R[1] * 0.0 + R[2] * 0.0) + l.p[0];
    r.p[1] = (R[3] * baseline + R[4] * 0.0 + R[5] * 0.0) + l.p[1];
    r.p[2] = (R[6] * baseline + R[7] * 0.0 + R[8] * 0.0) + l.p[2];
}
// lvt_local_map.cpp:62-82
__host__ __device__ inline bool is_point_visible(const double X[3], const double w[12], const Params &p, double &u, double &v) {
    const double cx = ((w[0] * X[0] + w[1] * X[1]) + w[2] * X[2]) + w[3] * 1.0;
    const double cy = ((w[4] * X[0] + w[5] * X[1]) + w[6] * X[2]) + w[7] * 1.0;
    const double cz = ((w[8] * X[0] + w[9] * X[1]) + w[10] * X[2]) + w[11] * 1.0;
    if (cz < p.near_plane || cz > p.far_plane) return false;
    const double inv_z = 1.0 / cz;
    const double uu = p.fx * cx * inv_z + p.cx;
    const double vv = p.fy * cy * inv_z + p.cy;
    if (uu < p.min_x || uu > p.max_x || vv < p.min_y || vv > p.max_y) return false;
    u = uu;
    v = vv;
    return true;
}
// SURVEY A.7
__host__ __device__ inline void undistort_point(const Params &p, float x_in, float y_in, float &x_out, float &y_out) {
    const double fx = p.fx, fy = p.fy, cx = p.cx, cy = p.cy;
    const double k1 = p.k1, k2 = p.k2, p1 = p.p1, p2 = p.p2, k3 = p.k3;
    double x = ((double)x_in - cx) / fx, y = ((double)y_in - cy) / fy;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
        const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        const double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    x_out = (float)(x * fx + cx);
    y_out = (float)(y * fy + cy);
}

// lvt_motion_model.cpp:42-65 (Eigen 3.3 slerp / inverse / normalize semantics).  PURE: reads the model state from
// `c`, returns the predicted pose and the model's next state {last_q[4], ang_vel[4], last_p[3], lin_vel[3]}.
__device__ inline void motion_predict(const Ctl &c, const Pose &cur, Pose &out, double next[14]) {
    double nv[3];
    for (int k = 0; k < 3; k++) nv[k] = ((cur.p[k] - c.mm_last_p[k]) + c.mm_lin_vel[k]) * 0.5;
    double inv[4];
    {
        const double *q = c.mm_last_q;
        const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        if (n2 > 0) {
            inv[0] = q[0] / n2, inv[1] = -q[1] / n2, inv[2] = -q[2] / n2, inv[3] = -q[3] / n2;
        } else {
            inv[0] = inv[1] = inv[2] = inv[3] = 0;
        }
    }
    double diff[4];
    q_mul(cur.q, inv, diff);
    double nav[4];
    {  // diff.slerp(0.5, ang_vel)
        const double *b = c.mm_ang_vel;
        const double one = 1.0 - 2.220446049250313e-16;
        const double d = diff[0] * b[0] + diff[1] * b[1] + diff[2] * b[2] + diff[3] * b[3];
        const double absD = fabs(d);
        double s0, s1;
        if (absD >= one) {
            s0 = 1.0 - 0.5;
            s1 = 0.5;
        } else {
            const double theta = acos(absD);
            const double sinTheta = sin(theta);
            s0 = sin((1.0 - 0.5) * theta) / sinTheta;
            s1 = sin(0.5 * theta) / sinTheta;
        }
        if (d < 0) s1 = -s1;
        for (int k = 0; k < 4; k++) nav[k] = s0 * diff[k] + s1 * b[k];
    }
    q_normalize(nav);
    for (int k = 0; k < 4; k++) {
        next[k] = cur.q[k];
        next[4 + k] = nav[k];
    }
    for (int k = 0; k < 3; k++) {
        next[8 + k] = cur.p[k];
        next[11 + k] = nv[k];
        out.p[k] = cur.p[k] + nv[k];
    }
    q_mul(cur.q, nav, out.q);
    q_normalize(out.q);
}

}  // namespace lvt
