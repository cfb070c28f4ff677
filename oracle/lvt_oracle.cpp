// lvt_oracle.cpp -- CPU ORACLE (test infrastructure; see lvt_oracle.h for scope and "parity unpinned").
//
// A dependency-free scalar C++17 restatement of the reference's per-frame tracking path.  Each
// function cites the reference file:line it follows (paths relative to /root/reference/) or the
// SURVEY.md appendix item restating the un-vendored third-party algorithm.  Built with
// -ffp-contract=off so no FMA contraction changes a gate.
#include "lvt_oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <deque>
#include <limits>
#include <set>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// constants -- lvt/src/lvt_definitions.h:29-34
// ---------------------------------------------------------------------------------------------
constexpr double kReprojTh2 = 5.991;
// edges whose squared error lies this close to the chi2 threshold are COUNTED (LVTO_C_PNP_BORDERLINE): an implementation that evaluates
// the same error in a different operation order agrees on e^2 to ~1e-11, so only those decisions could differ
constexpr double kGateMargin = 1e-8;
constexpr int kNMapPoints = 250;
constexpr int kRowRadius = 2;
constexpr int kHashCell = 25;
constexpr int kCornersLowTh = 200;
constexpr int kNMatchesTh = 50;

// ---------------------------------------------------------------------------------------------
// tiny fp64 linear algebra (instead of Eigen) -- lvt/src/lvt_pose.h:34-48
// ---------------------------------------------------------------------------------------------
struct V3 {
    double x = 0, y = 0, z = 0;
};
struct Quat {  // Eigen::Quaterniond coefficient semantics
    double w = 1, x = 0, y = 0, z = 0;
};
struct M33 {
    double m[3][3];
};
struct M34 {
    double m[3][4];
};

inline Quat qmul(const Quat &a, const Quat &b) {  // Eigen quat_product (generic)
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
inline double qsqn(const Quat &q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
inline Quat qnormalized(const Quat &q) {  // Eigen normalize(): if (z>0) /= sqrt(z)
    double z = qsqn(q);
    if (z > 0) {
        double n = std::sqrt(z);
        return Quat{q.w / n, q.x / n, q.y / n, q.z / n};
    }
    return q;
}
inline Quat qinverse(const Quat &q) {  // Eigen inverse(): conjugate / squaredNorm
    double n2 = qsqn(q);
    if (n2 > 0) return Quat{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
    return Quat{0, 0, 0, 0};
}
inline Quat qslerp(const Quat &a, double t, const Quat &b) {  // Eigen 3.3 QuaternionBase::slerp
    const double one = 1.0 - std::numeric_limits<double>::epsilon();
    double d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
    double absD = std::fabs(d);
    double s0, s1;
    if (absD >= one) {
        s0 = 1.0 - t;
        s1 = t;
    } else {
        double theta = std::acos(absD);
        double sinTheta = std::sin(theta);
        s0 = std::sin((1.0 - t) * theta) / sinTheta;
        s1 = std::sin(t * theta) / sinTheta;
    }
    if (d < 0) s1 = -s1;
    return Quat{s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}
inline M33 qtoR(const Quat &q) {  // Eigen toRotationMatrix
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M33 r;
    r.m[0][0] = 1 - (tyy + tzz);
    r.m[0][1] = txy - twz;
    r.m[0][2] = txz + twy;
    r.m[1][0] = txy + twz;
    r.m[1][1] = 1 - (txx + tzz);
    r.m[1][2] = tyz - twx;
    r.m[2][0] = txz - twy;
    r.m[2][1] = tyz + twx;
    r.m[2][2] = 1 - (txx + tyy);
    return r;
}

struct Pose {  // lvt/src/lvt_pose.h:51-79 : camera-to-world
    Quat q;
    V3 p;
};

// lvt/src/lvt_pose.cpp:36-43
inline M34 world_to_camera(const Pose &pose) {
    M33 R = qtoR(pose.q);
    M34 w;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) w.m[i][j] = R.m[j][i];
        w.m[i][3] = (-R.m[0][i]) * pose.p.x + (-R.m[1][i]) * pose.p.y + (-R.m[2][i]) * pose.p.z;
    }
    return w;
}
// lvt/src/lvt_pose.cpp:28-34
inline Pose right_camera_pose(const Pose &left, double baseline) {
    M33 R = qtoR(left.q);
    Pose r;
    r.q = left.q;
    r.p.x = (R.m[0][0] * baseline + R.m[0][1] * 0.0 + R.m[0][2] * 0.0) + left.p.x;
    r.p.y = (R.m[1][0] * baseline + R.m[1][1] * 0.0 + R.m[1][2] * 0.0) + left.p.y;
    r.p.z = (R.m[2][0] * baseline + R.m[2][1] * 0.0 + R.m[2][2] * 0.0) + left.p.z;
    return r;
}

// ---------------------------------------------------------------------------------------------
// BRIEF test pairs (stand-in table) -- data shared with the product as a table file
// ---------------------------------------------------------------------------------------------
const signed char kBriefPairs[256][4] = {
#include "../include/lvt_brief256_pattern.inc"
};

struct KeyPoint {
    float x, y, response;
};

// ---------------------------------------------------------------------------------------------
// AGAST OAST-9/16 -- SURVEY A.1 (OpenCV features2d agast.cpp / agast_score.cpp)
// ---------------------------------------------------------------------------------------------
const int kCircle[16][2] = {{-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}, {0, -3}, {1, -3}, {2, -2}, {3, -1},
                            {3, 0},  {3, 1},   {2, 2},   {1, 3},   {0, 3},  {-1, 3}, {-2, 2}, {-3, 1}};

// score = max{b <= 254 : exists 9 contiguous circle pixels all > p+b or all < p-b}; -1 if none at b=0.
// (the binary search of agast_cornerScore returns exactly this value when started from a threshold
//  t at which the pixel is a corner; corner(t) <=> score >= t.)
inline int oast9_score(const uint8_t *c, const int off[16]) {
    int d[16];
    const int p = *c;
    for (int i = 0; i < 16; i++) d[i] = int(c[off[i]]) - p;
    int best = 0;  // max over arcs of min(+-d) ; corner(b) <=> best >= b+1
    for (int s = 0; s < 16; s++) {
        int mb = 1 << 20, md = 1 << 20;
        for (int k = 0; k < 9; k++) {
            int v = d[(s + k) & 15];
            mb = std::min(mb, v);
            md = std::min(md, -v);
        }
        best = std::max(best, std::max(mb, md));
    }
    return best - 1;  // -1 when best == 0 (not a corner even at b = 0)
}

// one ROI as an isolated image; `stride` is the parent image's step (A.1)
void agast_detect_roi(const uint8_t *img, int rows, int cols, int stride, int threshold, bool nonmax,
                      std::vector<KeyPoint> &out) {
    out.clear();
    if (rows < 7 || cols < 7) return;
    int off[16];
    for (int i = 0; i < 16; i++) off[i] = kCircle[i][0] + kCircle[i][1] * stride;
    std::vector<KeyPoint> kpts;
    for (int y = 3; y <= rows - 4; y++) {
        const uint8_t *row = img + (size_t)y * stride;
        for (int x = 3; x <= cols - 4; x++) {
            // quick reject: a 9-arc contains at least one pixel of every opposite pair
            const int p = row[x];
            const int cb = p + threshold, c_b = p - threshold;
            const int a0 = row[x + off[0]], a8 = row[x + off[8]];
            const int a4 = row[x + off[4]], a12 = row[x + off[12]];
            bool br = (a0 > cb || a8 > cb) && (a4 > cb || a12 > cb);
            bool dk = (a0 < c_b || a8 < c_b) && (a4 < c_b || a12 < c_b);
            if (!br && !dk) continue;
            int s = oast9_score(row + x, off);
            if (s >= threshold) kpts.push_back(KeyPoint{(float)x, (float)y, (float)s});
        }
    }
    if (!nonmax) {
        out = kpts;
        return;
    }
    // AGAST's own non-maximum suppression: raster sweep, above/left links (A.1)
    const size_t n = kpts.size();
    std::vector<int> nms(n, -1);
    size_t lastRow = 0, next_lastRow = 0;
    size_t lastRowCorner_ind = 0, next_lastRowCorner_ind = 0;
    for (size_t cur = 0; cur < n; cur++) {
        const KeyPoint &cc = kpts[cur];
        if (lastRow + 1 < cc.y) {
            lastRow = next_lastRow;
            lastRowCorner_ind = next_lastRowCorner_ind;
        }
        if (next_lastRow != cc.y) {
            next_lastRow = (size_t)cc.y;
            next_lastRowCorner_ind = cur;
        }
        if (lastRow + 1 == cc.y) {
            while ((kpts[lastRowCorner_ind].x < cc.x) && (kpts[lastRowCorner_ind].y == lastRow))
                lastRowCorner_ind++;
            if ((kpts[lastRowCorner_ind].x == cc.x) && (lastRowCorner_ind != cur)) {
                size_t w = lastRowCorner_ind;
                while (nms[w] != -1) w = nms[w];
                if (kpts[cur].response < kpts[w].response)
                    nms[cur] = (int)w;
                else
                    nms[w] = (int)cur;
            }
        }
        int t = (int)cur - 1;
        if ((cur != 0) && (kpts[t].y == cc.y) && (kpts[t].x + 1 == cc.x)) {
            int above = nms[cur];
            while (nms[t] != -1) t = nms[t];
            if (above == -1) {
                if ((size_t)t != cur) {
                    if (kpts[cur].response < kpts[t].response)
                        nms[cur] = t;
                    else
                        nms[t] = (int)cur;
                }
            } else {
                if (t != above) {
                    if (kpts[above].response < kpts[t].response) {
                        nms[above] = t;
                        nms[cur] = t;
                    } else {
                        nms[t] = above;
                        nms[cur] = above;
                    }
                }
            }
        }
    }
    for (size_t i = 0; i < n; i++)
        if (nms[i] == -1) out.push_back(kpts[i]);
}

// ---------------------------------------------------------------------------------------------
// ANMS -- lvt/src/lvt_image_features_handler.cpp:34-83 (std::sort order is part of the result)
// ---------------------------------------------------------------------------------------------
void anms(std::vector<KeyPoint> &keypoints, const int num_to_keep, const float tx, const float ty) {
    std::sort(keypoints.begin(), keypoints.end(),
              [](const KeyPoint &l, const KeyPoint &r) { return l.response > r.response; });
    std::vector<KeyPoint> kept;
    kept.reserve(num_to_keep);
    const int n = (int)keypoints.size();
    std::vector<float> radii(n), radiiSorted(n);
    const float robustCoeff = 1.11;
    for (int i = 0; i < n; i++) {
        const float response = keypoints[i].response * robustCoeff;
        float radius = (std::numeric_limits<float>::max)();
        for (int j = 0; j < i && keypoints[j].response > response; j++) {
            const float dx = keypoints[i].x - keypoints[j].x;
            const float dy = keypoints[i].y - keypoints[j].y;
            radius = (std::min)(radius, dx * dx + dy * dy);
        }
        radius = sqrtf(radius);
        radii[i] = radius;
        radiiSorted[i] = radius;
    }
    std::sort(radiiSorted.begin(), radiiSorted.end(), [](const float &l, const float &r) { return l > r; });
    const float decisionRadius = radiiSorted[num_to_keep];
    for (int i = 0; i < n; i++) {
        if (radii[i] >= decisionRadius) {
            KeyPoint k = keypoints[i];
            k.x += tx;
            k.y += ty;
            kept.push_back(k);
        }
    }
    kept.swap(keypoints);
}

struct Rect {
    int x, y, w, h;
};

// grid rects -- lvt_image_features_handler.cpp:95-114
std::vector<Rect> make_grid(int img_w, int img_h, int s) {
    std::vector<Rect> r;
    int ny = 1 + ((img_h - 1) / s), nx = 1 + ((img_w - 1) / s);
    for (int i = 0; i < ny; i++)
        for (int k = 0; k < nx; k++) {
            int sy = s, sx = s;
            if ((i == ny - 1) && ((i + 1) * s > img_h)) sy = img_h - i * s;
            if ((k == nx - 1) && ((k + 1) * s > img_w)) sx = img_w - k * s;
            r.push_back(Rect{k * s, i * s, sx, sy});
        }
    return r;
}

// perform_detect_corners -- lvt_image_features_handler.cpp:131-154
void detect_corners(const uint8_t *img, int stride, const std::vector<Rect> &rects, int threshold,
                    int max_per_cell, std::vector<KeyPoint> &all) {
    std::vector<KeyPoint> kps;
    for (const Rect &rc : rects) {
        agast_detect_roi(img + (size_t)rc.y * stride + rc.x, rc.h, rc.w, stride, threshold, true, kps);
        if ((int)kps.size() > max_per_cell) {
            anms(kps, max_per_cell, (float)rc.x, (float)rc.y);
        } else {
            for (auto &k : kps) {
                k.x += (float)rc.x;
                k.y += (float)rc.y;
            }
        }
        all.insert(all.end(), kps.begin(), kps.end());
    }
}

// detection + low-count retry -- lvt_image_features_handler.cpp:158-169
void detect_with_retry(const uint8_t *img, int stride, const std::vector<Rect> &rects, const lvto_params &p,
                       std::vector<KeyPoint> &all, int *retry_used) {
    all.clear();
    detect_corners(img, stride, rects, p.agast_threshold, p.max_keypoints_per_cell, all);
    if (retry_used) *retry_used = 0;
    if ((int)all.size() < kCornersLowTh) {
        all.clear();
        int lowered = (double)p.agast_threshold * 0.5 + 0.5;
        detect_corners(img, stride, rects, lowered, p.max_keypoints_per_cell, all);
        if (retry_used) *retry_used = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// BRIEF-32 -- SURVEY A.3 (opencv_contrib xfeatures2d brief.cpp); call sites handler.cpp:172,190,247
// ---------------------------------------------------------------------------------------------
struct Integral {
    int rows, cols;  // image size; table is (rows+1) x (cols+1)
    std::vector<int> s;
    inline int at(int y, int x) const {  // zero-padded outside (only reachable via fractional corners)
        y = std::min(std::max(y, 0), rows);
        x = std::min(std::max(x, 0), cols);
        return s[(size_t)y * (cols + 1) + x];
    }
};
void build_integral(const uint8_t *img, int rows, int cols, int stride, Integral &I) {
    I.rows = rows;
    I.cols = cols;
    I.s.assign((size_t)(rows + 1) * (cols + 1), 0);
    for (int y = 0; y < rows; y++) {
        int run = 0;
        const uint8_t *r = img + (size_t)y * stride;
        int *cur = &I.s[(size_t)(y + 1) * (cols + 1)];
        const int *prev = &I.s[(size_t)y * (cols + 1)];
        for (int x = 0; x < cols; x++) {
            run += r[x];
            cur[x + 1] = prev[x + 1] + run;
        }
    }
}
inline int smoothed_sum(const Integral &I, float px, float py, int dy, int dx) {
    const int HALF = 4;
    const int iy = (int)(py + 0.5) + dy;
    const int ix = (int)(px + 0.5) + dx;
    return I.at(iy + HALF + 1, ix + HALF + 1) - I.at(iy + HALF + 1, ix - HALF) - I.at(iy - HALF, ix + HALF + 1) +
           I.at(iy - HALF, ix - HALF);
}
// keeps kp iff 28 <= cvRound(x) < W-28 and 28 <= cvRound(y) < H-28 (KeyPointsFilter::runByImageBorder)
inline bool brief_border_keep(float x, float y, int rows, int cols) {
    const int B = 28;
    if (rows <= B * 2 || cols <= B * 2) return false;
    long ix = lrintf(x), iy = lrintf(y);
    return ix >= B && ix < cols - B && iy >= B && iy < rows - B;
}
void brief_compute(const uint8_t *img, int rows, int cols, int stride, std::vector<KeyPoint> &kps,
                   std::vector<uint8_t> &desc) {
    Integral I;
    build_integral(img, rows, cols, stride, I);
    std::vector<KeyPoint> kept;
    kept.reserve(kps.size());
    for (const auto &k : kps)
        if (brief_border_keep(k.x, k.y, rows, cols)) kept.push_back(k);
    kps.swap(kept);
    desc.assign(kps.size() * 32, 0);
    for (size_t i = 0; i < kps.size(); i++) {
        uint8_t *d = &desc[i * 32];
        for (int j = 0; j < 32; j++) {
            int byte = 0;
            for (int k = 0; k < 8; k++) {
                const signed char *t = kBriefPairs[8 * j + k];
                int a = smoothed_sum(I, kps[i].x, kps[i].y, t[0], t[1]);
                int b = smoothed_sum(I, kps[i].x, kps[i].y, t[2], t[3]);
                byte |= (a < b) << (7 - k);
            }
            d[j] = (uint8_t)byte;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Hamming / masked 2-NN -- SURVEY A.4 (cv::BFMatcher(NORM_HAMMING).knnMatch k=2 with mask)
// ---------------------------------------------------------------------------------------------
inline int hamming32(const uint8_t *a, const uint8_t *b) {
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32);
    std::memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}
struct Top2 {
    int i1 = -1, d1 = INT_MAX, i2 = -1, d2 = INT_MAX;
    int count = 0;
    // candidates MUST be offered in ascending index order to reproduce "ties keep the lower index"
    inline void offer(int idx, int d) {
        count++;
        if (d < d1) {
            d2 = d1;
            i2 = i1;
            d1 = d;
            i1 = idx;
        } else if (d < d2) {
            d2 = d;
            i2 = idx;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// lvt_image_features_struct -- lvt/src/lvt_image_features_struct.{h,cpp}
// ---------------------------------------------------------------------------------------------
struct FeatureStruct {
    std::vector<KeyPoint> kps;
    std::vector<uint8_t> desc;  // N x 32
    std::vector<uint8_t> matched;
    std::vector<float> depths;
    int cell_size = 25, ccx = -1, ccy = -1, cell_search_radius = 0, tracking_radius = 0;
    int img_rows = 0, img_cols = 0, vertical_search_radius = 2;
    float tri_ratio = 0.6f, track_ratio = 0.8f, desc_th = 25.0f;
    std::vector<std::vector<int>> hash;  // ccy*ccx lists

    int count() const { return (int)kps.size(); }
    const uint8_t *descriptor(int i) const { return &desc[(size_t)i * 32]; }

    // struct.cpp:35-66
    void init(int rows, int cols, std::vector<KeyPoint> &in_kps, std::vector<uint8_t> &in_desc, int in_tracking_radius,
              int hashing_cell, int vrad, float tri_th, float track_th, float dth, std::vector<float> *kd = nullptr) {
        cell_size = hashing_cell;
        vertical_search_radius = vrad;
        tri_ratio = tri_th;
        track_ratio = track_th;
        desc_th = dth;
        img_rows = rows;
        img_cols = cols;
        tracking_radius = in_tracking_radius;
        const float k_cell = (float)cell_size;
        ccx = (int)std::ceil(img_cols / k_cell);
        ccy = (int)std::ceil(img_rows / k_cell);
        kps.swap(in_kps);
        desc.swap(in_desc);
        cell_search_radius = (tracking_radius == cell_size) ? 1 : (int)std::ceil((float)tracking_radius / k_cell);
        hash.assign((size_t)ccy * ccx, {});
        for (int i = 0; i < (int)kps.size(); i++) {
            int hy = (int)std::floor(kps[i].y / k_cell), hx = (int)std::floor(kps[i].x / k_cell);
            // reference indexes out of bounds for undistorted points outside the image;
            // such points are dropped before init() by the RGB-D path below.
            hash[(size_t)hy * ccx + hx].push_back(i);
        }
        matched.assign(kps.size(), 0);
        if (kd) depths = *kd;
    }

    // struct.cpp:68-120
    int find_match_index(double ptx_d, double pty_d, const uint8_t *qdesc, float *d1, float *d2) const {
        const float ptx = (float)ptx_d, pty = (float)pty_d;
        const int hy = (int)std::floor(pty / (float)cell_size), hx = (int)std::floor(ptx / (float)cell_size);
        int sy = std::max(hy - cell_search_radius, 0), ey = std::min(hy + cell_search_radius + 1, ccy);
        int sx = std::max(hx - cell_search_radius, 0), ex = std::min(hx + cell_search_radius + 1, ccx);
        const float r2 = (float)(tracking_radius * tracking_radius);
        // candidates collected then offered in ascending index order (BFMatcher scans train rows in order)
        std::vector<int> cand;
        for (int i = sy; i < ey; i++)
            for (int k = sx; k < ex; k++)
                for (int idx : hash[(size_t)i * ccx + k]) {
                    if (!matched[idx]) {
                        const float dx = kps[idx].x - ptx, dy = kps[idx].y - pty;
                        if ((dx * dx + dy * dy) < r2) cand.push_back(idx);
                    }
                }
        std::sort(cand.begin(), cand.end());
        Top2 t;
        for (int idx : cand) t.offer(idx, hamming32(qdesc, descriptor(idx)));
        if (t.count > 1) {
            float ratio = (float)t.d1 / (float)t.d2;
            if (ratio < track_ratio) {
                *d1 = (float)t.d1;
                *d2 = (float)t.d2;
                return t.i1;
            }
        } else if (t.count == 1 && (float)t.d1 <= desc_th) {
            *d1 = (float)t.d1;
            *d2 = 0.0f;  // reference leaves d2 with the (unset) second distance; unused downstream
            return t.i1;
        }
        return -1;
    }

    // struct.cpp:122-148
    int row_match(float ptx, float pty, const uint8_t *qdesc) const {
        (void)ptx;
        int start_y = (int)pty - vertical_search_radius;
        if (start_y < 0) start_y = 0;
        int end_y = (int)pty + vertical_search_radius;
        if (end_y > img_rows) end_y = img_rows;
        Top2 t;
        for (int i = 0, n = count(); i < n; i++) {
            if (!matched[i] && kps[i].y >= start_y && kps[i].y <= end_y) t.offer(i, hamming32(qdesc, descriptor(i)));
        }
        if ((t.count > 1 && ((float)t.d1 / (float)t.d2) < tri_ratio) || (t.count == 1 && (float)t.d1 <= desc_th))
            return t.i1;
        return -1;
    }
};

// ---------------------------------------------------------------------------------------------
// is_point_visible -- lvt/src/lvt_local_map.cpp:62-82
// ---------------------------------------------------------------------------------------------
struct Bounds {
    float min_x, max_x, min_y, max_y;
};
inline bool is_point_visible(const V3 &pt, const M34 &w, const lvto_params &p, const Bounds &b, double &u_out,
                             double &v_out) {
    const double cxm = ((w.m[0][0] * pt.x + w.m[0][1] * pt.y) + w.m[0][2] * pt.z) + w.m[0][3] * 1.0;
    const double cym = ((w.m[1][0] * pt.x + w.m[1][1] * pt.y) + w.m[1][2] * pt.z) + w.m[1][3] * 1.0;
    const double czm = ((w.m[2][0] * pt.x + w.m[2][1] * pt.y) + w.m[2][2] * pt.z) + w.m[2][3] * 1.0;
    if (czm < p.near_plane_distance || czm > p.far_plane_distance) return false;
    const double inv_z = 1.0 / czm;
    const double u = p.fx * cxm * inv_z + p.cx;
    const double v = p.fy * cym * inv_z + p.cy;
    if (u < b.min_x || u > b.max_x || v < b.min_y || v > b.max_y) return false;
    u_out = u;
    v_out = v;
    return true;
}

// cv::undistortPoints, fixed 5 iterations -- SURVEY A.7 (float in/out, double inside)
inline void undistort_point(const lvto_params &p, float x_in, float y_in, float &x_out, float &y_out) {
    const double fx = p.fx, fy = p.fy, cx = p.cx, cy = p.cy;
    const double k1 = p.k1, k2 = p.k2, p1 = p.p1, p2 = p.p2, k3 = p.k3;
    double x = ((double)x_in - cx) / fx, y = ((double)y_in - cy) / fy;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
        double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    x_out = (float)(x * fx + cx);
    y_out = (float)(y * fy + cy);
}

// ---------------------------------------------------------------------------------------------
// linear least squares 4x3 (stands in for Eigen::JacobiSVD<4x3>.solve): Householder QR
// ---------------------------------------------------------------------------------------------
bool ls_solve_4x3(double A[4][3], double b[4], double x[3]) {
    for (int k = 0; k < 3; k++) {
        double norm = 0;
        for (int i = k; i < 4; i++) norm += A[i][k] * A[i][k];
        norm = std::sqrt(norm);
        if (norm < 1e-300) return false;
        double alpha = (A[k][k] > 0) ? -norm : norm;
        double v[4] = {0, 0, 0, 0};
        for (int i = k; i < 4; i++) v[i] = A[i][k];
        v[k] -= alpha;
        double vnorm2 = 0;
        for (int i = k; i < 4; i++) vnorm2 += v[i] * v[i];
        if (vnorm2 > 0) {
            for (int j = k; j < 3; j++) {
                double dot = 0;
                for (int i = k; i < 4; i++) dot += v[i] * A[i][j];
                double f = 2.0 * dot / vnorm2;
                for (int i = k; i < 4; i++) A[i][j] -= f * v[i];
            }
            double dot = 0;
            for (int i = k; i < 4; i++) dot += v[i] * b[i];
            double f = 2.0 * dot / vnorm2;
            for (int i = k; i < 4; i++) b[i] -= f * v[i];
        }
    }
    // rank gate: JacobiSVD would return a finite minimum-norm solution for a rank-deficient system
    // (zero disparity); this build rejects such pairs instead (DESIGN.md "deviations").
    double rmax = std::max(std::fabs(A[0][0]), std::max(std::fabs(A[1][1]), std::fabs(A[2][2])));
    for (int k = 0; k < 3; k++)
        if (!(std::fabs(A[k][k]) > 1e-12 * rmax)) return false;
    x[2] = b[2] / A[2][2];
    x[1] = (b[1] - A[1][2] * x[2]) / A[1][1];
    x[0] = (b[0] - A[0][1] * x[1] - A[0][2] * x[2]) / A[0][0];
    return true;
}

// ---------------------------------------------------------------------------------------------
// g2o motion-only BA -- lvt/src/lvt_pnp_solver.cpp:60-128 + SURVEY A.6
// ---------------------------------------------------------------------------------------------
struct SBACam {
    Quat r;
    V3 t;
    double fx, fy, cx, cy;
    double w2n[3][4], w2i[3][4];
    double dRdx[3][3], dRdy[3][3], dRdz[3][3];
    void refresh() {
        M33 R = qtoR(r);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) w2n[i][j] = R.m[j][i];
        for (int i = 0; i < 3; i++) w2n[i][3] = -(w2n[i][0] * t.x + w2n[i][1] * t.y + w2n[i][2] * t.z);
        for (int j = 0; j < 4; j++) {
            w2i[0][j] = fx * w2n[0][j] + cx * w2n[2][j];
            w2i[1][j] = fy * w2n[1][j] + cy * w2n[2][j];
            w2i[2][j] = w2n[2][j];
        }
        // dRd{x,y,z} = dRid{x,y,z} * w2n[:, :3]
        for (int j = 0; j < 3; j++) {
            dRdx[0][j] = 0;
            dRdx[1][j] = 2.0 * w2n[2][j];
            dRdx[2][j] = -2.0 * w2n[1][j];
            dRdy[0][j] = -2.0 * w2n[2][j];
            dRdy[1][j] = 0;
            dRdy[2][j] = 2.0 * w2n[0][j];
            dRdz[0][j] = 2.0 * w2n[1][j];
            dRdz[1][j] = -2.0 * w2n[0][j];
            dRdz[2][j] = 0;
        }
    }
    void init(const Quat &q, const V3 &p) {  // SE3Quat ctor: normalizeRotation()
        r = q;
        if (r.w < 0) {
            r.w = -r.w;
            r.x = -r.x;
            r.y = -r.y;
            r.z = -r.z;
        }
        r = qnormalized(r);
        t = p;
        refresh();
    }
    void update(const double d[6]) {  // SBACam::update
        t.x += d[0];
        t.y += d[1];
        t.z += d[2];
        Quat qr;
        qr.x = d[3];
        qr.y = d[4];
        qr.z = d[5];
        qr.w = std::sqrt(1.0 - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]));
        r = qnormalized(qmul(r, qr));
        refresh();
    }
};

// exact SPD solve standing in for LinearSolverPCG with the exact block-Jacobi preconditioner (A.6)
bool solve6_spd(const double H[6][6], const double b[6], double x[6]) {
    double L[6][6] = {};
    for (int j = 0; j < 6; j++) {
        double s = H[j][j];
        for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
        if (!(s > 0) || !std::isfinite(s)) return false;
        L[j][j] = std::sqrt(s);
        for (int i = j + 1; i < 6; i++) {
            double v = H[i][j];
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
            L[i][j] = v / L[j][j];
        }
    }
    double y[6];
    for (int i = 0; i < 6; i++) {
        double v = b[i];
        for (int k = 0; k < i; k++) v -= L[i][k] * y[k];
        y[i] = v / L[i][i];
    }
    for (int i = 5; i >= 0; i--) {
        double v = y[i];
        for (int k = i + 1; k < 6; k++) v -= L[k][i] * x[k];
        x[i] = v / L[i][i];
    }
    return true;
}

struct PnpResult {
    Pose pose;
    int solve_calls = 0;
    int inliers = 0;
    int borderline = 0;      // gate decisions (both passes) taken within kGateMargin of the threshold
    double min_margin = 1e300;  // the closest any gate decision came to it
    // the branches of A.6 a benign scene never takes: LM trials (solve + update + chi2 per trial), trials REJECTED (rho <= 0 or a non-finite
    // chi2: lambda *= ni, pop()), passes ended by Terminate (qmax == 10 or rho == 0)
    int trials = 0, rejections = 0, terminates = 0;
};
// the last pnp_compute_pose of this thread: errors the gates saw (pass 2's), for tests that compare them edge by edge
static thread_local std::vector<double> g_last_pnp_err;
static thread_local PnpResult g_last_pnp;

PnpResult pnp_compute_pose(const lvto_params &prm, const Pose &prior, const std::vector<V3> &pts,
                           const std::vector<float> &obs /* n x 2 */, std::vector<int> *inlier_marks,
                           std::vector<double> *trace) {
    const int n = (int)pts.size();
    SBACam cam;
    cam.fx = prm.fx;
    cam.fy = prm.fy;
    cam.cx = prm.cx;
    cam.cy = prm.cy;
    cam.init(prior.q, prior.p);
    static const double mono_chi = std::sqrt(kReprojTh2);
    const double dsqr = mono_chi * mono_chi;
    const double dsqrReci = 1.0 / dsqr;

    std::vector<int> level(n, 0);
    std::vector<double> err(2 * (size_t)n, 0.0);  // last computeActiveErrors() result per edge
    std::vector<int> marks(n, 1);
    PnpResult res;

    auto compute_active_errors = [&](const SBACam &c) {
        for (int i = 0; i < n; i++) {
            if (level[i] != 0) continue;
            const V3 &X = pts[i];
            double px = ((c.w2i[0][0] * X.x + c.w2i[0][1] * X.y) + c.w2i[0][2] * X.z) + c.w2i[0][3];
            double py = ((c.w2i[1][0] * X.x + c.w2i[1][1] * X.y) + c.w2i[1][2] * X.z) + c.w2i[1][3];
            double pz = ((c.w2i[2][0] * X.x + c.w2i[2][1] * X.y) + c.w2i[2][2] * X.z) + c.w2i[2][3];
            err[2 * i] = px / pz - (double)obs[2 * i];
            err[2 * i + 1] = py / pz - (double)obs[2 * i + 1];
        }
    };
    auto active_robust_chi2 = [&]() {
        double chi = 0;
        for (int i = 0; i < n; i++) {
            if (level[i] != 0) continue;
            double e2 = err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1];
            double aux = dsqrReci * e2 + 1.0;
            chi += dsqr * std::log(aux);
        }
        return chi;
    };

    for (int pass = 0; pass < 2; pass++) {
        // initializeOptimization(0): active = level 0 edges.  optimize(5):
        double lambda = 0, ni = 2;
        bool ok = true;
        int n_active = 0;
        for (int i = 0; i < n; i++) n_active += (level[i] == 0);
        for (int iter = 0; iter < 5 && ok && n_active > 0; iter++) {
            res.solve_calls++;
            compute_active_errors(cam);
            double currentChi = active_robust_chi2();
            double tempChi = currentChi;
            // buildSystem: linearize at current estimate, accumulate H, b
            double H[6][6] = {}, b[6] = {};
            for (int i = 0; i < n; i++) {
                if (level[i] != 0) continue;
                const V3 &X = pts[i];
                double pcx = ((cam.w2n[0][0] * X.x + cam.w2n[0][1] * X.y) + cam.w2n[0][2] * X.z) + cam.w2n[0][3];
                double pcy = ((cam.w2n[1][0] * X.x + cam.w2n[1][1] * X.y) + cam.w2n[1][2] * X.z) + cam.w2n[1][3];
                double pcz = ((cam.w2n[2][0] * X.x + cam.w2n[2][1] * X.y) + cam.w2n[2][2] * X.z) + cam.w2n[2][3];
                double ipz2 = 1.0 / (pcz * pcz);
                double ipz2fx = ipz2 * cam.fx, ipz2fy = ipz2 * cam.fy;
                double pwt[3] = {X.x - cam.t.x, X.y - cam.t.y, X.z - cam.t.z};
                double J[2][6];
                const double(*dR[3])[3] = {cam.dRdx, cam.dRdy, cam.dRdz};
                for (int c = 0; c < 3; c++) {
                    double dp[3];
                    for (int r = 0; r < 3; r++) dp[r] = (dR[c][r][0] * pwt[0] + dR[c][r][1] * pwt[1]) + dR[c][r][2] * pwt[2];
                    J[0][3 + c] = (pcz * dp[0] - pcx * dp[2]) * ipz2fx;
                    J[1][3 + c] = (pcz * dp[1] - pcy * dp[2]) * ipz2fy;
                }
                for (int c = 0; c < 3; c++) {
                    double dp[3] = {-cam.w2n[0][c], -cam.w2n[1][c], -cam.w2n[2][c]};
                    J[0][c] = (pcz * dp[0] - pcx * dp[2]) * ipz2fx;
                    J[1][c] = (pcz * dp[1] - pcy * dp[2]) * ipz2fy;
                }
                double e0 = err[2 * i], e1 = err[2 * i + 1];
                double e2 = e0 * e0 + e1 * e1;
                double aux = dsqrReci * e2 + 1.0;
                double rho1 = 1.0 / aux;
                double wr0 = -e0 * rho1, wr1 = -e1 * rho1;  // omega_r *= rho[1]
                for (int a = 0; a < 6; a++) {
                    b[a] += J[0][a] * wr0 + J[1][a] * wr1;
                    for (int c = 0; c < 6; c++) H[a][c] += (J[0][a] * rho1) * J[0][c] + (J[1][a] * rho1) * J[1][c];
                }
            }
            if (iter == 0) {
                double maxDiag = 0;
                for (int j = 0; j < 6; j++) maxDiag = std::max(std::fabs(H[j][j]), maxDiag);
                lambda = 1e-5 * maxDiag;
                ni = 2;
            }
            double rho = 0;
            int qmax = 0;
            do {
                SBACam backup = cam;  // push()
                double Hl[6][6];
                for (int a = 0; a < 6; a++)
                    for (int c = 0; c < 6; c++) Hl[a][c] = H[a][c] + (a == c ? lambda : 0.0);
                double dx[6] = {0, 0, 0, 0, 0, 0};
                bool ok2 = solve6_spd(Hl, b, dx);
                cam.update(dx);
                compute_active_errors(cam);
                tempChi = active_robust_chi2();
                if (!ok2) tempChi = std::numeric_limits<double>::max();
                rho = (currentChi - tempChi);
                double scale = 0;
                for (int j = 0; j < 6; j++) scale += dx[j] * (lambda * dx[j] + b[j]);
                scale += 1e-3;
                rho /= scale;
                if (trace) {
                    trace->push_back(lambda);
                    trace->push_back(currentChi);
                    trace->push_back(tempChi);
                    trace->push_back(rho);
                }
                res.trials++;
                if (!(rho > 0 && std::isfinite(tempChi))) res.rejections++;
                if (rho > 0 && std::isfinite(tempChi)) {
                    double alpha = 1. - std::pow((2 * rho - 1), 3);
                    alpha = (std::min)(alpha, 2.0 / 3.0);
                    double scaleFactor = (std::max)(1.0 / 3.0, alpha);
                    lambda *= scaleFactor;
                    ni = 2;
                    currentChi = tempChi;
                } else {
                    lambda *= ni;
                    ni *= 2;
                    cam = backup;  // pop(): estimate restored, edge errors stay those of the rejected trial
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0) ok = false;  // Terminate
            if (!ok) res.terminates++;
        }
        // chi2 gate on the errors of the last computeActiveErrors() (pnp_solver.cpp:109-116)
        for (int k = 0; k < n; k++) {
            double chi2 = err[2 * k] * err[2 * k] + err[2 * k + 1] * err[2 * k + 1];
            if (level[k] == 0) {
                const double m = std::fabs(chi2 - kReprojTh2);
                res.min_margin = std::min(res.min_margin, m);
                if (m < kGateMargin) res.borderline++;
            }
            if (chi2 > kReprojTh2) {
                level[k] = 1;
                marks[k] = 0;
            }
        }
    }
    res.pose.q = cam.r;
    res.pose.p = cam.t;
    for (int k = 0; k < n; k++) res.inliers += marks[k];
    if (inlier_marks) *inlier_marks = marks;
    g_last_pnp_err = err;
    g_last_pnp = res;
    return res;
}

// ---------------------------------------------------------------------------------------------
// motion model -- lvt/src/lvt_motion_model.cpp:34-65
// ---------------------------------------------------------------------------------------------
struct MotionModel {
    Quat last_q, ang_vel;
    V3 last_p, lin_vel;
    void reset() {
        last_q = Quat{};
        ang_vel = Quat{};
        last_p = V3{};
        lin_vel = V3{};
    }
    Pose predict(const Pose &cur) {
        V3 nv{cur.p.x - last_p.x, cur.p.y - last_p.y, cur.p.z - last_p.z};
        nv = V3{(nv.x + lin_vel.x) * 0.5, (nv.y + lin_vel.y) * 0.5, (nv.z + lin_vel.z) * 0.5};
        Quat cq = cur.q;
        Quat diff = qmul(cq, qinverse(last_q));
        Quat nav = qnormalized(qslerp(diff, 0.5, ang_vel));
        last_q = cq;
        ang_vel = nav;
        last_p = cur.p;
        lin_vel = nv;
        Pose out;
        out.p = V3{last_p.x + lin_vel.x, last_p.y + lin_vel.y, last_p.z + lin_vel.z};
        out.q = qnormalized(qmul(cq, nav));
        return out;
    }
};

// ---------------------------------------------------------------------------------------------
// local map -- lvt/src/lvt_local_map.{h,cpp}
// ---------------------------------------------------------------------------------------------
struct MapPoint {
    uint8_t desc[32];
    V3 pos;
    int counter = 0, age = 0, match_idx = -1;
};

struct System {
    lvto_params prm;
    int sensor = 1;
    int state = 1;  // 1 NOT_INITIALIZED, 2 TRACKING, 3 LOST (lvt_system.h:45-50)
    int frame_number = 0;
    int n_threads = 2;
    std::vector<Rect> rects;
    Bounds bounds;
    MotionModel motion;
    std::vector<MapPoint> map, staged;
    Pose last_pose, predicted_pose;
    std::deque<int> last_matches;
    // introspection of the last frame
    int counts[LVTO_C__COUNT];
    FeatureStruct L, R;
    std::vector<int> dbg_match_feat;
    std::vector<V3> dbg_match_pos;
    std::vector<int> dbg_row_pairs;

    void init(const lvto_params &p, int sensor_type) {
        prm = p;
        sensor = sensor_type;
        rects = make_grid(p.img_width, p.img_height, p.detection_cell_size);
        // bounds -- local_map.cpp:84-123 (per instance here)
        if (std::fabs(p.k1) < 1e-5) {
            bounds = Bounds{0.0f, (float)p.img_width, 0.0f, (float)p.img_height};
        } else {
            float x[4], y[4];
            undistort_point(p, 0.0f, 0.0f, x[0], y[0]);
            undistort_point(p, (float)p.img_width, 0.0f, x[1], y[1]);
            undistort_point(p, 0.0f, (float)p.img_height, x[2], y[2]);
            undistort_point(p, (float)p.img_width, (float)p.img_height, x[3], y[3]);
            bounds.min_x = std::min(x[0], x[2]);
            bounds.max_x = std::max(x[1], x[3]);
            bounds.min_y = std::min(y[0], y[1]);
            bounds.max_y = std::max(y[2], y[3]);
        }
        reset();
    }
    void reset() {  // lvt_system.cpp:44-68
        map.clear();
        staged.clear();
        motion.reset();
        last_pose = Pose{};
        frame_number = 0;
        last_matches = std::deque<int>(3, std::numeric_limits<int>::max());
        state = 1;
        std::memset(counts, 0, sizeof(counts));
    }

    // perform_compute_features -- handler.cpp:156-176
    void compute_features_one(const uint8_t *img, int rows, int cols, FeatureStruct *out, int *retry) {
        std::vector<KeyPoint> kps;
        detect_with_retry(img, cols, rects, prm, kps, retry);
        std::vector<uint8_t> desc;
        brief_compute(img, rows, cols, cols, kps, desc);
        *out = FeatureStruct();
        out->init(rows, cols, kps, desc, prm.tracking_radius, kHashCell, kRowRadius, prm.triangulation_ratio_test_threshold,
                  prm.tracking_ratio_test_threshold, prm.descriptor_matching_threshold);
    }
    // handler.cpp:178-194
    void compute_descriptors_only_one(const uint8_t *img, int rows, int cols, const double *c, int nc, FeatureStruct *out) {
        std::vector<KeyPoint> kps;
        for (int i = 0; i < nc; i++) kps.push_back(KeyPoint{(float)c[2 * i], (float)c[2 * i + 1], 0.0f});
        std::vector<uint8_t> desc;
        brief_compute(img, rows, cols, cols, kps, desc);
        *out = FeatureStruct();
        out->init(rows, cols, kps, desc, prm.tracking_radius, kHashCell, kRowRadius, prm.triangulation_ratio_test_threshold,
                  prm.tracking_ratio_test_threshold, prm.descriptor_matching_threshold);
    }
    // compute_features_rgbd -- handler.cpp:227-300
    void compute_features_rgbd(const uint8_t *gray, const float *depth, int rows, int cols, FeatureStruct *out, int *retry) {
        std::vector<KeyPoint> kps;
        detect_with_retry(gray, cols, rects, prm, kps, retry);
        std::vector<uint8_t> desc;
        brief_compute(gray, rows, cols, cols, kps, desc);
        std::vector<float> depths;
        std::vector<KeyPoint> fk;
        std::vector<uint8_t> fd;
        const bool undist = std::fabs(prm.k1) > 1e-5;
        const float kc = (float)kHashCell;
        const int ccx = (int)std::ceil(cols / kc), ccy = (int)std::ceil(rows / kc);
        for (size_t i = 0; i < kps.size(); i++) {
            const float d = depth[(size_t)((int)kps[i].y) * cols + (int)kps[i].x];
            if (d >= prm.near_plane_distance && d <= prm.far_plane_distance) {
                KeyPoint k = kps[i];
                if (undist) {
                    undistort_point(prm, k.x, k.y, k.x, k.y);
                    int hy = (int)std::floor(k.y / kc), hx = (int)std::floor(k.x / kc);
                    if (hy < 0 || hy >= ccy || hx < 0 || hx >= ccx) continue;  // drop instead of OOB
                }
                depths.push_back(d);
                fk.push_back(k);
                fd.insert(fd.end(), desc.begin() + i * 32, desc.begin() + (i + 1) * 32);
            }
        }
        *out = FeatureStruct();
        out->init(rows, cols, fk, fd, prm.tracking_radius, kHashCell, kRowRadius, prm.triangulation_ratio_test_threshold,
                  prm.tracking_ratio_test_threshold, prm.descriptor_matching_threshold, &depths);
    }

    // find_matches -- local_map.cpp:136-229
    int find_matches(const Pose &cam_pose, FeatureStruct *ls, std::vector<V3> *out_pts, std::vector<int> *out_feat) {
        const M34 cml = world_to_camera(cam_pose);
        int matches_count = 0;
        const int M = (int)map.size();
        std::vector<int> matches(M, -2);
        std::vector<double> pu(M), pv(M);
        for (int i = 0; i < M; i++) {
            double u, v;
            if (!is_point_visible(map[i].pos, cml, prm, bounds, u, v)) {
                map[i].counter += 1;
                matches[i] = -2;
                continue;
            }
            pu[i] = u;
            pv[i] = v;
            float d1, d2;
            int mi = ls->find_match_index(u, v, map[i].desc, &d1, &d2);
            matches[i] = mi;
            if (mi != -1) {
                matches_count++;
                ls->matched[mi] = 1;
            }
        }
        counts[LVTO_C_SECOND_PASS] = 0;
        if (matches_count < kNMatchesTh) {
            counts[LVTO_C_SECOND_PASS] = 1;
            matches_count = 0;
            std::fill(ls->matched.begin(), ls->matched.end(), 0);
            int orig = ls->tracking_radius;
            ls->tracking_radius = 2 * orig;
            for (int i = 0; i < M; i++) {
                if (matches[i] == -2) continue;
                float d1, d2;
                int mi = ls->find_match_index(pu[i], pv[i], map[i].desc, &d1, &d2);
                matches[i] = mi;
                if (mi != -1) {
                    matches_count++;
                    ls->matched[mi] = 1;
                }
            }
            ls->tracking_radius = orig;
        }
        for (int i = 0; i < M; i++) {
            map[i].match_idx = matches[i];
            if (matches[i] == -2) continue;
            if (matches[i] == -1) {
                map[i].counter += 1;
                continue;
            }
            map[i].age += 1;
            out_pts->push_back(map[i].pos);
            out_feat->push_back(matches[i]);
        }
        return matches_count;
    }

    // handler.cpp:302-323
    void row_match(FeatureStruct *fl, FeatureStruct *fr, std::vector<int> *pairs) {
        for (int i = 0, n = fl->count(); i < n; i++) {
            if (fl->matched[i]) continue;
            const int mi = fr->row_match(fl->kps[i].x, fl->kps[i].y, fl->descriptor(i));
            if (mi != -1) {
                pairs->push_back(i);
                pairs->push_back(mi);
                fl->matched[i] = 1;
                fr->matched[mi] = 1;
            }
        }
    }

    bool triangulate_pair(const M34 &cml, const M34 &cmr, float u1x_f, float u1y_f, float u2x_f, float u2y_f, V3 &out) {
        const double cx = prm.cx, cy = prm.cy;
        const double inv_fx = 1.0 / prm.fx, inv_fy = 1.0 / prm.fy;
        double u1_x = (u1x_f - cx) * inv_fx, u1_y = (u1y_f - cy) * inv_fy;
        double u2_x = (u2x_f - cx) * inv_fx, u2_y = (u2y_f - cy) * inv_fy;
        double A[4][3], rhs[4];
        for (int j = 0; j < 3; j++) {
            A[0][j] = u1_x * cml.m[2][j] - cml.m[0][j];
            A[1][j] = u1_y * cml.m[2][j] - cml.m[1][j];
            A[2][j] = u2_x * cmr.m[2][j] - cmr.m[0][j];
            A[3][j] = u2_y * cmr.m[2][j] - cmr.m[1][j];
        }
        rhs[0] = -(u1_x * cml.m[2][3] - cml.m[0][3]);
        rhs[1] = -(u1_y * cml.m[2][3] - cml.m[1][3]);
        rhs[2] = -(u2_x * cmr.m[2][3] - cmr.m[0][3]);
        rhs[3] = -(u2_y * cmr.m[2][3] - cmr.m[1][3]);
        double x[3];
        if (!ls_solve_4x3(A, rhs, x)) return false;
        V3 wp{x[0], x[1], x[2]};
        double ul, vl, ur, vr;
        if (!is_point_visible(wp, cml, prm, bounds, ul, vl) || !is_point_visible(wp, cmr, prm, bounds, ur, vr)) return false;
        {
            double ex = ul - u1x_f, ey = vl - u1y_f;
            if ((ex * ex + ey * ey) > kReprojTh2) return false;
        }
        {
            double ex = ur - u2x_f, ey = vr - u2y_f;
            if ((ex * ex + ey * ey) > kReprojTh2) return false;
        }
        out = wp;
        return true;
    }

    // triangulate -- local_map.cpp:258-329
    void triangulate(const Pose &cam_pose, FeatureStruct *ls, FeatureStruct *rs, std::vector<MapPoint> *out) {
        std::vector<int> pairs;
        row_match(ls, rs, &pairs);
        dbg_row_pairs = pairs;
        counts[LVTO_C_N_ROW_MATCHES] = (int)pairs.size() / 2;
        if (pairs.empty()) return;
        const Pose right = right_camera_pose(cam_pose, prm.baseline);
        const M34 cml = world_to_camera(cam_pose), cmr = world_to_camera(right);
        for (size_t i = 0; i < pairs.size(); i += 2) {
            const KeyPoint &a = ls->kps[pairs[i]], &b = rs->kps[pairs[i + 1]];
            V3 wp;
            if (!triangulate_pair(cml, cmr, a.x, a.y, b.x, b.y, wp)) continue;
            MapPoint mp;
            mp.pos = wp;
            std::memcpy(mp.desc, ls->descriptor(pairs[i]), 32);
            out->push_back(mp);
        }
    }
    // triangulate_rgbd -- local_map.cpp:231-256
    void triangulate_rgbd(const Pose &cam_pose, FeatureStruct *s, std::vector<MapPoint> *out) {
        const float inv_fx = 1.0f / prm.fx, inv_fy = 1.0f / prm.fy;
        M33 R = qtoR(cam_pose.q);
        for (int i = 0, n = s->count(); i < n; i++) {
            const float u = s->kps[i].x, v = s->kps[i].y, z = s->depths[i];
            const float x = (u - prm.cx) * z * inv_fx, y = (v - prm.cy) * z * inv_fy;
            MapPoint mp;
            mp.pos.x = ((R.m[0][0] * x + R.m[0][1] * y) + R.m[0][2] * z) + cam_pose.p.x * 1.0;
            mp.pos.y = ((R.m[1][0] * x + R.m[1][1] * y) + R.m[1][2] * z) + cam_pose.p.y * 1.0;
            mp.pos.z = ((R.m[2][0] * x + R.m[2][1] * y) + R.m[2][2] * z) + cam_pose.p.z * 1.0;
            std::memcpy(mp.desc, s->descriptor(i), 32);
            out->push_back(mp);
        }
    }
    // update_with_new_triangulation -- local_map.cpp:331-353
    void update_with_new_triangulation(const Pose &cam_pose, FeatureStruct *ls, FeatureStruct *rs, bool dont_stage) {
        std::vector<MapPoint> nt;
        if (!ls->depths.empty())
            triangulate_rgbd(cam_pose, ls, &nt);
        else
            triangulate(cam_pose, ls, rs, &nt);
        counts[LVTO_C_TRIANGULATED] = 1;
        counts[LVTO_C_N_TRIANGULATED] = (int)nt.size();
        if (dont_stage || prm.staged_threshold == 0 || (int)map.size() < kNMapPoints)
            map.insert(map.end(), nt.begin(), nt.end());
        else
            staged.insert(staged.end(), nt.begin(), nt.end());
    }
    // update_staged_map_points -- local_map.cpp:355-391
    void update_staged(const Pose &cam_pose, FeatureStruct *ls) {
        const M34 cml = world_to_camera(cam_pose);
        std::vector<uint8_t> del(staged.size(), 0);
        for (int i = 0, n = (int)staged.size(); i < n; i++) {
            MapPoint *mp = &staged[i];
            double u, v;
            float d1, d2;
            int mi = -1;
            if (!is_point_visible(mp->pos, cml, prm, bounds, u, v) || (mi = ls->find_match_index(u, v, mp->desc, &d1, &d2)) == -1) {
                del[i] = 1;
                counts[LVTO_C_N_STAGED_ERASED]++;
                continue;
            }
            ls->matched[mi] = 1;
            mp->counter += 1;
            if (mp->counter == prm.staged_threshold || (int)map.size() < kNMapPoints) {
                map.push_back(staged[i]);
                del[i] = 1;
                counts[LVTO_C_N_STAGED_PROMOTED]++;
            }
        }
        std::vector<MapPoint> keep;
        for (size_t i = 0; i < staged.size(); i++)
            if (!del[i]) keep.push_back(staged[i]);
        staged.swap(keep);
    }
    // clean_untracked_points -- local_map.cpp:393-413
    void clean_untracked(FeatureStruct *ls) {
        const int th = prm.untracked_threshold;
        std::vector<MapPoint> keep;
        keep.reserve(map.size());
        for (auto &mp : map) {
            if (mp.counter >= th) {
                if (mp.match_idx >= 0) ls->matched[mp.match_idx] = 0;
                counts[LVTO_C_N_CULLED]++;
            } else
                keep.push_back(mp);
        }
        keep.swap(map);
    }

    // triangulation policies -- lvt_system.cpp:308-334
    bool need_new_triangulation() {
        if (prm.triangulation_policy == 2) return true;
        if (prm.triangulation_policy == 3) return (int)map.size() < 1000;
        const float ratio = 0.99;
        for (int i = 2; i > 0; --i)
            if (float(last_matches[i]) > ratio * float(last_matches[i - 1])) return false;
        return true;
    }

    // perform_tracking -- lvt_system.cpp:252-306
    Pose perform_tracking(const Pose &estimated, FeatureStruct *ls, FeatureStruct *rs, bool *is_tracking) {
        std::vector<V3> map_points;
        std::vector<int> matches_left;
        counts[LVTO_C_MAP_SIZE_AT_MATCH] = (int)map.size();
        find_matches(estimated, ls, &map_points, &matches_left);
        dbg_match_feat = matches_left;
        dbg_match_pos = map_points;
        const int matches_count = (int)map_points.size();
        counts[LVTO_C_N_MATCHES] = matches_count;
        if (matches_count < prm.min_num_matches_for_tracking) {
            *is_tracking = false;
            return last_pose;
        }
        last_matches.push_back(matches_count);
        last_matches.pop_front();
        std::vector<float> obs(2 * (size_t)matches_count);
        for (int i = 0; i < matches_count; i++) {
            obs[2 * i] = ls->kps[matches_left[i]].x;
            obs[2 * i + 1] = ls->kps[matches_left[i]].y;
        }
        PnpResult pr = pnp_compute_pose(prm, estimated, map_points, obs, nullptr, nullptr);
        counts[LVTO_C_PNP_ITERS] = pr.solve_calls;
        counts[LVTO_C_PNP_INLIERS] = pr.inliers;
        counts[LVTO_C_PNP_BORDERLINE] = pr.borderline;
        counts[LVTO_C_PNP_TRIALS] = pr.trials;
        counts[LVTO_C_PNP_REJECTIONS] = pr.rejections;
        counts[LVTO_C_PNP_TERMINATES] = pr.terminates;
        const Pose optimized = pr.pose;
        clean_untracked(ls);
        if (prm.staged_threshold > 0) update_staged(optimized, ls);
        if (need_new_triangulation()) update_with_new_triangulation(optimized, ls, rs, false);
        *is_tracking = true;
        return optimized;
    }

    Pose finish_track(FeatureStruct &ls, FeatureStruct &rs) {
        counts[LVTO_C_N_LEFT] = ls.count();
        counts[LVTO_C_N_RIGHT] = rs.count();
        Pose result;
        if (state == 1) {
            Pose identity;
            update_with_new_triangulation(identity, &ls, &rs, true);
            state = 2;
            last_matches[0] = (int)map.size();
            result = identity;
        } else {
            bool is_tracking = false;
            predicted_pose = motion.predict(last_pose);
            Pose computed = perform_tracking(predicted_pose, &ls, &rs, &is_tracking);
            if (!is_tracking) {
                state = 3;
                result = last_pose;
            } else {
                last_pose = computed;
                result = computed;
            }
        }
        counts[LVTO_C_MAP_SIZE] = (int)map.size();
        counts[LVTO_C_STAGED_SIZE] = (int)staged.size();
        L = std::move(ls);
        R = std::move(rs);
        return result;
    }

    void begin_frame() {
        std::memset(counts, 0, sizeof(counts));
        counts[LVTO_C_FRAME] = frame_number;
        dbg_match_feat.clear();
        dbg_match_pos.clear();
        dbg_row_pairs.clear();
        frame_number++;
    }

    // track -- lvt_system.cpp:157-207
    // LOST (lvt_system.cpp:161-166): nothing is detected or tracked; the introspection shows no features and the map as it is
    Pose lost_frame() {
        L = FeatureStruct();
        R = FeatureStruct();
        counts[LVTO_C_MAP_SIZE] = (int)map.size();
        counts[LVTO_C_STAGED_SIZE] = (int)staged.size();
        return last_pose;
    }
    Pose track(const uint8_t *img1, const void *img2, int rows, int cols) {
        begin_frame();
        if (state == 3) return lost_frame();
        FeatureStruct ls, rs;
        if (sensor == 1) {
            int r0 = 0, r1 = 0;
            if (n_threads >= 2) {
                std::thread th([&] { compute_features_one((const uint8_t *)img2, rows, cols, &rs, &r1); });
                compute_features_one(img1, rows, cols, &ls, &r0);
                th.join();
            } else {
                compute_features_one(img1, rows, cols, &ls, &r0);
                compute_features_one((const uint8_t *)img2, rows, cols, &rs, &r1);
            }
            counts[LVTO_C_RETRY_LEFT] = r0;
            counts[LVTO_C_RETRY_RIGHT] = r1;
        } else {
            int r0 = 0;
            compute_features_rgbd(img1, (const float *)img2, rows, cols, &ls, &r0);
            counts[LVTO_C_RETRY_LEFT] = r0;
        }
        return finish_track(ls, rs);
    }
    // track_with_external_corners -- lvt_system.cpp:209-250
    Pose track_ext(const uint8_t *l, const uint8_t *r, int rows, int cols, const double *cl, int ncl, const double *cr, int ncr) {
        begin_frame();
        if (state == 3) return lost_frame();
        FeatureStruct ls, rs;
        compute_descriptors_only_one(l, rows, cols, cl, ncl, &ls);
        compute_descriptors_only_one(r, rows, cols, cr, ncr, &rs);
        return finish_track(ls, rs);
    }
};

void pose_out(const Pose &p, double R[9], double t[3]) {
    M33 m = qtoR(p.q);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[3 * i + j] = m.m[i][j];
    t[0] = p.p.x;
    t[1] = p.p.y;
    t[2] = p.p.z;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

void lvto_default_params(lvto_params *p) {  // lvt_parameters.cpp:29-52
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

lvto_handle lvto_create(const lvto_params *p, int sensor_type) {
    if (sensor_type != 1 && sensor_type != 2) return nullptr;
    if (p->img_width <= 0 || p->img_height <= 0 || p->detection_cell_size <= 0) return nullptr;
    System *s = new System();
    s->init(*p, sensor_type);
    return s;
}
void lvto_destroy(lvto_handle h) { delete static_cast<System *>(h); }
void lvto_reset(lvto_handle h) { static_cast<System *>(h)->reset(); }
void lvto_set_threads(lvto_handle h, int n) { static_cast<System *>(h)->n_threads = n; }

void lvto_track(lvto_handle h, const uint8_t *left, const uint8_t *right, int rows, int cols, double R[9], double t[3]) {
    pose_out(static_cast<System *>(h)->track(left, right, rows, cols), R, t);
}
void lvto_track_rgbd(lvto_handle h, const uint8_t *gray, const float *depth, int rows, int cols, double R[9], double t[3]) {
    pose_out(static_cast<System *>(h)->track(gray, depth, rows, cols), R, t);
}
void lvto_track_with_external_corners(lvto_handle h, const uint8_t *left, const uint8_t *right, int rows, int cols,
                                      const double *cl, int ncl, const double *cr, int ncr, double R[9], double t[3]) {
    pose_out(static_cast<System *>(h)->track_ext(left, right, rows, cols, cl, ncl, cr, ncr), R, t);
}
int lvto_get_status(lvto_handle h) { return static_cast<System *>(h)->state; }

void lvto_get_counts(lvto_handle h, int out[LVTO_C__COUNT]) {
    std::memcpy(out, static_cast<System *>(h)->counts, sizeof(int) * LVTO_C__COUNT);
}
int lvto_get_features(lvto_handle h, int eye, float *xy, float *resp, uint8_t *desc, int cap) {
    System *s = static_cast<System *>(h);
    const FeatureStruct &f = eye ? s->R : s->L;
    int n = std::min(cap, f.count());
    for (int i = 0; i < n; i++) {
        if (xy) {
            xy[2 * i] = f.kps[i].x;
            xy[2 * i + 1] = f.kps[i].y;
        }
        if (resp) resp[i] = f.kps[i].response;
    }
    if (desc && n) std::memcpy(desc, f.desc.data(), (size_t)n * 32);
    return f.count();
}
int lvto_get_matches(lvto_handle h, int *feat_idx, double *xyz, int cap) {
    System *s = static_cast<System *>(h);
    int n = std::min(cap, (int)s->dbg_match_feat.size());
    for (int i = 0; i < n; i++) {
        if (feat_idx) feat_idx[i] = s->dbg_match_feat[i];
        if (xyz) {
            xyz[3 * i] = s->dbg_match_pos[i].x;
            xyz[3 * i + 1] = s->dbg_match_pos[i].y;
            xyz[3 * i + 2] = s->dbg_match_pos[i].z;
        }
    }
    return (int)s->dbg_match_feat.size();
}
int lvto_get_row_matches(lvto_handle h, int *pairs, int cap) {
    System *s = static_cast<System *>(h);
    int n = (int)s->dbg_row_pairs.size() / 2;
    for (int i = 0; i < std::min(cap, n) * 2; i++) pairs[i] = s->dbg_row_pairs[i];
    return n;
}
static int get_points(const std::vector<MapPoint> &v, double *xyz, int *counter, int *age, uint8_t *desc, int cap) {
    int n = std::min(cap, (int)v.size());
    for (int i = 0; i < n; i++) {
        if (xyz) {
            xyz[3 * i] = v[i].pos.x;
            xyz[3 * i + 1] = v[i].pos.y;
            xyz[3 * i + 2] = v[i].pos.z;
        }
        if (counter) counter[i] = v[i].counter;
        if (age) age[i] = v[i].age;
        if (desc) std::memcpy(desc + (size_t)i * 32, v[i].desc, 32);
    }
    return (int)v.size();
}
int lvto_get_map(lvto_handle h, double *xyz, int *counter, int *age, uint8_t *desc, int cap) {
    return get_points(static_cast<System *>(h)->map, xyz, counter, age, desc, cap);
}
int lvto_get_staged(lvto_handle h, double *xyz, int *counter, uint8_t *desc, int cap) {
    return get_points(static_cast<System *>(h)->staged, xyz, counter, nullptr, desc, cap);
}
void lvto_get_pose(lvto_handle h, double q[4], double p[3]) {
    const Pose &ps = static_cast<System *>(h)->last_pose;
    q[0] = ps.q.w, q[1] = ps.q.x, q[2] = ps.q.y, q[3] = ps.q.z;
    p[0] = ps.p.x, p[1] = ps.p.y, p[2] = ps.p.z;
}
void lvto_get_predicted_pose(lvto_handle h, double q[4], double p[3]) {
    const Pose &ps = static_cast<System *>(h)->predicted_pose;
    q[0] = ps.q.w, q[1] = ps.q.x, q[2] = ps.q.y, q[3] = ps.q.z;
    p[0] = ps.p.x, p[1] = ps.p.y, p[2] = ps.p.z;
}

// ---- primitives ----
void lvto_agast_score_map(const uint8_t *img, int rows, int cols, int stride, int16_t *out) {
    int off[16];
    for (int i = 0; i < 16; i++) off[i] = kCircle[i][0] + kCircle[i][1] * stride;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            if (y < 3 || y > rows - 4 || x < 3 || x > cols - 4)
                out[(size_t)y * cols + x] = -1;
            else
                out[(size_t)y * cols + x] = (int16_t)oast9_score(img + (size_t)y * stride + x, off);
        }
}
int lvto_agast_detect(const uint8_t *img, int rows, int cols, int stride, int threshold, int nonmax, float *xyr, int cap) {
    std::vector<KeyPoint> k;
    agast_detect_roi(img, rows, cols, stride, threshold, nonmax != 0, k);
    for (int i = 0; i < std::min(cap, (int)k.size()); i++) {
        xyr[3 * i] = k[i].x;
        xyr[3 * i + 1] = k[i].y;
        xyr[3 * i + 2] = k[i].response;
    }
    return (int)k.size();
}
int lvto_anms(float *xyr, int n, int num_to_keep, float tx, float ty) {
    std::vector<KeyPoint> k(n);
    for (int i = 0; i < n; i++) k[i] = KeyPoint{xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2]};
    anms(k, num_to_keep, tx, ty);
    for (size_t i = 0; i < k.size(); i++) {
        xyr[3 * i] = k[i].x;
        xyr[3 * i + 1] = k[i].y;
        xyr[3 * i + 2] = k[i].response;
    }
    return (int)k.size();
}
void lvto_sort_by_response(float *xyr, int n) {
    std::vector<KeyPoint> k(n);
    for (int i = 0; i < n; i++) k[i] = KeyPoint{xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2]};
    std::sort(k.begin(), k.end(), [](const KeyPoint &l, const KeyPoint &r) { return l.response > r.response; });
    for (int i = 0; i < n; i++) {
        xyr[3 * i] = k[i].x;
        xyr[3 * i + 1] = k[i].y;
        xyr[3 * i + 2] = k[i].response;
    }
}
int lvto_detect_grid(const uint8_t *img, int rows, int cols, const lvto_params *p, float *xyr, int cap, int *retry_used) {
    lvto_params q = *p;
    q.img_width = cols;
    q.img_height = rows;
    std::vector<Rect> rects = make_grid(cols, rows, q.detection_cell_size);
    std::vector<KeyPoint> k;
    detect_with_retry(img, cols, rects, q, k, retry_used);
    for (int i = 0; i < std::min(cap, (int)k.size()); i++) {
        xyr[3 * i] = k[i].x;
        xyr[3 * i + 1] = k[i].y;
        xyr[3 * i + 2] = k[i].response;
    }
    return (int)k.size();
}
int lvto_brief(const uint8_t *img, int rows, int cols, const float *xy, int n, int *kept, uint8_t *desc) {
    std::vector<KeyPoint> k(n);
    for (int i = 0; i < n; i++) k[i] = KeyPoint{xy[2 * i], xy[2 * i + 1], (float)i};  // response carries the index
    std::vector<uint8_t> d;
    brief_compute(img, rows, cols, cols, k, d);
    for (size_t i = 0; i < k.size(); i++)
        if (kept) kept[i] = (int)k[i].response;
    if (!d.empty()) std::memcpy(desc, d.data(), d.size());
    return (int)k.size();
}
int lvto_compute_features(const uint8_t *img, int rows, int cols, const lvto_params *p, float *xy, float *resp, uint8_t *desc,
                          int cap, int *retry_used) {
    System s;
    lvto_params q = *p;
    q.img_width = cols;
    q.img_height = rows;
    s.init(q, 1);
    FeatureStruct f;
    s.compute_features_one(img, rows, cols, &f, retry_used);
    int n = std::min(cap, f.count());
    for (int i = 0; i < n; i++) {
        xy[2 * i] = f.kps[i].x;
        xy[2 * i + 1] = f.kps[i].y;
        if (resp) resp[i] = f.kps[i].response;
    }
    if (n) std::memcpy(desc, f.desc.data(), (size_t)n * 32);
    return f.count();
}
void lvto_hamming_top2(const uint8_t *query, const uint8_t *train, int n, const uint8_t *mask, int out[4]) {
    Top2 t;
    for (int i = 0; i < n; i++)
        if (!mask || mask[i]) t.offer(i, hamming32(query, train + (size_t)i * 32));
    out[0] = t.i1;
    out[1] = t.d1;
    out[2] = t.i2;
    out[3] = t.d2;
}
int lvto_pnp(const lvto_params *p, const double q_in[4], const double p_in[3], const double *pts, const float *obs, int n,
             double q_out[4], double p_out[3], int *inlier_marks, double *trace, int trace_cap, int *solve_calls) {
    Pose prior;
    prior.q = Quat{q_in[0], q_in[1], q_in[2], q_in[3]};
    prior.p = V3{p_in[0], p_in[1], p_in[2]};
    std::vector<V3> P(n);
    for (int i = 0; i < n; i++) P[i] = V3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    std::vector<float> O(obs, obs + 2 * (size_t)n);
    std::vector<int> marks;
    std::vector<double> tr;
    PnpResult r = pnp_compute_pose(*p, prior, P, O, &marks, &tr);
    q_out[0] = r.pose.q.w, q_out[1] = r.pose.q.x, q_out[2] = r.pose.q.y, q_out[3] = r.pose.q.z;
    p_out[0] = r.pose.p.x, p_out[1] = r.pose.p.y, p_out[2] = r.pose.p.z;
    if (inlier_marks)
        for (int i = 0; i < n; i++) inlier_marks[i] = marks[i];
    if (solve_calls) *solve_calls = r.solve_calls;
    int rows = (int)tr.size() / 4;
    if (trace)
        for (int i = 0; i < std::min(rows, trace_cap) * 4; i++) trace[i] = tr[i];
    return rows;
}
int lvto_pnp_last_gate(double *err_out, int n, double *min_margin) {
    for (int i = 0; i < std::min(2 * n, (int)g_last_pnp_err.size()); i++) err_out[i] = g_last_pnp_err[i];
    if (min_margin) *min_margin = g_last_pnp.min_margin;
    return g_last_pnp.borderline;
}
void lvto_pnp_last_stats(int out[3]) {
    out[0] = g_last_pnp.trials, out[1] = g_last_pnp.rejections, out[2] = g_last_pnp.terminates;
}
int lvto_triangulate_one(const lvto_params *p, const double q[4], const double pos[3], float ulx, float uly, float urx, float ury,
                         double out_xyz[3]) {
    System s;
    s.init(*p, 1);
    Pose cp;
    cp.q = Quat{q[0], q[1], q[2], q[3]};
    cp.p = V3{pos[0], pos[1], pos[2]};
    const Pose right = right_camera_pose(cp, p->baseline);
    V3 wp;
    if (!s.triangulate_pair(world_to_camera(cp), world_to_camera(right), ulx, uly, urx, ury, wp)) return 0;
    out_xyz[0] = wp.x, out_xyz[1] = wp.y, out_xyz[2] = wp.z;
    return 1;
}
void lvto_motion_predict(double st[14], const double q[4], const double p[3], double q_out[4], double p_out[3]) {
    MotionModel m;
    m.last_q = Quat{st[0], st[1], st[2], st[3]};
    m.ang_vel = Quat{st[4], st[5], st[6], st[7]};
    m.last_p = V3{st[8], st[9], st[10]};
    m.lin_vel = V3{st[11], st[12], st[13]};
    Pose cur;
    cur.q = Quat{q[0], q[1], q[2], q[3]};
    cur.p = V3{p[0], p[1], p[2]};
    Pose o = m.predict(cur);
    st[0] = m.last_q.w, st[1] = m.last_q.x, st[2] = m.last_q.y, st[3] = m.last_q.z;
    st[4] = m.ang_vel.w, st[5] = m.ang_vel.x, st[6] = m.ang_vel.y, st[7] = m.ang_vel.z;
    st[8] = m.last_p.x, st[9] = m.last_p.y, st[10] = m.last_p.z;
    st[11] = m.lin_vel.x, st[12] = m.lin_vel.y, st[13] = m.lin_vel.z;
    q_out[0] = o.q.w, q_out[1] = o.q.x, q_out[2] = o.q.y, q_out[3] = o.q.z;
    p_out[0] = o.p.x, p_out[1] = o.p.y, p_out[2] = o.p.z;
}


// ---- EuRoC pre-step: stereo rectification as the reference's example runs it through OpenCV ------------------------
// cv::initUndistortRectifyMap, scalar path of OpenCV 3.x imgproc/src/undistort.cpp: iR = (Pnew * R)^-1 (closed-form 3x3
// inverse of cv::invert), per row the homogeneous coordinate is ACCUMULATED column by column (_x += ir[0] ...), then the
// Brown-Conrady model with k4..k6 = s1..s4 = tau = 0, result narrowed to float.
void lvto_init_undistort_rectify_map(const double K[9], const double D[5], const double R[9], const double P[9], int w, int h, float *map1,
                                     float *map2) {
    double AR[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += P[3 * i + k] * R[3 * k + j];
            AR[3 * i + j] = s;
        }
    double ir[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    {
        const double *S = AR;
        double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
        if (d != 0.) {
            d = 1. / d;
            ir[0] = (S[4] * S[8] - S[5] * S[7]) * d;
            ir[1] = (S[2] * S[7] - S[1] * S[8]) * d;
            ir[2] = (S[1] * S[5] - S[2] * S[4]) * d;
            ir[3] = (S[5] * S[6] - S[3] * S[8]) * d;
            ir[4] = (S[0] * S[8] - S[2] * S[6]) * d;
            ir[5] = (S[2] * S[3] - S[0] * S[5]) * d;
            ir[6] = (S[3] * S[7] - S[4] * S[6]) * d;
            ir[7] = (S[1] * S[6] - S[0] * S[7]) * d;
            ir[8] = (S[0] * S[4] - S[1] * S[3]) * d;
        }
    }
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
    for (int i = 0; i < h; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double iw = 1. / _w, x = _x * iw, y = _y * iw;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((0 * r2 + 0) * r2 + 0) * r2);
            const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + 0 * r2 + 0 * r2 * r2);
            const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + 0 * r2 + 0 * r2 * r2);
            const double u = fx * 1. * xd + u0;  // invProj = 1 (no tilt)
            const double v = fy * 1. * yd + v0;
            map1[(size_t)i * w + j] = (float)u;
            map2[(size_t)i * w + j] = (float)v;
        }
    }
}

// cv::remap, 8UC1, INTER_LINEAR, BORDER_CONSTANT(0): coordinates rounded to 1/32 px (cvRound = round-half-even of x*32),
// weights from initInterTab2D: (32-fx)(32-fy)*32 ... as shorts, except the all-integer entry, whose 32768 saturates to
// 32767 and is repaired by +1 on the LAST weight (the table's sum fix-up scans from index ksize/2 = 1).
void lvto_remap_bilinear(const uint8_t *src, int sw, int sh, int sstep, const float *map1, const float *map2, int dw, int dh, uint8_t *dst) {
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            const int sxf = (int)lrintf(map1[(size_t)y * dw + x] * 32.f), syf = (int)lrintf(map2[(size_t)y * dw + x] * 32.f);
            auto sat16 = [](int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); };
            const int sx = sat16(sxf >> 5), sy = sat16(syf >> 5), ax = sxf & 31, ay = syf & 31;
            int w0 = (32 - ax) * (32 - ay) * 32, w1 = ax * (32 - ay) * 32, w2 = (32 - ax) * ay * 32, w3 = ax * ay * 32;
            if (ax == 0 && ay == 0) w0 = 32767, w3 = 1;
            int v0, v1, v2, v3;
            if ((unsigned)sx < (unsigned)std::max(sw - 1, 0) && (unsigned)sy < (unsigned)std::max(sh - 1, 0)) {
                const uint8_t *S = src + (size_t)sy * sstep + sx;
                v0 = S[0], v1 = S[1], v2 = S[sstep], v3 = S[sstep + 1];
            } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
                dst[(size_t)y * dw + x] = 0;
                continue;
            } else {
                const uint8_t *S0 = src + (size_t)sy * sstep, *S1 = src + (size_t)(sy + 1) * sstep;
                v0 = (sx >= 0 && sy >= 0) ? S0[sx] : 0;
                v1 = (sx + 1 < sw && sy >= 0) ? S0[sx + 1] : 0;
                v2 = (sx >= 0 && sy + 1 < sh) ? S1[sx] : 0;
                v3 = (sx + 1 < sw && sy + 1 < sh) ? S1[sx + 1] : 0;
            }
            const int acc = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
            const int r = (acc + (1 << 14)) >> 15;
            dst[(size_t)y * dw + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
}

}  // extern "C"
