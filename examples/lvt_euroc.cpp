// lvt_euroc -- EuRoC MAV stereo command line harness over the C-ABI (SURVEY 8(f) rows 1 + 2).
//
// Same argv, inputs and outputs as the reference's example binary (examples/euroc/euroc_example.cpp:49-175 there):
//     lvt_euroc <euroc_root_dir> <stamps_dir> <dataset_name> <config_file_name> [--max-frames N] [--out name.txt]
// reads <stamps_dir>/<dataset_name>.txt (one nanosecond stamp per line = the image file stem), the cam0 / cam1 PNGs below
// <root>/<dataset_name>/mav0/, rectifies both on the GPU with the calibration the reference hard-codes
// (lvt_amd_rectifier_*: cv::initUndistortRectifyMap + cv::remap INTER_LINEAR), tracks, maps the camera pose into the body
// frame (T_BS * T_cam) and writes <dataset_name>.txt in the TUM format "t x y z qx qy qz qw" (6 / 7 digits).
#include "../include/lvt_amd_ext.h"
#include "../include/lvt_c.h"
#include "image_io.h"

#include <chrono>
#include <iomanip>
#include <iostream>
#include <sstream>

using namespace lvt_io;

namespace {

// Eigen::Quaternion(Matrix3) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>): w x y z
void rotation_to_quaternion(const double m[3][3], double q[4]) {
    double t = m[0][0] + m[1][1] + m[2][2];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (m[2][1] - m[1][2]) * t;
        q[2] = (m[0][2] - m[2][0]) * t;
        q[3] = (m[1][0] - m[0][1]) * t;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[k][j] - m[j][k]) * t;
        q[1 + j] = (m[j][i] + m[i][j]) * t;
        q[1 + k] = (m[k][i] + m[i][k]) * t;
    }
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 5) {
        std::cout << "Usage ./lvt_euroc euroc_root_dir stamps_dir dataset_name config_file_name [--max-frames N] [--out name.txt]" << std::endl;
        return -1;
    }
    const std::string root_dir = argv[1], stamps_dir = argv[2], dataset_name = argv[3], config_file_name = argv[4];
    const std::string seq_dir = root_dir + "/" + dataset_name + "/mav0";
    std::string out_name = dataset_name + ".txt";
    long max_frames = -1;
    for (int i = 5; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--out") out_name = argv[i + 1];
        else if (k == "--max-frames") max_frames = std::atol(argv[i + 1]);
    }
    std::vector<std::string> titles;
    std::vector<double> time_stamps;
    {
        const std::string path = stamps_dir + "/" + dataset_name + ".txt";
        std::ifstream f(path);
        if (!f.is_open()) {
            std::cout << "Unable to open stamps files " << path << std::endl;
            return -1;
        }
        std::string line;
        while (std::getline(f, line)) {
            while (!line.empty() && std::isspace((unsigned char)line.back())) line.pop_back();
            if (line.empty()) continue;
            titles.push_back(line + ".png");
            time_stamps.push_back(std::atof(line.c_str()) / 1e9);
        }
    }
    lvt_amd_params params;
    if (!lvt_amd_params_from_file(config_file_name.c_str(), &params)) {
        std::cout << "Failed to initialize from " << config_file_name << std::endl;
        return -1;
    }
    // the calibration of the reference's example (euroc_example.cpp:95-119)
    const double kl[9] = {458.654, 0.0, 367.215, 0.0, 457.296, 248.375, 0.0, 0.0, 1.0};
    const double kr[9] = {457.587, 0.0, 379.999, 0.0, 456.134, 255.238, 0.0, 0.0, 1.0};
    const double pn[9] = {435.2046959714599, 0, 367.4517211914062, 0, 435.2046959714599, 252.2008514404297, 0, 0, 1};
    const double rl[9] = {0.999966347530033, -0.001422739138722922, 0.008079580483432283, 0.001365741834644127, 0.9999741760894847,
                          0.007055629199258132, -0.008089410156878961, -0.007044357138835809, 0.9999424675829176};
    const double rr[9] = {0.9999633526194376, -0.003625811871560086, 0.007755443660172947, 0.003680398547259526, 0.9999684752771629,
                          -0.007035845251224894, -0.007729688520722713, 0.007064130529506649, 0.999945173484644};
    const double dl[5] = {-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0};
    const double dr[5] = {-0.28368365, 0.07451284, -0.00010473, -3.555907e-05, 0.0};
    const int W = 752, H = 480;
    params.fx = 435.2046959714599f, params.fy = 435.2046959714599f, params.cx = 367.4517211914062f, params.cy = 252.2008514404297f;
    params.baseline = 0.110077842f;
    params.img_width = W, params.img_height = H;
    const double Tbs[4][4] = {{0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975},
                              {0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768},
                              {-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949},
                              {0.0, 0.0, 0.0, 1.0}};
    lvt_amd_rectifier rect_l = lvt_amd_rectifier_create(kl, dl, rl, pn, W, H), rect_r = lvt_amd_rectifier_create(kr, dr, rr, pn, W, H);
    lvt_handle vo = lvt_amd_create(&params, 1 /* STEREO */);
    if (!rect_l || !rect_r || !vo) {
        std::cout << "failed to create the tracker: " << lvt_amd_last_error(nullptr) << std::endl;
        return -1;
    }
    long frame_count = (long)titles.size();
    if (max_frames >= 0 && max_frames < frame_count) frame_count = max_frames;
    std::vector<double> poses;  // q (w x y z), p of the BODY per processed frame
    std::vector<unsigned char> rl_img((size_t)W * H), rr_img((size_t)W * H);
    double total_time = 0;
    long n = 0;
    for (long i = 0; i < frame_count; i++) {
        std::cout << "Frame number: " << i << "/" << frame_count << "\r" << std::flush;
        Gray left, right;
        std::string err;
        if (!load_image(seq_dir + "/cam0/data/" + titles[i], left, err) || !load_image(seq_dir + "/cam1/data/" + titles[i], right, err) || left.w != W ||
            left.h != H || right.w != W || right.h != H) {
            std::cout << "Failed to load image " << titles[i] << std::endl;
            break;
        }
        const auto t0 = std::chrono::steady_clock::now();
        if (lvt_amd_rectify(rect_l, left.px.data(), rl_img.data()) != 0 || lvt_amd_rectify(rect_r, right.px.data(), rr_img.data()) != 0) {
            std::cout << "rectification failed" << std::endl;
            break;
        }
        double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, t[3] = {0, 0, 0};
        lvt_track(vo, rl_img.data(), rr_img.data(), H, W, R, t);
        total_time += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double body[3][4];  // Tbs * [R t; 0 1]
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += Tbs[r][k] * (c < 3 ? R[k][c] : t[k]);
                body[r][c] = s + (c == 3 ? Tbs[r][3] : 0.0);
            }
        const double Rb[3][3] = {{body[0][0], body[0][1], body[0][2]}, {body[1][0], body[1][1], body[1][2]}, {body[2][0], body[2][1], body[2][2]}};
        double q[4];
        rotation_to_quaternion(Rb, q);
        for (int k = 0; k < 4; k++) poses.push_back(q[k]);
        for (int k = 0; k < 3; k++) poses.push_back(body[k][3]);
        n++;
        if (lvt_get_status(vo) == 3) break;  // LOST
    }
    std::ofstream file(out_name.c_str());
    file << std::fixed;
    for (long i = 0; i < (long)titles.size(); i++) {
        double q[4] = {1, 0, 0, 0}, p[3] = {0, 0, 0};
        if (i < n) {
            for (int k = 0; k < 4; k++) q[k] = poses[(size_t)i * 7 + k];
            for (int k = 0; k < 3; k++) p[k] = poses[(size_t)i * 7 + 4 + k];
        }
        file << std::setprecision(6) << time_stamps[i] << std::setprecision(7) << " " << p[0] << " " << p[1] << " " << p[2] << " " << q[1] << " " << q[2]
             << " " << q[3] << " " << q[0] << std::endl;
    }
    file.close();
    lvt_destroy(vo);
    lvt_amd_rectifier_destroy(rect_l);
    lvt_amd_rectifier_destroy(rect_r);
    std::cout << std::endl << "Frames: " << n << "/" << frame_count << "  Average frame processing time (rectify + track): " << (n ? total_time / (double)n : 0.0) << std::endl;
    return 0;
}
