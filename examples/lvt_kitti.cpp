// lvt_kitti -- KITTI-odometry command line harness, written against the header-compatible C++ layer
// include/lvt_system.h -- the calls below are the reference example's own (kitti_example.cpp:82-131: lvt_parameters,
// lvt_system::create, vo->track, lvt_pose::get_position / get_orientation_matrix), with lvt_image_view where it has cv::Mat.
//
// Same argv, inputs and outputs as the reference's example binary (examples/kitti/kitti_example.cpp:49-152 there):
//     lvt_kitti <sequences_dir> <seq_number> [--config vo_config.yaml] [--calib calib/NN.yml] [--max-frames N] [--out NN.txt]
// reads <sequences_dir>/NN/image_0/%06d.{png,pgm} and image_1/..., the OpenCV-YAML calibration (camera_matrix, baseline)
// and vo_config.yaml, tracks every frame until the sequence ends or the tracker is LOST, writes NN.txt in the KITTI
// 3x4 row format with the reference's precision (fixed, 9 digits) and prints the mean per-frame track() time.
// No OpenCV: the PNG reader below handles what KITTI ships (8-bit gray / RGB / RGBA / gray+alpha, non-interlaced), colour
// is reduced to gray with cv::cvtColor's fixed-point weights, and PGM (P5) is accepted for pre-converted data.
#include "../include/lvt_system.h"

#include "image_io.h"

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

using namespace lvt_io;

namespace {

// the two entries the reference reads from calib/NN.yml (OpenCV FileStorage YAML): camera_matrix.data and baseline
bool read_calib(const std::string &path, double K[9], double &baseline) {
    std::ifstream f(path);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string s = ss.str();
    const size_t cm = s.find("camera_matrix");
    const size_t d0 = (cm == std::string::npos) ? cm : s.find("data:", cm);
    const size_t b0 = s.find("baseline:");
    if (d0 == std::string::npos || b0 == std::string::npos) return false;
    const size_t lb = s.find('[', d0), rb = s.find(']', d0);
    if (lb == std::string::npos || rb == std::string::npos) return false;
    std::string list = s.substr(lb + 1, rb - lb - 1);
    for (char &c : list)
        if (c == ',') c = ' ';
    std::stringstream ls(list);
    for (int i = 0; i < 9; i++)
        if (!(ls >> K[i])) return false;
    baseline = std::atof(s.c_str() + b0 + 9);
    return true;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 3) {
        std::cout << "Usage ./lvt_kitti sequences_dir seq_number [--config vo_config.yaml] [--calib calib/NN.yml] [--max-frames N] [--out NN.txt]" << std::endl;
        return -1;
    }
    char seq_cstr[16];
    std::snprintf(seq_cstr, sizeof seq_cstr, "%02d", std::atoi(argv[2]));
    const std::string seq_str(seq_cstr), dir_prefix = std::string(argv[1]) + "/" + seq_str;
    std::string config = "vo_config.yaml", calib = "calib/" + seq_str + ".yml", out_name = seq_str + ".txt";
    long max_frames = -1;
    for (int i = 3; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--config") config = argv[i + 1];
        else if (k == "--calib") calib = argv[i + 1];
        else if (k == "--out") out_name = argv[i + 1];
        else if (k == "--max-frames") max_frames = std::atol(argv[i + 1]);
    }
    double K[9], baseline = 0;
    if (!read_calib(calib, K, baseline)) {
        std::cout << "failed to open camera matrix yml file" << std::endl;
        return -1;
    }
    lvt_parameters params;
    if (!params.init_from_file(config.c_str())) {
        std::cout << "failed to initialize from vo_config.yml file." << std::endl;
        return -1;
    }
    auto stem = [&](int cam, long i) {
        char name[32];
        std::snprintf(name, sizeof name, "/image_%d/%06ld", cam, i);
        return dir_prefix + name;
    };
    Gray left, right;
    std::string err;
    if (!load_gray(stem(0, 0), left, err) || !load_gray(stem(1, 0), right, err)) {
        std::cout << "failed to get image sequences (" << err << ")" << std::endl;
        return -1;
    }
    params.fx = (float)K[0], params.fy = (float)K[4], params.cx = (float)K[2], params.cy = (float)K[5];
    params.baseline = (float)baseline;
    params.img_width = left.w, params.img_height = left.h;
    lvt_system *vo = lvt_system::create(params, lvt_system::eSensor_STEREO);
    if (!vo) {
        std::cout << "failed to create the tracker: " << lvt_amd_last_error(nullptr) << std::endl;
        return -1;
    }
    // like the reference, the trajectory has one row per frame of the sequence; frames after a LOST keep the default pose
    long frame_count = 0;
    for (;; frame_count++) {
        std::ifstream pl(stem(0, frame_count) + ".png"), gl(stem(0, frame_count) + ".pgm");
        if (!pl && !gl) break;
        if (max_frames >= 0 && frame_count >= max_frames) break;
    }
    std::vector<double> rows;  // 12 per processed frame
    double total_time = 0.0;
    long n = 0;
    // Frame i is enqueued (lvt_system::track_async: its images are the library's when the call returns), frame i + 1 is read and decoded while it
    // tracks, then frame i's pose is collected: the reference's loop (kitti_example.cpp:113-138) does the two in turn.  Same poses, same file.
    Gray nleft, nright;
    bool have = true;
    for (long i = 0; i < frame_count && have; i++) {
        if (left.w != params.img_width || left.h != params.img_height || right.w != left.w || right.h != left.h) {
            std::cout << "frame " << i << ": image size changed" << std::endl;
            break;
        }
        std::cout << "Frame number: " << i << "\r" << std::flush;
        const auto t0 = std::chrono::steady_clock::now();
        const bool queued = vo->track_async(lvt_image_view(left.px.data(), left.h, left.w), lvt_image_view(right.px.data(), right.h, right.w));
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (!queued) break;
        have = i + 1 < frame_count && load_gray(stem(0, i + 1), nleft, err) && load_gray(stem(1, i + 1), nright, err);  // (an unreadable frame ends the run)
        const auto t1 = std::chrono::steady_clock::now();
        const lvt_pose pose = vo->wait_pose();
        dt += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        total_time += dt;  // (time spent inside the tracker's calls, as the reference reports it)
        const lvt_vector3 pos = pose.get_position();  // kitti_example.cpp:40-47
        const lvt_matrix33 R = pose.get_orientation_matrix();
        for (int r = 0; r < 3; r++) {
            rows.push_back(R(r, 0)), rows.push_back(R(r, 1)), rows.push_back(R(r, 2));
            rows.push_back(pos(r));
        }
        n++;
        if (vo->get_state() == lvt_system::eState_LOST) break;
        if (have) std::swap(left, nleft), std::swap(right, nright);
    }
    std::ofstream file(out_name.c_str());
    file << std::fixed;
    static const double identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (long i = 0; i < frame_count; i++) {
        const double *row = (i < n) ? &rows[(size_t)i * 12] : identity;
        for (int k = 0; k < 12; k++) file << std::setprecision(9) << row[k] << (k == 11 ? "" : " ");
        file << std::endl;
    }
    file.close();
    lvt_system::destroy(vo);
    std::cout << std::endl << "Frames: " << n << "/" << frame_count << "  Average frame processing time: " << (frame_count ? total_time / (double)frame_count : 0.0) << std::endl;
    return 0;
}
