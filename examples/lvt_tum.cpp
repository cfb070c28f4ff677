// lvt_tum -- TUM RGB-D command line harness over the C-ABI.
//
// Same argv, inputs and outputs as the reference's example binary (examples/tum_rgbd/tum_rgbd_example.cpp:49-150 there):
//     lvt_tum <tum_sequences_root_dir> <associations_dir> <dataset_name> <config_file_name> [--max-frames N] [--out name.txt]
// reads <associations_dir>/<dataset_name>.txt ("t_rgb rgb/xxx.png t_depth depth/xxx.png" per line), the colour and the 16-bit
// depth PNGs below <root>/<dataset_name>/, converts colour to gray (cv::cvtColor weights) and depth to metres in float
// (value * (1.0f / 5000.0f), as cv::Mat::convertTo does), tracks through lvt_amd_track_rgbd and writes <dataset_name>.txt in
// the TUM trajectory format "t x y z qx qy qz qw" with the reference's precision (6 digits for t, 7 for the rest).
#include "../include/lvt_amd_ext.h"
#include "../include/lvt_c.h"
#include "image_io.h"

#include <chrono>
#include <iomanip>
#include <iostream>
#include <sstream>

using namespace lvt_io;

int main(int argc, char **argv) {
    if (argc < 5) {
        std::cout << "Usage ./lvt_tum tum_sequences_root_dir associations_dir dataset_name config_file_name [--max-frames N] [--out name.txt]" << std::endl;
        return -1;
    }
    const std::string root_dir = argv[1], associations_dir = argv[2], dataset_name = argv[3], config_file_name = argv[4];
    std::string out_name = dataset_name + ".txt";
    long max_frames = -1;
    for (int i = 5; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--out") out_name = argv[i + 1];
        else if (k == "--max-frames") max_frames = std::atol(argv[i + 1]);
    }
    std::vector<std::string> rgb_titles, depth_titles;
    std::vector<double> time_stamps;
    {
        const std::string path = associations_dir + "/" + dataset_name + ".txt";
        std::ifstream f(path);
        if (!f.is_open()) {
            std::cout << "Unable to open asscoiations files " << path << std::endl;
            return -1;
        }
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::stringstream ss(line);
            double t = 0, t2 = 0;
            std::string rgb, depth;
            if (!(ss >> t >> rgb >> t2 >> depth)) continue;
            time_stamps.push_back(t);
            rgb_titles.push_back(rgb);
            depth_titles.push_back(depth);
        }
    }
    if (rgb_titles.empty()) {
        std::cout << "Image asscoiations was not read correctly" << std::endl;
        return -1;
    }
    lvt_amd_params params;
    if (!lvt_amd_params_from_file(config_file_name.c_str(), &params)) {
        std::cout << "Failed to initialize from " << config_file_name << std::endl;
        return -1;
    }
    lvt_handle vo = lvt_amd_create(&params, 2 /* RGBD */);
    if (!vo) {
        std::cout << "failed to create the tracker: " << lvt_amd_last_error(nullptr) << std::endl;
        return -1;
    }
    long frame_count = (long)rgb_titles.size();
    if (max_frames >= 0 && max_frames < frame_count) frame_count = max_frames;
    const float depth_scale = 1.0f / 5000.0f;  // depth values in TUM are scaled
    std::vector<double> poses;                 // q (w x y z), p per processed frame
    double total_time = 0;
    long n = 0;
    // frame i is enqueued (lvt_amd_track_rgbd_async: pageable buffers are the library's copy when the call returns), frame i + 1 is read, decoded and
    // converted to metres while it tracks, then frame i's pose is collected -- the reference's loop (tum_rgbd_example.cpp:83-103) does the two in turn
    struct Frame {
        Gray rgb;
        std::vector<float> depth_m;
        bool ok = false;
    };
    auto load = [&](long i, Frame &f) {
        Gray depth;
        std::string err;
        f.ok = i < frame_count && load_image(root_dir + "/" + dataset_name + "/" + rgb_titles[i], f.rgb, err) &&
               load_image(root_dir + "/" + dataset_name + "/" + depth_titles[i], depth, err) && !depth.px16.empty() && depth.w == f.rgb.w && depth.h == f.rgb.h;
        if (!f.ok) return;
        f.depth_m.resize(depth.px16.size());
        for (size_t k = 0; k < f.depth_m.size(); k++) f.depth_m[k] = (float)depth.px16[k] * depth_scale;
    };
    Frame cur, nxt;
    load(0, cur);
    for (long i = 0; i < frame_count; i++) {
        std::cout << "Frame number: " << i << "/" << frame_count << "\r" << std::flush;
        if (!cur.ok) {
            std::cout << "Failed to load image " << std::endl;
            break;
        }
        const auto t0 = std::chrono::steady_clock::now();
        const int queued = lvt_amd_track_rgbd_async(vo, cur.rgb.px.data(), cur.depth_m.data(), cur.rgb.h, cur.rgb.w);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (queued != 0) break;  // (a frame of another size: the reference would throw inside OpenCV)
        load(i + 1, nxt);
        const auto t1 = std::chrono::steady_clock::now();
        double q[4], p[3];
        const int state = lvt_amd_wait_pose(vo, q, p);
        total_time += dt + std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        for (int k = 0; k < 4; k++) poses.push_back(q[k]);
        for (int k = 0; k < 3; k++) poses.push_back(p[k]);
        n++;
        if (state == 3) break;  // LOST
        std::swap(cur, nxt);
    }
    std::ofstream file(out_name.c_str());
    file << std::fixed;
    for (long i = 0; i < (long)rgb_titles.size(); i++) {  // one row per association line, default pose after the end / a LOST
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
    std::cout << std::endl << "Frames: " << n << "/" << frame_count << "  Average frame processing time: " << (n ? total_time / (double)n : 0.0) << std::endl;
    return 0;
}
