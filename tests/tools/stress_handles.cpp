// Stress / throughput run of SEVERAL lvt handles in one process, one host thread per handle, in C++ (no interpreter lock between the
// threads: tests/tools/stress_handles.py drives the same experiment from Python threads, whose numbers include GIL hand-overs).
//   stress_handles <frames.bin> <n_frames> <rows> <cols> <pitch> <handles> <frames per handle> [vo_config.yaml]
// frames.bin: n_frames stereo pairs, each 2 x rows x pitch bytes (left plane, right plane), rendered by stress_handles_cpp.py.
// Every handle tracks the same device-resident sequence, walking it back and forth (0 .. n-1, n-2 .. 0, ...), 4 frames in flight;
// all handles must report TRACKING throughout and identical final poses.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "lvt_amd_ext.h"
#include "lvt_c.h"

int main(int argc, char **argv) {
    if (argc < 8) {
        std::fprintf(stderr, "usage: %s frames.bin n rows cols pitch handles frames_per_handle [config.yaml]\n", argv[0]);
        return 2;
    }
    const int n = std::atoi(argv[2]), rows = std::atoi(argv[3]), cols = std::atoi(argv[4]), pitch = std::atoi(argv[5]);
    const int H = std::atoi(argv[6]), per = std::atoi(argv[7]);
    const size_t plane = (size_t)rows * pitch, bytes = (size_t)n * 2 * plane;
    std::vector<unsigned char> host(bytes);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(host.data(), 1, bytes, f) != bytes) {
        std::fprintf(stderr, "cannot read %zu bytes from %s\n", bytes, argv[1]);
        return 2;
    }
    std::fclose(f);
    unsigned char *dev = nullptr;
    if (hipMalloc((void **)&dev, bytes) != hipSuccess || hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 3;
    lvt_amd_params prm;
    if (argc > 8) {
        if (lvt_amd_params_from_file(argv[8], &prm) != 1) return 4;
    } else
        lvt_amd_default_params(&prm);
    prm.img_width = cols, prm.img_height = rows;

    std::vector<std::vector<double>> last((size_t)H, std::vector<double>(12, 0.0));
    std::vector<int> lost((size_t)H, 0), ordering((size_t)H, -1);
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    auto work = [&](int k) {
        // the reference's own create call when a configuration file was given (round 6: handles created before any of them has tracked become seats of one chain by
        // themselves), the extension's otherwise
        lvt_handle h = (argc > 8) ? lvt_create(argv[8], 1) : lvt_amd_create(&prm, 1);
        if (!h) {
            lost[(size_t)k] = -1;
            ready++;
            return;
        }
        ordering[(size_t)k] = lvt_amd_get_ordering(h);
        ready++;
        while (!go.load()) std::this_thread::yield();
        double R[3][3], t[3];
        int inflight = 0;
        for (int i = 0; i < per; i++) {
            int j = i % (2 * n - 2);
            if (j >= n) j = 2 * n - 2 - j;  // back and forth: consecutive frames stay neighbours
            const unsigned char *l = dev + (size_t)j * 2 * plane;
            lvt_amd_track_device_async(h, l, l + plane, rows, cols, pitch);
            if (++inflight >= 4) {
                if (lvt_amd_wait_status(h, R, t) != 2) lost[(size_t)k]++;
                inflight--;
            }
        }
        while (inflight--)
            if (lvt_amd_wait_status(h, R, t) != 2) lost[(size_t)k]++;
        for (int a = 0; a < 3; a++) {
            for (int b = 0; b < 3; b++) last[(size_t)k][(size_t)(3 * a + b)] = R[a][b];
            last[(size_t)k][(size_t)(9 + a)] = t[a];
        }
        const char *e = lvt_amd_last_error(h);
        if (e && *e) std::fprintf(stderr, "handle %d: %s\n", k, e);
        lvt_destroy(h);
    };
    std::vector<std::thread> th;
    for (int k = 0; k < H; k++) th.emplace_back(work, k);
    while (ready.load() < H) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true);
    for (auto &x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    bool same = true;
    int bad = 0;
    for (int k = 0; k < H; k++) {
        bad += lost[(size_t)k] != 0;
        for (int i = 0; i < 12; i++) same = same && last[(size_t)k][(size_t)i] == last[0][(size_t)i];
    }
    std::printf("STRESS-C++ handles=%d frames_each=%d ordering=", H, per);
    for (int k = 0; k < H; k++) std::printf("%d", ordering[(size_t)k]);
    std::printf(" identical_final_pose=%d handles_not_tracking=%d aggregate %.0f frames/s (%.1f us per frame and handle)\n", same ? 1 : 0, bad,
                H * (double)per / dt, 1e6 * dt / per);
    (void)hipFree(dev);
    return (same && bad == 0) ? 0 : 1;
}
