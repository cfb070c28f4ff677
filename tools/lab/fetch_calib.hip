// lab: what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for a KNOWN byte count, by access width?  (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE shows one half of a wide
// coalesced streaming read; other widths and WRITE_SIZE are "uncalibrated: calibrate on a known byte count in your own access pattern".)
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/bin/fetch_calib tools/lab/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -d out -o c --output-format csv -- tools/lab/bin/fetch_calib      (and again with WRITE_SIZE; tools/pmc_frame.sh does both)
// Every kernel streams NBYTES = 1 GiB (past the 256-MB Infinity Cache) once: rd<W> reads it with W bytes per lane, lanes of a wavefront on consecutive addresses, and
// writes 4 bytes per workgroup; wr<W> writes it the same way and reads nothing; rd_rows reads it the way k_score reads an image -- 64-byte row segments of 16 lanes x 4 B,
// rows a 1280-byte pitch apart (a wavefront touches four rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr size_t NBYTES = 1ull << 30;
template <class T>
__global__ __launch_bounds__(256) void rd(const T *src, uint32_t *sink, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const T v = src[i];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;  // (never true for the zero-filled buffer XOR pattern used here: the loads stay, nothing is written)
}
template <class T>
__global__ __launch_bounds__(256) void wr(T *dst, size_t n, uint32_t val) {
    const size_t stride = (size_t)gridDim.x * 256;
    T v;
    uint32_t *w = reinterpret_cast<uint32_t *>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) w[k] = val + k;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}
__global__ __launch_bounds__(256) void rd_rows(const uint32_t *src, uint32_t *sink, size_t n_words) {
    // word index of lane l of wavefront-step s: 16 lanes cover 64 bytes of a row, the wavefront's four 16-lane groups sit on four consecutive rows of 320 words
    const size_t steps = (n_words / 1280) * 20;  // whole tiles of 4 rows x 320 words only (the last partial tile is not read: 1 GiB = 209 715 tiles + 256 words)
    uint32_t acc = 0;
    for (size_t s = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < steps; s += (size_t)gridDim.x * 4) {
        const int lane = threadIdx.x & 63;
        const size_t tile = s / 20, col = s % 20;               // 20 segments of 64 bytes per 1280-byte row, 4 rows per tile
        const size_t w = tile * 4 * 320 + (size_t)(lane >> 4) * 320 + col * 16 + (lane & 15);
        acc ^= src[w];
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}
int main() {
    void *buf; uint32_t *sink;
    if (hipMalloc(&buf, NBYTES) != hipSuccess || hipMalloc((void **)&sink, 1 << 20) != hipSuccess) return 1;
    (void)hipMemset(buf, 0x5a, NBYTES);
    (void)hipDeviceSynchronize();
    const int grid = 256 * 8;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(rd<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, sink, NBYTES / 4);
        hipLaunchKernelGGL(rd<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2 *)buf, sink, NBYTES / 8);
        hipLaunchKernelGGL(rd<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4 *)buf, sink, NBYTES / 16);
        hipLaunchKernelGGL(rd_rows, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, sink, NBYTES / 4);
        hipLaunchKernelGGL(wr<uint32_t>, dim3(grid), dim3(256), 0, 0, (uint32_t *)buf, NBYTES / 4, 7u);
        hipLaunchKernelGGL(wr<uint2>, dim3(grid), dim3(256), 0, 0, (uint2 *)buf, NBYTES / 8, 7u);
        hipLaunchKernelGGL(wr<uint4>, dim3(grid), dim3(256), 0, 0, (uint4 *)buf, NBYTES / 16, 7u);
    }
    (void)hipDeviceSynchronize();
    std::printf("fetch_calib: 3 x 7 kernels over %zu bytes each\n", NBYTES);
    return 0;
}
