// lab check of lvt_amd/csrc/wave_reduce.h on the device:  hipcc --offload-arch=gfx950 -O3 -o /tmp/wrt tools/lab/wave_reduce_test.hip && /tmp/wrt
#include "../../lvt_amd/csrc/wave_reduce.h"
#include <cmath>
#include <cstdio>
#include <vector>
template <int NV>
__global__ void k(const double *in, double *out) {  // in: [NV][64], out: [32]
    double v[NV];
    for (int k2 = 0; k2 < NV; k2++) v[k2] = in[k2 * 64 + threadIdx.x];
    const double s = lvt::wave_reduce_scatter<NV>(v);
    const int idx = lvt::wave_rs_index(threadIdx.x);
    if (!(threadIdx.x & 1)) out[idx] = s;
    if ((threadIdx.x & 1) && idx < NV) out[32 + idx] = s;  // the pair's other lane holds the same total
}
template <int NV>
int run() {
    std::vector<double> h(NV * 64), o(64, -1.0);
    unsigned long long st = 88172645463325252ull + NV;
    for (auto &x : h) {
        st ^= st << 13, st ^= st >> 7, st ^= st << 17;
        x = (double)(st % 2000001) / 1000.0 - 1000.0;
    }
    double *di, *dout;
    (void)hipMalloc(&di, sizeof(double) * h.size());
    (void)hipMalloc(&dout, sizeof(double) * 64);
    (void)hipMemcpy(di, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    (void)hipMemset(dout, 0, sizeof(double) * 64);
    hipLaunchKernelGGL(k<NV>, dim3(1), dim3(64), 0, 0, di, dout);
    (void)hipMemcpy(o.data(), dout, sizeof(double) * 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int k2 = 0; k2 < NV; k2++) {
        long double s = 0;
        for (int l = 0; l < 64; l++) s += h[k2 * 64 + l];
        if (std::fabs((double)s - o[k2]) > 1e-9 || o[k2] != o[32 + k2]) bad++, std::printf("NV %d value %d: %.12f / %.12f, expected %.12f\n", NV, k2, o[k2], o[32 + k2], (double)s);
    }
    std::printf("NV = %d: %s\n", NV, bad ? "WRONG" : "ok");
    return bad;
}
int main() { return (run<28>() + run<32>() + run<7>() + run<1>() + run<17>()) ? 1 : 0; }
