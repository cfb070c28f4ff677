// lab: how fast does ONE compute unit issue fp64 instructions, as a function of the wavefronts per SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f64 tools/lab/fp64_issue.hip && /tmp/f64
// One workgroup of 256 / 512 / 1024 threads (1 / 2 / 4 wavefronts per SIMD); every thread runs CH independent chains of REP dependent
// operations (fma, mul, add); the workgroup's clock64 span / (REP * CH) = cycles per wave-instruction slot of a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP, int CH>
__global__ void k(double *out, long long *cyc, int rep, double a, double b) {
    double v[CH];
    for (int c = 0; c < CH; c++) v[c] = (double)threadIdx.x * 1e-3 + c;
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < rep; r += 16) {  // sixteen rounds per trip: the loop's own scalar instructions and branch (~28 cycles) stay below 5 %
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (OP == 0) v[c] = __builtin_fma(v[c], a, b);
                if (OP == 1) v[c] = v[c] * a;
                if (OP == 2) v[c] = v[c] + b;
            }
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CH; c++) s += v[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP, int CH>
void run(const char *name, double *out, long long *cyc) {
    const int rep = 4096;
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL((k<OP, CH>), dim3(1), dim3(threads), 0, 0, out, cyc, rep, 1.0000001, 1e-9);
        hipLaunchKernelGGL((k<OP, CH>), dim3(1), dim3(threads), 0, 0, out, cyc, rep, 1.0000001, 1e-9);
        long long c = 0;
        (void)hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        const double per = (double)c / ((double)rep * CH);
        const int wps = threads >= 256 ? threads / 256 : 1;
        std::printf("%-4s chains %d  threads %4d (%d wave(s) per SIMD%s): %.2f cycles per instruction of a wave, %.2f per SIMD issue slot\n", name, CH, threads, wps,
                    threads == 64 ? ", one SIMD only" : "", per, per / wps);
    }
}
int main() {
    double *out;
    long long *cyc;
    (void)hipMalloc(&out, 1024 * sizeof(double));
    (void)hipMalloc(&cyc, sizeof(long long));
    run<0, 1>("fma", out, cyc);
    run<0, 2>("fma", out, cyc);
    run<0, 4>("fma", out, cyc);
    run<0, 8>("fma", out, cyc);
    run<1, 1>("mul", out, cyc);
    run<1, 2>("mul", out, cyc);
    run<1, 8>("mul", out, cyc);
    run<2, 1>("add", out, cyc);
    run<2, 8>("add", out, cyc);
    return 0;
}
