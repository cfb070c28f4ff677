// lab: how many cycles does a SIMD need per wave64 INTEGER / fp32 VALU instruction (the batched matcher's instruction mix)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/bin/int_issue tools/lab/int_issue.hip && tools/lab/bin/int_issue
// One workgroup of 256 / 512 / 1024 threads (1 / 2 / 4 wavefronts per SIMD); every thread runs CH independent chains of REP dependent
// operations; the workgroup's clock64 span / (REP * CH * waves per SIMD) = cycles per wave-instruction slot of a SIMD.
// The operations are inline asm: nothing is folded, every chain is a true dependency chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
enum { XOR, BCNT, ADD, MINU, FMA32, PKFMA, CNDMASK, CMP_ADDC, LSHL_OR, PERM, MAD24, POPC_PAIR };
template <int OP>
__device__ __forceinline__ void op(uint32_t &v, uint32_t &w, uint32_t a, uint32_t b) {
    if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v) : "v"(a));
    if (OP == BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v) : "v"(a));
    if (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(a));
    if (OP == MINU) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v) : "v"(a));
    if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
    if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<uint64_t *>(&v)) : "v"((uint64_t)a | ((uint64_t)b << 32)));
    if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(a) : "vcc");
    if (OP == CMP_ADDC) asm volatile("v_cmp_gt_u32 vcc, %1, %0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(v) : "v"(a) : "vcc");
    if (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(v) : "v"(a));
    if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
    if (OP == MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
    if (OP == POPC_PAIR) asm volatile("v_xor_b32 %1, %0, %2\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(v), "+v"(w) : "v"(a));
}
template <int OP, int CH>
__global__ __launch_bounds__(1024) void k(uint32_t *out, long long *cyc, int rep, uint32_t a, uint32_t b) {
    uint32_t v[CH], w[CH];
    for (int c = 0; c < CH; c++) v[c] = threadIdx.x * 2654435761u + c, w[c] = v[c] ^ a;
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < rep; r += 16) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (OP == PKFMA) {
                    if (c % 2 == 0 && c + 1 < CH) op<OP>(v[c], w[c], a, b);  // v[c], v[c+1] as one 64-bit pair
                } else
                    op<OP>(v[c], w[c], a, b);
            }
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    uint32_t s = 0;
    for (int c = 0; c < CH; c++) s += v[c] + w[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP, int CH>
void run(const char *name, int per_op, uint32_t *out, long long *cyc) {
    const int rep = 4096;
    for (int threads : {256, 512, 1024}) {
        for (int blocks : {1, 512}) {  // one CU alone, and the whole chip busy (2 workgroups per CU: the power-limited clock)
            hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(threads), 0, 0, out, cyc, rep, 0x9E3779B9u, 0x7F4A7C15u);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(threads), 0, 0, out, cyc, rep, 0x9E3779B9u, 0x7F4A7C15u);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            long long c = 0;
            (void)hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
            const int nch = (OP == PKFMA) ? CH / 2 : CH;
            const double insts = (double)rep * nch * per_op;
            const int wps = threads / 256;
            std::printf("%-9s chains %d threads %4d x %3d blocks: %.2f cycles per instruction per SIMD (clock64), launch %.1f us\n", name, nch, threads, blocks,
                        (double)c / insts / wps, ms * 1e3);
        }
    }
}
int main() {
    uint32_t *out;
    long long *cyc;
    (void)hipMalloc(&out, 512 * 1024 * sizeof(uint32_t));
    (void)hipMalloc(&cyc, sizeof(long long));
    run<XOR, 8>("xor", 1, out, cyc);
    run<BCNT, 8>("bcnt", 1, out, cyc);
    run<ADD, 8>("add_u32", 1, out, cyc);
    run<MINU, 8>("min_u32", 1, out, cyc);
    run<FMA32, 8>("fma_f32", 1, out, cyc);
    run<PKFMA, 8>("pk_fma", 1, out, cyc);
    run<CNDMASK, 8>("cndmask", 1, out, cyc);
    run<CMP_ADDC, 8>("cmp+addc", 2, out, cyc);
    run<LSHL_OR, 8>("lshl_or", 1, out, cyc);
    run<PERM, 8>("perm", 1, out, cyc);
    run<MAD24, 8>("mad_u24", 1, out, cyc);
    run<POPC_PAIR, 8>("xor+bcnt", 2, out, cyc);
    run<XOR, 2>("xor", 1, out, cyc);
    run<XOR, 1>("xor", 1, out, cyc);
    return 0;
}
