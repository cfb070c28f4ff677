// Development harness (not part of the product): ONE instantiation of the batched matcher behind a C entry, so that a kernel
// experiment compiles in seconds.  tools/hamming_lab/lab.py builds it, times it and diffs its output against the shipped library.
#include LAB_KERNEL
#include <cstdio>
#include <cstdlib>
using namespace lvt;
extern "C" float lab_run(const void *q_desc, const void *q_xy, const void *t_desc, const void *t_xy, const void *t_flag, int B, int M, int N,
                         float r2, int img_rows, int img_cols, void *out, int launches, long long *dbg_host) {
    HammingArgs a;
    a.q_desc = (const uint64_t *)q_desc, a.q_xy = (const float2 *)q_xy, a.t_desc = (const uint64_t *)t_desc, a.t_xy = (const float2 *)t_xy;
    a.t_flag = (const uint8_t *)t_flag, a.out = (int4 *)out;
    a.M = M, a.N = N, a.r2 = r2, a.img_rows = img_rows, a.img_cols = img_cols;
#ifdef LAB_HAS_B
    a.B = B;
#endif
#ifndef LAB_MODE
#define LAB_MODE 0
#endif
    if (LAB_MODE == 1) a.nbx = 1, a.nby = img_rows + 1, a.csr = 0;
    else a.nbx = (img_cols + HASH_CELL - 1) / HASH_CELL, a.nby = (img_rows + HASH_CELL - 1) / HASH_CELL, a.csr = 1;
    long long *d_dbg = nullptr;
    a.dbg = nullptr;
    const size_t dbg_bytes = (dbg_host && dbg_host[15] == 12345) ? (16 + 4 * (size_t)B) * 8 : 128;
    if (dbg_host && hipMalloc((void **)&d_dbg, dbg_bytes) == hipSuccess && hipMemset(d_dbg, 0, dbg_bytes) == hipSuccess) {
        a.dbg = d_dbg;
        hipMemcpy(d_dbg + 15, dbg_host + 15, 8, hipMemcpyHostToDevice);
    }
#ifdef LAB_SPLIT
    SplitPlan pl = hamming_split_plan<LAB_MODE>(a, B, LAB_SPLIT, LAB_THREADS);
    pl.sa.h.dbg = a.dbg;
    size_t lds = pl.lds;
    auto split_kern = hamming_split_pick<LAB_MODE, LAB_THREADS, LAB_SPLIT, LAB_WPE>(pl);
    if (!split_kern) {
        std::fprintf(stderr, "lab: no instance for N=%d M=%d ncap=%d mcap=%d T=%d\n", N, M, pl.sa.ncap, pl.sa.mcap, LAB_THREADS);
        return -4.f;
    }
    if (dbg_host && dbg_host[14] == 777) {
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, split_kern, LAB_THREADS, lds);
        std::fprintf(stderr, "split plan: S=%d T=%d ncap=%d mcap=%d nbins_max=%d dynamic LDS %zu B, grid %d, occupancy query: %d workgroups per CU\n", LAB_SPLIT, LAB_THREADS, pl.sa.ncap,
                     pl.sa.mcap, pl.sa.nbins_max, lds, pl.grid, nb);
    }
#define a pl.sa
#elif defined(LAB_BAND)
    BandArgs ba;
    ba.h = a;
    if (!hamming_band_setup(ba)) return -3.f;
    size_t lds = hamming_band_lds_bytes(N, M, ba.nbands * ba.nxb);
#define a ba
#else
    size_t lds = hamming_lds_bytes(N, M, a.nbx * a.nby);
#endif
    if (getenv("LAB_LDS_EXTRA")) lds += atoi(getenv("LAB_LDS_EXTRA"));  // e.g. force one workgroup per CU
#ifdef LAB_SPLIT
    auto kern = split_kern;
#else
    auto kern = LAB_INSTANCE;
#endif
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    int grid = B, threads = HB_THREADS;
#ifdef LAB_SPLIT
    grid = pl.grid, threads = LAB_THREADS;
#endif
#ifdef LAB_HAS_B
    grid = B < 512 ? B : 512;  // resident workgroups: two per CU
    if (getenv("LAB_GRID")) grid = atoi(getenv("LAB_GRID"));
#endif
    for (int l = 0; l < launches; l++) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, a);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -2.f;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0), hipEventDestroy(e1);
#if defined(LAB_BAND) || defined(LAB_SPLIT)
#undef a
#endif
    if (d_dbg) {
        hipMemcpy(dbg_host, d_dbg, dbg_bytes, hipMemcpyDeviceToHost);
        hipFree(d_dbg);
    }
    return ms * 1000.f / launches;
}
