// k_hamming.hip -- batched masked 2-NN Hamming matcher (the matcher of lvt_image_features_struct.cpp:68-148
// + cv::BFMatcher::knnMatch(k=2, mask)) as ONE launch over B independent problems.
//
// One 512-thread workgroup per problem, two workgroups per CU (72 KB of LDS each at N = 1500).
//   1. every thread fetches "its" train features (coordinates, flag, 32-B descriptor) and its first queries into
//      registers -- all global loads of the problem are in flight at once;
//   2. the train set is counting-sorted into the reference's 25-px hash cells (tracking mode) or image rows (row
//      mode): the LDS atomicAdd that counts a bin also returns the feature's rank inside it, so one scan of the
//      bin counts later each feature is stored at start[bin] + rank -- coordinates, packed (flag | index) word and
//      descriptor all in BIN ORDER, the query loop then walks contiguous LDS with no indirection;
//   3. each lane owns queries (descriptor in 4 x u64 VGPR pairs), visits only the bins its mask admits, evaluates
//      4 x (xor, popcount) on the few candidates that pass the coordinate test, and keeps the running top-2 as
//      packed (distance << 16 | index) keys so ties resolve to the lowest index exactly as batchDistance does.
// Algorithmic HBM bytes per problem: 40*(M+N) + N + 16*M; every byte is read or written exactly once.
#include "lvt_dev.h"

namespace lvt {

struct HammingArgs {
    const uint64_t *q_desc;  // [B][M][4]
    const float2 *q_xy;      // [B][M]
    const uint64_t *t_desc;  // [B][N][4]
    const float2 *t_xy;      // [B][N]
    const uint8_t *t_flag;   // [B][N]
    int4 *out;               // [B][M] (idx1, d1, idx2, d2)
    int M, N;
    float r2;
    int img_rows, img_cols;
    int nbx, nby, csr;       // bins: hash cells (mode 0) or rows (mode 1: nbx = 1, nby = rows + 1)
    long long *dbg;          // optional: phase cycle stamps of workgroup 0
};

constexpr int HB_THREADS = 512;
constexpr int HB_TPT = 4;          // train features per thread  => N <= 2048
constexpr int HB_NMAX = HB_THREADS * HB_TPT;

template <int MODE>
__global__ __launch_bounds__(HB_THREADS) void k_hamming_batched(HammingArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int N = a.N, M = a.M;
    const int nbins = a.nbx * a.nby;
    // carve: desc_s [N][2] uint4 | rec_s [N] {x, y, flag|index, -} (16 B) | start [nbins+1] | cursor [nbins]
    uint4 *s_desc = reinterpret_cast<uint4 *>(smem);
    float4 *s_rec = reinterpret_cast<float4 *>(s_desc + (size_t)N * 2);
    int *s_start = reinterpret_cast<int *>(s_rec + N);
    int *s_cur = s_start + nbins + 1;
    __shared__ int s_scan[32];

    const int b = blockIdx.x, tid = threadIdx.x;
    const uint4 *td = reinterpret_cast<const uint4 *>(a.t_desc + (size_t)b * N * 4);
    const float2 *txy = a.t_xy + (size_t)b * N;
    const uint8_t *tf = a.t_flag + (size_t)b * N;
    const uint4 *qd = reinterpret_cast<const uint4 *>(a.q_desc + (size_t)b * M * 4);
    const float2 *qxy = a.q_xy + (size_t)b * M;
    int4 *out = a.out + (size_t)b * M;

    long long *dbg = (a.dbg && blockIdx.x == gridDim.x / 2 && tid == 0) ? a.dbg : nullptr;
    if (dbg) dbg[0] = clock64();
    auto bin_of = [&](float x, float y) -> int {
        if (MODE == 1) return min(max((int)floorf(y), 0), a.nby - 1);
        const int cy = min(max((int)floorf(y / (float)HASH_CELL), 0), a.nby - 1);
        const int cx = min(max((int)floorf(x / (float)HASH_CELL), 0), a.nbx - 1);
        return cy * a.nbx + cx;
    };

    // ---- 1. everything this thread will need from HBM, issued back to back
    float2 tp[HB_TPT];
    uint8_t tfl[HB_TPT];
    uint4 tdlo[HB_TPT], tdhi[HB_TPT];
#pragma unroll
    for (int k = 0; k < HB_TPT; k++) {
        const int j = tid + k * HB_THREADS;
        if (j < N) {
            tp[k] = txy[j];
            tfl[k] = tf[j];
            tdlo[k] = td[2 * j];
            tdhi[k] = td[2 * j + 1];
        }
    }
    uint4 q0lo = make_uint4(0, 0, 0, 0), q0hi = q0lo;
    float2 q0p = make_float2(0, 0);
    if (tid < M) {
        q0lo = qd[2 * tid];
        q0hi = qd[2 * tid + 1];
        q0p = qxy[tid];
    }
    for (int i = tid; i < nbins; i += HB_THREADS) s_cur[i] = 0;
    __syncthreads();
    if (dbg) dbg[1] = clock64();
    // ---- 2. counting sort into bins: the counting atomic returns the rank inside the bin
    int tbin[HB_TPT], trank[HB_TPT];
#pragma unroll
    for (int k = 0; k < HB_TPT; k++) {
        const int j = tid + k * HB_THREADS;
        if (j < N) {
            tbin[k] = bin_of(tp[k].x, tp[k].y);
            trank[k] = atomicAdd(&s_cur[tbin[k]], 1);
        }
    }
    __syncthreads();
    if (dbg) dbg[2] = clock64();
    {
        int run = 0;
        for (int base = 0; base < nbins; base += HB_THREADS) {
            const int i = base + tid;
            const int v = (i < nbins) ? s_cur[i] : 0;
            int total;
            const int ex = run + block_excl_scan(v, s_scan, &total);
            if (i < nbins) s_start[i] = ex;
            run += total;
        }
        if (tid == 0) s_start[nbins] = run;
    }
    __syncthreads();
    if (dbg) dbg[3] = clock64();
#pragma unroll
    for (int k = 0; k < HB_TPT; k++) {
        const int j = tid + k * HB_THREADS;
        if (j < N) {
            const int pos = s_start[tbin[k]] + trank[k];
            s_rec[pos] = make_float4(tp[k].x, tp[k].y, __uint_as_float((uint32_t)j | (tfl[k] ? 0x80000000u : 0u)), 0.0f);
            s_desc[2 * pos] = tdlo[k];
            s_desc[2 * pos + 1] = tdhi[k];
        }
    }
    __syncthreads();
    if (dbg) dbg[4] = clock64();

    // ---- 3. queries
    for (int q = tid; q < M; q += HB_THREADS) {
        uint4 w0, w1;
        float2 p;
        if (q == tid) {
            w0 = q0lo, w1 = q0hi, p = q0p;
        } else {
            w0 = qd[2 * q];
            w1 = qd[2 * q + 1];
            p = qxy[q];
        }
        const uint64_t d0 = ((uint64_t)w0.y << 32) | w0.x, d1 = ((uint64_t)w0.w << 32) | w0.z;
        const uint64_t d2 = ((uint64_t)w1.y << 32) | w1.x, d3 = ((uint64_t)w1.w << 32) | w1.z;
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        int y0, y1, x0, x1;
        if (MODE == 1) {  // struct.cpp:124-131: [int(y)-2, int(y)+2] clipped to [0, rows]
            y0 = max((int)p.y - ROW_RADIUS, 0);
            y1 = min((int)p.y + ROW_RADIUS, a.img_rows);
            x0 = x1 = 0;
        } else {  // struct.cpp:71-83
            const int hy = (int)floorf(p.y / (float)HASH_CELL), hx = (int)floorf(p.x / (float)HASH_CELL);
            y0 = max(hy - a.csr, 0);
            y1 = min(hy + a.csr, a.nby - 1);
            x0 = max(hx - a.csr, 0);
            x1 = min(hx + a.csr, a.nbx - 1);
        }
        // one candidate, branch-free: every LDS read is unconditional so the loads of the next candidates pipeline
        auto eval = [&](int it, bool valid, float fy0, float fy1) {
            const float4 r = s_rec[it];
            const uint4 a0 = s_desc[2 * it], a1 = s_desc[2 * it + 1];
            const uint32_t m = __float_as_uint(r.z);
            bool ok;
            if (MODE == 1) ok = (r.y >= fy0) && (r.y <= fy1);
            else {
                const float dx = r.x - p.x, dy = r.y - p.y;
                ok = (dx * dx + dy * dy) < a.r2;
            }
            ok = ok && valid && !(m >> 31);
            const int d = __popcll(d0 ^ (((uint64_t)a0.y << 32) | a0.x)) + __popcll(d1 ^ (((uint64_t)a0.w << 32) | a0.z)) +
                          __popcll(d2 ^ (((uint64_t)a1.y << 32) | a1.x)) + __popcll(d3 ^ (((uint64_t)a1.w << 32) | a1.z));
            const uint32_t key = ok ? (((uint32_t)d << 16) | (m & 0xFFFFu)) : 0xFFFFFFFFu;
            k2 = min(k2, max(k1, key));
            k1 = min(k1, key);
        };
        auto span = [&](int s, int e, float fy0, float fy1) {
#pragma unroll 2
            for (int it = s; it < e; it += 2) {
                eval(it, true, fy0, fy1);
                eval(min(it + 1, e - 1), it + 1 < e, fy0, fy1);
            }
        };
        if (MODE == 1) {
            // rows y0..y1 are contiguous bins: one span
            y1 = min(y1, a.nby - 1);
            const int s = (y0 <= y1) ? s_start[y0] : 0, e = (y0 <= y1) ? s_start[y1 + 1] : 0;
            span(s, e, (float)y0, (float)min((int)p.y + ROW_RADIUS, a.img_rows));
        } else {
            // the bins of one window row are contiguous in the CSR: at most 2*csr+1 spans; their bounds are read up front
            int ss[3], ee[3];
            const bool small = (y1 - y0) <= 2;
            if (small) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int by = min(y0 + k, y1);
                    ss[k] = s_start[by * a.nbx + x0];
                    ee[k] = (y0 + k <= y1) ? s_start[by * a.nbx + x1 + 1] : ss[k];
                }
#pragma unroll
                for (int k = 0; k < 3; k++) span(ss[k], ee[k], 0.f, 0.f);
            } else {
                for (int by = y0; by <= y1; by++) span(s_start[by * a.nbx + x0], s_start[by * a.nbx + x1 + 1], 0.f, 0.f);
            }
        }
        int4 o;
        o.x = (k1 == 0xFFFFFFFFu) ? -1 : (int)(k1 & 0xFFFFu);
        o.y = (k1 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k1 >> 16);
        o.z = (k2 == 0xFFFFFFFFu) ? -1 : (int)(k2 & 0xFFFFu);
        o.w = (k2 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k2 >> 16);
        out[q] = o;
    }
    if (dbg) dbg[5] = clock64();
}

static inline size_t hamming_lds_bytes(int N, int nbins) {
    return (size_t)N * 32 + (size_t)N * 16 + (size_t)(2 * nbins + 1) * 4 + 64;
}

}  // namespace lvt
