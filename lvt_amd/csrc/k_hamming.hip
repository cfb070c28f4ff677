// k_hamming.hip -- batched masked 2-NN Hamming matcher (the matcher of lvt_image_features_struct.cpp:68-148
// + cv::BFMatcher::knnMatch(k=2, mask)) as ONE launch over B independent problems.
//
// One workgroup per problem.  The train set (descriptors 32 B, coordinates, matched flags) is staged in
// LDS once and binned into the reference's 25-px hash cells (tracking mode) or image rows (row mode), so a
// query only visits the handful of candidates its mask admits instead of all N.  Each lane owns queries
// (descriptor held in 4 x u64 VGPR pairs), distances are 4 x (xor, popcount) and the running top-2 is kept
// as packed (distance << 16 | index) keys so that ties resolve to the lowest index exactly as
// batchDistance does.  Algorithmic HBM bytes per problem: 40*(M+N) + N + 16*M  (SURVEY 8d).
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
};

template <int MODE>
__global__ __launch_bounds__(256) void k_hamming_batched(HammingArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int N = a.N, M = a.M;
    const int nbins = a.nbx * a.nby;
    // carve: desc [N][4] u64 | xy [N] float2 | bin_start [nbins+1] int | cursor [nbins] int | items [N] u16 | flag [N] u8
    uint64_t *s_desc = reinterpret_cast<uint64_t *>(smem);
    float2 *s_xy = reinterpret_cast<float2 *>(s_desc + (size_t)N * 4);
    int *s_start = reinterpret_cast<int *>(s_xy + N);
    int *s_cur = s_start + nbins + 1;
    uint16_t *s_items = reinterpret_cast<uint16_t *>(s_cur + nbins);
    uint8_t *s_flag = reinterpret_cast<uint8_t *>(s_items + ((N + 1) & ~1));
    __shared__ int s_scan[32];

    const int b = blockIdx.x, tid = threadIdx.x;
    const uint64_t *td = a.t_desc + (size_t)b * N * 4;
    const float2 *txy = a.t_xy + (size_t)b * N;
    const uint8_t *tf = a.t_flag + (size_t)b * N;

    for (int i = tid; i < nbins; i += 256) s_cur[i] = 0;
    // descriptors: 16-byte coalesced loads
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(td);
        uint4 *dst = reinterpret_cast<uint4 *>(s_desc);
        for (int i = tid; i < N * 2; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    auto bin_of = [&](float x, float y) -> int {
        if (MODE == 1) {
            int r = (int)floorf(y);
            r = min(max(r, 0), a.nby - 1);
            return r;
        }
        int cy = (int)floorf(y / (float)HASH_CELL), cx = (int)floorf(x / (float)HASH_CELL);
        cy = min(max(cy, 0), a.nby - 1);
        cx = min(max(cx, 0), a.nbx - 1);
        return cy * a.nbx + cx;
    };
    for (int j = tid; j < N; j += 256) {
        const float2 p = txy[j];
        s_xy[j] = p;
        s_flag[j] = tf[j];
        atomicAdd(&s_cur[bin_of(p.x, p.y)], 1);
    }
    __syncthreads();
    // exclusive scan of the bin counts
    {
        int run = 0;
        for (int base = 0; base < nbins; base += 256) {
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
    for (int i = tid; i < nbins; i += 256) s_cur[i] = s_start[i];
    __syncthreads();
    for (int j = tid; j < N; j += 256) {
        const float2 p = s_xy[j];
        const int pos = atomicAdd(&s_cur[bin_of(p.x, p.y)], 1);
        s_items[pos] = (uint16_t)j;
    }
    __syncthreads();

    const uint64_t *qd = a.q_desc + (size_t)b * M * 4;
    const float2 *qxy = a.q_xy + (size_t)b * M;
    int4 *out = a.out + (size_t)b * M;
    for (int q = tid; q < M; q += 256) {
        const uint4 w0 = reinterpret_cast<const uint4 *>(qd)[2 * q], w1 = reinterpret_cast<const uint4 *>(qd)[2 * q + 1];
        const uint64_t d0 = ((uint64_t)w0.y << 32) | w0.x, d1 = ((uint64_t)w0.w << 32) | w0.z;
        const uint64_t d2 = ((uint64_t)w1.y << 32) | w1.x, d3 = ((uint64_t)w1.w << 32) | w1.z;
        const float2 p = qxy[q];
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
        for (int by = y0; by <= y1 && by < a.nby; by++) {
            // bins of one row of the window are contiguous in the CSR
            const int s = s_start[by * a.nbx + x0], e = s_start[by * a.nbx + x1 + 1];
            for (int it = s; it < e; it++) {
                const int j = s_items[it];
                if (s_flag[j]) continue;
                const float2 t = s_xy[j];
                bool ok;
                if (MODE == 1) ok = (t.y >= (float)y0) && (t.y <= (float)y1);
                else {
                    const float dx = t.x - p.x, dy = t.y - p.y;
                    ok = (dx * dx + dy * dy) < a.r2;
                }
                if (!ok) continue;
                const uint64_t *tdj = s_desc + (size_t)j * 4;
                const int d = __popcll(d0 ^ tdj[0]) + __popcll(d1 ^ tdj[1]) + __popcll(d2 ^ tdj[2]) + __popcll(d3 ^ tdj[3]);
                const uint32_t key = ((uint32_t)d << 16) | (uint32_t)j;
                if (key < k1) {
                    k2 = k1;
                    k1 = key;
                } else if (key < k2)
                    k2 = key;
            }
        }
        int4 o;
        o.x = (k1 == 0xFFFFFFFFu) ? -1 : (int)(k1 & 0xFFFFu);
        o.y = (k1 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k1 >> 16);
        o.z = (k2 == 0xFFFFFFFFu) ? -1 : (int)(k2 & 0xFFFFu);
        o.w = (k2 == 0xFFFFFFFFu) ? 0x7FFFFFFF : (int)(k2 >> 16);
        out[q] = o;
    }
}

static inline size_t hamming_lds_bytes(int N, int nbins) {
    return (size_t)N * 32 + (size_t)N * 8 + (size_t)(2 * nbins + 1) * 4 + (size_t)((N + 1) & ~1) * 2 + (size_t)N + 64;
}

}  // namespace lvt
