// k_features.hip -- per-frame feature extraction on gfx950:
//   k_score   : OAST-9/16 corner score map + 9x9 box-sum map, one coalesced pass over the image rows
//   k_cells   : per detection cell: raster compaction, AGAST NMS, LVT's ANMS (handler.cpp:34-83,131-154)
//   k_gather  : concatenate cells, BRIEF border filter, RGB-D depth filter (handler.cpp:156-176,227-300)
//   k_brief   : BRIEF-256, one wavefront per key point
// Integer stages are bit-exact restatements; see DESIGN.md for the layout and roofline of each kernel.
#include "lvt_dev.h"
#include "lvt_math.h"
#include <type_traits>

namespace lvt {

__constant__ signed char c_brief[256][4] = {
#include "../../include/lvt_brief256_pattern.inc"
};
// the same table for compile-time checks (k_brief sizes its window from the largest offset)
constexpr signed char BRIEF_TABLE[256][4] = {
#include "../../include/lvt_brief256_pattern.inc"
};
constexpr int brief_max_offset() {
    int m = 0;
    for (int i = 0; i < 256; i++)
        for (int k = 0; k < 4; k++) m = (BRIEF_TABLE[i][k] > m) ? BRIEF_TABLE[i][k] : (-BRIEF_TABLE[i][k] > m) ? -BRIEF_TABLE[i][k] : m;
    return m;
}

// =================================================================================================
// k_score : OAST-9/16 score + 9x9 box sums, tile 64x16, halo 4
// =================================================================================================
constexpr int TS_W = 64, TS_H = 16, HALO = 4;
constexpr int TILE_W = TS_W + 2 * HALO;   // 72
constexpr int TILE_H = TS_H + 2 * HALO;   // 24

// two horizontally adjacent pixels at once, one in each 16-bit half (differences fit in 9 bits + sign): the packed min / max of
// VOP3P halve the instruction count of the ring arithmetic, which is what k_score spends its VALU time on
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pk_min(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s16x2 oast9_score2(const s16x2 d[16]) {
    // score = max{b : 9 contiguous ring pixels all > p+b or all < p-b} = max_arc min(+-d) - 1,  d = ring - centre
    s16x2 mn2[16], mx2[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        mn2[i] = pk_min(d[i], d[(i + 1) & 15]);
        mx2[i] = pk_max(d[i], d[(i + 1) & 15]);
    }
    s16x2 mn4[16], mx4[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        mn4[i] = pk_min(mn2[i], mn2[(i + 2) & 15]);
        mx4[i] = pk_max(mx2[i], mx2[(i + 2) & 15]);
    }
    s16x2 best_b = (s16x2)(-1000), best_d = (s16x2)(1000);  // best_d = min over arcs of max(d)  (dark score = -best_d)
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const s16x2 mn9 = pk_min(pk_min(mn4[i], mn4[(i + 4) & 15]), d[(i + 8) & 15]);
        const s16x2 mx9 = pk_max(pk_max(mx4[i], mx4[(i + 4) & 15]), d[(i + 8) & 15]);
        best_b = pk_max(best_b, mn9);
        best_d = pk_min(best_d, mx9);
    }
    const s16x2 best = pk_max(pk_max(best_b, -best_d), (s16x2)(0));
    return best - (s16x2)(1);
}
// bytes j and j + 1 (j = 0 .. 10, a compile-time constant after unrolling) of the 12-byte row segment (w0, w1, w2), widened to the two
// halves of one register: a single v_perm_b32
__device__ __forceinline__ s16x2 seg_pair(uint32_t w0, uint32_t w1, uint32_t w2, int j) {
    const uint32_t lo = (j <= 6) ? w0 : w1, hi = (j <= 6) ? w1 : w2;
    const uint32_t jj = (uint32_t)((j <= 6) ? j : j - 4);
    const uint32_t r = __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | jj | ((jj + 1u) << 16));
    return __builtin_bit_cast(s16x2, r);
}

// raw-corner key: (ly << 18) | (lx << 8) | score    (cell-local coords < 1024, score <= 254)
__device__ __forceinline__ uint32_t mk_key(int ly, int lx, int s) { return ((uint32_t)ly << 18) | ((uint32_t)lx << 8) | (uint32_t)s; }
__device__ __forceinline__ int key_y(uint32_t k) { return (int)(k >> 18); }
__device__ __forceinline__ int key_x(uint32_t k) { return (int)((k >> 8) & 1023); }
__device__ __forceinline__ int key_r(uint32_t k) { return (int)(k & 255); }
__device__ __forceinline__ uint32_t key_pos(uint32_t k) { return k >> 8; }

// per-frame, per-sequence inputs (pinned host memory read by k_feat_begin for batches; a kernel argument of k_score for
// a single sequence, which then needs no separate "begin" launch)
struct FrameArgs {
    const uint8_t *img[2];
    const float *depth;
    int img_pitch, depth_pitch;
    int ext_corners, n_ext[2];
    const float *ext_xy[2];  // external corner lists of THIS frame (a pooled handle's: every seat has its own); nullptr: the context's lists (FrameBuf::ext_xy as created)
    int absent;  // (pooled handles, lvt_host.hip) this sequence has no frame in this lock-step step: every kernel of the step leaves it exactly as it is
};

// publish this frame's inputs, clear the feature stage's control block
__device__ __forceinline__ void feat_begin(Seq &S, const FrameArgs &f, int par) {
    FrameBuf &FB = S.fb[par];
    FB.img[0] = f.img[0];
    FB.img[1] = f.img[1];
    FB.depth_img = f.depth;
    FB.img_pitch = f.img_pitch;
    FB.depth_pitch = f.depth_pitch;
    FeatCtl &c = *FB.fc;
    c.absent = f.absent;
    if (c.poison) return;  // (k_gate_buf: the buffer still belongs to an older frame)
    c.ext_corners = f.ext_corners;
    if (f.ext_corners) {  // a pooled seat's frame brings its own lists (freed with the seat: never kept beyond the frame); any other frame reads the context's
        FB.ext_xy[0] = f.ext_xy[0] ? f.ext_xy[0] : FB.ext_xy_own[0];
        FB.ext_xy[1] = f.ext_xy[0] ? f.ext_xy[1] : FB.ext_xy_own[1];
    }
    c.n_ext[0] = f.n_ext[0];
    c.n_ext[1] = f.n_ext[1];
    c.n_detected[0] = c.n_detected[1] = 0;
    c.retry[0] = c.retry[1] = 0;
    c.overflow = 0;
    // a sequence without a frame in this step rides the machinery of a frame whose features never arrive: the feature kernels leave the buffer
    // alone (poison), "no features" is published (skip_seq), the early stream stands down and the tracking chain's prologue finds the frame
    // skipped -- and, because it was ABSENT rather than late, does not even count it (frame_prologue)
    if (f.absent) c.poison = 1;
}

// BEGIN = single sequence: `fa` carries the frame's inputs, block (0, 0, 0) publishes them for the later kernels
template <bool BEGIN>
__global__ __launch_bounds__(256) void k_score(Seq *seqs, FrameArgs fa, int par, int z0, int box) {  // z0: first image of this launch (a batch's images may come in several launches); box: 0 = no box-sum plane (k_brief_img builds the sums from the image)
    const int seq = (blockIdx.z + z0) >> 1, eye = (blockIdx.z + z0) & 1;
    const Seq &S = seq_const(seqs, seq);  // (read through the constant address space: global, not flat, accesses -- lvt_dev.h; the fields written below are not read here)
    const FrameBuf &FB = S.fb[par];
    if (FB.fc->poison) return;
    if (BEGIN && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) feat_begin(seqs[seq], fa, par);
    if (eye == 1 && S.prm.sensor == 2) return;
    const int W = S.prm.W, H = S.prm.H;
    // workgroups go to the 8 XCDs round-robin by their linear id and every XCD has its own L2: with the plain mapping the four neighbours of a
    // tile -- which share its halo rows and its 128-byte lines -- sit on other XCDs, and an image was fetched 2.9 times (rocprofv3 FETCH_SIZE
    // 2.7 MB per stereo pair).  When a plane's tile count is a multiple of 8, XCD x takes the tiles [x n/8, (x + 1) n/8) of the row-major
    // order: a band of whole tile rows, whose lines are pulled into ONE L2.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int n_tiles = gridDim.x * gridDim.y;
        if ((n_tiles & 7) == 0) {
            const int lid = blockIdx.x + gridDim.x * blockIdx.y;
            const int t = (lid & 7) * (n_tiles >> 3) + (lid >> 3);
            bx = t % gridDim.x, by = t / gridDim.x;
        }
    }
    const int x0 = bx * TS_W, y0 = by * TS_H;
    if (x0 >= W || y0 >= H) return;
    const uint8_t *img = BEGIN ? fa.img[eye] : FB.img[eye];
    const int pitch = BEGIN ? fa.img_pitch : FB.img_pitch;

    // every LDS access below is a whole word (a thread owns 4 horizontally adjacent pixels, so a 12-byte row segment = 3 words holds
    // all it needs from one tile row): 35 LDS instructions per thread where byte-wise reads needed 158, and the kernel was LDS-issue-bound
    __shared__ __attribute__((aligned(16))) uint32_t tile[TILE_H][TILE_W / 4];
    __shared__ __attribute__((aligned(8))) uint2 hs[TILE_H][TS_W / 4];  // horizontal 9-sums, four u16 per entry

    const int tid = threadIdx.x;
    // ---- load tile + halo as 32-bit words (rows are 4-byte aligned: x0 % 64 == 0, pitch % 16 == 0)
    for (int idx = tid; idx < TILE_H * (TILE_W / 4); idx += 256) {
        const int r = idx / (TILE_W / 4), wc = idx % (TILE_W / 4);
        const int gy = y0 - HALO + r, gx = x0 - HALO + 4 * wc;
        uint32_t v = 0;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            if (gx + 3 < pitch) {
                v = *reinterpret_cast<const uint32_t *>(img + (size_t)gy * pitch + gx);
            } else {
                for (int b = 0; b < 4; b++)
                    if (gx + b < pitch) v |= (uint32_t)img[(size_t)gy * pitch + gx + b] << (8 * b);
            }
            // zero everything right of the image (row padding is not image content)
            if (gx + 3 >= W) {
                const int keep = W - gx;  // 1..3 valid bytes
                v &= (keep >= 4) ? 0xFFFFFFFFu : ((1u << (8 * keep)) - 1u);
            }
        }
        tile[r][wc] = v;
    }
    __syncthreads();
    // ---- horizontal 9-sums of four adjacent columns from one 12-byte segment: v_sad_u8 against 0 adds the four bytes of a word
    for (int idx = tid; box && idx < TILE_H * (TS_W / 4); idx += 256) {
        const int r = idx >> 4, j = idx & 15;
        const uint32_t w0 = tile[r][j], w1 = tile[r][j + 1], w2 = tile[r][j + 2];
        const uint32_t s0 = __builtin_amdgcn_sad_u8(w0, 0u, __builtin_amdgcn_sad_u8(w1, 0u, w2 & 255u));
        const uint32_t s1 = s0 - (w0 & 255u) + ((w2 >> 8) & 255u);
        const uint32_t s2 = s1 - ((w0 >> 8) & 255u) + ((w2 >> 16) & 255u);
        const uint32_t s3 = s2 - ((w0 >> 16) & 255u) + (w2 >> 24);
        hs[r][j] = make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
    }
    // ---- corner score.  A thread owns 4 horizontally adjacent pixels = two packed PAIRS.  Round 3: the quick test (four ring pixels) runs
    // for every pair in place, but the 16-pixel ring ladder (~150 packed instructions) only for the pairs that pass it -- ~15 % on textured
    // images -- and those are first COMPACTED into a list the workgroup's threads share out: before, a wave ran the ladder for both of its
    // pairs as soon as ONE of its 64 lanes passed the test, i.e. always (the kernel is VALU-bound: 88 us for a batch of 16).
    const int tx = tid & 15, ty = tid >> 4;
    const int gy = y0 + ty;
    const int cs = S.prm.cell_size;
    const int t_low = S.prm.agast_th_low;
    const int t_hi = S.prm.agast_th;
    // A runtime integer division is ~20 VALU instructions, and every thread did four or five of them (its row's and its pixels' cells): a fifth of the
    // kernel's issue.  x / cs is one v_mul_hi_u32 with the host's ceil(2^32 / cs) (exact while x * cs < 2^32: pixel coordinates).
    const uint32_t cmag = S.prm.cell_magic;
    auto cell_of_x = [&](int gx) -> int { return cmag ? (int)__umulhi((uint32_t)gx, cmag) : gx; };  // (cell size 1: 2^32 does not fit, the host stores 0)
    auto cell_of_y = cell_of_x;
    const int cell_x0 = cell_of_x(x0);  // cell of the tile's first pixel; at most one cell boundary inside a tile when cs >= 64
    __shared__ uint16_t s_item[2 * 256];                                    // passing pairs: thread << 1 | pair
    __shared__ __attribute__((aligned(4))) uint8_t s_sc[TS_H][TS_W];        // thresholded scores of the tile
    __shared__ int s_wcnt[4];
    const s16x2 T = (s16x2)((short)t_low);
#define LVT_RING_AT(W_, c_, dy, dx) (seg_pair(W_[3 + (dy)][0], W_[3 + (dy)][1], W_[3 + (dy)][2], (c_) + (dx)) - P)
    bool pass_q[2] = {false, false};
    {
        const int gx0 = x0 + 4 * tx;
        bool row_ok = false;
        if (gy < H) {
            const int cy = cell_of_y(gy), ly = gy - cy * cs, ch = min(cs, H - cy * cs);
            row_ok = (ly >= 3) && (ly <= ch - 4) && gx0 < W;
        }
        if (row_ok) {
            uint32_t w[7][3];  // only rows -3, 0, +3 are needed here (the others are never read: dead after unrolling)
#pragma unroll
            for (int dy = 0; dy < 7; dy += 3) {
                w[dy][0] = tile[ty + HALO - 3 + dy][tx];
                w[dy][1] = tile[ty + HALO - 3 + dy][tx + 1];
                w[dy][2] = tile[ty + HALO - 3 + dy][tx + 2];
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int c = 4 + 2 * q;  // segment byte of the pair's first centre
                const s16x2 P = seg_pair(w[3][0], w[3][1], w[3][2], c);
                const s16x2 d0 = LVT_RING_AT(w, c, 0, -3), d8 = LVT_RING_AT(w, c, 0, 3), d4 = LVT_RING_AT(w, c, -3, 0), d12 = LVT_RING_AT(w, c, 3, 0);
                // quick reject: every 9-arc contains one pixel of each opposite pair
                const s16x2 e1 = pk_min(pk_max(d0, d8), pk_max(d4, d12));   // > t_low: could be a bright corner
                const s16x2 e2 = pk_max(pk_min(d0, d8), pk_min(d4, d12));   // < -t_low: could be a dark corner
                const s16x2 pass = pk_max(e1, -e2) - T;
                pass_q[q] = pass.x > 0 || pass.y > 0;  // (a pixel that fails the test alone scores below t_low: same outcome as skipping it)
            }
        }
    }
    *reinterpret_cast<uint32_t *>(&s_sc[ty][4 * tx]) = 0u;
    {   // compaction: wave ballots give the rank inside the wave, four wave counts the base
        const int lane = tid & 63, wv = tid >> 6;
        const uint64_t m0 = __ballot(pass_q[0]), m1 = __ballot(pass_q[1]);
        const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int n0 = __popcll(m0), n1 = __popcll(m1);
        if (lane == 0) s_wcnt[wv] = n0 + n1;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) base += (k < wv) ? s_wcnt[k] : 0;
        if (pass_q[0]) s_item[base + __popcll(m0 & lt)] = (uint16_t)(tid << 1);
        if (pass_q[1]) s_item[base + n0 + __popcll(m1 & lt)] = (uint16_t)((tid << 1) | 1);
    }
    __syncthreads();
    const int n_items = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    for (int e = tid; e < n_items; e += 256) {
        const int it = s_item[e], t = it >> 1, q = it & 1;
        const int itx = t & 15, ity = t >> 4;
        const int igy = y0 + ity, gx0 = x0 + 4 * itx;
        const int cx0 = cell_of_x(gx0), lx0 = gx0 - cx0 * cs;
        uint32_t w[7][3];  // rows r0-3 .. r0+3, byte columns 4 tx .. 4 tx + 11 of the tile (pixel k's centre is byte 4 + k)
#pragma unroll
        for (int dy = 0; dy < 7; dy++) {
            w[dy][0] = tile[ity + HALO - 3 + dy][itx];
            w[dy][1] = tile[ity + HALO - 3 + dy][itx + 1];
            w[dy][2] = tile[ity + HALO - 3 + dy][itx + 2];
        }
        // the second pair's centres sit two bytes further right: its rows are shifted down by 16 bits (one v_alignbit per word, shift
        // amount 16 q), and ONE ladder with compile-time byte selectors serves both kinds of item -- no divergence inside a wave
        const uint32_t sh = 16u * (uint32_t)q;
#pragma unroll
        for (int dy = 0; dy < 7; dy++) {
            w[dy][0] = __builtin_amdgcn_alignbit(w[dy][1], w[dy][0], sh);
            w[dy][1] = __builtin_amdgcn_alignbit(w[dy][2], w[dy][1], sh);
            w[dy][2] = w[dy][2] >> sh;
        }
        s16x2 s2;
        {
            const int c = 4;
            const s16x2 P = seg_pair(w[3][0], w[3][1], w[3][2], c);
            s16x2 d[16];
            d[0] = LVT_RING_AT(w, c, 0, -3), d[8] = LVT_RING_AT(w, c, 0, 3), d[4] = LVT_RING_AT(w, c, -3, 0), d[12] = LVT_RING_AT(w, c, 3, 0);
            d[1] = LVT_RING_AT(w, c, -1, -3), d[2] = LVT_RING_AT(w, c, -2, -2), d[3] = LVT_RING_AT(w, c, -3, -1), d[5] = LVT_RING_AT(w, c, -3, 1);
            d[6] = LVT_RING_AT(w, c, -2, 2), d[7] = LVT_RING_AT(w, c, -1, 3), d[9] = LVT_RING_AT(w, c, 1, 3), d[10] = LVT_RING_AT(w, c, 2, 2);
            d[11] = LVT_RING_AT(w, c, 3, 1), d[13] = LVT_RING_AT(w, c, 3, -1), d[14] = LVT_RING_AT(w, c, 2, -2), d[15] = LVT_RING_AT(w, c, 1, -3);
            s2 = oast9_score2(d);
        }
        uint32_t two = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = 2 * q + h, sv = h ? (int)s2.y : (int)s2.x;
            int cx = cx0, lx = lx0 + k;
            if (lx >= cs) {
                const int qq = cell_of_x(lx);
                cx += qq;
                lx -= qq * cs;
            }
            const int cw = min(cs, W - cx * cs);
            if (gx0 + k < W && lx >= 3 && lx <= cw - 4 && sv >= t_low) two |= (uint32_t)sv << (8 * h);
        }
        (void)igy;
        *reinterpret_cast<uint16_t *>(&s_sc[ity][4 * itx + 2 * q]) = (uint16_t)two;
    }
#undef LVT_RING_AT
    __syncthreads();
    // back in place: the thread's four thresholded scores; raw corners of pass 0 (score >= agast_th) as cell-local keys -- k_cells gathers
    // them per (row, tile) segment instead of re-reading the score map with one CU per cell
    const uint32_t packed = *reinterpret_cast<const uint32_t *>(&s_sc[ty][4 * tx]);
    uint32_t ckey[4] = {0, 0, 0, 0};
    uint32_t cvalid = 0;  // bit k: pixel k of this thread is a pass-0 corner
    int ncl = 0;
    if (gy < H && packed != 0) {
        const int cy = cell_of_y(gy), ly = gy - cy * cs;
        const int gx0 = x0 + 4 * tx;
        const int cx0 = cell_of_x(gx0), lx0 = gx0 - cx0 * cs;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int sv = (int)((packed >> (8 * k)) & 255u);
            if (sv >= t_hi) {
                int cx = cx0, lx = lx0 + k;
                if (lx >= cs) {
                    const int qq = cell_of_x(lx);
                    cx += qq;
                    lx -= qq * cs;
                }
                ckey[k] = mk_key(ly, lx, sv);
                cvalid |= 1u << k;
                ncl += (cx == cell_x0) ? 1 : 0;
            }
        }
    }
    {  // the 16 lanes of one DPP row hold one tile row, x-ascending: a row scan of the counts places the keys
        const int nc = __popc(cvalid);
        const int incl = row16_incl_scan(nc | (ncl << 8));
        const int excl = (incl & 0xFF) - nc;
        if (gy < H) {
            const size_t seg = (size_t)gy * gridDim.x + bx;
            uint32_t *dst = FB.seg_keys[eye] + seg * TS_W + excl;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((cvalid >> k) & 1u) dst[__popc(cvalid & ((1u << k) - 1u))] = ckey[k];
            if (tx == 15) FB.seg_cnt[eye][seg] = (uint16_t)incl;  // count | left-of-boundary count << 8
        }
    }
    __syncthreads();  // hs complete
    if (gy < H) {
        const int pp = S.plane_pitch;
        *reinterpret_cast<uint32_t *>(FB.score[eye] + (size_t)gy * pp + x0 + 4 * tx) = packed;
        if (box) {
            uint2 o = make_uint2(0u, 0u);  // four u16 sums; 81 * 255 < 2^16, so whole-word adds never carry between the halves
#pragma unroll
            for (int d = 0; d < 9; d++) {
                const uint2 h = hs[ty + d][tx];
                o.x += h.x;
                o.y += h.y;
            }
            *reinterpret_cast<uint2 *>(FB.boxsum[eye] + (size_t)gy * pp + x0 + 4 * tx) = o;
        }
    }
}

// =================================================================================================
// k_cells : one 1024-thread workgroup per (cell, eye, sequence)
// =================================================================================================
// index traits: 16-bit indices while a cell's corners fit in LDS, 32-bit in the global-memory path
template <typename I> struct IdxT;
template <> struct IdxT<uint16_t> {
    static constexpr uint32_t NONE = 0x3FFFu, LEFT = 0x8000u, MAXF = 0xFFFFu;
};
template <> struct IdxT<uint32_t> {
    static constexpr uint32_t NONE = 0x3FFFFFFFu, LEFT = 0x80000000u, MAXF = 0xFFFFFFFFu;
};

// ---- libstdc++ std::sort emulation pieces (comp(a,b) := resp(a) > resp(b); handler.cpp:38-41) ----
__device__ __forceinline__ bool scomp(uint32_t a, uint32_t b) { return key_r(a) > key_r(b); }

// heap fallback of introsort (std::__partial_sort(first,last,last)), sequential, one lane
__device__ __forceinline__ void heap_adjust(uint32_t *f, int hole, int len, uint32_t value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (scomp(f[child], f[child - 1])) child--;
        f[hole] = f[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        f[hole] = f[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && scomp(f[parent], value)) {
        f[hole] = f[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    f[hole] = value;
}
__device__ __forceinline__ void heap_sort_seq(uint32_t *f, int len) {
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            uint32_t v = f[parent];
            heap_adjust(f, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    int last = len;
    while (last > 1) {
        --last;
        uint32_t v = f[last];
        f[last] = f[0];
        heap_adjust(f, 0, last, v);
    }
}

// std::__unguarded_partition_pivot on arr[first,last) executed by ONE full wavefront.
// Hoare partition evaluated with ballots: the k-th "left stop" (ascending) swaps with the k-th
// "right stop" (descending) while they have not crossed (see DESIGN.md "std::sort emulation").
// Only this wavefront touches arr[first,last) and its slice of posL / posR, and a wavefront's LDS operations execute in program
// order: no fences between the steps when the arrays live in LDS (16-bit indices).  The global-memory path (32-bit indices) keeps them.
template <typename I>
__device__ __forceinline__ void part_fence() {
    if (sizeof(I) == 4) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
}
template <typename I>
__device__ __forceinline__ int wave_partition_pivot(uint32_t *arr, int first, int last, I *posL, I *posR) {
    const int lane = lane_id();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int mid = first + (last - first) / 2;
    int piv;
    {  // std::__move_median_to_first(first, first+1, mid, last-1): the four words are read by every lane at once (one round trip)
        const int a = first + 1, b = mid, c = last - 1;
        const uint32_t va = arr[a], vb = arr[b], vc = arr[c], vf = arr[first];
        int pick;
        if (scomp(va, vb)) {
            if (scomp(vb, vc)) pick = b;
            else if (scomp(va, vc)) pick = c;
            else pick = a;
        } else if (scomp(va, vc)) pick = a;
        else if (scomp(vb, vc)) pick = c;
        else pick = b;
        const uint32_t vp = (pick == a) ? va : (pick == b) ? vb : vc;
        if (lane == 0) {
            arr[first] = vp;
            arr[pick] = vf;
        }
        piv = key_r(vp);
    }
    part_fence<I>();
    const int lo = first + 1, hi = last;
    int cntL = 0, cntR = 0;
    // the left-stop scan (ascending) and the right-stop scan (descending) are independent: both run in the same trip, two 64-element
    // groups each, so four LDS reads are in flight where the one-group-per-trip loops waited for one (the scans are latency-bound)
    for (int base = lo, top = hi; base < hi; base += 128, top -= 128) {
        const int p0 = base + lane, p1 = p0 + 64;
        const int q0 = top - 1 - lane, q1 = q0 - 64;
        // (clamped addresses, unconditional loads: a load under a branch gets its own wait and the four would run one after the other)
        const uint32_t va0 = arr[min(p0, hi - 1)], va1 = arr[min(p1, hi - 1)], vb0 = arr[max(q0, lo)], vb1 = arr[max(q1, lo)];
        const int a0 = (p0 < hi) ? key_r(va0) : 0x7FFF, a1 = (p1 < hi) ? key_r(va1) : 0x7FFF;
        const int b0 = (q0 >= lo) ? key_r(vb0) : -1, b1 = (q1 >= lo) ? key_r(vb1) : -1;
        const bool l0 = !(a0 > piv), l1 = !(a1 > piv);  // (the out-of-range fillers never stop a scan)
        const bool r0 = !(piv > b0), r1 = !(piv > b1);
        const uint64_t ml0 = __ballot(l0), ml1 = __ballot(l1), mr0 = __ballot(r0), mr1 = __ballot(r1);
        if (l0) posL[first + cntL + __popcll(ml0 & lt_mask)] = (I)p0;
        cntL += __popcll(ml0);
        if (l1) posL[first + cntL + __popcll(ml1 & lt_mask)] = (I)p1;
        cntL += __popcll(ml1);
        if (r0) posR[first + cntR + __popcll(mr0 & lt_mask)] = (I)q0;
        cntR += __popcll(mr0);
        if (r1) posR[first + cntR + __popcll(mr1 & lt_mask)] = (I)q1;
        cntR += __popcll(mr1);
    }
    part_fence<I>();
    const int K = min(cntL, cntR);
    int m = 0;
    for (int base = 0; base < K; base += 64) {
        const int k = base + lane;
        const bool ok = (k < K) && (posL[first + k] < posR[first + k]);
        const uint64_t b = __ballot(ok);
        m += __popcll(b);
        if (b != ~0ull) break;  // monotone: once a pair has crossed, all later ones have
    }
    for (int k = lane; k < m; k += 64) {
        const int p = (int)posL[first + k], q = (int)posR[first + k];
        const uint32_t t = arr[p];
        arr[p] = arr[q];
        arr[q] = t;
    }
    const int pm = (m < cntL) ? (int)posL[first + m] : 0x7FFFFFFF;
    const int qm1 = (m > 0) ? (int)posR[first + m - 1] : last;
    part_fence<I>();
    return min(pm, qm1);
}

// std::__introsort_loop on arr[first,last) with `depth` splits left, depth-first by ONE wavefront (partitions only; the final insertion
// sort is a stable sort and is done afterwards by ranking).  `stack` = 3 ints per pending segment: at most one per halving of depth.
template <typename I>
__device__ __forceinline__ void wave_introsort_segment(uint32_t *arr, int first, int last, int depth, I *posL, I *posR, int *stack) {
    int sp = 0;
    while (true) {
        while (last - first > 16) {
            if (depth == 0) {
                if (lane_id() == 0) heap_sort_seq(arr + first, last - first);
                part_fence<I>();
                break;
            }
            --depth;
            const int cut = wave_partition_pivot<I>(arr, first, last, posL, posR);
            if (last - cut > 16) {  // the recursive call introsort_loop(cut, last, depth)
                if (lane_id() == 0) {
                    stack[3 * sp] = cut;
                    stack[3 * sp + 1] = last;
                    stack[3 * sp + 2] = depth;
                }
                sp++;
            }
            last = cut;
        }
        if (sp == 0) break;
        sp--;
        part_fence<I>();
        first = stack[3 * sp];
        last = stack[3 * sp + 1];
        depth = stack[3 * sp + 2];
    }
}

// LDS carve (bytes): keys 4*RAW | uf 4*RAW | root 2*RAW | abv 2*RAW | nms 2*RAW | row_first/row_end | misc
constexpr int cells_lds_bytes(int raw_cap) { return raw_cap * 15 + 2 * 1024 * 4 + 64 * 4 + 3 * 64 * 4 + 64; }
constexpr int CELLS_LDS_BYTES = cells_lds_bytes(RAW_CAP);
constexpr int RAW_CAP_SMALL = 4700;  // 80 KB: two k_cells workgroups per CU (lvt_host.hip, Context::cells_raw_cap)

__device__ __forceinline__ int uf_find(volatile uint32_t *parent, int i) {
    while (true) {
        const int p = (int)parent[i];
        if (p == i) return i;
        i = p;
    }
}

struct CellGeom {
    int X0, Y0, cw, ch, threshold, pp;
    const uint8_t *score;
};

// raster-order compaction of the cell's raw corners into keys[0..cap); returns the true count.
// Every thread owns a CONTIGUOUS run of 16-pixel chunks (so one block scan orders the whole cell); the chunks
// are read twice (count, then emit) -- the second read hits L2.
// Thresholds up to 128 (every shipped configuration: 25 / 12) take the byte-parallel path: the four scores of a word are
// compared at once (s >= th  <=>  bit 7 of (s & 0x7F) + (0x80 - th), or bit 7 of s itself), so a 16-pixel chunk costs ~25
// instructions to count and a find-first-set walk over its corners to emit, instead of ~100 + ~160 for the scalar loops
// -- one CU runs all 16 wavefronts of the cell, so this phase is VALU-bound.
__device__ __forceinline__ int cell_compact(const CellGeom &g, uint32_t *keys, int cap, int *scan, long long *dbg = nullptr, int ly_off = 0) {
    const int tid = threadIdx.x;
    const int xa = g.X0 + 3, xb = g.X0 + g.cw - 4;  // inclusive pixel range
    const int c0 = xa >> 4, c1 = xb >> 4;
    const int nchunk = c1 - c0 + 1;
    const int nrows = g.ch - 6;
    const int items = nrows * nchunk;
    const int per = (items + 1023) / 1024;
    const int it0 = tid * per, it1 = min(items, it0 + per);
    const bool swar = g.threshold >= 1 && g.threshold <= 128;
    const uint32_t addk = (uint32_t)(0x80 - g.threshold) * 0x01010101u;
    // bit 7 of byte b of the result <=> pixel gx0 + 4 * word + b is a corner of this cell at this threshold
    auto mask4 = [&](uint32_t w, int gx) -> uint32_t {
        uint32_t m = ((((w & 0x7F7F7F7Fu) + addk) | w) & 0x80808080u);
        if (gx < xa || gx + 3 > xb) {  // the cell's first / last chunk only
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (gx + b < xa || gx + b > xb) m &= ~(0x80u << (8 * b));
        }
        return m;
    };
    auto count16 = [&](const uint4 &v, int gx0) -> int {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        int cnt = 0;
        if (swar) {
#pragma unroll
            for (int k = 0; k < 4; k++) cnt += __popc(mask4(w[k], gx0 + 4 * k));
            return cnt;
        }
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const int s = (w[b >> 2] >> (8 * (b & 3))) & 255;
            const int gx = gx0 + b;
            cnt += (s >= g.threshold && gx >= xa && gx <= xb) ? 1 : 0;
        }
        return cnt;
    };
    // up to 4 chunks per thread stay in registers between the counting and the emitting pass (KITTI: 4, EuRoC: 2)
    constexpr int KEEP = 4;
    uint4 kv[KEEP];
    int cnt = 0;
    if (per <= KEEP) {
#pragma unroll
        for (int u = 0; u < KEEP; u++) {
            const int it = min(it0 + u, items - 1);
            const int ly = 3 + it / nchunk, gx0 = (c0 + it % nchunk) << 4;
            kv[u] = *reinterpret_cast<const uint4 *>(g.score + (size_t)(g.Y0 + ly) * g.pp + gx0);
        }
#pragma unroll
        for (int u = 0; u < KEEP; u++) {
            const int it = it0 + u;
            if (it < it1) cnt += count16(kv[u], (c0 + it % nchunk) << 4);
        }
    } else {
#pragma unroll 4
        for (int it = it0; it < it1; it++) {
            const int ly = 3 + it / nchunk, gx0 = (c0 + it % nchunk) << 4;
            const uint4 v = *reinterpret_cast<const uint4 *>(g.score + (size_t)(g.Y0 + ly) * g.pp + gx0);
            cnt += count16(v, gx0);
        }
    }
    int total;
    int off = block_excl_scan(cnt, scan, &total);
    auto emit16 = [&](const uint4 &v, int ly, int gx0) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if ((w[0] | w[1] | w[2] | w[3]) == 0) return;
        if (swar) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t m = mask4(w[k], gx0 + 4 * k);
                while (m) {  // ascending x: lowest set byte first
                    const int b = (__ffs((int)m) - 1) >> 3;
                    m &= m - 1;
                    const int s = (w[k] >> (8 * b)) & 255;
                    if (off < cap) keys[off] = mk_key(ly + ly_off, gx0 + 4 * k + b - g.X0, s);
                    off++;
                }
            }
            return;
        }
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const int s = (w[b >> 2] >> (8 * (b & 3))) & 255;
            const int gx = gx0 + b;
            if (s >= g.threshold && gx >= xa && gx <= xb) {
                if (off < cap) keys[off] = mk_key(ly + ly_off, gx - g.X0, s);
                off++;
            }
        }
    };
    if (cnt) {
        if (per <= KEEP) {
#pragma unroll
            for (int u = 0; u < KEEP; u++) {
                const int it = it0 + u;
                if (it < it1) emit16(kv[u], 3 + it / nchunk, (c0 + it % nchunk) << 4);
            }
        } else {
            for (int it = it0; it < it1; it++) {
                const int ly = 3 + it / nchunk, gx0 = (c0 + it % nchunk) << 4;
                emit16(*reinterpret_cast<const uint4 *>(g.score + (size_t)(g.Y0 + ly) * g.pp + gx0), ly, gx0);
            }
        }
    }
    __syncthreads();
    return total;
}

// Pass 0: the raw corners were already extracted by k_score (all CUs) as one x-ascending list per (image row, 64-px tile);
// the cell's raster-ordered list is the concatenation of its (row, tile) segments: one scan over the segment counts and a
// copy.  Returns the number of raw corners; copies nothing when they do not fit `cap` (caller: global-memory path).
__device__ __forceinline__ int cell_gather_segments(const FrameBuf &FB, int eye, const CellGeom &g, int cs, int cell_x, int tiles_x, uint32_t *keys,
                                                    int cap, int *scan) {
    const int tid = threadIdx.x;
    const int t0 = g.X0 / TS_W, t1 = (g.X0 + g.cw - 1) / TS_W, nt = t1 - t0 + 1;
    const int nrows = g.ch - 6;
    const int items = nrows * nt;
    const int per = (items + 1023) / 1024;
    const int it0 = min(tid * per, items), it1 = min(items, it0 + per);
    const uint16_t *cnt = FB.seg_cnt[eye];
    const uint32_t *sk = FB.seg_keys[eye];
    auto range_of = [&](int it, int &start, int &n) -> size_t {
        const int r = it / nt, t = t0 + (it - r * nt);
        const size_t seg = (size_t)(g.Y0 + 3 + r) * tiles_x + t;
        const int c = cnt[seg], total = c & 0xFF, left = c >> 8;
        const int cl = (t * TS_W) / cs, cr = (t * TS_W + TS_W - 1) / cs;  // cells of the tile's first / last pixel
        if (cl == cr) start = 0, n = (cl == cell_x) ? total : 0;
        else if (cl == cell_x) start = 0, n = left;
        else start = left, n = (cr == cell_x) ? total - left : 0;
        return seg;
    };
    int sum = 0;
    for (int it = it0; it < it1; it++) {
        int st, n;
        range_of(it, st, n);
        sum += n;
    }
    int total;
    int off = block_excl_scan(sum, scan, &total);
    if (total <= cap) {
        for (int it = it0; it < it1; it++) {
            int st, n;
            const size_t seg = range_of(it, st, n);
            const uint32_t *src = sk + seg * TS_W + st;
            for (int k = 0; k < n; k++) keys[off + k] = src[k];
            off += n;
        }
    }
    __syncthreads();
    return total;
}

// AGAST NMS + LVT ANMS of one cell on arrays that live either in LDS (I = u16) or in global scratch
// (I = u32).  Writes the cell's key points to `out` and returns how many.
#define STAMP(k) do { if (dbg && threadIdx.x == 0) dbg[k] = clock64(); } while (0)
// first half: AGAST's NMS.  keys[0..n_raw) = raw corners in raster order; on return uf[0..n_kp) (= `arr`) holds the survivors in raster
// order, keys / root are intact (root[i] = representative of corner i's 4-connected component).
template <typename I>
__device__ __forceinline__ int cell_nms(uint32_t *keys, uint32_t *uf, I *root, I *abv, I *nms, uint8_t *tie, int n_raw, int *row_first, int *row_end, int *scan,
                                        long long *dbg) {
    constexpr uint32_t NONE = IdxT<I>::NONE, LEFT = IdxT<I>::LEFT, MAXF = IdxT<I>::MAXF;
    const int tid = threadIdx.x;
    STAMP(2);
    // ---------------- neighbour links, union-find over 4-connected corner pixels
    for (int i = tid; i < 1024; i += 1024) {
        row_first[i] = -1;
        row_end[i] = 0;
    }
    __syncthreads();
    for (int i = tid; i < n_raw; i += 1024) {
        const int y = key_y(keys[i]);
        if (i == 0 || key_y(keys[i - 1]) != y) row_first[y] = i;
        if (i == n_raw - 1 || key_y(keys[i + 1]) != y) row_end[y] = i + 1;
        uf[i] = (uint32_t)i;
        nms[i] = (I)MAXF;
    }
    __syncthreads();
    for (int i = tid; i < n_raw; i += 1024) {
        const uint32_t k = keys[i];
        const int y = key_y(k), x = key_x(k);
        const bool left = (i > 0) && (key_pos(keys[i - 1]) + 1 == key_pos(k));
        uint32_t above = NONE;
        if (y > 0 && row_first[y - 1] >= 0) {
            int lo = row_first[y - 1], hi = row_end[y - 1];  // [lo,hi)
            const int end = hi;
            while (lo < hi) {
                const int m = (lo + hi) >> 1;
                const int mx = key_x(keys[m]);
                if (mx < x) lo = m + 1;
                else hi = m;
            }
            if (lo < end && key_x(keys[lo]) == x) above = (uint32_t)lo;
        }
        abv[i] = (I)(above | (left ? LEFT : 0u));
    }
    __syncthreads();
    for (int i = tid; i < n_raw; i += 1024) {
        const uint32_t a = abv[i];
        int nb[2];
        int nn = 0;
        if (a & LEFT) nb[nn++] = i - 1;
        if ((a & NONE) != NONE) nb[nn++] = (int)(a & NONE);
        for (int e = 0; e < nn; e++) {
            int u = i, v = nb[e];
            while (true) {
                u = uf_find(uf, u);
                v = uf_find(uf, v);
                if (u == v) break;
                if (u < v) {
                    const int t = u;
                    u = v;
                    v = t;
                }  // u > v : hang u under v
                const uint32_t old = atomicMin(&uf[u], (uint32_t)v);
                if (old == (uint32_t)u) break;
                u = (int)old;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < n_raw; i += 1024) root[i] = (I)uf_find(uf, i);
    __syncthreads();
    // ---- component maximum: a component whose maximum response is attained ONCE keeps exactly that pixel
    // (every merge of the sweep lets the larger response win), no replay needed.  uf[root] := max(resp, -index)
    for (int i = tid; i < n_raw; i += 1024) {
        uf[i] = 0;
        tie[i] = 0;
    }
    __syncthreads();
    for (int i = tid; i < n_raw; i += 1024)
        atomicMax(&uf[(int)root[i]], ((uint32_t)key_r(keys[i]) << 24) | (0xFFFFFFu - (uint32_t)i));
    __syncthreads();
    for (int i = tid; i < n_raw; i += 1024) {  // pass 1 over ALL pixels: is the component's maximum tied?
        const int r = (int)root[i];
        const uint32_t m = uf[r];
        const int argmax = (int)(0xFFFFFFu - (m & 0xFFFFFFu));
        if ((uint32_t)key_r(keys[i]) == (m >> 24) && i != argmax) tie[r] = 1;  // same value from every writer
    }
    __syncthreads();
    for (int i = tid; i < n_raw; i += 1024) {  // pass 2: unique maximum -> everybody else is suppressed by it
        const int r = (int)root[i];
        const int argmax = (int)(0xFFFFFFu - (uf[r] & 0xFFFFFFu));
        if (i != argmax && !tie[r]) nms[i] = (I)argmax;
    }
    __syncthreads();
    for (int i = tid; i < n_raw; i += 1024) uf[i] = (uint32_t)i;  // uf[root] := last member of the component
    __syncthreads();
    for (int i = tid; i < n_raw; i += 1024) {
        const int r = (int)root[i];
        if (r != i && tie[r]) atomicMax(&uf[r], (uint32_t)i);
    }
    __syncthreads();
    // uf[non-root member] := next member of its component in raster order (roots keep "last"); tied components only
    for (int base = 0; base < n_raw; base += 1024) {
        const int i = base + tid;
        uint32_t link = 0;
        bool wr = false;
        if (i < n_raw) {
            const int r = (int)root[i];
            if (r != i && tie[r]) {
                wr = true;
                const int lastm = (int)uf[r];
                link = (uint32_t)i;  // self = end of chain
                if (i != lastm) {
                    int j = i + 1;
                    while ((int)root[j] != r) j++;
                    link = (uint32_t)j;
                }
            }
        }
        __syncthreads();
        if (wr) uf[i] = link;
    }
    __syncthreads();

    STAMP(3);
    // ---------------- exact replay of AGAST's NMS sweep inside every multi-pixel component.  Path compression
    // in the "find the maximum of that area" walks changes pointers of suppressed corners only, never a decision.
    auto find_max = [&](int w) -> int {
        int r = w;
        while ((uint32_t)nms[r] != MAXF) r = (int)nms[r];
        while (w != r) {
            const int nx = (int)nms[w];
            nms[w] = (I)r;
            w = nx;
        }
        return r;
    };
    for (int i = tid; i < n_raw; i += 1024) {
        if ((int)root[i] != i || !tie[i]) continue;  // only components whose maximum is tied need the sweep
        const int lastm = (int)uf[i];
        int cur = i + 1;
        while ((int)root[cur] != i) cur++;  // first successor of the root
        while (true) {                        // the root itself has no above/left neighbour: nothing to do for it
            const uint32_t a = abv[cur];
            const int rc = key_r(keys[cur]);
            if ((a & NONE) != NONE) {
                const int w = find_max((int)(a & NONE));
                if (rc < key_r(keys[w])) nms[cur] = (I)w;
                else nms[w] = (I)cur;
            }
            if (a & LEFT) {
                const uint32_t above_root = (uint32_t)nms[cur];
                const int t = find_max(cur - 1);
                if (above_root == MAXF) {
                    if (t != cur) {
                        if (rc < key_r(keys[t])) nms[cur] = (I)t;
                        else nms[t] = (I)cur;
                    }
                } else if (t != (int)above_root) {
                    if (key_r(keys[above_root]) < key_r(keys[t])) {
                        nms[above_root] = (I)t;
                        nms[cur] = (I)t;
                    } else {
                        nms[t] = (I)above_root;
                        nms[cur] = (I)above_root;
                    }
                }
            }
            if (cur == lastm) break;
            cur = (int)uf[cur];
        }
    }
    __syncthreads();

    STAMP(4);
    // ---------------- survivors, raster order -> arr (aliases uf)
    uint32_t *arr = uf;
    int n_kp = 0;
    for (int base = 0; base < n_raw; base += 1024) {
        const int i = base + tid;
        const bool keep = (i < n_raw) && ((uint32_t)nms[i] == MAXF);
        const uint32_t k = keep ? keys[i] : 0;
        int total;
        const int off = n_kp + block_excl_scan(keep ? 1 : 0, scan, &total);
        if (keep) arr[off] = k;  // off <= i, and uf[] is dead after the replay
        n_kp += total;
    }
    __syncthreads();

    STAMP(5);
    return n_kp;
}

// second half: LVT's ANMS when the cell is too dense (handler.cpp:140-143, 34-83), else the survivors as they are.  arr = uf[0..n_kp) in
// raster order; keys / root / abv / nms are scratch of n_cap elements each.
// `parts`: 1 = std::sort emulation + rank (-> sorted[] in keys), 2 = suppression radii of the key points i_first, i_first + i_stride, ...
// (sorted[] in keys -> r2[] in uf), 4 = decision radius + emit (sorted[] in keys, r2[] in uf).  7 = the whole thing on one workgroup;
// the oversized-cell kernels run the parts in three launches, the radii on several workgroups (they are n^2 / 2 distances).
constexpr int ANMS_SORT = 1, ANMS_RADII = 2, ANMS_SELECT = 4, ANMS_ALL = 7;
template <typename I>
__device__ __forceinline__ int cell_anms(const Seq &S, const CellGeom &g, uint32_t *keys, uint32_t *uf, I *root, I *abv, I *nms, int n_kp, int n_cap, int *row_first,
                                         int *row_end, int *scan, int *misc, float *out, long long *dbg, int n_raw_dbg, int parts = ANMS_ALL, int i_first = 0,
                                         int i_stride = 1) {
    constexpr uint32_t NONE = IdxT<I>::NONE, LEFT = IdxT<I>::LEFT, MAXF = IdxT<I>::MAXF;
    (void)NONE, (void)LEFT, (void)MAXF;
    const int tid = threadIdx.x;
    uint32_t *arr = uf;
    const int n_raw = n_raw_dbg;
    // ---------------- ANMS when the cell is too dense (handler.cpp:140-143, 34-83)
    const int max_kp = S.prm.max_kp_cell;
    const float fX0 = (float)g.X0, fY0 = (float)g.Y0;
    int n_out = 0;
    if (n_kp > max_kp) {
        I *posL = root;  // free after NMS
        I *posR = abv;
        uint32_t *sorted = keys;  // raw keys no longer needed
        if (parts & ANMS_SORT) {
        // ---- std::__introsort_loop.  The segments of one recursion level are disjoint: the first levels run level by level, one
        // wavefront per segment with a workgroup barrier between levels; as soon as a level holds one segment per wavefront each
        // wavefront finishes its segments depth-first on its own (no more barriers, nobody waits for the level's largest segment).
        // Queues and stacks live in the dead nms[] array.
        {
            int *q = reinterpret_cast<int *>(nms);
            constexpr int DFS_AT = 64;  // (also bounds the queues: a level never holds more than 2 * DFS_AT segments)
            constexpr int qcap = 2 * DFS_AT;  // per level (a level is handed over to the depth-first part at DFS_AT segments)
            static_assert((6 * qcap + 16 * 96) * sizeof(int) <= (size_t)RAW_CAP_SMALL * sizeof(uint16_t), "queues + stacks must fit the nms[] array of the smallest LDS instance");
            int *qa = q, *qb = q + 3 * qcap;
            int *stacks = q + 6 * qcap;  // [16 wavefronts][3 * 32]
            const int nw = blockDim.x >> 6;
            if (tid == 0) {
                int depth0 = 0;
                for (int v = n_kp; v > 1; v >>= 1) depth0++;
                qa[0] = 0, qa[1] = n_kp, qa[2] = 2 * depth0;
                misc[1] = (n_kp > 16) ? 1 : 0;
                misc[2] = 0;
            }
            __syncthreads();
            int levels = 0;
            while (true) {
                const int cnt = misc[1];
                if (cnt == 0) break;
                levels++;
                if (cnt >= DFS_AT) {  // depth-first from here
                    for (int sgi = wave_id(); sgi < cnt; sgi += nw)
                        wave_introsort_segment<I>(arr, qa[3 * sgi], qa[3 * sgi + 1], qa[3 * sgi + 2], posL, posR, stacks + 96 * wave_id());
                    break;
                }
                for (int sgi = wave_id(); sgi < cnt; sgi += nw) {
                    const int first = qa[3 * sgi], last = qa[3 * sgi + 1], depth = qa[3 * sgi + 2];
                    if (depth == 0) {
                        if (lane_id() == 0) heap_sort_seq(arr + first, last - first);
                        continue;
                    }
                    const int cut = wave_partition_pivot<I>(arr, first, last, posL, posR);
                    if (lane_id() == 0) {
                        if (last - cut > 16) {
                            const int o = atomicAdd(&misc[2], 1);
                            qb[3 * o] = cut, qb[3 * o + 1] = last, qb[3 * o + 2] = depth - 1;
                        }
                        if (cut - first > 16) {
                            const int o = atomicAdd(&misc[2], 1);
                            qb[3 * o] = first, qb[3 * o + 1] = cut, qb[3 * o + 2] = depth - 1;
                        }
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    misc[1] = misc[2];
                    misc[2] = 0;
                }
                int *t = qa;
                qa = qb;
                qb = t;
                __syncthreads();
            }
            __syncthreads();
            if (dbg && tid == 0) dbg[41] = levels;
        }
        STAMP(6);
        // ---- final insertion sort == stable sort by response (descending).  rank = #(greater response) +
        // #(equal response earlier in the array); the second term comes from one wavefront walking the array in
        // 64-element steps with a running 256-bin histogram, equal keys inside a step grouped by 8 ballots.
        // More than 512 key points: every wavefront walks ITS block of the array with a histogram of its own (in the dead nms[] area), the
        // blocks' counts are prefix-summed per key afterwards -- the single-wavefront walk over 1 650 key points took 17k cycles.
        int *hist = row_first;    // [256] (row tables are dead)
        int *gtab = row_end;      // [256] #elements with a strictly greater response
        const int nwv = blockDim.x >> 6;
        // (the per-wavefront histograms live in the dead nms[] array: [nwv][256] ints -- with the small LDS instance of a batch, RAW_CAP_SMALL, that array is too short)
        const bool par_rank = n_kp > 512 && (size_t)n_cap * sizeof(I) >= (size_t)nwv * 256 * sizeof(int);
        const int blk = par_rank ? (((n_kp + nwv - 1) / nwv + 63) & ~63) : n_kp;  // elements per wavefront (a multiple of 64)
        int *hw = par_rank ? reinterpret_cast<int *>(nms) : hist;                  // [nwv][256] | [256]
        if (par_rank) {
            for (int k = tid; k < nwv * 256; k += 1024) hw[k] = 0;
            __syncthreads();
        }
        if (par_rank || wave_id() == 0) {
            const int lane = lane_id();
            const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            int *myh = par_rank ? hw + 256 * wave_id() : hist;
            if (!par_rank) {
                for (int k = lane; k < 256; k += 64) hist[k] = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            }
            const int b_lo = par_rank ? wave_id() * blk : 0, b_hi = min(n_kp, b_lo + blk);
            for (int base = b_lo; base < b_hi; base += 64) {
                const int i = base + lane;
                const bool valid = i < b_hi;
                const int key = valid ? key_r(arr[i]) : 0;
                uint64_t m = __ballot(valid);
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    const uint64_t bb = __ballot((key >> b) & 1);
                    m &= ((key >> b) & 1) ? bb : ~bb;
                }
                if (valid) {
                    const int hb = myh[key];
                    posL[i] = (I)(hb + __popcll(m & lt_mask));
                    if ((m & lt_mask) == 0ull) myh[key] = hb + __popcll(m);  // group leader
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            }
        }
        if (par_rank) {  // per key: the blocks' counts -> exclusive prefix over the blocks (in place), total -> hist
            __syncthreads();
            if (tid < 256) {
                int run = 0;
                for (int w = 0; w < nwv; w++) {
                    const int t = hw[256 * w + tid];
                    hw[256 * w + tid] = run;
                    run += t;
                }
                hist[tid] = run;
            }
            __syncthreads();
        }
        if (wave_id() == 0) {
            const int lane = lane_id();
            // suffix sums: gtab[k] = sum_{k' > k} hist[k']
            int h[4], tot = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                h[q] = hist[4 * lane + q];
                tot += h[q];
            }
            int incl = tot;  // inclusive suffix scan over lanes
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_down(incl, d, 64);
                if (lane + d < 64) incl += t;
            }
            int above = incl - tot;
#pragma unroll
            for (int q = 3; q >= 0; q--) {
                gtab[4 * lane + q] = above;
                above += h[q];
            }
        }
        __syncthreads();
        for (int i = tid; i < n_kp; i += 1024) {
            const uint32_t k = arr[i];
            const int kr = key_r(k);
            sorted[gtab[kr] + (par_rank ? hw[256 * (i / blk) + kr] : 0) + (int)posL[i]] = k;
        }
        __syncthreads();
        }
        STAMP(7);
        if (!(parts & (ANMS_RADII | ANMS_SELECT))) return n_kp;
        // ---- suppression radius^2 (integers: exact in the reference's float arithmetic too).  sorted[] is descending, so the points
        // stronger than 1.11 * response are a prefix [0, lo) whose length a binary search finds, and lo never decreases along the
        // array.  One lane per key point; the lanes of a wave walk the prefix TOGETHER (every lane reads the same word: an LDS
        // broadcast), coordinates packed as two 16-bit halves so that a distance is one packed subtract and one dot product.  The
        // phase is VALU-bound on the cell's single CU (n^2 / 2 distances): 3 instructions per distance where the scalar form had 8.
        uint32_t *r2 = arr;  // arr consumed
        if (parts & ANMS_RADII) {
        uint32_t *sxy = reinterpret_cast<uint32_t *>(root);  // posL (and, with 16-bit indices, the adjacent posR) are dead: x | y << 16
        for (int i = tid; i < n_kp; i += 1024) {
            const uint32_t k = sorted[i];
            sxy[i] = (uint32_t)key_x(k) | ((uint32_t)key_y(k) << 16);
        }
        __syncthreads();
        // this workgroup's key points: i_first + k * i_stride, k = 0 .. n_pts - 1 (all of them when the cell has one workgroup)
        const int n_pts = (n_kp > i_first) ? (n_kp - i_first + i_stride - 1) / i_stride : 0;
        // few key points: 2, 4 or 8 lanes share one (its prefix in interleaved groups of four), so that all 16 wavefronts have work
        auto radii = [&](auto lpp_log_c) {
            constexpr int lpp_log = decltype(lpp_log_c)::value, lpp = 1 << lpp_log, ppp = 1024 >> lpp_log, step = 4 * lpp;
            for (int base = 0; base < n_pts; base += ppp) {
                const int k = base + (tid >> lpp_log), sub = tid & (lpp - 1);
                const int i = i_first + k * i_stride;
                const bool valid = k < n_pts;
                int lo = 0;
                if (valid) {
                    const float response = (float)key_r(sorted[i]) * 1.11f;
                    int hi = i;  // first index in [0,i) whose response is NOT > `response`
                    while (lo < hi) {
                        const int m = (lo + hi) >> 1;
                        if ((float)key_r(sorted[m]) > response) lo = m + 1;
                        else hi = m;
                    }
                }
                const int wave_first = base + ((tid & ~63) >> lpp_log);
                if (wave_first < n_pts) {  // (wave-uniform)
                    const int n_valid = min(64 >> lpp_log, n_pts - wave_first);
                    const int lo_min = __shfl(lo, 0, 64), lo_max = __shfl(lo, (n_valid << lpp_log) - 1, 64);
                    const s16x2 pi = __builtin_bit_cast(s16x2, sxy[min(i, n_kp - 1)]);
                    uint32_t best = 0xFFFFFFFFu;
                    int j = 4 * sub;
                    for (; j + 4 <= lo_min; j += step) {  // every lane's prefix covers these
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const s16x2 d = pi - __builtin_bit_cast(s16x2, sxy[j + c]);
                            best = min(best, (uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false));
                        }
                    }
                    for (; j < lo_max; j += step) {
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const s16x2 d = pi - __builtin_bit_cast(s16x2, sxy[j + c]);  // (reads at most 3 words past the prefix: inside the arrays)
                            const uint32_t d2 = (uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false);
                            best = (j + c < lo) ? min(best, d2) : best;
                        }
                    }
                    if (lpp > 1) best = min(best, (uint32_t)__shfl_xor((int)best, 1, 64));
                    if (lpp > 2) best = min(best, (uint32_t)__shfl_xor((int)best, 2, 64));
                    if (lpp > 4) best = min(best, (uint32_t)__shfl_xor((int)best, 4, 64));
                    if (valid && sub == 0) r2[i] = best;
                }
            }
        };
        if (n_pts <= 128) radii(std::integral_constant<int, 3>{});
        else if (n_pts <= 512) radii(std::integral_constant<int, 2>{});
        else if (n_pts <= 1024) radii(std::integral_constant<int, 1>{});
        else radii(std::integral_constant<int, 0>{});
        __syncthreads();
        }
        if (!(parts & ANMS_SELECT)) return n_kp;
        STAMP(8);
        // ---- decisionRadius = (max_kp)-th element (0-based) of the radii sorted descending: radix select over
        // three 8-bit digits (finite radii^2 < 2^24; 0xFFFFFFFF stands for sqrt(FLT_MAX))
        {
            int *h = row_first;  // [256]
            if (tid == 0) {
                misc[0] = 0;      // selected prefix
                misc[3] = max_kp; // remaining rank
                misc[4] = 0;      // #infinite
                misc[5] = 0;      // done flag
            }
            __syncthreads();
            int ninf = 0;
            for (int i = tid; i < n_kp; i += 1024) ninf += (r2[i] == 0xFFFFFFFFu) ? 1 : 0;
            if (ninf) atomicAdd(&misc[4], ninf);
            __syncthreads();
            if (tid == 0) {
                if (misc[3] < misc[4]) {
                    misc[0] = (int)0xFFFFFFFFu;
                    misc[5] = 1;
                } else
                    misc[3] -= misc[4];
            }
            __syncthreads();
            if (!misc[5]) {
                for (int shift = 16; shift >= 0; shift -= 8) {
                    for (int k = tid; k < 256; k += 1024) h[k] = 0;
                    __syncthreads();
                    const uint32_t prefix = (uint32_t)misc[0];
                    const uint32_t himask = (shift == 16) ? 0u : (0xFFFFFFu & ~((1u << (shift + 8)) - 1u));
                    for (int i = tid; i < n_kp; i += 1024) {
                        const uint32_t v = r2[i];
                        if (v != 0xFFFFFFFFu && (v & himask) == (prefix & himask)) atomicAdd(&h[(v >> shift) & 255u], 1);
                    }
                    __syncthreads();
                    if (wave_id() == 0) {  // digit d with #(digit > d) <= rem < #(digit >= d): suffix scan over 256 bins
                        const int lane = lane_id();
                        const int rem = misc[3];
                        int hh[4], tot = 0;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            hh[q] = h[4 * lane + q];
                            tot += hh[q];
                        }
                        int incl = tot;
                        for (int d = 1; d < 64; d <<= 1) {
                            const int t = __shfl_down(incl, d, 64);
                            if (lane + d < 64) incl += t;
                        }
                        int above = incl - tot;
                        int found_d = -1, found_above = 0;
#pragma unroll
                        for (int q = 3; q >= 0; q--) {
                            if (found_d < 0 && above <= rem && rem < above + hh[q]) {
                                found_d = 4 * lane + q;
                                found_above = above;
                            }
                            above += hh[q];
                        }
                        const uint64_t fm = __ballot(found_d >= 0);
                        if (fm == 0ull) {  // rem >= total (cannot happen: n_kp > max_kp): fall to digit 0
                            if (lane == 0) misc[0] = (int)prefix;
                        } else if (found_d >= 0) {
                            misc[3] = rem - found_above;
                            misc[0] = (int)(prefix | ((uint32_t)found_d << shift));
                        }
                    }
                    __syncthreads();
                }
            }
        }
        STAMP(9);
        const uint32_t decision = (uint32_t)misc[0];
        __syncthreads();
        for (int base = 0; base < n_kp; base += 1024) {
            const int i = base + tid;
            const bool keep = (i < n_kp) && (r2[i] >= decision);
            int total;
            const int off = n_out + block_excl_scan(keep ? 1 : 0, scan, &total);
            if (keep && off < CELL_OUT_CAP) {
                const uint32_t k = sorted[i];
                out[3 * off] = (float)key_x(k) + fX0;
                out[3 * off + 1] = (float)key_y(k) + fY0;
                out[3 * off + 2] = (float)key_r(k);
            }
            n_out += total;
        }
    } else {
        for (int i = tid; i < n_kp; i += 1024) {
            if (i < CELL_OUT_CAP) {
                const uint32_t k = arr[i];
                out[3 * i] = (float)key_x(k) + fX0;
                out[3 * i + 1] = (float)key_y(k) + fY0;
                out[3 * i + 2] = (float)key_r(k);
            }
        }
        n_out = n_kp;
    }
    STAMP(10);
    if (dbg && threadIdx.x == 0) dbg[39] = (long long)n_raw | ((long long)n_kp << 16) | ((long long)n_out << 32);  // (one free slot)
    return n_out;
}
#undef STAMP

// AGAST NMS + LVT ANMS of one cell on arrays that live either in LDS (I = u16) or in global scratch (I = u32).
template <typename I>
__device__ __forceinline__ int cell_nms_anms(const Seq &S, const CellGeom &g, uint32_t *keys, uint32_t *uf, I *root, I *abv, I *nms, uint8_t *tie,
                                             int n_raw, int n_cap, int *row_first, int *row_end, int *scan, int *stack, int *misc, float *out, long long *dbg) {
    (void)stack;
    const int n_kp = cell_nms<I>(keys, uf, root, abv, nms, tie, n_raw, row_first, row_end, scan, dbg);
    return cell_anms<I>(S, g, keys, uf, root, abv, nms, n_kp, n_cap, row_first, row_end, scan, misc, out, dbg, n_raw);
}

// the LDS image of one cell's workgroup
struct CellLds {
    uint32_t *keys, *uf;
    uint16_t *root16, *abv16, *nms16;
    int *row_first, *row_end, *scan, *stack, *misc;
    uint8_t *tie8;
};
__device__ __forceinline__ CellLds carve_cell_lds(uint8_t *smem, int raw_cap = RAW_CAP) {  // raw_cap even
    CellLds L;
    L.keys = reinterpret_cast<uint32_t *>(smem);
    L.uf = L.keys + raw_cap;
    L.root16 = reinterpret_cast<uint16_t *>(L.uf + raw_cap);
    L.abv16 = L.root16 + raw_cap;
    L.nms16 = L.abv16 + raw_cap;
    L.row_first = reinterpret_cast<int *>(L.nms16 + raw_cap);  // [1024]
    L.row_end = L.row_first + 1024;                            // [1024]
    L.scan = L.row_end + 1024;                                 // [64]
    L.stack = L.scan + 64;                                     // [192]
    L.misc = L.stack + 192;                                    // [16]
    L.tie8 = reinterpret_cast<uint8_t *>(L.misc + 16);          // [RAW_CAP]
    return L;
}
// shared prologue of the three cell kernels: false = nothing to do for this (cell, eye, pass)
__device__ __forceinline__ bool cell_begin(const Seq &S, const FrameBuf &FB, int eye, int cell, int pass, CellGeom &g, int &cxi) {
    const FeatCtl &ctl = *FB.fc;
    if (ctl.poison || ctl.ext_corners) return false;
    if (eye == 1 && S.prm.sensor == 2) return false;
    if (cell >= S.prm.n_cells) return false;
    g.threshold = S.prm.agast_th;
    if (pass == 1) {
        if (ctl.n_detected[eye] >= CORNERS_LOW_TH) return false;  // handler.cpp:161
        g.threshold = S.prm.agast_th_low;
    }
    const int cs = S.prm.cell_size;
    cxi = cell % S.prm.cells_x;
    const int cyi = cell / S.prm.cells_x;
    g.X0 = cxi * cs;
    g.Y0 = cyi * cs;
    g.cw = min(cs, S.prm.W - g.X0);
    g.ch = min(cs, S.prm.H - g.Y0);
    g.score = FB.score[eye];
    g.pp = S.plane_pitch;
    return true;
}
__device__ __forceinline__ void cell_finish(const FrameBuf &FB, int eye, int cell, int pass, int n_out) {
    FeatCtl &ctl = *FB.fc;
    if (threadIdx.x == 0) {
        if (n_out > CELL_OUT_CAP) {
            atomicOr(&ctl.overflow, OVF_CELL_OUT);
            n_out = CELL_OUT_CAP;
        }
        FB.cell_n[eye][cell] = n_out;
        if (pass == 0) atomicAdd(&ctl.n_detected[eye], n_out);
        else if (cell == 0) ctl.retry[eye] = 1;
    }
}
// dense / very large cell on ONE workgroup: the same algorithm on global scratch sized for every pixel of the cell
__device__ __forceinline__ int cell_global_path(const Seq &S, const FrameBuf &FB, int eye, int cell, int cxi, bool segs, const CellGeom &g, const CellLds &L, float *out,
                                                long long *dbg) {
    const size_t cap = (size_t)g.cw * g.ch;
    uint32_t *gk = S.cell_scratch[eye] + S.cell_scratch_off[cell];
    uint32_t *guf = gk + cap, *groot = guf + cap, *gabv = groot + cap, *gnms = gabv + cap;
    const int n_raw = segs ? cell_gather_segments(FB, eye, g, S.prm.cell_size, cxi, (S.prm.W + TS_W - 1) / TS_W, gk, (int)cap, L.scan) : cell_compact(g, gk, (int)cap, L.scan);
    return cell_nms_anms<uint32_t>(S, g, gk, guf, groot, gabv, gnms, reinterpret_cast<uint8_t *>(gnms + cap), n_raw, (int)cap, L.row_first, L.row_end, L.scan, L.stack, L.misc, out,
                                   dbg);
}

// one detection cell of one image: AGAST NMS + LVT's ANMS (or the hand-over of an oversized cell to the strip kernels)
__device__ __forceinline__ void cells_work(const Seq &S, const FrameBuf &FB, int eye, int cell, int pass, const CellLds &L, int raw_cap = RAW_CAP) {
    FeatCtl &ctl = *FB.fc;
    const int tid = threadIdx.x;
    CellGeom g;
    int cxi;
    if (!cell_begin(S, FB, eye, cell, pass, g, cxi)) return;
    const int cs = S.prm.cell_size;
    float *out = FB.cell_kp[eye] + (size_t)cell * CELL_OUT_CAP * 3;

    long long *dbg = (cell == 0 && eye == 0 && pass == 0) ? S.ctl->dbg : nullptr;
    if (dbg && tid == 0) dbg[0] = clock64();
    int n_out = 0;
    if (g.cw > 1024 || g.ch > 1024) {
        if (tid == 0) atomicOr(&ctl.overflow, OVF_CELL_DIM);
    } else if (g.cw >= 7 && g.ch >= 7) {
        // pass 0 with cells of at least one tile width: gather k_score's segments; otherwise compact the score map here
        const bool segs = (pass == 0) && (cs >= TS_W);
        const int n_raw = segs ? cell_gather_segments(FB, eye, g, cs, cxi, (S.prm.W + TS_W - 1) / TS_W, L.keys, raw_cap, L.scan)
                               : cell_compact(g, L.keys, raw_cap, L.scan, dbg);
        if (dbg && tid == 0) dbg[1] = clock64();
        if (n_raw <= raw_cap) {
            n_out = cell_nms_anms<uint16_t>(S, g, L.keys, L.uf, L.root16, L.abv16, L.nms16, L.tie8, n_raw, raw_cap, L.row_first, L.row_end, L.scan, L.stack, L.misc, out, dbg);
        } else if (S.prm.big_cell_strips && pass == 0) {
            // more raw corners than this workgroup's LDS holds, in a cell tall enough to cut: k_cells_strip (NMS of row strips on several
            // CUs) and the kernels behind it take over; they also write cell_n / n_detected
            if (tid == 0) S.cell_big[eye][cell] = 1;
            return;
        } else {
            n_out = cell_global_path(S, FB, eye, cell, cxi, segs, g, L, out, dbg);
        }
    }
    if (dbg && tid == 0) dbg[11] = clock64();
    cell_finish(FB, eye, cell, pass, n_out);
}

// ---- a single sequence's tall cells as TWO (or three) co-operating workgroups of the same launch --------------------------------------
// k_cells is the longest kernel of the feature stage (66 us of one CU per cell) and half of it is AGAST's NMS, which can be cut by rows exactly as
// for the oversized cells below (one survivor per 4-connected blob, decided by the blob's pixels alone): workgroup "strip s" gathers the raw corners of
// its core rows + SPLIT_HALO rows on either side, runs the same cell_nms, vouches that no blob reaches from a core row to the outermost halo row, and
// keeps the survivors of its core rows.  Strips 1.. ("helpers") put theirs into global memory and publish a count + this launch's token; strip 0
// ("main") appends them behind its own -- strip order IS raster order -- and runs LVT's ANMS on the merged list as before.  A strip that cannot vouch
// (or overflows) reports -1 and the main workgroup runs the whole cell alone, as before.
// No deadlock: the helpers have the LOWER workgroup ids and the same id modulo 8 as their main workgroup (cells_entry), i.e. they sit in the same XCD's
// dispatch order in front of it: when a main workgroup runs, its helpers are resident or done.  The wait is bounded all the same (2 ms -> whole cell alone).
constexpr int STRIPS = 8;  // (row strips of an oversized cell, k_cells_strip below; the per-cell stride of Seq::strip_n)
constexpr int SPLIT_HALO = 16, SPLIT_MIN_ROWS = 160, SPLIT_MAX = 3, SPLIT_KP_CAP = 2048;
__device__ __forceinline__ void cells_work_split(const Seq &S, const FrameBuf &FB, int eye, int cell, int strip, int nsplit, unsigned token, const CellLds &L, int raw_cap) {
    const int tid = threadIdx.x;
    CellGeom g;
    int cxi;
    if (!cell_begin(S, FB, eye, cell, 0, g, cxi)) return;
    const int cs = S.prm.cell_size;
    const bool splittable = g.cw >= 7 && g.ch >= SPLIT_MIN_ROWS && g.cw <= 1024 && g.ch <= 1024 && cs >= TS_W && !S.prm.big_cell_strips;
    if (!splittable) {
        if (strip == 0) cells_work(S, FB, eye, cell, 0, L, raw_cap);
        return;
    }
    long long *dbg = (cell == 0 && eye == 0 && strip == 0) ? S.ctl->dbg : nullptr;
    if (dbg && tid == 0) dbg[0] = clock64();
    int *cnt_all = S.strip_n[eye] + cell * STRIPS;          // [1, nsplit): the helpers' counts; [4 + s]: their tokens
    const int per = (g.ch - 6 + nsplit - 1) / nsplit;
    const int c0 = 3 + strip * per, c1 = min(c0 + per, g.ch - 3);
    const int e0 = max(3, c0 - SPLIT_HALO), e1 = min(g.ch - 3, c1 + SPLIT_HALO);
    CellGeom g2 = g;  // the extended strip as a "cell" whose interior rows are [e0, e1)
    g2.Y0 = g.Y0 + e0 - 3;
    g2.ch = (e1 - e0) + 6;
    int n_core = -1;
    const int n_raw = (c0 < c1) ? cell_gather_segments(FB, eye, g2, cs, cxi, (S.prm.W + TS_W - 1) / TS_W, L.keys, raw_cap, L.scan) : 0;
    if (dbg && tid == 0) dbg[1] = clock64();
    uint32_t *dst = S.strip_kp[eye] + ((size_t)cell * SPLIT_MAX + strip) * SPLIT_KP_CAP;  // (helpers)
    if (n_raw <= raw_cap) {
        const int n_kp = cell_nms<uint16_t>(L.keys, L.uf, L.root16, L.abv16, L.nms16, L.tie8, n_raw, L.row_first, L.row_end, L.scan, dbg);
        // a blob with a pixel in an outermost extended row (that is not the cell's own first / last interior row) AND one in a core row?
        for (int i = tid; i < n_raw; i += 1024) L.tie8[i] = 0;
        __syncthreads();
        for (int i = tid; i < n_raw; i += 1024) {
            const int y = key_y(L.keys[i]);
            if ((e0 > 3 && y == e0) || (e1 < g.ch - 3 && y == e1 - 1)) L.tie8[L.root16[i]] = 1;
        }
        __syncthreads();
        int unsafe = 0;
        for (int i = tid; i < n_raw; i += 1024) {
            const int y = key_y(L.keys[i]);
            if (y >= c0 && y < c1 && L.tie8[L.root16[i]]) unsafe = 1;
        }
        if (!__syncthreads_or(unsafe)) {
            // survivors of the core rows, raster order: the main workgroup keeps them where they are (uf[0 ..): off <= i), a helper sends them out
            int n_out = 0;
            for (int base = 0; base < n_kp; base += 1024) {
                const int i = base + tid;
                const uint32_t k = (i < n_kp) ? L.uf[i] : 0u;
                const bool keep = (i < n_kp) && key_y(k) >= c0 && key_y(k) < c1;
                int total;
                const int off = n_out + block_excl_scan(keep ? 1 : 0, L.scan, &total);
                if (keep) {
                    if (strip == 0) L.uf[off] = k;
                    else if (off < SPLIT_KP_CAP) dst[off] = k;
                }
                n_out += total;
            }
            n_core = (strip != 0 && n_out > SPLIT_KP_CAP) ? -1 : n_out;
        }
    }
    if (strip != 0) {  // helper: the survivors, then the count, then the token.  strip_kp / strip_n are UNCACHED device memory (lvt_host.hip): a store is
        // performed when its wave's vmcnt says so -- no release fence, whose L2 write-back costs ~13 us behind k_score's planes (measured)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(cnt_all + strip, n_core, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(reinterpret_cast<unsigned *>(cnt_all + 4 + strip), token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // main: wait for the helpers
    __shared__ int s_helper[SPLIT_MAX];
    if (tid == 0) {
        for (int h = 1; h < nsplit; h++) {
            const unsigned *fl = reinterpret_cast<const unsigned *>(cnt_all + 4 + h);
            const long long t0 = wall_clock64();
            int n = -1;
            while (true) {
                if (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == token) {
                    n = __hip_atomic_load(cnt_all + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                if (wall_clock64() - t0 > 200000) break;  // 2 ms of the 100-MHz clock: the whole cell on this workgroup
                __builtin_amdgcn_s_sleep(2);
            }
            s_helper[h] = n;
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    bool bad = n_core < 0;
    int total = max(n_core, 0);
    for (int h = 1; h < nsplit; h++) {
        bad = bad || s_helper[h] < 0;
        total += max(s_helper[h], 0);
    }
    if (bad || total > raw_cap) {
        __syncthreads();
        cells_work(S, FB, eye, cell, 0, L, raw_cap);
        if (dbg && tid == 0) dbg[10] = 2003;  // (tests: a strip could not vouch or overflowed -- the whole cell on this workgroup, as without the split)
        return;
    }
    {
        int off = n_core;
        for (int h = 1; h < nsplit; h++) {
            const uint32_t *src = S.strip_kp[eye] + ((size_t)cell * SPLIT_MAX + h) * SPLIT_KP_CAP;
            for (int i = tid; i < s_helper[h]; i += 1024) L.uf[off + i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            off += s_helper[h];
        }
    }
    __syncthreads();
    if (dbg && tid == 0) dbg[5] = clock64();
    float *out = FB.cell_kp[eye] + (size_t)cell * CELL_OUT_CAP * 3;
    const int n_out = cell_anms<uint16_t>(S, g, L.keys, L.uf, L.root16, L.abv16, L.nms16, total, raw_cap, L.row_first, L.row_end, L.scan, L.misc, out, dbg, total);
    if (dbg && tid == 0) dbg[11] = clock64();
    cell_finish(FB, eye, cell, 0, n_out);
}

// ---- host images into the pitched planes (the body of k_stage_in, lvt_host.hip; also the pull workgroups of k_cells below) -------------------
// tightly packed source (stride == cols, ANY byte address and byte count) -> pitched plane: bytes up to the first 16-byte boundary and behind the
// last whole vector travel one by one, everything between as aligned 16-byte loads; no load reaches outside [src, src + W H).
__device__ __forceinline__ void stage_put(uint8_t *dst, size_t i, int W, int pitch, uint8_t v) {
    const size_t y = i / (size_t)W;
    dst[y * pitch + (i - y * W)] = v;
}
__device__ __forceinline__ void stage_in_body(const uint8_t *src, uint8_t *dst, int W, int H, int pitch, size_t t0, size_t stride) {
    const size_t n = (size_t)W * H;
    const size_t head = min((size_t)((16 - ((uintptr_t)src & 15)) & 15), n);
    const size_t nv = (n - head) / 16, tail0 = head + nv * 16;
    for (size_t v = t0; v < nv; v += stride) {
        const uint4 q = *reinterpret_cast<const uint4 *>(src + head + v * 16);
        const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
        size_t i = head + v * 16;
        int y = (int)(i / (size_t)W), x = (int)(i - (size_t)y * W);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            dst[(size_t)y * pitch + x] = (uint8_t)(wds[k >> 2] >> (8 * (k & 3)));
            if (++x == W) x = 0, y++;
        }
    }
    if (t0 < head) stage_put(dst, t0, W, pitch, src[t0]);
    if (t0 < n - tail0) stage_put(dst, tail0 + t0, W, pitch, src[tail0 + t0]);
}
// Asynchronous host frames of a single sequence: k_cells is the longest kernel of the feature stage (66 us) and occupies 40 of the 256 CUs -- its
// launch carries extra workgroups (blockIdx.x >= n_cells) that pull the NEXT frame's images over PCIe into their planes (24 us, beside the cells):
// the pull leaves the feature stream's chain, which was the longest of the three with it (lvt_host.hip, upload_async).
struct NextPull {
    const uint8_t *src[2];  // nullptr: nothing to pull
    uint8_t *dst[2];
    int W, H, pitch;
};

// pass 0 of the detection (the <200-corner retry, handler.cpp:161-169, runs inside k_gather)
// A lock-step batch launches its cells as ONE row of workgroups, the cells with the largest area first (CellOrder, filled by the host): with more
// workgroups than CUs (16 sequences x 20 cells on 256) the workgroups that start late -- on the CUs the first small cells leave -- are then small
// cells too, and the launch ends with the first round's 250 x 250 cells instead of a second round of them (KITTI, 16 sequences: 112 -> 7x us).
struct CellOrder {
    uint8_t v[CELLS_MAX];
};
template <bool BV>  // (a single sequence's descriptor travels in the kernel arguments: one dependent memory hop less at the head of the longest kernel)
__device__ __forceinline__ void cells_entry(const SeqArg<BV> &sa, int pass, int par, const CellOrder &ord, int lanes, int raw_cap, const NextPull &np, int nsplit, unsigned token) {
    int eye = blockIdx.y, cell = blockIdx.x;
    int strip = -1;  // >= 0: this workgroup is one row strip of a split cell (cells_work_split)
    if constexpr (BV) {
        // nsplit >= 2: x in [0, NH (nsplit - 1)) helpers (strip 1 + x / NH of cell x % NH), [NH (nsplit - 1), NH nsplit) the cells' main workgroups,
        // behind them the pull workgroups; NH = n_cells rounded up to 8 and gridDim.x % 8 == 0, so a cell's workgroups share their id modulo 8
        const int nc = sa.v.prm.n_cells;
        const int first_pull = (nsplit >= 2) ? ((nc + 7) & ~7) * nsplit : nc;
        if ((int)blockIdx.x >= first_pull) {  // a pull workgroup (or grid padding when there is nothing to pull)
            if (!np.src[eye]) return;
            const int k = (int)blockIdx.x - first_pull, nk = (int)gridDim.x - first_pull;
            stage_in_body(np.src[eye], np.dst[eye], np.W, np.H, np.pitch, (size_t)k * blockDim.x + threadIdx.x, (size_t)nk * blockDim.x);
            return;
        }
        if (nsplit >= 2) {
            const int NH = (nc + 7) & ~7, role = (int)blockIdx.x / NH;
            cell = (int)blockIdx.x - role * NH;
            if (cell >= nc) return;
            strip = (role == nsplit - 1) ? 0 : role + 1;
        }
    }
    const Seq *Sp;
    if constexpr (BV) {
        Sp = &sa.v;
    } else {  // lanes = 2 x sequences: workgroup L is (cell ord[L / lanes], eye L & 1, sequence (L % lanes) >> 1)
        const int slot = (int)blockIdx.x / lanes, r = (int)blockIdx.x - slot * lanes;
        cell = ord.v[slot], eye = r & 1;
        Sp = &seq_const(sa.p, r >> 1);
    }
    const Seq &S = *Sp;
    const FrameBuf &FB = S.fb[par];
    if (threadIdx.x == 0 && cell < CELLS_MAX && strip <= 0) S.cell_big[eye][cell] = 0;  // (nobody reads it before this launch is over)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CellLds L = carve_cell_lds(smem, raw_cap);
    if (strip >= 0) cells_work_split(S, FB, eye, cell, strip, nsplit, token, L, raw_cap);
    else cells_work(S, FB, eye, cell, pass, L, raw_cap);
}
// A single sequence runs one workgroup per CU with the 159-KB carve: it may use the 128 registers a 1024-thread workgroup can have.  The batch
// instance is held to 64 so that TWO of its 80-KB workgroups share a CU (at 66 registers it ran one per CU whatever its LDS) -- an occupancy
// attribute on the template would cap the single-sequence instance too, where a second workgroup can never fit and the cap can only spill.
template <bool BV>
__global__ __launch_bounds__(1024) void k_cells(SeqArg<BV> sa, int pass, int par, CellOrder ord, int lanes, int raw_cap, NextPull np, int nsplit, unsigned token) {
    cells_entry<BV>(sa, pass, par, ord, lanes, raw_cap, np, nsplit, token);
}
template <>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_cells<false>(SeqArg<false> sa, int pass, int par, CellOrder ord, int lanes, int raw_cap, NextPull np, int nsplit, unsigned token) {
    cells_entry<false>(sa, pass, par, ord, lanes, raw_cap, np, 0, token);
}

// ---- an oversized cell as row strips --------------------------------------------------------------------------------------------
// AGAST's NMS keeps one corner per 4-connected blob of corner pixels, decided by the blob's pixels alone, so the rows
// of a cell can be cut: strip s decides the corners of its CORE rows from the raw corners of core + STRIP_HALO rows on either side
// (the fast LDS path of k_cells, one CU per strip), and every decision is exact as long as no blob reaches from a core row to the
// outermost halo row of the extended strip -- such a blob might continue outside.  That is checked per blob; a strip that cannot
// vouch (or whose extended rows overflow the LDS) reports -1 and k_cells_big runs the whole cell on the single-workgroup global path
// instead, as before.  Survivors of the strips, concatenated in strip order, ARE the cell's survivors in raster order.
constexpr int STRIP_HALO = 16;
__global__ __launch_bounds__(1024) void k_cells_strip(Seq *seqs, int pass, int par) {
    const Seq &S = seq_const(seqs, blockIdx.z);
    const int eye = blockIdx.y, cell = blockIdx.x / STRIPS, strip = blockIdx.x % STRIPS;
    const FrameBuf &FB = S.fb[par];
    CellGeom g;
    int cxi;
    if (!cell_begin(S, FB, eye, cell, pass, g, cxi)) return;
    if (S.cell_big[eye][cell] != 1) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CellLds L = carve_cell_lds(smem);
    const int tid = threadIdx.x, cs = S.prm.cell_size;
    int *cnt = S.strip_n[eye] + cell * STRIPS + strip;
    uint32_t *dst = S.strip_kp[eye] + ((size_t)cell * STRIPS + strip) * RAW_CAP;
    // interior rows of the cell: ly in [3, ch - 3)
    const int per = (g.ch - 6 + STRIPS - 1) / STRIPS;
    const int c0 = 3 + strip * per, c1 = min(c0 + per, g.ch - 3);
    if (c0 >= c1) {
        if (tid == 0) *cnt = 0;
        return;
    }
    const int e0 = max(3, c0 - STRIP_HALO), e1 = min(g.ch - 3, c1 + STRIP_HALO);
    CellGeom g2 = g;  // the extended strip as a "cell" whose interior rows are [e0, e1)
    g2.Y0 = g.Y0 + e0 - 3;
    g2.ch = (e1 - e0) + 6;
    const bool segs = (pass == 0) && (cs >= TS_W);
    const int n_raw = segs ? cell_gather_segments(FB, eye, g2, cs, cxi, (S.prm.W + TS_W - 1) / TS_W, L.keys, RAW_CAP, L.scan)
                           : cell_compact(g2, L.keys, RAW_CAP, L.scan, nullptr, e0 - 3);
    if (n_raw > RAW_CAP) {
        if (tid == 0) *cnt = -1;
        return;
    }
    const int n_kp = cell_nms<uint16_t>(L.keys, L.uf, L.root16, L.abv16, L.nms16, L.tie8, n_raw, L.row_first, L.row_end, L.scan, nullptr);
    // a blob with a pixel in an outermost extended row (that is not the cell's own first / last interior row) AND one in a core row?
    for (int i = tid; i < n_raw; i += 1024) L.tie8[i] = 0;
    __syncthreads();
    for (int i = tid; i < n_raw; i += 1024) {
        const int y = key_y(L.keys[i]);
        if ((e0 > 3 && y == e0) || (e1 < g.ch - 3 && y == e1 - 1)) L.tie8[L.root16[i]] = 1;
    }
    __syncthreads();
    int unsafe = 0;
    for (int i = tid; i < n_raw; i += 1024) {
        const int y = key_y(L.keys[i]);
        if (y >= c0 && y < c1 && L.tie8[L.root16[i]]) unsafe = 1;
    }
    if (__syncthreads_or(unsafe)) {
        if (tid == 0) *cnt = -1;
        return;
    }
    // survivors of the core rows, raster order
    int n_out = 0;
    for (int base = 0; base < n_kp; base += 1024) {
        const int i = base + tid;
        const uint32_t k = (i < n_kp) ? L.uf[i] : 0u;
        const bool keep = (i < n_kp) && key_y(k) >= c0 && key_y(k) < c1;
        int total;
        const int off = n_out + block_excl_scan(keep ? 1 : 0, L.scan, &total);
        if (keep) dst[off] = k;  // (at most n_raw <= RAW_CAP)
        n_out += total;
    }
    if (tid == 0) *cnt = n_out;
}

// the oversized cell's second half: its strips' survivors, merged in LDS, through LVT's ANMS (or the whole cell on the old path)
__global__ __launch_bounds__(1024) void k_cells_big(Seq *seqs, int pass, int par) {
    const Seq &S = seq_const(seqs, blockIdx.z);
    const int eye = blockIdx.y, cell = blockIdx.x;
    const FrameBuf &FB = S.fb[par];
    CellGeom g;
    int cxi;
    if (!cell_begin(S, FB, eye, cell, pass, g, cxi)) return;
    if (S.cell_big[eye][cell] != 1) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CellLds L = carve_cell_lds(smem);
    const int tid = threadIdx.x;
    float *out = FB.cell_kp[eye] + (size_t)cell * CELL_OUT_CAP * 3;
    const int *cnt = S.strip_n[eye] + cell * STRIPS;
    int total = 0;
    bool bad = false;
    for (int s = 0; s < STRIPS; s++) {  // (block-uniform: every thread reads the same words)
        const int n = cnt[s];
        bad = bad || n < 0;
        total += max(n, 0);
    }
    int n_out;
    long long *route = (cell == 0 && eye == 0 && pass == 0 && tid == 0) ? S.ctl->dbg + 10 : nullptr;  // (tests: which way the cell went)
    if (bad || total > RAW_CAP) {
        if (route) *route = 1003;  // a strip could not vouch or overflowed: the whole cell on one workgroup
        n_out = cell_global_path(S, FB, eye, cell, cxi, (pass == 0) && (S.prm.cell_size >= TS_W), g, L, out, nullptr);
    } else {
        if (route) *route = (total > S.prm.max_kp_cell) ? 1001 : 1002;  // strips + three-launch ANMS / strips, no ANMS needed
        int off = 0;
        for (int s = 0; s < STRIPS; s++) {
            const uint32_t *src = S.strip_kp[eye] + ((size_t)cell * STRIPS + s) * RAW_CAP;
            for (int i = tid; i < cnt[s]; i += 1024) L.uf[off + i] = src[i];
            off += cnt[s];
        }
        __syncthreads();
        long long *dbg = (cell == 0 && eye == 0 && pass == 0) ? S.ctl->dbg : nullptr;  // (phase stamps 5 .. 10: tools/cells_phases.py)
        if (dbg && tid == 0) dbg[5] = clock64();
        if (total > S.prm.max_kp_cell) {
            // ANMS in three launches: the sort here, the radii (n^2 / 2 distances: VALU-bound on one CU) on RADII_WGS workgroups, the
            // selection on one again.  sorted[] travels through strip 0's (now dead) buffer, the radii through strip 1's.
            cell_anms<uint16_t>(S, g, L.keys, L.uf, L.root16, L.abv16, L.nms16, total, RAW_CAP, L.row_first, L.row_end, L.scan, L.misc, out, dbg, total, ANMS_SORT);
            uint32_t *gs = S.strip_kp[eye] + (size_t)cell * STRIPS * RAW_CAP;
            for (int i = tid; i < total; i += 1024) gs[i] = L.keys[i];
            if (tid == 0) {
                S.strip_n[eye][cell * STRIPS] = total;
                S.cell_big[eye][cell] = 2;
            }
            if (dbg && tid == 0) dbg[11] = clock64();
            return;
        }
        n_out = cell_anms<uint16_t>(S, g, L.keys, L.uf, L.root16, L.abv16, L.nms16, total, RAW_CAP, L.row_first, L.row_end, L.scan, L.misc, out, dbg, total);
        if (dbg && tid == 0) dbg[11] = clock64();
    }
    cell_finish(FB, eye, cell, pass, n_out);
}

constexpr int RADII_WGS = 16;
static_assert(STRIPS >= 2, "the ANMS launches pass sorted[] and r2[] through the first two strip buffers");
__global__ __launch_bounds__(1024) void k_cells_radii(Seq *seqs, int pass, int par) {
    const Seq &S = seq_const(seqs, blockIdx.z);
    const int eye = blockIdx.y, cell = blockIdx.x / RADII_WGS, wg = blockIdx.x % RADII_WGS;
    const FrameBuf &FB = S.fb[par];
    CellGeom g;
    int cxi;
    if (!cell_begin(S, FB, eye, cell, pass, g, cxi)) return;
    if (S.cell_big[eye][cell] != 2) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CellLds L = carve_cell_lds(smem);
    const int tid = threadIdx.x;
    const int n = S.strip_n[eye][cell * STRIPS];
    const uint32_t *gs = S.strip_kp[eye] + (size_t)cell * STRIPS * RAW_CAP;
    uint32_t *gr = S.strip_kp[eye] + ((size_t)cell * STRIPS + 1) * RAW_CAP;
    for (int i = tid; i < n; i += 1024) L.keys[i] = gs[i];
    __syncthreads();
    cell_anms<uint16_t>(S, g, L.keys, L.uf, L.root16, L.abv16, L.nms16, n, RAW_CAP, L.row_first, L.row_end, L.scan, L.misc, nullptr, nullptr, n, ANMS_RADII, wg, RADII_WGS);
    for (int i = wg + tid * RADII_WGS; i < n; i += 1024 * RADII_WGS) gr[i] = L.uf[i];
}

__global__ __launch_bounds__(1024) void k_cells_select(Seq *seqs, int pass, int par) {
    const Seq &S = seq_const(seqs, blockIdx.z);
    const int eye = blockIdx.y, cell = blockIdx.x;
    const FrameBuf &FB = S.fb[par];
    CellGeom g;
    int cxi;
    if (!cell_begin(S, FB, eye, cell, pass, g, cxi)) return;
    if (S.cell_big[eye][cell] != 2) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CellLds L = carve_cell_lds(smem);
    const int tid = threadIdx.x;
    const int n = S.strip_n[eye][cell * STRIPS];
    const uint32_t *gs = S.strip_kp[eye] + (size_t)cell * STRIPS * RAW_CAP;
    const uint32_t *gr = S.strip_kp[eye] + ((size_t)cell * STRIPS + 1) * RAW_CAP;
    for (int i = tid; i < n; i += 1024) {
        L.keys[i] = gs[i];
        L.uf[i] = gr[i];
    }
    __syncthreads();
    float *out = FB.cell_kp[eye] + (size_t)cell * CELL_OUT_CAP * 3;
    const int n_out = cell_anms<uint16_t>(S, g, L.keys, L.uf, L.root16, L.abv16, L.nms16, n, RAW_CAP, L.row_first, L.row_end, L.scan, L.misc, out, nullptr, n, ANMS_SELECT);
    cell_finish(FB, eye, cell, pass, n_out);
}

// =================================================================================================
// k_gather : cells (rect order) -> feature struct, BRIEF border filter, RGB-D depth filter + undistort
// =================================================================================================
__device__ __forceinline__ bool brief_border_keep(float x, float y, int rows, int cols) {
    const int B = 28;
    if (rows <= 2 * B || cols <= 2 * B) return false;
    const int ix = __float2int_rn(x), iy = __float2int_rn(y);  // cvRound
    return ix >= B && ix < cols - B && iy >= B && iy < rows - B;
}

// (`seqs` = the descriptors in device memory: the per-frame inputs -- image / depth pointers and pitches -- are published there by the
//  head of the feature stage and are NOT part of a descriptor that travels by value)
template <bool BV>
__global__ __launch_bounds__(1024) void k_gather(SeqArg<BV> sa, const Seq *seqs, int par) {
    const Seq &S = sa.get();
    const int eye = blockIdx.y;
    const FrameBuf &FB = S.fb[par];
    const FrameBuf &FBd = seq_const(seqs, blockIdx.z).fb[par];
    FeatCtl &ctl = *FB.fc;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CellLds L = carve_cell_lds(smem);
    int *cell_off = L.row_first;  // [CELLS_MAX + 1]
    int *scan = L.scan;
    const int tid = threadIdx.x;
    const Feat &F = FB.feat[eye];
    if (ctl.poison) return;
    // handler.cpp:161-169: fewer than 200 corners in the whole image -> every cell again with the lowered threshold.  Almost never
    // taken, so it has no launch of its own (that launch cost every frame 6.5 us of its longest chain): this workgroup runs the
    // image's cells one after the other, then gathers.
    if (!ctl.ext_corners && !(eye == 1 && S.prm.sensor == 2) && ctl.n_detected[eye] < CORNERS_LOW_TH) {
        for (int cell = 0; cell < S.prm.n_cells; cell++) {
            cells_work(S, FB, eye, cell, 1, L);
            __syncthreads();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (eye == 1 && S.prm.sensor == 2) {
        if (tid == 0) *F.n = 0;
        return;
    }
    const int nc = ctl.ext_corners ? 1 : S.prm.n_cells;
    if (tid == 0) {
        int acc = 0;
        for (int c = 0; c < nc; c++) {
            cell_off[c] = acc;
            acc += ctl.ext_corners ? ctl.n_ext[eye] : FB.cell_n[eye][c];
        }
        cell_off[nc] = acc;
    }
    __syncthreads();
    const int total_in = cell_off[nc];
    const int W = S.prm.W, H = S.prm.H;
    const bool rgbd = (S.prm.sensor == 2);
    int n_out = 0;
    for (int base = 0; base < total_in; base += 1024) {
        const int g = base + tid;
        bool keep = false;
        float x = 0, y = 0, r = 0, ox = 0, oy = 0, dep = 0;
        if (g < total_in) {
            if (ctl.ext_corners) {
                x = FB.ext_xy[eye][2 * g];
                y = FB.ext_xy[eye][2 * g + 1];
            } else {
                int c = 0;
                while (g >= cell_off[c + 1]) c++;
                const float *kp = FB.cell_kp[eye] + ((size_t)c * CELL_OUT_CAP + (g - cell_off[c])) * 3;
                x = kp[0];
                y = kp[1];
                r = kp[2];
            }
            ox = x;
            oy = y;
            keep = brief_border_keep(x, y, H, W);
            if (keep && rgbd) {  // handler.cpp:255-265 (depth at the distorted pixel), :268-294
                dep = FBd.depth_img[(size_t)((int)y) * FBd.depth_pitch + (int)x];
                keep = (dep >= S.prm.near_plane && dep <= S.prm.far_plane);
                if (keep && S.prm.undistort) {
                    undistort_point(S.prm, x, y, x, y);
                    const int hy = (int)floorf(y / (float)HASH_CELL), hx = (int)floorf(x / (float)HASH_CELL);
                    if (hy < 0 || hy >= S.prm.hash_ccy || hx < 0 || hx >= S.prm.hash_ccx) keep = false;
                }
            }
        }
        int total;
        const int off = n_out + block_excl_scan(keep ? 1 : 0, scan, &total);
        if (keep && off < NF_MAX) {
            F.x[off] = x;
            F.y[off] = y;
            F.resp[off] = r;
            F.bx[off] = ox;
            F.by[off] = oy;
            F.depth[off] = dep;
            F.flag[off] = 0;
            F.hcy[off] = (int16_t)floorf(y / (float)HASH_CELL);
            F.hcx[off] = (int16_t)floorf(x / (float)HASH_CELL);
        }
        n_out += total;
    }
    if (tid == 0) {
        if (n_out > NF_MAX) {
            atomicOr(&ctl.overflow, OVF_FEATURES);
            n_out = NF_MAX;
        }
        *F.n = n_out;
    }
}

// =================================================================================================
// k_brief : one wavefront per key point, lane l evaluates tests 4l..4l+3
// =================================================================================================
__device__ __forceinline__ int box_at(const Seq &S, const FrameBuf &FB, const FrameBuf &FBd, int eye, int iy, int ix) {
    const int W = S.prm.W, H = S.prm.H;
    if (ix >= 0 && ix < W && iy >= 0 && iy < H) return FB.boxsum[eye][(size_t)iy * S.plane_pitch + ix];
    // centre outside the image (fractional external corners at the border only): clipped window
    int s = 0;
    for (int y = max(iy - 4, 0); y <= min(iy + 4, H - 1); y++)
        for (int x = max(ix - 4, 0); x <= min(ix + 4, W - 1); x++) s += FBd.img[eye][(size_t)y * FBd.img_pitch + x];
    return s;
}

// publish_seq != 0: k_brief is the last kernel of the feature stage and its last workgroup publishes the frame's sequence number
// for the gates of the other streams (a one-thread kernel behind it cost the feature chain a launch and a boundary)
// The 512 box sums of a key point lie in the 49x49 window around it (|offset| <= 24: a 48-px patch).  Fetching them one by one made every wave touch
// ~50 cache lines per load instruction, eight times over (4 waves x 70 lines do not stay in the L1): the batched launch ran at the
// L2 -> L1 rate.  Now the wave copies the window ONCE, row-contiguous, into its own 4.9 KB of LDS (20 coalesced loads per lane, issued
// for the next key point before the current one is evaluated) and samples it there.  Lane l evaluates, in round w, the test whose
// result is bit l of descriptor word w (test 64 w + 8 (l / 8) + 7 - l % 8: OpenCV packs test 8 j + k into bit 7 - k of byte j), so
// four ballots ARE the descriptor -- no cross-lane traffic.
constexpr int BR_R = 24, BR_ROWS = 2 * BR_R + 1, BR_DW = 25;      // window rows; dwords per window row (49 u16 columns + 1 for an odd start)
static_assert(brief_max_offset() <= BR_R, "k_brief's window does not cover the test pattern");
constexpr int BR_WIN_DW = BR_ROWS * BR_DW, BR_LOADS = (BR_WIN_DW + 63) / 64;
template <bool BV>
__global__ __launch_bounds__(256) void k_brief(SeqArg<BV> sa, const Seq *seqs, int par, seq_t publish_seq) {
    // workgroups go to the 8 XCDs round-robin by their linear id, and every XCD has its own L2: with the plain mapping each L2 sees the
    // box-sum planes of ALL sequences (30 MB for a batch of 16) and every window comes from the Infinity Cache.  When the number of planes
    // is a multiple of 8, XCD x takes planes x, x + 8, ... whole, so a plane is pulled into ONE L2, once.
    const int planes = gridDim.y * gridDim.z;
    int lid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    int plane = lid / gridDim.x, bx = lid % gridDim.x;
    if ((planes & 7) == 0) {
        const int j = lid >> 3;
        plane = (lid & 7) + 8 * (j / gridDim.x);
        bx = j % gridDim.x;
    }
    const Seq &S = BV ? sa.get() : seq_const(seqs, plane / gridDim.y);  // (sa.get() indexes by blockIdx.z: the batch form picks its plane's sequence itself)
    const FrameBuf &FBd = seq_const(seqs, plane / gridDim.y).fb[par];  // per-frame image pointers (border fall-back only)
    const int eye = plane % gridDim.y;
    const FrameBuf &FB = S.fb[par];
    const Feat &F = FB.feat[eye];
    const bool poison = FB.fc->poison != 0;  // (block-uniform; reset by the publishing workgroup below, after every workgroup has read it)
    const int n = poison ? 0 : *F.n;
    const int lane = lane_id();
    const int wpb = blockDim.x >> 6;
    __shared__ uint32_t s_win[4][BR_WIN_DW + 64];
    uint32_t *win = s_win[wave_id()];
    typedef uint16_t __attribute__((may_alias)) u16_alias;  // (the window is written as words and sampled as halves)
    const u16_alias *win16 = reinterpret_cast<const u16_alias *>(win);
    const int W = S.prm.W, H = S.prm.H, ppd = S.plane_pitch >> 1;  // (the pitch is a multiple of 64)
    const uint32_t *box32 = reinterpret_cast<const uint32_t *>(FB.boxsum[eye]);
    // per-lane constants: where this lane's window words come from, and which eight window entries its four tests compare
    int goff[BR_LOADS];
#pragma unroll
    for (int i = 0; i < BR_LOADS; i++) {
        const int e = min(lane + 64 * i, BR_WIN_DW - 1);
        goff[i] = (e / BR_DW) * ppd + (e % BR_DW);
    }
    int ta[4], tb[4];
    signed char tq[4][4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const signed char *t = c_brief[64 * w + 8 * (lane >> 3) + 7 - (lane & 7)];
        tq[w][0] = t[0], tq[w][1] = t[1], tq[w][2] = t[2], tq[w][3] = t[3];
        ta[w] = (t[0] + BR_R) * (2 * BR_DW) + t[1] + BR_R;
        tb[w] = (t[2] + BR_R) * (2 * BR_DW) + t[3] + BR_R;
    }
    auto centre = [&](int i, int &cy, int &cx) {
        cy = (int)((double)F.by[i] + 0.5), cx = (int)((double)F.bx[i] + 0.5);
    };
    auto inside = [&](int cy, int cx) { return cx - BR_R >= 0 && cx + BR_R < W && cy - BR_R >= 0 && cy + BR_R < H; };
    // (a single sequence: 2 planes, so the plane-per-XCD mapping above does not apply and every XCD would sample windows all over both
    //  planes -- 5.4 MB fetched for 1.9 MB of box sums.  The key points are stored in detection-cell order: XCD x takes the x-th EIGHTH of
    //  them, i.e. the windows of about one cell and a quarter, and its L2 pulls that part of the plane only.)
    int stride = gridDim.x * wpb;
    int i = bx * wpb + wave_id();
    int i_end = n;
    if ((planes & 7) != 0 && (gridDim.x & 7) == 0) {
        const int eighth = (n + 7) >> 3, x = bx & 7;   // (planes are whole multiples of gridDim.x workgroups: bx & 7 is this workgroup's XCD)
        stride = (gridDim.x >> 3) * wpb;
        i = x * eighth + (bx >> 3) * wpb + wave_id();
        i_end = min(n, (x + 1) * eighth);
    }
    uint32_t v[BR_LOADS];
    int cy = 0, cx = 0;
    bool fast = false;
    auto fetch = [&](int cy_, int cx_) {  // the window's words, row-contiguous from the even column at or left of cx - 24
        // (address space 1 spelled out: a flat load would also count on lgkmcnt, and the LDS waits below would then wait for this prefetch)
        typedef const uint32_t __attribute__((address_space(1))) gu32;
        gu32 *src = (gu32 *)(box32 + (size_t)(cy_ - BR_R) * ppd + ((cx_ - BR_R) >> 1));
#pragma unroll
        for (int k = 0; k < BR_LOADS; k++) v[k] = src[goff[k]];
    };
    if (i < i_end) {
        centre(i, cy, cx);
        fast = inside(cy, cx);
        if (fast) fetch(cy, cx);
    }
    for (; i < i_end; i += stride) {
        uint64_t word[4];
        const int inext = i + stride;
        int ncy = 0, ncx = 0;
        bool nfast = false;
        if (inext < i_end) {
            centre(inext, ncy, ncx);
            nfast = inside(ncy, ncx);
        }
        if (fast) {
#pragma unroll
            for (int k = 0; k < BR_LOADS; k++) win[lane + 64 * k] = v[k];  // (the tail of the last round lands in the 64 spare words)
            if (nfast) fetch(ncy, ncx);  // in flight while this key point is evaluated
            const int odd = (cx - BR_R) & 1;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int a = win16[ta[w] + odd], b = win16[tb[w] + odd];
                word[w] = __ballot(a < b);
            }
        } else {  // window not inside the image (external corners at the border only): element-wise, clipped box sums
            if (nfast) fetch(ncy, ncx);
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int a = box_at(S, FB, FBd, eye, cy + tq[w][0], cx + tq[w][1]);
                const int b = box_at(S, FB, FBd, eye, cy + tq[w][2], cx + tq[w][3]);
                word[w] = __ballot(a < b);
            }
        }
        if (lane < 4) F.desc[(size_t)i * 4 + lane] = (lane == 0) ? word[0] : (lane == 1) ? word[1] : (lane == 2) ? word[2] : word[3];
        cy = ncy, cx = ncx, fast = nfast;
    }
    if (publish_seq) {
        __syncthreads();  // every wave's descriptors are written ...
        if (threadIdx.x == 0) {
            FeatCtl &fc = *FB.fc;
            __threadfence();  // ... and visible before this workgroup counts itself
            if (atomicAdd(&fc.done_blocks, 1u) == gridDim.x * gridDim.y - 1) {
                fc.done_blocks = 0;
                if (poison) {  // the frame has no features: say so to the gates of the other streams, then release the flag
                    fc.skip_seq = publish_seq;
                    fc.poison = 0;
                }
                S.ctl->dbg[47] = (long long)wall_clock64();  // (written from the feature stream: the frame it belongs to may differ)
                __threadfence();
                atomicExch(&fc.feat_seq, publish_seq);
            }
        }
    }
}


// =================================================================================================
// k_brief_img : BRIEF-256 WITHOUT the box-sum plane (LVT_AMD_BRIEF_FROM_IMAGE=1; verdict r4 item 2).  The wave copies the key point's 57 x 57 u8 patch
// (the 49 x 49 sample window + the 9 x 9 kernel's margin) into LDS, builds the horizontal 9-sums of all its rows there (v_sad_u8 on whole words, as
// k_score does), and every test adds nine of them per box.  Same sums of the same 81 pixels as the plane holds: descriptors bit-identical.
// =================================================================================================
constexpr int BI_R = BR_R + 4, BI_ROWS = 2 * BI_R + 1;   // patch rows (57)
constexpr int BI_DW = 16;                               // dwords per patch row: columns x0 .. x0 + 63, x0 = (cx - 28) & ~3 (57 + 3 <= 60 used)
constexpr int BI_G = 13;                                // groups of four horizontal sums per row: local columns 0 .. 51
constexpr int BI_PATCH_DW = BI_ROWS * BI_DW, BI_LOADS = (BI_PATCH_DW + 63) / 64;
constexpr int BI_GROUPS = BI_ROWS * BI_G, BI_GLOOPS = (BI_GROUPS + 63) / 64;
template <bool BV>
__global__ __launch_bounds__(256) void k_brief_img(SeqArg<BV> sa, const Seq *seqs, int par, seq_t publish_seq) {
    const int planes = gridDim.y * gridDim.z;
    int lid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    int plane = lid / gridDim.x, bx = lid % gridDim.x;
    if ((planes & 7) == 0) {
        const int j = lid >> 3;
        plane = (lid & 7) + 8 * (j / gridDim.x);
        bx = j % gridDim.x;
    }
    const Seq &S = BV ? sa.get() : seq_const(seqs, plane / gridDim.y);
    const FrameBuf &FBd = seq_const(seqs, plane / gridDim.y).fb[par];  // per-frame image pointers
    const int eye = plane % gridDim.y;
    const FrameBuf &FB = S.fb[par];
    const Feat &F = FB.feat[eye];
    const bool poison = FB.fc->poison != 0;
    const int n = poison ? 0 : *F.n;
    const int lane = lane_id();
    const int wpb = blockDim.x >> 6;
    __shared__ uint32_t s_patch[4][BI_PATCH_DW + 64];
    __shared__ __attribute__((aligned(8))) uint2 s_hs[4][BI_GROUPS + 64];
    uint32_t *pat = s_patch[wave_id()];
    uint2 *hs = s_hs[wave_id()];
    typedef uint16_t __attribute__((may_alias)) u16_alias;
    const u16_alias *hs16 = reinterpret_cast<const u16_alias *>(hs);   // [row][52] horizontal sums
    const int W = S.prm.W, H = S.prm.H;
    const uint8_t *img = FBd.img[eye];
    const int pitch = FBd.img_pitch;  // (a multiple of 16: rows are word-aligned)
    int goff[BI_LOADS];
#pragma unroll
    for (int i = 0; i < BI_LOADS; i++) {
        const int e = min(lane + 64 * i, BI_PATCH_DW - 1);
        goff[i] = (e / BI_DW) * (pitch >> 2) + (e % BI_DW);
    }
    int ta[4], tb[4];
    signed char tq[4][4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const signed char *t = c_brief[64 * w + 8 * (lane >> 3) + 7 - (lane & 7)];
        tq[w][0] = t[0], tq[w][1] = t[1], tq[w][2] = t[2], tq[w][3] = t[3];
        // box (dy, dx): rows cy + dy - 4 .. + 4 = patch rows dy + 24 .. dy + 32; left column cx + dx - 4 = local column dx + 24 + (cx - 28 - x0)
        ta[w] = (t[0] + BR_R) * (4 * BI_G) + t[1] + BR_R;
        tb[w] = (t[2] + BR_R) * (4 * BI_G) + t[3] + BR_R;
    }
    auto centre = [&](int i, int &cy, int &cx) {
        cy = (int)((double)F.by[i] + 0.5), cx = (int)((double)F.bx[i] + 0.5);
    };
    // the patch's 64 columns must lie inside the row pitch and its rows inside the image
    auto inside = [&](int cy, int cx) { return cx - BI_R >= 0 && ((cx - BI_R) & ~3) + 64 <= pitch && cx + BI_R < W && cy - BI_R >= 0 && cy + BI_R < H; };
    int stride = gridDim.x * wpb;
    int i = bx * wpb + wave_id();
    int i_end = n;
    if ((planes & 7) != 0 && (gridDim.x & 7) == 0) {
        const int eighth = (n + 7) >> 3, x = bx & 7;
        stride = (gridDim.x >> 3) * wpb;
        i = x * eighth + (bx >> 3) * wpb + wave_id();
        i_end = min(n, (x + 1) * eighth);
    }
    uint32_t v[BI_LOADS];
    int cy = 0, cx = 0;
    bool fast = false;
    auto fetch = [&](int cy_, int cx_) {
        typedef const uint32_t __attribute__((address_space(1))) gu32;
        gu32 *src = (gu32 *)(reinterpret_cast<const uint32_t *>(img + (size_t)(cy_ - BI_R) * pitch) + (((cx_ - BI_R) & ~3) >> 2));
#pragma unroll
        for (int k = 0; k < BI_LOADS; k++) v[k] = src[goff[k]];
    };
    // clipped 9 x 9 sum around (iy, ix) straight from the image (key points whose patch leaves the image: external corners at the border only)
    auto box_slow = [&](int iy, int ix) -> int {
        int sum = 0;
        for (int y = max(iy - 4, 0); y <= min(iy + 4, H - 1); y++)
            for (int x = max(ix - 4, 0); x <= min(ix + 4, W - 1); x++) sum += img[(size_t)y * pitch + x];
        return sum;
    };
    if (i < i_end) {
        centre(i, cy, cx);
        fast = inside(cy, cx);
        if (fast) fetch(cy, cx);
    }
    for (; i < i_end; i += stride) {
        uint64_t word[4];
        const int inext = i + stride;
        int ncy = 0, ncx = 0;
        bool nfast = false;
        if (inext < i_end) {
            centre(inext, ncy, ncx);
            nfast = inside(ncy, ncx);
        }
        if (fast) {
#pragma unroll
            for (int k = 0; k < BI_LOADS; k++) pat[lane + 64 * k] = v[k];
            if (nfast) fetch(ncy, ncx);  // in flight while this key point is evaluated
            // horizontal 9-sums: group g = (row, four local columns 4 j .. 4 j + 3) from the 12-byte segment at word j
#pragma unroll
            for (int k = 0; k < BI_GLOOPS; k++) {
                const int g = min(lane + 64 * k, BI_GROUPS - 1);
                const int r = g / BI_G, j = g - r * BI_G;
                const uint32_t w0 = pat[r * BI_DW + j], w1 = pat[r * BI_DW + j + 1], w2 = pat[r * BI_DW + j + 2];
                const uint32_t s0 = __builtin_amdgcn_sad_u8(w0, 0u, __builtin_amdgcn_sad_u8(w1, 0u, w2 & 255u));
                const uint32_t s1 = s0 - (w0 & 255u) + ((w2 >> 8) & 255u);
                const uint32_t s2 = s1 - ((w0 >> 8) & 255u) + ((w2 >> 16) & 255u);
                const uint32_t s3 = s2 - ((w0 >> 16) & 255u) + (w2 >> 24);
                hs[lane + 64 * k] = make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));  // (the tail of the last round lands in the 64 spare entries)
            }
            const int off = (cx - BI_R) & 3;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                int a = 0, b = 0;
#pragma unroll
                for (int r = 0; r < 9; r++) {
                    a += hs16[ta[w] + off + r * (4 * BI_G)];
                    b += hs16[tb[w] + off + r * (4 * BI_G)];
                }
                word[w] = __ballot(a < b);
            }
        } else {
            if (nfast) fetch(ncy, ncx);
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int a = box_slow(cy + tq[w][0], cx + tq[w][1]);
                const int b = box_slow(cy + tq[w][2], cx + tq[w][3]);
                word[w] = __ballot(a < b);
            }
        }
        if (lane < 4) F.desc[(size_t)i * 4 + lane] = (lane == 0) ? word[0] : (lane == 1) ? word[1] : (lane == 2) ? word[2] : word[3];
        cy = ncy, cx = ncx, fast = nfast;
    }
    if (publish_seq) {
        __syncthreads();
        if (threadIdx.x == 0) {
            FeatCtl &fc = *FB.fc;
            __threadfence();
            if (atomicAdd(&fc.done_blocks, 1u) == gridDim.x * gridDim.y - 1) {
                fc.done_blocks = 0;
                if (poison) {
                    fc.skip_seq = publish_seq;
                    fc.poison = 0;
                }
                S.ctl->dbg[47] = (long long)wall_clock64();
                __threadfence();
                atomicExch(&fc.feat_seq, publish_seq);
            }
        }
    }
}

}  // namespace lvt
