// 7x7 Gaussian blur of every level, then per keypoint: intensity-centroid orientation on the
// unblurred level, 256-bit steered BRIEF on the blurred level, scale-back and level-major output
// (GaussianBlur call src/ORBextractor.cc:1157-1159; IC_Angle :93-124; computeOrbDescriptor :127-169;
// assembly :1147-1181 of the reference).
#include "orbx_internal.h"


// ---------------------------------------------------------------- blur

struct BlurArgs {
    int nlevels, slab, bslab, variant, batch, tiles_per_frame;
    OrbxLevel0 l0;
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS], pitch[ORBX_MAX_LEVELS], off[ORBX_MAX_LEVELS], boff[ORBX_MAX_LEVELS];
};

#include "blur_stage.h"

#ifndef BLUR_TPW
#define BLUR_TPW 1  // tiles per workgroup (4, with the next tile's pieces fetched ahead, shortens the kernel alone by
                    // 2 % and costs a batch 2.7 % of its frames/s: fewer, longer workgroups interleave worse)
#endif

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
}

// Separable fixed-point filter: horizontal sums fit 16 bits (kernel sum <= 257), the vertical pass
// rounds with +32768 and shifts by 16; variant 1 (kernel sum 257) saturates to 255.
// Horizontal pass: a work item is 4 pixels of two rows; each sum is two or three 4-byte dot products of the
// three staged words around the pixels with the taps laid out at the pixel's offset; the two rows' sums of a
// column share one LDS word.
// Vertical pass: a thread takes 4 pixels of four rows; the 7 taps of a pixel are four 2 x u16 dot
// products on those row-pair words (the row's parity decides where the zero tap sits).
template <int VARIANT>  // 0: taps 18 34 48 56 (sum 256), 1: 18 34 49 55 (sum 257)
__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur,
                                              const uint32_t *__restrict__ tile_tab, BlurArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t img[BL_SH * BL_SWORDS];
    __shared__ __attribute__((aligned(16))) uint32_t hs[BL_PAIRS * BL_W];  // h(row 2p, x) | h(row 2p+1, x) << 16
    const int f = ORBX_BLOCK_FRAME;
    if (f >= a.batch)
        return;
    const uint32_t k0 = 18, k1 = 34, k2 = VARIANT ? 49 : 48, k3 = VARIANT ? 55 : 56;
    // the seven taps k0 k1 k2 k3 k2 k1 k0 laid over bytes o+1 .. o+7 of three words, for o = 0 .. 3
    const uint32_t t0a = (k0 << 8) | (k1 << 16) | (k2 << 24), t0b = k3 | (k2 << 8) | (k1 << 16) | (k0 << 24);
    const uint32_t t1a = (k0 << 16) | (k1 << 24), t1b = k2 | (k3 << 8) | (k2 << 16) | (k1 << 24), t1c = k0;
    const uint32_t t2a = k0 << 24, t2b = k1 | (k2 << 8) | (k3 << 16) | (k2 << 24), t2c = k1 | (k0 << 8);
    const uint32_t t3b = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24), t3c = k2 | (k1 << 8) | (k0 << 16);

    // A workgroup takes BLUR_TPW tiles in turn.  The level, row and column of a tile come from the host's table
    // (one scalar load instead of a search over the levels and a division in every wave: scalar instructions are
    // issued at the rate of the slower vector ones).  The next tile's pieces are fetched into registers while the
    // current one is filtered: with one tile's loads in flight per workgroup the bytes in flight per compute unit
    // do not cover the memory latency.
    const int unit0 = ORBX_BLOCK_UNIT * BLUR_TPW;
    const int nunits = min(BLUR_TPW, a.tiles_per_frame - unit0);
    auto tile_of = [&](int unit, int &lvl) -> BlurTile {
        const uint32_t te = tile_tab[unit];
        lvl = te >> 24;
        const int spitch = lvl == 0 ? a.l0.pitch : a.pitch[lvl];
        const uint8_t *src = lvl == 0 ? a.l0.ptr + (size_t)f * a.l0.fstride : pyr + (size_t)f * a.slab + a.off[lvl];
        return BlurTile{src, spitch, a.w[lvl], a.h[lvl], (int)(te & 0xfffu) * BL_W, (int)((te >> 12) & 0xfffu) * BL_H,
                        lvl == 0};
    };
    int lvl_next = 0;
    BlurTile next = tile_of(unit0, lvl_next);
    BlurFetch fx;
    blur_stage_fetch<BLUR_TPW == 1>(next, fx, img);
    for (int tile = 0; tile < nunits; tile++) {
    const BlurTile cur = next;
    const int lvl = lvl_next;
    const int tx0 = cur.tx0, ty0 = cur.ty0, h = cur.h, pitch = a.pitch[lvl];
    uint8_t *dst = blur + (size_t)f * a.bslab + a.boff[lvl];  // tiled: ORBX_BLUR_ADDR
    const int npairs = cur.npairs();
    // (every wave has left the previous tile's horizontal pass, the last reader of img, at the barrier behind it)
    blur_stage_commit(img, cur, fx);
    __syncthreads();
    if (tile + 1 < nunits) {
        next = tile_of(unit0 + tile + 1, lvl_next);
        blur_stage_fetch(next, fx);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int item = threadIdx.x + 256 * k;  // 19 row pairs x 32 groups = 608 items
        if (item < npairs * BL_GROUPS) {
            const int p = item >> 5, cg = item & 31;
            uint32_t hv[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                // staged words cg+3 .. cg+5 hold columns x-4 .. x+7 of the group's first pixel x; pixel x+o
                // has its taps at bytes o+1 .. o+7 of the three words: the taps, not the bytes, are shifted
                // (constant words), so a pixel is two or three 4-byte dot products and nothing else
                const uint32_t *wp = &img[(2 * p + rr) * BL_SWORDS + cg + 3];
                const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
                hv[rr][0] = __builtin_amdgcn_udot4(w1, t0b, __builtin_amdgcn_udot4(w0, t0a, 0u, false), false);
                hv[rr][1] = __builtin_amdgcn_udot4(
                    w2, t1c, __builtin_amdgcn_udot4(w1, t1b, __builtin_amdgcn_udot4(w0, t1a, 0u, false), false), false);
                hv[rr][2] = __builtin_amdgcn_udot4(
                    w2, t2c, __builtin_amdgcn_udot4(w1, t2b, __builtin_amdgcn_udot4(w0, t2a, 0u, false), false), false);
                hv[rr][3] = __builtin_amdgcn_udot4(w2, t3c, __builtin_amdgcn_udot4(w1, t3b, 0u, false), false);
            }
            uint4 pk;
            pk.x = hv[0][0] | (hv[1][0] << 16);
            pk.y = hv[0][1] | (hv[1][1] << 16);
            pk.z = hv[0][2] | (hv[1][2] << 16);
            pk.w = hv[0][3] | (hv[1][3] << 16);
            *reinterpret_cast<uint4 *>(&hs[p * BL_W + cg * 4]) = pk;
        }
    }
    __syncthreads();
    const int cg = threadIdx.x & 31, rb = threadIdx.x >> 5;  // rows 4rb .. 4rb+3 of the tile
    const int x = tx0 + cg * 4;
    if (ty0 + 4 * rb < h && x < pitch) {
        // output row r sums staged rows r .. r+6 with taps k0 k1 k2 k3 k2 k1 k0; a word of hs holds an
        // even staged row in its low half and the next (odd) row in its high half
        const uint32_t e01 = k0 | (k1 << 16), e23 = k2 | (k3 << 16), e45 = k2 | (k1 << 16), e67 = k0;
        const uint32_t o01 = k0 << 16, o23 = k1 | (k2 << 16), o45 = k3 | (k2 << 16), o67 = k1 | (k0 << 16);
        uint4 q[5];
#pragma unroll
        for (int j = 0; j < 5; j++)
            q[j] = *reinterpret_cast<const uint4 *>(&hs[(2 * rb + j) * BL_W + cg * 4]);
        // the thread's 4 x 4 pixels are one 16-byte piece of the tiled level (a group of 4 pixels, its 4 rows);
        // rows below the level fall into the padding tile rows
        uint32_t out[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool odd = j & 1;
            const int p0 = j >> 1;
            const uint32_t c01 = odd ? o01 : e01, c23 = odd ? o23 : e23, c45 = odd ? o45 : e45, c67 = odd ? o67 : e67;
            uint32_t acc[4];
            acc[0] = dot2_u16(q[p0 + 3].x, c67, dot2_u16(q[p0 + 2].x, c45, dot2_u16(q[p0 + 1].x, c23, dot2_u16(q[p0].x, c01, 32768u))));
            acc[1] = dot2_u16(q[p0 + 3].y, c67, dot2_u16(q[p0 + 2].y, c45, dot2_u16(q[p0 + 1].y, c23, dot2_u16(q[p0].y, c01, 32768u))));
            acc[2] = dot2_u16(q[p0 + 3].z, c67, dot2_u16(q[p0 + 2].z, c45, dot2_u16(q[p0 + 1].z, c23, dot2_u16(q[p0].z, c01, 32768u))));
            acc[3] = dot2_u16(q[p0 + 3].w, c67, dot2_u16(q[p0 + 2].w, c45, dot2_u16(q[p0 + 1].w, c23, dot2_u16(q[p0].w, c01, 32768u))));
            // byte 2 of each sum is the result; a sum of 2^24 (variant 1 only: its taps add up to 257)
            // saturates to 255
            if (VARIANT) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc[i] = min(acc[i], 0x00ffffffu);
            }
            const uint32_t lo = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u);  // bytes: acc0.b2, acc1.b2
            const uint32_t hi = __builtin_amdgcn_perm(acc[3], acc[2], 0x06020c0cu);  // upper half: acc2.b2, acc3.b2
            out[j] = lo | hi;
        }
        const int y0 = ty0 + 4 * rb;
        if (ty0 + ORBX_FT_H > h) {
            // bottom tile of the level (uniform branch): rows below the level were filtered from unstaged rows;
            // they land in the padding tile rows, which stay zero instead of run-dependent bytes
#pragma unroll
            for (int j = 0; j < 4; j++)
                out[j] = y0 + j < h ? out[j] : 0u;
        }
        *reinterpret_cast<uint4 *>(dst + (uint32_t)ORBX_BLUR_ADDR(x, y0, pitch)) = make_uint4(out[0], out[1], out[2], out[3]);
    }
    }  // tiles of the workgroup (the barrier behind the next tile's staging also guards hs)
}

hipError_t orbx_launch_blur(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    BlurArgs a;
    a.nlevels = g.nlevels, a.slab = g.slab, a.bslab = g.bslab, a.variant = g.blur_variant, a.batch = g.batch;
    a.tiles_per_frame = g.tiles_per_frame;
    a.l0 = b.l0;
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &L = g.lv[l];
        a.w[l] = L.w, a.h[l] = L.h, a.pitch[l] = L.pitch, a.off[l] = L.off, a.boff[l] = L.boff;
    }
    const dim3 grid = orbx_frame_grid((g.tiles_per_frame + BLUR_TPW - 1) / BLUR_TPW, g.batch);
    if (g.blur_variant)
        k_blur<1><<<grid, 256, 0, s>>>(b.pyr, b.blur, b.tile_tab, a);
    else
        k_blur<0><<<grid, 256, 0, s>>>(b.pyr, b.blur, b.tile_tab, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------- orientation + descriptor

// 512 sample points of the rBRIEF test table, (x,y) int8 pairs, point-major
constexpr int8_t k_pattern[1024] = {
#include "brief_pattern.inc"
};
// the 256 tests as floats (x0, x1, y0, y1): a lane reads the four it evaluates with four 16-byte loads, and the
// two points of a test sit side by side for the packed single-precision instructions
struct PatternF {
    float v[1024];
};
constexpr PatternF make_pattern_f()
{
    PatternF t{};
    for (int i = 0; i < 256; i++) {
        t.v[4 * i + 0] = (float)k_pattern[4 * i + 0];
        t.v[4 * i + 1] = (float)k_pattern[4 * i + 2];
        t.v[4 * i + 2] = (float)k_pattern[4 * i + 1];
        t.v[4 * i + 3] = (float)k_pattern[4 * i + 3];
    }
    return t;
}
__constant__ __attribute__((aligned(16))) PatternF c_pattern_f = make_pattern_f();

// Weights of the orientation patch, one signed byte per pixel, for the words a wave holds in registers.
// Lanes 0 .. 59 of register j hold word i = lane + 60 * j of the patch rows (12 words each, row i / 12,
// v = row - 15); once shifted to start at x - 15 the word's bytes are u = 4 * (i % 12) + b - 15.
// wu holds u, wv holds v, both 0 outside the circular patch, in rows past the 31st and in lanes 60 .. 63.
#define PAT_REGS 7
struct PatchWeights {
    uint32_t wu[PAT_REGS][64], wv[PAT_REGS][64];
};
constexpr PatchWeights make_patch_weights()
{
    constexpr int UMAX[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    PatchWeights t{};
    for (int j = 0; j < PAT_REGS; j++)
        for (int l = 0; l < 60; l++) {
            const int i = l + 60 * j, v = i / 12 - 15;
            for (int b = 0; b < 4; b++) {
                const int u = 4 * (i % 12) + b - 15;
                const int au = u < 0 ? -u : u, av = v < 0 ? -v : v;
                if (av <= 15 && au <= UMAX[av]) {
                    t.wu[j][l] |= (uint32_t)(u & 0xff) << (8 * b);
                    t.wv[j][l] |= (uint32_t)(v & 0xff) << (8 * b);
                }
            }
        }
    return t;
}
__constant__ PatchWeights c_patch_w = make_patch_weights();
// half-width of the circular patch at each |row| (umax, src/ORBextractor.cc:483-505):
// 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3

// cv::fastAtan2 (degrees).  Every operation is rounded separately, as on the host.
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.141592653589793238462643383279502884);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(
            90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0)
        a = __fsub_rn(180.f, a);
    if (y < 0)
        a = __fsub_rn(360.f, a);
    return a;
}

struct DescArgs {
    int nlevels, slab, bslab, kp_per_frame, out_cap, batch;
    OrbxLevel0 l0;
    int pitch[ORBX_MAX_LEVELS], off[ORBX_MAX_LEVELS], boff[ORBX_MAX_LEVELS], kp_base[ORBX_MAX_LEVELS],
        patch_size[ORBX_MAX_LEVELS];
    float scale[ORBX_MAX_LEVELS];
};

// Keypoints per wave, one after the other (the next one's window in flight while the current one is computed).
// With few keypoints per wave a frame's 2000 keypoints spread over more workgroups (125 at 4 per wave, 32 at 16),
// so fewer frames are in flight on an XCD at a time and more of a frame's window rows are still in its L2 when the
// next keypoint wants them: 16 -> 4 took the kernel from 0.088 to 0.075 ms per 64 KITTI frames (8: 0.077, 2: 0.083).  With the later, leaner
// keypoint loops the per-workgroup set-up (weights, test table, keypoint records) weighs more: per 256 frames
// 4: 0.2945 ms, 6: 0.286, 8: 0.287, 10: 0.291, 12: 0.294, 16: 0.299; in a batch 8 .. 12 give 2 % more frames/s
// than 4, 16 gives 5 % fewer.
#ifndef DESC_KPW
#define DESC_KPW 8
#endif
#define DESC_SLOTS (4 * DESC_KPW)  // keypoint slots per workgroup
#define PAT_ROWS 31    // orientation patch: rows y-15 .. y+15
#define PAT_WORDS 12   // 48 fetched bytes per row from the aligned word at or below x-15 (36 are needed)
// Descriptor window on the blurred level (rows y-18 .. y+18, columns x-18 .. x+18): fetched as the 64 x 40-pixel
// block of the tiled level that starts at the tile column of x-18 and the 4-row half tile of y-18, ten loads of
// four whole 64-byte pieces each (lanes 16t .. 16t+15 take the piece of tile column t), and laid out in LDS as
// rows of 64 bytes.
#define BLP_PITCH 64   // bytes per window row in LDS
#define BLP_REGS 10    // loads per window: one per half tile row
#define BLP_SLOTS (64 * BLP_REGS)  // words of a wave's window buffer

// One workgroup per DESC_SLOTS consecutive output slots of a frame, DESC_KPW keypoints per wave.  A
// keypoint is the same for all lanes of its wave, so the start of its windows is a scalar address and
// a lane adds only a fixed offset of its own.  The address unit spends a clock per 64-byte piece that a load
// instruction touches (tools/ubench_ta.hip), and a line filled into the L1 costs two more, so both fetches
// are shaped to touch few pieces: the orientation patch (unblurred, row-major level) as aligned words, 12 to a
// row from the word at or below x - 15 (9 would do, but rows of 9 or 11 words cost three times as much; words
// at odd byte addresses four times as much again); the descriptor window from the tiled blurred level as whole
// 64-byte pieces (BLP_* above).
// Orientation: the lane keeps its words of the 31 x 48 patch in registers, shifts each to start at
// x - 15 with the next word from its neighbour lane, and takes two signed 4 x 8-bit dot products
// with constant weights, u and v of the circular patch (0 outside): the sums of u * (I - 128) and
// v * (I - 128) are m10 and m01, because u and v sum to 0 over the patch.
// In between, one thread per keypoint evaluates cos/sin in double precision (once per keypoint instead
// of once per lane).  Descriptor: the wave copies the window block into LDS and each lane samples four
// tests.  The loads of keypoint i+1 are in flight while keypoint i is computed.
// The fetched patch rows may run a few bytes past the image width and the window block past the level's last
// rows and columns: both stay inside the slabs, which are allocated with slack.
// measurement hook: with DESC_EXP_ROWMASK=m a patch row r reads image row r & ~m (fewer cache lines, wrong results)
#ifdef DESC_EXP_ROWMASK
#define DESC_EXP_ROW(r) ((r) & ~(uint32_t)DESC_EXP_ROWMASK)
#else
#define DESC_EXP_ROW(r) (r)
#endif
#ifdef DESC_STAMPS
// phase stamps of sampled workgroups (measurement builds only, tools/desc_stamps.py)
__device__ unsigned long long g_desc_stamps[1024][8];
__device__ unsigned int g_desc_nstamp;
extern "C" int orbx_debug_desc_stamps(unsigned long long *out)
{
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_desc_nstamp), sizeof(n)) != hipSuccess)
        return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_desc_stamps), sizeof(unsigned long long) * 1024 * 8) != hipSuccess)
        return -1;
    const unsigned int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_desc_nstamp), &zero, sizeof(zero));
    return (int)(n < 1024 ? n : 1024);
}
#define STAMP(k) st[k] = __builtin_amdgcn_s_memrealtime()
#else
#define STAMP(k)
#endif

__global__ __launch_bounds__(256) void k_describe(const uint8_t *__restrict__ pyr, const uint8_t *__restrict__ blur,
                                                  const uint32_t *__restrict__ kept_xy,
                                                  const uint8_t *__restrict__ kept_resp, const int *__restrict__ nkept,
                                                  orbx_keypoint *__restrict__ out_kp, uint8_t *__restrict__ out_desc,
                                                  int *__restrict__ out_n, DescArgs a)
{
    __shared__ int s_first[ORBX_MAX_LEVELS + 1];
    __shared__ float s_angle[DESC_SLOTS], s_ca[DESC_SLOTS], s_sa[DESC_SLOTS];
    __shared__ int s_m01[DESC_SLOTS], s_m10[DESC_SLOTS];  // patch moments of the block's keypoints
    __shared__ int2 s_quad[4][DESC_KPW][16];              // per wave and keypoint: the moments summed over each quad of lanes
    __shared__ uint32_t s_win[4 * BLP_SLOTS];  // a window per wave
    __shared__ uint32_t s_xy[DESC_SLOTS];                // level coordinates of the block's keypoints
    __shared__ int s_lvl[DESC_SLOTS], s_ki[DESC_SLOTS];  // their level and index into the kept arrays
    __shared__ int s_pitch[DESC_SLOTS], s_off[DESC_SLOTS];  // row pitch of that level and its offset in the blurred slab
    __shared__ int s_upitch[DESC_SLOTS];                    // row pitch of the unblurred level
    __shared__ long long s_usrc[DESC_SLOTS];                // byte offset of the unblurred level from pyr
    __shared__ float s_scale[DESC_SLOTS], s_size[DESC_SLOTS], s_resp[DESC_SLOTS];
    const int f = ORBX_BLOCK_FRAME, unit = ORBX_BLOCK_UNIT;
    if (f >= a.batch)
        return;
#ifdef DESC_STAMPS
    unsigned long long st[8];
#endif
    STAMP(0);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // the constant weight words of the orientation patch: loaded first, so that they have arrived long before the
    // keypoint loop (see the wait in front of it)
    uint32_t wu[PAT_REGS], wvv[PAT_REGS];
#pragma unroll
    for (int j = 0; j < PAT_REGS; j++)
        wu[j] = c_patch_w.wu[j][lane], wvv[j] = c_patch_w.wv[j][lane];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int l = 0; l < a.nlevels; l++) {
            s_first[l] = acc;
            acc += nkept[f * a.nlevels + l];
        }
        s_first[a.nlevels] = acc;
        if (unit == 0)
            out_n[f] = acc;
    }
    __syncthreads();
    const int total = s_first[a.nlevels];
    const int base = unit * DESC_SLOTS;
    if (base >= total)
        return;
    STAMP(1);
    // all keypoints of the block are located with one round of loads, so that the window fetches
    // below depend on LDS only
    if (threadIdx.x < DESC_SLOTS && base + (int)threadIdx.x < total) {
        const int slot = base + threadIdx.x;
        int lvl = 0;
        while (slot >= s_first[lvl + 1])
            lvl++;
        const int ki = a.kp_base[lvl] + (slot - s_first[lvl]);
        s_lvl[threadIdx.x] = lvl;
        s_ki[threadIdx.x] = ki;
        s_pitch[threadIdx.x] = a.pitch[lvl];
        s_off[threadIdx.x] = a.boff[lvl];
        s_upitch[threadIdx.x] = lvl == 0 ? a.l0.pitch : a.pitch[lvl];
        s_usrc[threadIdx.x] = lvl == 0 ? (long long)(a.l0.ptr - pyr) + (long long)f * a.l0.fstride
                                       : (long long)f * a.slab + a.off[lvl];
        s_scale[threadIdx.x] = a.scale[lvl];
        s_size[threadIdx.x] = (float)a.patch_size[lvl];
        s_resp[threadIdx.x] = (float)kept_resp[(size_t)f * a.kp_per_frame + ki];
        s_xy[threadIdx.x] = kept_xy[(size_t)f * a.kp_per_frame + ki];
    }
    __syncthreads();
    STAMP(2);
    // everything a window fetch needs comes from LDS: no global load sits in front of it
    const int nmine = min(DESC_KPW, total - (base + wv * DESC_KPW));  // keypoints of this wave (may be <= 0)

    // ---- IC_Angle (src/ORBextractor.cc:85-117 of the reference) ----
    uint32_t prow[PAT_REGS], pcol[PAT_REGS];
#pragma unroll
    for (int j = 0; j < PAT_REGS; j++) {
        const uint32_t idx = min(lane, 59) + 60 * j;  // lanes 60 .. 63 repeat lane 59 and weigh 0
        const uint32_t q = __umul24(idx, 5462u) >> 16;  // idx / 12
        const uint32_t r = q < PAT_ROWS ? q : PAT_ROWS - 1;  // words of rows past the 31st weigh 0 and reread it
        prow[j] = r, pcol[j] = 4 * (idx - __umul24(q, PAT_WORDS));
    }
    auto fetch_patch = [&](int ls, uint32_t (&tmp)[PAT_REGS], uint32_t &ox) {
        const uint32_t xy = __builtin_amdgcn_readfirstlane(s_xy[ls]);
        const int x = xy & 0xffff, y = xy >> 16;
        const int pitch = __builtin_amdgcn_readfirstlane(s_upitch[ls]);
        const long long usrc = ((long long)__builtin_amdgcn_readfirstlane((int)(s_usrc[ls] >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)s_usrc[ls]);
        const int xa = (x - 15) & ~3;
        ox = (uint32_t)((x - 15) - xa);
        const uint8_t *src = pyr + usrc + (long long)(y - 15) * pitch + xa;
#pragma unroll
        for (int j = 0; j < PAT_REGS; j++)  // rows and pitch are below 2^13
            tmp[j] = *reinterpret_cast<const uint32_t *>(src + (__umul24(DESC_EXP_ROW(prow[j]), (uint32_t)pitch) + pcol[j]));
    };
    // Every earlier load has landed here (the weights among them).  Without this statement the compiler waits for
    // the weights inside the loop, at their first use, and that wait covers the loads just issued for the next
    // keypoint as well: every keypoint would then pay a whole memory round trip.
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    uint32_t pt[PAT_REGS], pox = 0;
    if (nmine > 0)
        fetch_patch(wv * DESC_KPW, pt, pox);
    for (int i = 0; i < nmine; i++) {
        // the four pixels from x - 15 + 4 * word on: this word and the next one (the neighbour lane's),
        // shifted by the patch's offset in its first word; then I - 128 as signed bytes
        const uint32_t ox = pox;
        uint32_t px[PAT_REGS];
#pragma unroll
        for (int j = 0; j < PAT_REGS; j++) {
            const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pt[j], 0x130, 0xf, 0xf, true);  // wave_shl:1
            px[j] = __builtin_amdgcn_alignbyte(nxt, pt[j], ox) ^ 0x80808080u;
        }
        if (i + 1 < nmine)
            fetch_patch(wv * DESC_KPW + i + 1, pt, pox);
        int m10 = 0, m01 = 0;
#pragma unroll
        for (int j = 0; j < PAT_REGS; j++) {
            m10 = __builtin_amdgcn_sdot4((int)px[j], (int)wu[j], m10, false);
            m01 = __builtin_amdgcn_sdot4((int)px[j], (int)wvv[j], m01, false);
        }
        // Sums over the wave: here only over the four lanes of a quad (two cross-lane additions each);
        // the sixteen quad sums of every keypoint are added up after the loop, one lane per keypoint,
        // instead of a full wave-wide reduction per keypoint.
        m10 += __builtin_amdgcn_update_dpp(0, m10, 0xb1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
        m01 += __builtin_amdgcn_update_dpp(0, m01, 0xb1, 0xf, 0xf, false);
        m10 += __builtin_amdgcn_update_dpp(0, m10, 0x4e, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
        m01 += __builtin_amdgcn_update_dpp(0, m01, 0x4e, 0xf, 0xf, false);
        if ((lane & 3) == 0)
            s_quad[wv][i][lane >> 2] = make_int2(m10, m01);
    }
    STAMP(3);
    __builtin_amdgcn_wave_barrier();
    if (lane < nmine) {
        int m10 = 0, m01 = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int2 p = s_quad[wv][lane][k];
            m10 += p.x, m01 += p.y;
        }
        s_m01[wv * DESC_KPW + lane] = m01, s_m10[wv * DESC_KPW + lane] = m10;
    }
    __syncthreads();
    STAMP(4);
    // the angles of the block's keypoints, one thread each (inside the loop every lane of a wave would
    // step through the polynomial for one keypoint)
    if (threadIdx.x < DESC_SLOTS && base + (int)threadIdx.x < total) {
        const float ang = fast_atan2_deg((float)s_m01[threadIdx.x], (float)s_m10[threadIdx.x]);
        s_angle[threadIdx.x] = ang;
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        const float rad = __fmul_rn(ang, factorPI);
        double sn, cs;
#ifdef DESC_FAST_ANGLE
        sn = __sinf(rad), cs = __cosf(rad);
#else
        sincos((double)rad, &sn, &cs);
#endif
        s_ca[threadIdx.x] = (float)cs;
        s_sa[threadIdx.x] = (float)sn;
    }
    __syncthreads();
    STAMP(5);

    // ---- steered BRIEF: lane L evaluates tests L, L+64, L+128, L+192; a ballot yields 64 descriptor bits ----
    float4 pat[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        pat[j] = *reinterpret_cast<const float4 *>(&c_pattern_f.v[(lane + 64 * j) * 4]);
    // this lane's bytes inside a 4-row strip of the block: tile column lane / 16, then the piece's own order
    // (group of 4 pixels, row); in LDS the same word lies at row lane & 3, byte column 16 * (lane / 16) + 4 * group
    const uint32_t lane_src = (uint32_t)(lane >> 4) * 128u + (uint32_t)(lane & 15) * 4u;
    const uint32_t lane_dst = (uint32_t)(lane & 3) * (BLP_PITCH / 4) + (uint32_t)(lane >> 4) * 4u + (uint32_t)((lane & 15) >> 2);
    auto fetch_window = [&](int ls, uint32_t (&tmp)[BLP_REGS], uint32_t &oxy) {
        const uint32_t xy = __builtin_amdgcn_readfirstlane(s_xy[ls]);
        const int x = xy & 0xffff, y = xy >> 16;
        const int pitch = __builtin_amdgcn_readfirstlane(s_pitch[ls]);
        const int off = __builtin_amdgcn_readfirstlane(s_off[ls]);
        const int ax = (x - 18) & ~15, ay = (y - 18) & ~3;
        oxy = (uint32_t)((y - 18 - ay) * BLP_PITCH + (x - 18 - ax));  // the window's first pixel inside the block
        const uint8_t *src = blur + (size_t)f * a.bslab + off + (ax >> 4) * 128;
        const int hr0 = ay >> 2;  // first half tile row
#pragma unroll
        for (int j = 0; j < BLP_REGS; j++) {
            const int hr = hr0 + j;
            tmp[j] = *reinterpret_cast<const uint32_t *>(src + ((uint32_t)((hr >> 1) * (pitch * 8) + (hr & 1) * 64) + lane_src));
        }
    };
    uint32_t *win = s_win + wv * BLP_SLOTS;
    // Rounding to the nearest integer, ties to even, by adding 1.5 * 2^23: the sum's bit pattern is
    // 0x4b400000 + n.  With window rows of 64 bytes, row * 64 + column comes out as 64 * n_row + n_col + RND_BIAS
    // (modulo 2^32) in one shift-and-add of the two bit patterns.
    const float RND = 12582912.f;
    const uint32_t RND_BIAS = (0x4b400000u << 6) + 0x4b400000u;  // modulo 2^32, as the sums below
    const uint32_t ctr0 = (uint32_t)(wv * BLP_SLOTS * 4 + 18 * BLP_PITCH + 18) - RND_BIAS;
    const uint8_t *winb = reinterpret_cast<const uint8_t *>(s_win);
    uint32_t wt[BLP_REGS], wox = 0;
    if (nmine > 0)
        fetch_window(wv * DESC_KPW, wt, wox);
    for (int i = 0; i < nmine; i++) {
        const int ls = wv * DESC_KPW + i, slot = base + ls;
        __builtin_amdgcn_wave_barrier();  // the previous keypoint's samples are read before this overwrite
        const uint32_t ctr = ctr0 + wox;  // the window's offset in its first word
#pragma unroll
        for (int j = 0; j < BLP_REGS; j++)
            win[lane_dst + 64 * j] = wt[j];  // strip j: rows 4j .. 4j+3 of the block
        if (i + 1 < nmine)
            fetch_window(ls + 1, wt, wox);
        __builtin_amdgcn_wave_barrier();
        const float ca = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__float_as_int(s_ca[ls])));
        const float sa = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__float_as_int(s_sa[ls])));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 sa2 = {sa, sa}, ca2 = {ca, ca}, nsa2 = {-sa, -sa}, rnd2 = {RND, RND};
        unsigned long long bits[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // rows x * sin + y * cos, columns x * cos - y * sin, each operation rounded (src/ORBextractor.cc:124-135)
            // The two points of a test at a time in the packed single-precision instructions: every operation
            // still rounds on its own (no contraction: the file is compiled with -ffp-contract=off), and
            // x * cos - y * sin is x * cos + y * (-sin) exactly.
            const f2 X = {pat[j].x, pat[j].y}, Y = {pat[j].z, pat[j].w};
            const f2 R = (X * sa2 + Y * ca2) + rnd2;
            const f2 Q = (X * ca2 + Y * nsa2) + rnd2;
            const uint32_t k0 = (__float_as_uint(R.x) << 6) + __float_as_uint(Q.x) + ctr;
            const uint32_t k1 = (__float_as_uint(R.y) << 6) + __float_as_uint(Q.y) + ctr;
            const int t0 = winb[k0], t1 = winb[k1];
            bits[j] = __ballot(t0 < t1);
        }
        const size_t o = (size_t)f * a.out_cap + slot;
        if (lane < 4)
            *reinterpret_cast<unsigned long long *>(out_desc + o * 32 + lane * 8) =
                lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
    }
    STAMP(6);
    // the keypoint records, one thread each
    if (threadIdx.x < DESC_SLOTS && base + (int)threadIdx.x < total) {
        const int ls = threadIdx.x;
        const uint32_t xy = s_xy[ls];
        orbx_keypoint kp;
        kp.x = (float)(xy & 0xffff), kp.y = (float)(xy >> 16);
        if (s_lvl[ls] != 0) {
            kp.x = __fmul_rn(kp.x, s_scale[ls]);
            kp.y = __fmul_rn(kp.y, s_scale[ls]);
        }
        kp.size = s_size[ls];
        kp.angle = s_angle[ls];
        kp.response = s_resp[ls];
        kp.octave = s_lvl[ls];
        kp.class_id = -1;
        out_kp[(size_t)f * a.out_cap + base + ls] = kp;
    }
#ifdef DESC_STAMPS
    STAMP(7);
    if (threadIdx.x == 0 && (unit % 5) == 0) {
        const unsigned int k = atomicAdd(&g_desc_nstamp, 1u);
        if (k < 1024)
            for (int i = 0; i < 8; i++)
                g_desc_stamps[k][i] = st[i];
    }
#endif
}

hipError_t orbx_launch_describe(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    DescArgs a;
    a.nlevels = g.nlevels, a.slab = g.slab, a.bslab = g.bslab, a.kp_per_frame = g.kp_per_frame, a.out_cap = g.out_cap;
    a.batch = g.batch;
    a.l0 = b.l0;
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &L = g.lv[l];
        a.pitch[l] = L.pitch, a.off[l] = L.off, a.boff[l] = L.boff, a.kp_base[l] = L.kp_base, a.patch_size[l] = L.patch_size;
        a.scale[l] = L.scale;
    }
    const dim3 grid = orbx_frame_grid((g.out_cap + DESC_SLOTS - 1) / DESC_SLOTS, g.batch);
    k_describe<<<grid, 256, 0, s>>>(b.pyr, b.blur, b.kept_xy, b.kept_resp, b.nkept, b.out_kp, b.out_desc, b.out_n, a);
    return hipGetLastError();
}
