// Per-cell FAST-9/16 detection (ORBextractor::ComputeKeyPointsOctTree, src/ORBextractor.cc:815-893
// of the reference, which calls cv::FAST(cell, keys, threshold, nonmax=true) per 30-px cell).
//
// The score cv::FAST attaches to a corner, cornerScore = max(t, A) - 1, rests on a quantity that does
// not depend on the threshold t:  A(p) = max over the 16 arcs of 9 contiguous ring pixels of
// min(v - ring) and of min(ring - v).  p is a corner at t  <=>  A(p) > t, and it survives the
// non-maximum suppression of its cell  <=>  A(p) > A(q) for the 8 neighbours q that lie in the same
// cell's detect region (pixels 3 or more inside the cell window); neighbours with A(q) <= t lose
// against A(p) > t anyway.  The detect regions of the cells tile [19, w-19) x [19, h-19) exactly.
// So one pass per level serves both thresholds: every strict per-cell local maximum with
// A > min(iniThFAST, minThFAST) is appended to the level's candidate list, and a per-cell counter records how many
// of them exceed iniThFAST.  The consumer keeps a candidate iff A > iniThFAST, or its cell has no
// such candidate (the fallback of :874-878).  Candidates are appended to the level's arrays in arbitrary
// order (a workgroup reserves the slots of its tile with one atomic on the level's counter); each
// carries an order key (cell-major, then row-major inside the cell), the order in which the
// reference appends them.
//
// Work per 64x64 tile:  stage the image tile (+4 px) in LDS; test every pixel of the tile (+1 px
// ring) against the four compass points, four pixels per 32-bit operation (a 9-arc contains at
// least one pixel of each antipodal pair); queue the few that pass; compute A only for those, both
// polarities at once with packed 16-bit min/max (half-precision instructions on values whose float order is their
// integer order: they have three-input forms and operand modifiers that widen the bytes for free); suppress
// non-maxima; append to the level's candidate arrays.
#include "orbx_internal.h"

#ifndef FAST_CUT
#define FAST_CUT(n)  // measurement hook of tools/exp_fast.hip: leave the kernel after phase n
#endif

#define IMG_W 96                 // staged bytes per row: columns tx0-16 .. tx0+79 (16-byte pieces)
#define IMG_X0 16                // staged column of pixel tx0
#define IMG_H (ORBX_FAST_H + 8)  // 72 staged rows: ty0-4 .. ty0+67
#define IMG_WORDS (IMG_W / 4)    // 24 words per staged row
#define GRP_0 3                  // first word of a row that matters: columns tx0-4 .. tx0-1
#define GRP_W 18                 // words that matter: columns tx0-4 .. tx0+67
#define SC_W (ORBX_FAST_W + 2)   // 66: tile plus a one-pixel ring
#define SC_H (ORBX_FAST_H + 2)   // 66: staged rows 3..68
#define QUEUE_CAP (SC_H * SC_W)   // every pixel of the tile and its ring may pass the compass test
#define SC_P 72                  // bytes per row of the score map: staged columns 12 .. 83 (the 18 words that matter)
#define SC_AT(r, c) (((r) - 3) * SC_P + (c) - 4 * GRP_0)  // score of staged row r (3 .. 68), staged column c

// Compass test of four pixels at once, one per byte, on pixel values halved to 7 bits (the eighth bit of
// every byte is the room the comparisons need).  A 9-arc holds at least one pixel of each antipodal pair,
// so a corner at threshold t needs a ring pixel r with r < v - t in both compass pairs (dark arc) or with
// r > v + t in both (bright arc).  With q = value >> 1 and qt = (t + 1) >> 1:
//     r < v - t  =>  q(r) <= q(v) - qt        r > v + t  =>  q(r) >= q(v) + qt
// (checked for every v, r and t), so the halved test passes every pixel the exact one passes and about 3 %
// more; it only selects the pixels whose exact score is computed.  Bytes never borrow from or carry into
// their neighbours: every minuend has its top bit set and every subtrahend is a 7-bit value, and the top bit
// of a difference is the comparison.  Where q(v) - qt < 0 or q(v) + qt > 127 no ring pixel can pass; there the
// low bits of x and z below are meaningless, and their top bits, which say so, mask the result.  All of it is
// 32-bit add / subtract / logic, which this chip issues at twice the rate of the packed 16-bit minimum /
// maximum instructions.  Returns the pass bits in bit 7 of each byte.
__device__ __forceinline__ uint32_t compass4(uint32_t vw, uint32_t r0w, uint32_t r8w, uint32_t r4w, uint32_t r12w,
                                             uint32_t qt4)
{
    const uint32_t H = 0x80808080u, L7 = 0x7f7f7f7fu;
    const uint32_t hv = vw >> 1, q0 = (r0w >> 1) & L7, q8 = (r8w >> 1) & L7, q4 = (r4w >> 1) & L7,
                   q12 = (r12w >> 1) & L7;
    const uint32_t x = (hv | H) - qt4;   // top bit: q(v) >= qt; then the low bits are q(v) - qt
    const uint32_t z = (~hv | H) - qt4;  // top bit: q(v) + qt <= 127; then the low bits are 127 - q(v) - qt
    const uint32_t lo = x | H;           // lo - q has its top bit set iff q <= q(v) - qt
    const uint32_t c = (z | H) - L7;     // 128 - q(v) - qt: q + c has its top bit set iff q >= q(v) + qt
    const uint32_t dark = ((lo - q0) | (lo - q8)) & ((lo - q4) | (lo - q12)) & x;
    const uint32_t bright = ((q0 + c) | (q8 + c)) & ((q4 + c) | (q12 + c)) & z;
    return dark | bright;
}

// Packed maximum / minimum of the arc score.  A register holds a value in 0 .. 255 in its low half and the
// NEGATED value of another quantity in its high half (sign bit | value), both read as half-precision floats:
// the bit patterns 0x0000 .. 0x00ff are +0 and the smallest subnormal numbers, ordered like the integers they
// spell, and 0x8000 | x is -x, so a maximum of the high halves is minus the minimum of the values (the kernel runs
// with subnormal numbers kept, the compiler's setting for this chip).  The chip has three-input packed maximum /
// minimum for this type only.
__device__ __forceinline__ uint32_t pkf_max(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pkf_min(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pkf_max3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pkf_min3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// From two plain bytes (low halves): max(a, b) | -min(a, b) << 16 and min(a, b) | -max(a, b) << 16.  Both halves
// of the result read the low halves of the operands, the upper one negated: the widening costs no instruction.
__device__ __forceinline__ uint32_t pkf_max_raw(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_max_f16 %0, %1, %2 op_sel_hi:[0,0] neg_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pkf_min_raw(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_min_f16 %0, %1, %2 op_sel_hi:[0,0] neg_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// t points at the centre byte of the staged tile.  Both polarities at once: the low halves carry maxima of ring
// pixels (the dark test), the high halves minima of ring pixels, negated (the bright test; see above):
//   dark   (ring < v):  max_arcs min(v - ring) = v - min_arcs max_9(ring)
//   bright (ring > v):  max_arcs min(ring - v) = max_arcs min_9(ring) - v = -min_arcs max_9(-ring) - v
__device__ __forceinline__ int fast_arc_score(const uint8_t *t)
{
    const int p = IMG_W;
    uint32_t d[16];
    d[0] = t[3 * p];
    d[1] = t[3 * p + 1];
    d[2] = t[2 * p + 2];
    d[3] = t[p + 3];
    d[4] = t[3];
    d[5] = t[-p + 3];
    d[6] = t[-2 * p + 2];
    d[7] = t[-3 * p + 1];
    d[8] = t[-3 * p];
    d[9] = t[-3 * p - 1];
    d[10] = t[-2 * p - 2];
    d[11] = t[-p - 3];
    d[12] = t[-3];
    d[13] = t[p - 3];
    d[14] = t[2 * p - 2];
    d[15] = t[3 * p - 1];
    // Arc k holds ring pixels k .. k+8.  Arcs k and k-1 share the eight pixels k .. k+7, so
    //   min(max_arc(k), max_arc(k-1)) = max(max8(k), min(d[k+8], d[k-1]))
    // and the even k cover all sixteen arcs: the eight-pixel maxima are needed at even k only, and each joins
    // the minimum of its two outer pixels in one three-input maximum of two four-pixel maxima and that minimum.
    uint32_t m2[8], m4[8], x[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
        m2[k] = pkf_max_raw(d[2 * k], d[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 8; k++)
        m4[k] = pkf_max(m2[k], m2[(k + 1) & 7]);
#pragma unroll
    for (int k = 0; k < 8; k++)  // pixels 2k .. 2k+7 and the outer pair
        x[k] = pkf_max3(m4[k], m4[(k + 2) & 7], pkf_min_raw(d[(2 * k + 8) & 15], d[(2 * k + 15) & 15]));
    const uint32_t best = pkf_min(pkf_min3(pkf_min3(x[0], x[1], x[2]), x[3], x[4]), pkf_min3(x[5], x[6], x[7]));
    const int v = t[0];
    const int dark = v - (int)(best & 0xffff);                // v - min_arcs max_9(ring)
    const int bright = (int)((best >> 16) & 0x7fff) - v;      // max_arcs min_9(ring) - v
    return max(dark, bright);
}

struct FastArgs {
    int nlevels, slab, cells_per_frame, cand_per_frame, tiles_per_frame, batch;
    OrbxLevel0 l0;
    int t_low, t_ini;  // min(iniThFAST, minThFAST) and iniThFAST, clamped to [1,255]
    int tile_base[ORBX_MAX_LEVELS + 1];
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS], pitch[ORBX_MAX_LEVELS], off[ORBX_MAX_LEVELS];
    int wcell[ORBX_MAX_LEVELS], hcell[ORBX_MAX_LEVELS], ncols[ORBX_MAX_LEVELS], cell_base[ORBX_MAX_LEVELS];
    int cand_base[ORBX_MAX_LEVELS], cand_cap[ORBX_MAX_LEVELS];
    uint32_t wc_magic[ORBX_MAX_LEVELS], hc_magic[ORBX_MAX_LEVELS];  // ceil(2^20 / cell size)
};


__device__ __forceinline__ int lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__global__ __launch_bounds__(256) void k_fast(const uint8_t *__restrict__ pyr, int *__restrict__ ncand,
                                              uint32_t *__restrict__ cand_xy, uint8_t *__restrict__ cand_resp,
                                              uint32_t *__restrict__ cand_key, uint8_t *__restrict__ cell_flag,
                                              int *__restrict__ status, const uint32_t *__restrict__ ftile_tab,
                                              FastArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[IMG_H * IMG_W];  // staged pixels
    __shared__ __attribute__((aligned(16))) uint8_t sc[SC_H * SC_P];     // A where it exceeds t_low, else 0
    // Pixels that passed the compass test (staged row<<7 | staged column), filled from the bottom.  The corners
    // among them that lie inside the tile are collected from the top of the same array downwards while the
    // queue is consumed: about 500 entries and 300 corners per tile, so the two meet only on images where
    // most pixels pass; such a tile visits all its pixels in the suppression pass instead.
    __shared__ uint16_t queue[QUEUE_CAP];
    __shared__ int qn, sn, cn;

    FAST_CUT(-1);
    const int f = ORBX_BLOCK_FRAME, unit = ORBX_BLOCK_UNIT;
    if (f >= a.batch)
        return;
    // level, tile row and tile column from the host's table (one scalar load; a search over the levels and a
    // division in every wave cost as many issue slots as a tenth of the kernel's vector work)
    const uint32_t te = ftile_tab[unit];
    const int lvl = te >> 24;
    const int tx0 = ORBX_FAST_X0 + (int)(te & 0xfffu) * ORBX_FAST_W;
    const int ty0 = ORBX_FAST_Y0 + (int)((te >> 12) & 0xfffu) * ORBX_FAST_H;
    const int w = a.w[lvl], h = a.h[lvl], pitch = lvl == 0 ? a.l0.pitch : a.pitch[lvl];
    const int wc = a.wcell[lvl], hc = a.hcell[lvl];
    const int x_lo = ORBX_CELL_ORG, x_hi = w - ORBX_CELL_ORG, y_lo = ORBX_CELL_ORG, y_hi = h - ORBX_CELL_ORG;
    // tiles that hold no detectable pixel, and levels without a cell grid, produce nothing
    if (wc == 0 || tx0 + ORBX_FAST_W <= x_lo || tx0 >= x_hi || ty0 + ORBX_FAST_H <= y_lo || ty0 >= y_hi)
        return;
    FAST_CUT(0);
    const uint8_t *src = lvl == 0 ? a.l0.ptr + (size_t)f * a.l0.fstride : pyr + (size_t)f * a.slab + a.off[lvl];
    const int lane = threadIdx.x & 63;
    uint32_t *img32 = reinterpret_cast<uint32_t *>(img);

    if (threadIdx.x == 0)
        qn = 0, sn = 0, cn = 0;
    // a thread keeps one word column (of the 18 that matter) and walks down the rows in steps of 14
    const int g = threadIdx.x % GRP_W, r0 = threadIdx.x / GRP_W;
    const bool walker = threadIdx.x < 14 * GRP_W;
    // Stage rows ty0-4 .. ty0+67 in whole 16-byte pieces (the tile grid starts at column 16 and row 19, so the
    // staged window never begins outside the level).  Only pixels up to 3 around the detectable range are ever
    // read: rows from y_hi + 3 on and pieces behind the row pitch stay unstaged.
    const bool aligned = (pitch & 15) == 0 && ((uintptr_t)src & 15) == 0;
    const int nrows = min(IMG_H, y_hi + 3 - (ty0 - 4));                       // staged rows 0 .. nrows-1
    const int npieces = min(IMG_W / 16, (pitch - (tx0 - IMG_X0)) >> 4);        // pieces 0 .. npieces-1 of a row
    // the score map starts at zero: 297 pieces of 16 bytes
    static_assert(SC_H * SC_P % 16 == 0 && SC_H * SC_P / 16 <= 512, "two pieces per thread cover the score map");
    auto zero_scores = [&]() {
        uint4 *z = reinterpret_cast<uint4 *>(sc);
        z[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
        if (threadIdx.x + 256 < SC_H * SC_P / 16)
            z[threadIdx.x + 256] = make_uint4(0u, 0u, 0u, 0u);
    };
    if (aligned) {
        const uint8_t *s0 = src + (size_t)(ty0 - 4) * pitch + (tx0 - IMG_X0);
        // 72 rows x 6 pieces = 432 items; a thread takes items i and i + 256 (256 = 42 * 6 + 4)
        static_assert(IMG_W / 16 == 6, "the item arithmetic below is for six pieces per row");
        const int ra = (int)(__umul24(threadIdx.x, 10923u) >> 16), ca = (int)threadIdx.x - 6 * ra;
        const int rb = ra + (ca >= 2 ? 43 : 42), cb = ca >= 2 ? ca - 2 : ca + 4;
        const bool oka = ra < nrows && ca < npieces, okb = rb < nrows && cb < npieces;
        // straight from global memory into LDS: item i of the tile lands at byte 16 i of img, i.e. at 16 bytes per
        // lane from the wave's base, which is where this load instruction puts a lane's piece
        typedef __attribute__((address_space(1))) const void gvoid;
        typedef __attribute__((address_space(3))) void lvoid;
        uint8_t *wbase = img + (threadIdx.x >> 6) * (64 * 16);
        if (oka)
            __builtin_amdgcn_global_load_lds((gvoid *)(s0 + (uint32_t)(__umul24(ra, pitch) + 16 * ca)), (lvoid *)wbase, 16, 0, 0);
        if (okb)
            __builtin_amdgcn_global_load_lds((gvoid *)(s0 + (uint32_t)(__umul24(rb, pitch) + 16 * cb)),
                                             (lvoid *)(wbase + 256 * 16), 16, 0, 0);
        zero_scores();
    } else {
        // a level at an odd address or pitch (a caller's frame): word by word, the 18 words of a row that matter
        zero_scores();
        if (walker) {
            const int x = tx0 - 4 + 4 * g;
            for (int r = r0; r < IMG_H; r += 14) {
                const int y = ty0 - 4 + r;
                uint32_t v = 0;
                if (y >= 0 && y < h && x >= 0 && x < pitch)
                    v = *reinterpret_cast<const uint32_t *>(src + (uint32_t)(__umul24(y, pitch) + x));
                img32[r * IMG_WORDS + GRP_0 + g] = v;
            }
        }
    }
    __syncthreads();
    FAST_CUT(1);

    // ---- compass test, four pixels per thread, on staged rows 3..68 (tile rows -1..64) ----
    const int tm = a.t_low;
    const uint32_t qt4 = (uint32_t)min((tm + 1) >> 1, 127) * 0x01010101u;
    {
        // bit 7 of byte o of a pass word stands for pixel o of the group; the groups left and right
        // of the tile hold one needed pixel each
        const int xg = tx0 - 4 + 4 * g;
        uint32_t gmask = !walker ? 0u : g == 0 ? 0x80000000u : g == GRP_W - 1 ? 0x00000080u : 0x80808080u;
        if (tx0 - 1 < x_lo || tx0 + ORBX_FAST_W + 1 > x_hi) {  // (uniform) a tile at the left or right border
#pragma unroll
            for (int o = 0; o < 4; o++)
                if (xg + o < x_lo || xg + o >= x_hi)
                    gmask &= ~(0x80u << (8 * o));
        }
        // Score rows 0 .. lim-1 (staged rows 3 ..) hold detectable pixels; above the first of them lies an
        // undetectable one only in the tiles of the top row, whose ring row it is.  Trip k takes score row
        // r0 + 14 k: one comparison per trip, and every address is a constant away from the first trip's.
        static_assert(ORBX_FAST_Y0 == ORBX_CELL_ORG, "the tile grid starts at the first detectable row");
        const int lim = min(SC_H, y_hi - (ty0 - 1));
        const uint32_t gmask0 = (ty0 - 1 < y_lo && r0 == 0) ? 0u : gmask;
        const uint32_t *wrow0 = img32 + (r0 + 3) * IMG_WORDS + GRP_0 + g;
        // Pass bits of the five trips collect in one word: trip k's bits (7, 15, 23, 31) move down by k.
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint32_t gm = k == 0 ? gmask0 : gmask;
            if (gm != 0u && r0 < lim - 14 * k) {
                const uint32_t *wrow = wrow0 + 14 * k * IMG_WORDS;
                const uint32_t vw = wrow[0], upw = wrow[-3 * IMG_WORDS], dnw = wrow[3 * IMG_WORDS];
                const uint32_t lw = wrow[-1], rw = wrow[1];
                const uint32_t r4w = __builtin_amdgcn_alignbyte(rw, vw, 3);   // pixels x+3
                const uint32_t r12w = __builtin_amdgcn_alignbyte(vw, lw, 1);  // pixels x-3
                acc |= (compass4(vw, dnw, upw, r4w, r12w, qt4) & gm) >> k;
            }
        }
        // queue slots: a wave-wide prefix sum of the per-lane counts, one LDS atomic per wave
        const int cnt = __popc(acc);
        const int incl = orbx_wave_scan_add_i32(cnt);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total) {
            int base = 0;
            if (lane == 0)
                base = atomicAdd(&qn, total);
            int slot = __builtin_amdgcn_readfirstlane(base) + incl - cnt;
            const uint32_t e0 = (uint32_t)((r0 + 3) << 7) | (uint32_t)(4 * (GRP_0 + g));
            while (acc) {
                const int p = __builtin_ctz(acc);   // bit 8o + 7 - k: pixel o of trip k
                const int k = 7 - (p & 7);
                queue[slot++] = (uint16_t)(e0 + (uint32_t)(k * (14 << 7)) + (uint32_t)(p >> 3));
                acc &= acc - 1;
            }
        }
    }
    __syncthreads();
    FAST_CUT(2);

    // ---- full score for the queued pixels ----
    // About 0.6 of the queued pixels are corners, and only corners inside the tile proper take part in the
    // suppression pass: they are compacted (ballot, one LDS atomic per wave) so that it runs on full waves.
    const int nq = qn;
    for (int q0 = 0; q0 < nq; q0 += 256) {
        const int qi = q0 + threadIdx.x;
        bool corner = false;
        int e = 0;
        if (qi < nq) {
            e = queue[qi];
            const int r = e >> 7, c = e & 127;  // staged row / column
            const int A = fast_arc_score(&img[r * IMG_W + c]);
            if (A > tm) {
                sc[SC_AT(r, c)] = (uint8_t)A;
                // ring pixels belong to a neighbouring tile
                corner = r >= 4 && r < 4 + ORBX_FAST_H && c >= IMG_X0 && c < IMG_X0 + ORBX_FAST_W;
            }
        }
        const unsigned long long bal = __ballot(corner);
        if (bal) {
            int base = 0;
            if (lane == 0)
                base = atomicAdd(&cn, __popcll(bal));
            const int at = QUEUE_CAP - 1 - (__builtin_amdgcn_readfirstlane(base) + lanes_below(bal));
            if (corner && at >= nq)  // never onto a queue entry: those may not have been read yet
                queue[at] = (uint16_t)e;
        }
    }
    __syncthreads();
    FAST_CUT(3);

    // ---- non-maximum suppression inside the pixel's cell, on the tile proper ----
    // Survivors are listed at the bottom of the queue array (its entries are spent; the corner list sits at the
    // top, and survivors + corners stay below the array's size because survivors <= corners <= half of it).
    // Then the workgroup reserves its share of the level's candidate arrays with ONE atomic on the level's
    // counter (an atomic per wave and trip made the kernel 2.5 times slower: a level's counter took a thousand of
    // them per launch) and writes the records.  The arrays are filled in no particular order; every candidate
    // carries its order key, which is all the quadtree needs.
    const uint32_t wc_magic = a.wc_magic[lvl], hc_magic = a.hc_magic[lvl];
    // All eight neighbours are read (the score map has a one-pixel ring); those of another cell are masked out
    // by the pixel's position inside its cell, without branches.
    const uint32_t wcu = (uint32_t)wc, hcu = (uint32_t)hc;
    auto is_maximum = [&](int e) -> bool {
        const int r = e >> 7, c = e & 127;
        const uint8_t *p = &sc[SC_AT(r, c)];
        // x - 19 < 4096 and cells are 30 .. 60 pixels: the 20-bit reciprocal divides exactly
        const uint32_t ux = (uint32_t)(tx0 - IMG_X0 - ORBX_CELL_ORG + c), uy = (uint32_t)(ty0 - 4 - ORBX_CELL_ORG + r);
        const uint32_t xm = ux - __umul24(__umul24(ux, wc_magic) >> 20, wcu);  // column and row inside the cell
        const uint32_t ym = uy - __umul24(__umul24(uy, hc_magic) >> 20, hcu);
        const uint32_t A = p[0], nl = p[-1], nr = p[1];
        const uint32_t ul = p[-SC_P - 1], uc = p[-SC_P], ur = p[-SC_P + 1];
        const uint32_t dl = p[SC_P - 1], dc = p[SC_P], dr = p[SC_P + 1];
        const uint32_t ml = xm != 0u ? ~0u : 0u, mr = xm != wcu - 1u ? ~0u : 0u;
        const uint32_t mu = ym != 0u ? ~0u : 0u, md = ym != hcu - 1u ? ~0u : 0u;
        const uint32_t left = max(max(ul & mu, nl), dl & md) & ml;
        const uint32_t right = max(max(ur & mu, nr), dr & md) & mr;
        const uint32_t mid = max(uc & mu, dc & md);
        return A > max(max(left, right), mid);
    };
    auto list_survivors = [&](bool have, int e) {
        const bool keep = have && is_maximum(e);
        const unsigned long long bal = __ballot(keep);
        if (bal == 0ull)
            return;
        int base = 0;
        if (lane == 0)
            base = atomicAdd(&sn, __popcll(bal));  // LDS
        const int slot = __builtin_amdgcn_readfirstlane(base) + lanes_below(bal);  // (lane 0's value: all lanes active here)
        if (keep)
            queue[slot] = (uint16_t)e;
    };
    const int nc = cn;
    if (nc <= QUEUE_CAP - nq) {
        for (int i0 = 0; i0 < nc; i0 += 256) {
            const int i = i0 + threadIdx.x;
            list_survivors(i < nc, i < nc ? queue[QUEUE_CAP - 1 - i] : 0);
        }
    } else {
        // the corner list ran into the queue: visit every pixel of the tile (at most a quarter of them can be
        // strict maxima, so the list below stays inside the array)
        for (int i = threadIdx.x; i < ORBX_FAST_W * ORBX_FAST_H; i += 256) {
            const int r = 4 + i / ORBX_FAST_W, c = IMG_X0 + i % ORBX_FAST_W;
            list_survivors(sc[SC_AT(r, c)] != 0, (r << 7) | c);
        }
    }
    __syncthreads();
    FAST_CUT(4);
    // the first wave alone writes the records (a tile keeps about 45 candidates): no second barrier, and the
    // other waves retire
    const int ns = sn;
    if (ns == 0 || threadIdx.x >= 64)
        return;
    int first = 0;
    if (lane == 0)
        first = atomicAdd(ncand + f * a.nlevels + lvl, ns);  // the slot of the tile's first record
    first = __builtin_amdgcn_readfirstlane(first);
    uint8_t *flags = cell_flag + (size_t)f * a.cells_per_frame + a.cell_base[lvl];
    const size_t cb = (size_t)f * a.cand_per_frame + a.cand_base[lvl];
    const int cap = a.cand_cap[lvl];
    if (first + ns > cap && lane == 0)
        atomicMax(&status[ORBX_ST_CAND_OVERFLOW], first + ns);
    for (int i = lane; i < ns; i += 64) {
        const int e = queue[i];
        const int r = e >> 7, c = e & 127;
        const int A = sc[SC_AT(r, c)];
        const int y = ty0 - 4 + r, x = tx0 - IMG_X0 + c;
        const int cj = (int)(((uint32_t)(x - ORBX_CELL_ORG) * wc_magic) >> 20);
        const int ci = (int)(((uint32_t)(y - ORBX_CELL_ORG) * hc_magic) >> 20);
        const int cell = ci * a.ncols[lvl] + cj;
        const int k = first + i;
        if (k < cap) {
            // x | y<<16 relative to (16,16), cornerScore, order key of the reference's candidate list
            cand_xy[cb + k] = (uint32_t)(x - ORBX_BORDER) | ((uint32_t)(y - ORBX_BORDER) << 16);
            cand_resp[cb + k] = (uint8_t)(A - 1);
            cand_key[cb + k] = ((uint32_t)cell << 12) | ((uint32_t)(y - (ORBX_CELL_ORG + ci * hc)) << 6) |
                               (uint32_t)(x - (ORBX_CELL_ORG + cj * wc));
        }
        if (A > a.t_ini)
            flags[cell] = 1;  // the cell has a corner above iniThFAST
    }
}

// part 0: clear the status words, the candidate counts and the per-cell flags, 1: detection
hipError_t orbx_launch_fast(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s, int part)
{
    FastArgs a;
    a.nlevels = g.nlevels;
    a.batch = g.batch;
    a.slab = g.slab;
    a.l0 = b.l0;
    a.cells_per_frame = g.cells_per_frame;
    a.cand_per_frame = g.cand_per_frame;
    a.tiles_per_frame = g.ftiles_per_frame;
    // cv::FAST clamps the threshold to [0,255]; at t = 0 a corner still needs score A-1 > 0 to beat
    // its non-corner neighbours, which is A > 1
    a.t_ini = min(max(g.ini_th, 1), 255);
    a.t_low = min(a.t_ini, min(max(g.min_th, 1), 255));
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &L = g.lv[l];
        a.tile_base[l] = L.ftile_base;
        a.w[l] = L.w, a.h[l] = L.h, a.pitch[l] = L.pitch, a.off[l] = L.off;
        a.wcell[l] = L.wcell, a.hcell[l] = L.hcell, a.ncols[l] = L.ncols, a.cell_base[l] = L.cell_base;
        a.cand_base[l] = L.cand_base, a.cand_cap[l] = L.cand_cap;
        a.wc_magic[l] = L.wcell > 0 ? ((1u << 20) + L.wcell - 1) / L.wcell : 0u;
        a.hc_magic[l] = L.hcell > 0 ? ((1u << 20) + L.hcell - 1) / L.hcell : 0u;
    }
    a.tile_base[g.nlevels] = g.ftiles_per_frame;
    if (part == 0) {
        // the counts lie behind the status words and the flags behind the counts in one allocation: one fill
        const uint8_t *lo = reinterpret_cast<const uint8_t *>(b.status);
        return hipMemsetAsync(b.status, 0, (size_t)(b.cell_flag - lo) + (size_t)g.batch * g.cells_per_frame, s);
    }
    const dim3 grid = orbx_frame_grid(g.ftiles_per_frame, g.batch);
    k_fast<<<grid, 256, 0, s>>>(b.pyr, b.ncand, b.cand_xy, b.cand_resp, b.cand_key, b.cell_flag, b.status,
                                b.tile_tab + g.tiles_per_frame, a);
    return hipGetLastError();
}
