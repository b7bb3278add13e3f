// Staging of one 128 x 32 blur tile (+3 pixels around it, BORDER_REFLECT_101 at the level's borders) in LDS,
// shared by the vector-ALU blur kernel and the matrix-core one.
#pragma once

#include "orbx_internal.h"

__device__ __forceinline__ int reflect101(int i, int n)
{
    // BORDER_REFLECT_101: ... 2 1 | 0 1 2 ... n-2 n-1 | n-2 n-3 ...
    if (n == 1)
        return 0;
    while (i < 0 || i >= n)
        i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

#define BL_W ORBX_FT_W          // 128 output columns per tile
#define BL_H ORBX_FT_H          // 32 output rows per tile
#define BL_LEFT 16              // staged columns start at tx0-16 (16-byte aligned); 3 are needed
#define BL_SWORDS 40            // staged words per row: columns tx0-16 .. tx0+143
#define BL_SH (BL_H + 6)        // staged rows ty0-3 .. ty0+34 (38 rows = 19 row pairs)
#define BL_PAIRS (BL_SH / 2)
#define BL_GROUPS (BL_W / 4)    // 4-pixel groups per row


// One tile's source: level rows / pitch / size and the tile's first pixel.
struct BlurTile {
    const uint8_t *src;
    int spitch, w, h, tx0, ty0;
    bool level0;  // the level is a caller's frame, which may end with the last row's pixels
    __device__ __forceinline__ int npairs() const { return min(BL_PAIRS, (h - ty0 + 7) >> 1); }
    __device__ __forceinline__ bool aligned() const { return (spitch & 15) == 0 && ((uintptr_t)src & 15) == 0; }
    // every needed pixel (3 around the tile) exists and whole 16-byte pieces can be read
    __device__ __forceinline__ bool interior() const
    {
        return aligned() && tx0 >= BL_LEFT && tx0 + BL_W + 3 <= w && tx0 + BL_W + 16 <= spitch && ty0 >= 3 &&
               ty0 + BL_H + 3 <= h;
    }
};
// What a thread has fetched of a tile and not yet written to LDS: up to two 16-byte pieces and one border byte,
// each with its place in the staged tile (negative: nothing to store).
struct BlurFetch {
    uint4 v[2];
    int dst[2];  // word index of the piece in the staged tile
    uint32_t patch;
    int pdst;    // byte index of the border byte
};

// Staging in two steps, so that a kernel that loops over tiles can have the next tile's loads in flight while it
// works on the current one.  img: BL_SH rows of BL_SWORDS words, staged row r = level row ty0 - 3 + r (mirrored
// outside the level), staged byte column c = level column tx0 - BL_LEFT + c.  Rows behind 2 * npairs() are not
// written.  256 threads.
//
// Whole 16-byte pieces wherever the level is 16-byte aligned.  Tiles that are not interior mirror the rows
// outside the image (reflect 101: a staged row is a source row, the same for the whole row), skip the pieces left
// of column 0 or behind the row pitch, and afterwards patch the three columns left of the image and the three
// right of it byte by byte.  Nearly half of the tiles of a pyramid touch a border, so they must not cost more
// than this.  A tile at the bottom of a level holds h - ty0 < 32 rows: only the row pairs those rows read are staged.
// blur_stage_fetch issues the loads (aligned levels; an unaligned level is read in blur_stage_commit).
// Direct: whole pieces go straight from global memory into `img` (a load that writes LDS puts a lane's 16 bytes
// at 16 bytes per lane from a wave-uniform base, and item i of a tile belongs at byte 16 i of the staged tile);
// only for kernels that stage one tile per workgroup, where `img` is free when the fetch is issued.
template <bool Direct = false>
__device__ __forceinline__ void blur_stage_fetch(const BlurTile &t, BlurFetch &fx, uint32_t *img = nullptr)
{
    if (!t.aligned())
        return;
    const bool interior = t.interior();
    const int nitems = 2 * t.npairs() * (BL_SWORDS / 4);  // 38 rows x 10 pieces = 380 items at most
    // readable bytes of the level
    const uint32_t src_bytes = (uint32_t)t.spitch * (uint32_t)(t.h - 1) + (uint32_t)(t.level0 ? t.w : t.spitch);
    // the thread's items i = threadIdx.x and i + 256: row i / 10, piece i % 10; 256 = 25 * 10 + 6
    const int r0 = (int)(__umul24(threadIdx.x, 6554u) >> 16), c0 = (int)threadIdx.x - 10 * r0;
    const bool wrap = c0 >= 4;
    const int rr[2] = {r0, r0 + (wrap ? 26 : 25)}, cc[2] = {c0, wrap ? c0 - 4 : c0 + 6};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int r = rr[k], c = cc[k];
        const int x0 = t.tx0 - BL_LEFT + 16 * c;
        fx.dst[k] = -1;
        if ((int)threadIdx.x + 256 * k < nitems && (interior || (x0 >= 0 && x0 + 16 <= t.spitch))) {
            fx.dst[k] = r * BL_SWORDS + 4 * c;
            const int sy = interior ? t.ty0 - 3 + r : reflect101(t.ty0 - 3 + r, t.h);
            const uint32_t o = (uint32_t)(__umul24(sy, t.spitch) + x0);
            if (interior || o + 16u <= src_bytes) {
                if (Direct) {
                    static_assert(BL_SWORDS * 4 % 16 == 0, "staged rows are whole 16-byte pieces");
                    typedef __attribute__((address_space(1))) const void gvoid;
                    typedef __attribute__((address_space(3))) void lvoid;
                    uint8_t *wbase = reinterpret_cast<uint8_t *>(img) + ((threadIdx.x >> 6) * 64 + 256 * k) * 16;
                    __builtin_amdgcn_global_load_lds((gvoid *)(t.src + o), (lvoid *)wbase, 16, 0, 0);
                    fx.dst[k] = -1;  // nothing left to commit
                } else {
                    fx.v[k] = *reinterpret_cast<const uint4 *>(t.src + o);
                }
            } else {
                // the last pieces of the last row: pixels only
                uint32_t q[4] = {0, 0, 0, 0};
                for (int b = 0; b < 16; b++)
                    if (x0 + b < t.w)
                        q[b >> 2] |= (uint32_t)t.src[o + b] << (8 * (b & 3));
                fx.v[k] = make_uint4(q[0], q[1], q[2], q[3]);
            }
        }
    }
    fx.pdst = -1;
    if (!interior && threadIdx.x < BL_SH * 6) {
        const int r = (int)(__umul24(threadIdx.x, 10923u) >> 16), k = (int)threadIdx.x - 6 * r;
        const int x = k < 3 ? -1 - k : t.w + k - 3;
        const int col = x - (t.tx0 - BL_LEFT);  // staged byte column
        if (col >= 0 && col < 4 * BL_SWORDS) {
            fx.pdst = r * (4 * BL_SWORDS) + col;
            fx.patch = t.src[(uint32_t)__umul24(reflect101(t.ty0 - 3 + r, t.h), t.spitch) + reflect101(x, t.w)];
        }
    }
}

// Writes the fetched pieces to LDS (with a barrier of its own before the border bytes of a tile that is not
// interior); the caller synchronises the workgroup afterwards.  Flip: every staged byte is stored as
// pixel ^ 0x80, i.e. pixel - 128 as a signed byte (the matrix-core kernel's operand form).
template <bool Flip = false>
__device__ __forceinline__ void blur_stage_commit(uint32_t *img, const BlurTile &t, const BlurFetch &fx)
{
    const uint32_t flip = Flip ? 0x80808080u : 0u;
    const int tx0 = t.tx0, ty0 = t.ty0, w = t.w, h = t.h, spitch = t.spitch;
    const uint8_t *src = t.src;
    if (t.aligned()) {
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (fx.dst[k] >= 0)
                *reinterpret_cast<uint4 *>(&img[fx.dst[k]]) =
                    make_uint4(fx.v[k].x ^ flip, fx.v[k].y ^ flip, fx.v[k].z ^ flip, fx.v[k].w ^ flip);
        if (!t.interior()) {
            __syncthreads();
            if (fx.pdst >= 0)
                reinterpret_cast<uint8_t *>(img)[fx.pdst] = (uint8_t)(fx.patch ^ (flip & 0x80u));
        }
    } else {
        // unaligned level (a caller's level 0 at an odd address or pitch): rows and columns outside the image
        // mirror about the border pixel; words that lie inside the row are still loaded whole.  Only words 3 .. 36 of a staged row
        // (columns tx0-4 .. tx0+131) are needed.  A thread keeps one word column, so the column's
        // reflected byte offsets are computed once, and walks down the rows.
        const int c4 = threadIdx.x & 63;
        if (c4 < 34) {
            const int x0 = tx0 - 4 + c4 * 4;
            const bool whole = x0 >= 0 && x0 + 3 < w;
            const bool unused = x0 >= w + 3;  // right of every needed column
            const int o0 = whole ? x0 : reflect101(x0, w), o1 = reflect101(x0 + 1, w);
            const int o2 = reflect101(x0 + 2, w), o3 = reflect101(x0 + 3, w);
            // all loads of the thread's rows are issued before the first store
            uint32_t v[(BL_SH + 3) / 4];
#pragma unroll
            for (int k = 0; k < (BL_SH + 3) / 4; k++) {
                const int r = (threadIdx.x >> 6) + 4 * k;
                v[k] = 0;
                if (r < BL_SH && !unused) {
                    const uint8_t *row = src + (uint32_t)__umul24(reflect101(ty0 - 3 + r, h), spitch);
                    if (whole)
                        v[k] = *reinterpret_cast<const uint32_t *>(row + o0);
                    else
                        v[k] = (uint32_t)row[o0] | ((uint32_t)row[o1] << 8) | ((uint32_t)row[o2] << 16) |
                               ((uint32_t)row[o3] << 24);
                }
            }
#pragma unroll
            for (int k = 0; k < (BL_SH + 3) / 4; k++) {
                const int r = (threadIdx.x >> 6) + 4 * k;
                if (r < BL_SH)
                    img[r * BL_SWORDS + 3 + c4] = v[k] ^ flip;
            }
        }
    }
}

// both steps at once; returns the number of staged row pairs
__device__ __forceinline__ int blur_stage_tile(uint32_t *img, const uint8_t *__restrict__ src, int spitch, int w, int h,
                                               int tx0, int ty0, bool level0)
{
    const BlurTile t{src, spitch, w, h, tx0, ty0, level0};
    BlurFetch fx;
    blur_stage_fetch(t, fx);
    blur_stage_commit(img, t, fx);
    return t.npairs();
}
