// 7x7 Gaussian blur (GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101), src/ORBextractor.cc:1157-1159 of the
// reference; taps 18 34 48 56 48 34 18 / 256) on the matrix cores.
//
// Both passes of the separable filter are products with a banded tap matrix, exact in integers:
//   horizontal  H' = P' * Bh      P' = pixels - 128 (i8), taps (i8)      H' = H - 32768 fits 16 bits
//   vertical    V  = H'^T * Bv    H' split into its two bytes (high byte signed, low byte - 128), one product each
//   out = (256 * V_hi + V_lo + 8454144) >> 16 = (sum_j k_j H_j + 32768) >> 16
// (8454144 = 32768 rounding + 128 * 256 for the low bytes' offset + 32768 * 256 for H's offset.)
// v_mfma_i32_16x16x64_i8 takes 16 k-bytes per lane: lane l supplies row (A) / column (B) l & 15 and k-slots
// (l >> 4, 0..15); result register reg of lane l is C[row][col] with col = l & 15 (B's lane index) and
// row = reg + 4 * (l >> 4) (A's lane index).  So:
//   * the four 16-byte pieces of 64 staged bytes of an image row are the A operand of a row as they are read from
//     LDS, and 64 inputs cover the 16 + 6 that 16 output columns read;
//   * the horizontal result has a COLUMN on the lane and 4 rows in the registers; the results of three row blocks,
//     packed to bytes, are the A operand of the transposed vertical product (k = 48 staged rows, in that register
//     order), so H never moves between lanes;
//   * the vertical result has an output ROW on the lane and 4 consecutive columns in the registers: they pack into
//     the 4-pixel row of a 4 x 4 group of the tiled blurred level, and the 64 lanes of one product write two whole
//     128-byte tiles.
// A workgroup stages a 128 x 32 tile (+3 pixels around it, borders mirrored) in LDS exactly as the vector-ALU
// kernel does; each of its four waves takes 32 columns as two blocks of 16.  14 matrix instructions per wave and
// tile, no second LDS array, one barrier per tile.  The three tap operands are constants.
#include <vector>

#include "blur_stage.h"
#include "orbx_internal.h"

typedef int bm_v4i __attribute__((ext_vector_type(4)));

#define BM_CONST 8454144
#define BM_TPW 1  // tiles per workgroup (2 or 4 with the next tile's pieces fetched ahead: 2.5 % fewer frames/s in a batch)

struct BlurMArgs {
    int nlevels, slab, bslab, batch, tiles_per_frame;
    OrbxLevel0 l0;
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS], pitch[ORBX_MAX_LEVELS], off[ORBX_MAX_LEVELS], boff[ORBX_MAX_LEVELS];
};

// operand table: [0 horizontal, 1 vertical for output rows 0..15, 2 vertical for output rows 16..31][lane][16]
#define BM_BAND_BYTES 1024

__global__ __launch_bounds__(256) void k_blur_mfma(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur,
                                                   const uint32_t *__restrict__ tile_tab,
                                                   const uint8_t *__restrict__ bands, BlurMArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t img[2][BL_SH * BL_SWORDS];  // tiles in turn: one barrier per tile
    const int f = ORBX_BLOCK_FRAME;
    if (f >= a.batch)
        return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    const uint8_t *lb = bands + lane * 16;
    const bm_v4i bh = *reinterpret_cast<const bm_v4i *>(lb);
    const bm_v4i bv0 = *reinterpret_cast<const bm_v4i *>(lb + BM_BAND_BYTES);
    const bm_v4i bv1 = *reinterpret_cast<const bm_v4i *>(lb + 2 * BM_BAND_BYTES);
    const bm_v4i zero4 = {0, 0, 0, 0};
    const bm_v4i cinit = {BM_CONST >> 8, BM_CONST >> 8, BM_CONST >> 8, BM_CONST >> 8};  // joins the high-byte product
    // staged byte offset of the lane's piece in row block 0 of column block 0: row n, columns 16 q .. of the 64
    // that start 16 left of the wave's first column
    const int o0 = n * (4 * BL_SWORDS) + BL_LEFT + 32 * wv - 16 + 16 * q;
    const int o2 = min(32 + n, BL_SH - 1) * (4 * BL_SWORDS) + BL_LEFT + 32 * wv - 16 + 16 * q;  // row block 2: 6 rows exist

    const int unit0 = ORBX_BLOCK_UNIT * BM_TPW;
    const int nunits = min(BM_TPW, a.tiles_per_frame - unit0);
    uint8_t *dst_frame = blur + (size_t)f * a.bslab;
    auto tile_of = [&](int unit, int &lvl) -> BlurTile {
        const uint32_t te = tile_tab[unit];
        lvl = te >> 24;
        const int tx0 = (int)(te & 0xfffu) * BL_W, ty0 = (int)((te >> 12) & 0xfffu) * BL_H;
        const int spitch = lvl == 0 ? a.l0.pitch : a.pitch[lvl];
        const uint8_t *src = lvl == 0 ? a.l0.ptr + (size_t)f * a.l0.fstride : pyr + (size_t)f * a.slab + a.off[lvl];
        return BlurTile{src, spitch, a.w[lvl], a.h[lvl], tx0, ty0, lvl == 0};
    };
    // the next tile's pieces are fetched into registers while the current one is filtered
    int lvl_next = 0;
    BlurTile next = tile_of(unit0, lvl_next);
    BlurFetch fx;
    blur_stage_fetch(next, fx);
    for (int k = 0; k < nunits; k++) {
        const BlurTile cur = next;
        const int lvl = lvl_next;
        const int tx0 = cur.tx0, ty0 = cur.ty0, h = cur.h, pitch = a.pitch[lvl];
        uint8_t *dst = dst_frame + a.boff[lvl];
        uint32_t *stage = img[k & 1];
        blur_stage_commit<true>(stage, cur, fx);  // staged as pixel - 128
        __syncthreads();
        if (k + 1 < nunits) {
            next = tile_of(unit0 + k + 1, lvl_next);
            blur_stage_fetch(next, fx);
        }
        const uint8_t *sb = reinterpret_cast<const uint8_t *>(stage);
        auto piece = [&](int o) -> bm_v4i { return *reinterpret_cast<const bm_v4i *>(sb + o); };
        // lane = output row (n of 16) and 4-pixel group (q of 4): first byte of its quad in column block 0, row block 0
        const int y = ty0 + n, x = tx0 + 32 * wv + 4 * q;
        uint8_t *d0 = dst + (uint32_t)ORBX_BLUR_ADDR(x, y, pitch);
        const int h8 = (h + 7) & ~7;  // rows of the level's last tile row are written: zeros below the level
        const bool cols_ok = x < pitch;
        const bool full = ty0 + BL_H <= h;  // every row of the tile exists (all tiles but a level's last row of them)
#pragma unroll
        for (int cb = 0; cb < 2; cb++) {
            // horizontal pass of 16 columns: staged rows 0..15, 16..31, 32..37
            bm_v4i c[3];
            c[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(piece(o0 + 16 * cb), bh, zero4, 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(piece(o0 + 16 * cb + 16 * (4 * BL_SWORDS)), bh, zero4, 0, 0, 0);
            c[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(piece(o2 + 16 * cb), bh, zero4, 0, 0, 0);
            // the byte planes of H' in the register order: k-slot (q, 4 rb + i) = staged row 16 rb + 4 q + i
            bm_v4i hi = zero4, lo = zero4;
#pragma unroll
            for (int rb = 0; rb < 3; rb++) {
                const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)c[rb][1], (uint32_t)c[rb][0], 0x05010400u);
                const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)c[rb][3], (uint32_t)c[rb][2], 0x05010400u);
                lo[rb] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
                hi[rb] = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302u);
            }
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {
                const bm_v4i vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(hi, ob ? bv1 : bv0, cinit, 0, 0, 0);
                const bm_v4i vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(lo, ob ? bv1 : bv0, zero4, 0, 0, 0);
                uint32_t s[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    s[i] = ((uint32_t)vh[i] << 8) + (uint32_t)vl[i];
                const uint32_t o = __builtin_amdgcn_perm(s[1], s[0], 0x0c0c0602u) |
                                   __builtin_amdgcn_perm(s[3], s[2], 0x06020c0cu);
                const int yy = y + 16 * ob;
                // 16 columns on: the next tile of the blurred level; 16 rows on: two tile rows down
                uint32_t *dq = reinterpret_cast<uint32_t *>(d0 + cb * 128 + ob * (16 * pitch));
                if (full) {
                    if (cols_ok)
                        *dq = o;
                } else if (cols_ok && yy < h8) {
                    *dq = yy < h ? o : 0u;
                }
            }
        }
    }
}

// ---- host side: the three constant tap operands ----

// One 1024-byte operand: lane (n = lane & 15, quarter = lane >> 4), byte e = tap that the input in k-slot
// (quarter, e) contributes to output n.  Output n reads seven consecutive inputs with taps 18 34 48 56 48 34 18.
static void bm_band(uint8_t *out, int which)
{
    static const int taps[7] = {18, 34, 48, 56, 48, 34, 18};
    for (int lane = 0; lane < 64; lane++) {
        const int n = lane & 15, quarter = lane >> 4;
        for (int e = 0; e < 16; e++) {
            int d;  // index of the slot's input among the seven of output n
            if (which == 0)
                d = 16 * quarter + e - (16 + n - 3);  // slot = staged column, 16 left of output 0
            else
                d = 16 * (e >> 2) + 4 * quarter + (e & 3) - (16 * (which - 1) + n);  // slot = staged row; output row's first input
            out[lane * 16 + e] = (uint8_t)((d >= 0 && d <= 6) ? taps[d] : 0);
        }
    }
}

size_t orbx_blur_mfma_band_bytes(void) { return 3 * BM_BAND_BYTES; }

void orbx_blur_mfma_tables(std::vector<uint8_t> &bands)
{
    bands.assign(orbx_blur_mfma_band_bytes(), 0);
    for (int which = 0; which < 3; which++)
        bm_band(&bands[(size_t)which * BM_BAND_BYTES], which);
}

hipError_t orbx_launch_blur_mfma(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    BlurMArgs a;
    a.nlevels = g.nlevels, a.slab = g.slab, a.bslab = g.bslab, a.batch = g.batch;
    a.tiles_per_frame = g.tiles_per_frame;
    a.l0 = b.l0;
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &L = g.lv[l];
        a.w[l] = L.w, a.h[l] = L.h, a.pitch[l] = L.pitch, a.off[l] = L.off, a.boff[l] = L.boff;
    }
    k_blur_mfma<<<orbx_frame_grid((g.tiles_per_frame + BM_TPW - 1) / BM_TPW, g.batch), 256, 0, s>>>(
        b.pyr, b.blur, b.tile_tab, b.blur_bands, a);
    return hipGetLastError();
}
