// Scale pyramid: level 0 ingest and the chained bilinear down-scaling
// (ORBextractor::ComputePyramid, src/ORBextractor.cc:1191-1221 of the reference).
#include "orbx_internal.h"

// ---- level 0: copy the caller's frames (arbitrary stride) into the slab layout ----
__global__ __launch_bounds__(256) void k_ingest(const uint8_t *__restrict__ src, int stride, int64_t fstride,
                                                uint8_t *__restrict__ pyr, int slab, int off, int w, int h, int pitch)
{
    const int f = blockIdx.z;
    const int y = blockIdx.y;
    const uint8_t *s = src + (size_t)f * fstride + (size_t)y * stride;
    uint8_t *d = pyr + (size_t)f * slab + off + (size_t)y * pitch;
    // destination rows are 64-byte aligned; the source may not be, so read bytes, write words
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x4 >= w)
        return;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = x4 + i;
        const uint32_t px = x < w ? s[x] : 0u;
        v |= px << (8 * i);
    }
    *reinterpret_cast<uint32_t *>(d + x4) = v;
}

hipError_t orbx_launch_ingest(const uint8_t *src, int stride, int64_t fstride, const OrbxGeom &g, const OrbxBuffers &b,
                              hipStream_t s)
{
    const OrbxLevel &L = g.lv[0];
    dim3 grid((L.w + 1023) / 1024, L.h, g.batch);
    k_ingest<<<grid, 256, 0, s>>>(src, stride, fstride, b.pyr, g.slab, L.off, L.w, L.h, L.pitch);
    return hipGetLastError();
}

// ---- level l from level l-1: cv::resize INTER_LINEAR, 8-bit, 11-bit fixed-point coefficients ----
// Horizontal pass in int32 (S[sx]*a0 + S[sx+1]*a1), vertical pass
// ((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2.  Source positions and coefficients come from
// tables built on the host for the current image size (they depend on the column or row only).
#ifndef RS_ROWS
// Destination rows per wave of the batch kernel.  With 8 the kernel alone is fastest (0.181 ms per 256 frames against
// 0.197 with 16: twice the waves), but a wave filters the one or two source rows above its strip again, and in a
// batch, where the kernels of other instances fill the machine anyway, the work saved counts: 16 gives 1.2 % more
// frames/s than 8; 12, 20, 24 and 32 give less than 16.
#define RS_ROWS 16
#endif
#define RS2_ROWS 8  // the two-level kernel: four waves compute the 32 rows of a tile of the middle level

typedef unsigned short rs_us2 __attribute__((ext_vector_type(2)));

// A wave produces ROWS destination rows of 256 pixels, each lane 4 adjacent pixels of every row.  The
// rows of a wave are uniform, so the wave walks down the source rows its destination rows read and
// keeps the horizontal pass of the row before: at a scale factor of 1.2 the upper source row of a
// destination row is the lower one of the row before in four cases of five, so a source row is read
// and filtered once, not twice.  The row bookkeeping is scalar.
// When the source bytes of a lane's four pixels lie within 8 bytes of the first one (any scale
// factor up to 2), a source row costs one 12-byte load from the aligned word below the first pixel,
// two byte alignments that bring the first pixel to byte 0, and per pixel one byte permute (the two
// neighbours as a 2 x u16 vector) and one 2 x u16 dot product with the packed coefficient pair.  Lanes
// whose 12 bytes would end behind `src_bytes` (the readable bytes of the source) take the byte-wise path.
//
// resize_rows: destination rows dy0 .. dy1-1 (wave-uniform) of the lane's columns dx0 .. dx0+3.  `sbase`
// points at source pixel (0, 0) (global memory or LDS); `put(dy, word)` receives the four result bytes of a row.
// LdsSrc: the source is a staged tile in LDS (every 12-byte read that starts at a lane's own first word stays
// inside the array by construction of its margins; `src_bytes` is not a limit there).
template <int ROWS, bool LdsSrc = false, typename Put>
__device__ __forceinline__ void resize_rows(const uint8_t *sbase, int spitch, uint32_t src_bytes, int dx0, int dy0,
                                            int dy1, const OrbxXCoef *__restrict__ xt,
                                            const OrbxYCoef *__restrict__ yt, Put put)
{
    // the column table is padded to a multiple of 4 columns with copies of the last entry
    const uint4 t0 = *reinterpret_cast<const uint4 *>(xt + dx0), t1 = *reinterpret_cast<const uint4 *>(xt + dx0 + 2);
    const int sx[4] = {(int)t0.x, (int)t0.z, (int)t1.x, (int)t1.z};
    const uint32_t coef[4] = {t0.y, t0.w, t1.y, t1.w};  // a0 | a1 << 16
    const uint32_t wb = (uint32_t)sx[0] & ~3u;  // aligned byte offset of the first word
    const uint32_t s0 = (uint32_t)sx[0] & 3u;
    const uint32_t last_row = (uint32_t)yt[dy1 - 1].sy1;  // source rows ascend with the destination row
    const bool narrow = sx[3] + 1 - sx[0] <= 7 && __umul24(last_row, (uint32_t)spitch) + wb + 12u <= src_bytes;
    uint32_t sel[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        sel[i] = (uint32_t)(sx[i] - sx[0]) * 0x00010001u + 0x0c010c00u;  // bytes d, 0, d+1, 0

    // horizontal pass of one source row, (S[sx] * a0 + S[sx + 1] * a1) >> 4 for the lane's four pixels (the shift
    // is the vertical pass's, applied once per source row here), from the three words `w` loaded ahead (narrow
    // lanes) or from single bytes
    // other lanes load words they ignore: at the row's start in global memory (always readable), at their own
    // first word in LDS (the row's start may lie before the array there)
    const uint32_t wb_safe = (narrow || LdsSrc) ? wb : 0u;
    auto load_row = [&](int srow, uint32_t (&w)[3]) {
        const uint32_t *row =
            reinterpret_cast<const uint32_t *>(sbase + __umul24((uint32_t)srow, (uint32_t)spitch) + wb_safe);
        w[0] = row[0], w[1] = row[1], w[2] = row[2];  // rows and pitch are below 2^13
    };
    auto hpass = [&](int srow, const uint32_t (&w)[3], uint32_t (&o)[4]) {
        if (narrow) {
            const uint32_t lo = __builtin_amdgcn_alignbyte(w[1], w[0], s0);  // source bytes sx0 .. sx0+3
            const uint32_t hi = __builtin_amdgcn_alignbyte(w[2], w[1], s0);  // source bytes sx0+4 .. sx0+7
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t pq = __builtin_amdgcn_perm(hi, lo, sel[i]);
                o[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_us2, pq), __builtin_bit_cast(rs_us2, coef[i]), 0u,
                                              false) >> 4;
            }
        } else {
            const uint8_t *row = sbase + __umul24((uint32_t)srow, (uint32_t)spitch);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // a1 == 0 marks columns whose right neighbour must not be read
                const uint32_t a0 = coef[i] & 0xffffu, a1 = coef[i] >> 16;
                const int s1 = a1 ? sx[i] + 1 : sx[i];
                o[i] = (row[sx[i]] * a0 + row[s1] * a1) >> 4;
            }
        }
    };

    // The wave's row coefficients, all at once (its rows are uniform: scalar loads), so that the loop below
    // holds no load of them and no wait.  Entries past dy1 repeat the last row's.
    OrbxYCoef C[ROWS];
#pragma unroll
    for (int d = 0; d < ROWS; d++)
        C[d] = yt[__builtin_amdgcn_readfirstlane(min(dy0 + d, dy1 - 1))];
    // Destination rows in turn.  A row reads source rows (sy0, sy1) = (sy1 - 1, sy1), or (sy1, sy1) where the table
    // clamps at the border, and sy1 grows by one from row to row mostly, by two or more where the ratio skips source
    // rows.  The horizontal passes of the two rows sit in H[0] and H[1]; before destination row d the newest one
    // (source row j) is in H[d & 1].  One step: the row keeps it as its upper row and filters sy1 into the other
    // array; more: it filters both its rows anew; none (clamped rows at the bottom): the arrays change places.  So
    // no register is copied on the common paths.  The raw words of a row's source rows are loaded while the row
    // before is computed (R[d & 1]: [0] for sy0, [1] for sy1).
    uint32_t H[2][4], R[2][2][3];
    int j = C[0].sy1 - 2;  // row 0 filters both its source rows
    load_row(C[0].sy0, R[0][0]);
    load_row(C[0].sy1, R[0][1]);
#pragma unroll
    for (int d = 0; d < ROWS; d++) {
        if (dy0 + d < dy1) {
            const OrbxYCoef c = C[d];
            const int q = d & 1;
            if (d + 1 < ROWS && dy0 + d + 1 < dy1) {
                const int step = C[d + 1].sy1 - c.sy1;
                if (step >= 2)
                    load_row(C[d + 1].sy0, R[q ^ 1][0]);
                if (step >= 1)
                    load_row(C[d + 1].sy1, R[q ^ 1][1]);
            }
            if (c.sy1 - j >= 2) {
                hpass(c.sy0, R[q][0], H[q]);
                hpass(c.sy1, R[q][1], H[q ^ 1]);
            } else if (c.sy1 - j == 1) {
                hpass(c.sy1, R[q][1], H[q ^ 1]);
                if (c.sy0 == c.sy1) {  // clamped: both rows are sy1
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        H[q][i] = H[q ^ 1][i];
                }
            } else {
                // the same lower row as the row before: H[q] holds it, H[q ^ 1] the row above it
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t t = H[q][i];
                    H[q][i] = c.sy0 == c.sy1 ? t : H[q ^ 1][i];
                    H[q ^ 1][i] = t;
                }
            }
            j = c.sy1;
            const uint32_t(&prev)[4] = H[q], (&cur)[4] = H[q ^ 1];
            // (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 for four pixels: the upper halves
            // of the eight products are gathered two pixels to a register and added and shifted as packed 16-bit
            // values (both products stay below 2^27, so 24-bit multiplies are exact; the rounding 2 enters the
            // upper half of the first product with the multiplication)
            uint32_t p0[4], p1[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                p0[i] = __umul24((uint32_t)c.b0, prev[i]) + 0x20000u;
                p1[i] = __umul24((uint32_t)c.b1, cur[i]);
            }
            // The halves' sums stay below 2^12, so one 32-bit addition adds both without a carry between them, and
            // of the 32-bit shift only the low byte of each half is kept (the two bits that enter the lower half
            // from the upper one land in its bits 14 and 15): plain additions and shifts are issued at nearly
            // twice the rate of the packed 16-bit forms.
            auto pair = [&](int i) -> uint32_t {
                const uint32_t t = __builtin_amdgcn_perm(p0[i + 1], p0[i], 0x07060302u);  // hi16(p0[i]) | hi16(p0[i+1]) << 16
                const uint32_t u = __builtin_amdgcn_perm(p1[i + 1], p1[i], 0x07060302u);
                return (t + u) >> 2;
            };
            put(dy0 + d, __builtin_amdgcn_perm(pair(2), pair(0), 0x06040200u));  // the four low bytes
        }
    }
}

template <int ROWS>
__global__ __launch_bounds__(256) void k_resize(const uint8_t *__restrict__ src_lvl, long long src_fstride,
                                                uint32_t src_bytes, uint8_t *__restrict__ pyr_out, int slab, int spitch,
                                                int doff, int dpitch, int dw, int dh,
                                                const OrbxXCoef *__restrict__ xt, const OrbxYCoef *__restrict__ yt, int batch,
                                                int fshift)
{
    // blockIdx.x = column block << fshift | frame inside its group of 1 << fshift frames: workgroups are handed to
    // the XCDs in turn, so with groups of 8 every workgroup of a frame runs on the same XCD and the source lines that
    // neighbouring column blocks share (a block's 307 source bytes touch three or four 128-byte lines) are fetched
    // into one L2 once instead of into two
    const int f = (int)(blockIdx.z << fshift) + (int)(blockIdx.x & ((1u << fshift) - 1u));
    const int dy0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * 4 + (threadIdx.x >> 6)) * ROWS);
    const int dx0 = ((blockIdx.x >> fshift) * 64 + (threadIdx.x & 63)) * 4;
    if (dy0 >= dh || dx0 >= dw || f >= batch)
        return;
    uint8_t *dbase = pyr_out + (size_t)f * slab + doff;
    resize_rows<ROWS>(src_lvl + (size_t)f * src_fstride, spitch, src_bytes, dx0, dy0, min(dy0 + ROWS, dh), xt, yt,
                         [&](int dy, uint32_t out) {
                             *reinterpret_cast<uint32_t *>(dbase + (uint32_t)(__umul24(dy, dpitch) + dx0)) = out;
                         });
}

// ---- two levels in one launch (calls of a few frames, whose time is mostly launches) ----
// A workgroup computes 256 x 32 pixels of the middle level (from global memory, as above), writes them out and
// keeps them in LDS; it owns 252 x 31 of them (the last four columns and the last row are computed again by the
// neighbours that own them, so that every pixel of the upper level has both its source columns and both its
// source rows inside the tile that holds the left / upper one).  Then it computes from LDS the pixels of the
// upper level whose left / upper source pixel it owns: columns pcol[tx] .. pcol[tx + 1] - 1, rows
// prow[ty] .. prow[ty + 1] - 1 (tables from the host).  Lanes take aligned groups of four columns as before; a
// group that straddles the tile's first or last column stores its own bytes only.  Four launches per pyramid
// instead of seven; for batches the plain kernel is faster (this one is bound by its instruction count and
// computes 5 % of the middle level twice).
#define RS2_PAD 16     // readable bytes left of the staged columns (groups of the upper level that start before the tile)
#define RS2_PITCH 304  // 16 + 256 + 32 (what a group that ends behind the tile may touch at a level ratio up to 2.5)
__global__ __launch_bounds__(256) void k_resize_pair(const uint8_t *__restrict__ src_lvl, long long src_fstride,
                                                     uint32_t src_bytes, uint8_t *__restrict__ pyr_out, int slab,
                                                     int spitch, int moff, int mpitch, int mw, int mh,
                                                     const OrbxXCoef *__restrict__ mxt, const OrbxYCoef *__restrict__ myt,
                                                     int toff, int tpitch, const OrbxXCoef *__restrict__ txt,
                                                     const OrbxYCoef *__restrict__ tyt, const int *__restrict__ pcol,
                                                     const int *__restrict__ prow)
{
    __shared__ __attribute__((aligned(16))) uint8_t mid[32 * RS2_PITCH];
    const int f = blockIdx.z, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int X1 = blockIdx.x * ORBX_RS2_TW, Y1 = blockIdx.y * ORBX_RS2_TH;
    {
        const int dx0 = X1 + 4 * lane;
        const int dy0 = __builtin_amdgcn_readfirstlane(Y1 + RS2_ROWS * wv);
        if (dy0 < mh && dx0 < mw) {
            uint8_t *dbase = pyr_out + (size_t)f * slab + moff;
            uint8_t *lbase = mid + RS2_PAD + 4 * lane;
            resize_rows<RS2_ROWS>(src_lvl + (size_t)f * src_fstride, spitch, src_bytes, dx0, dy0, min(dy0 + RS2_ROWS, mh), mxt, myt,
                        [&](int dy, uint32_t out) {
                            *reinterpret_cast<uint32_t *>(dbase + (uint32_t)(__umul24(dy, mpitch) + dx0)) = out;
                            *reinterpret_cast<uint32_t *>(lbase + __umul24(dy - Y1, RS2_PITCH)) = out;
                        });
        }
    }
    __syncthreads();
    const int x2a = pcol[blockIdx.x], x2b = pcol[blockIdx.x + 1];
    const int y2a = prow[blockIdx.y], y2b = prow[blockIdx.y + 1];
    const int gx0 = (x2a & ~3) + 4 * lane;
    const int ry0 = __builtin_amdgcn_readfirstlane(y2a + RS2_ROWS * wv);
    if (ry0 < y2b && gx0 < x2b) {
        uint8_t *tbase = pyr_out + (size_t)f * slab + toff;
        const bool whole = gx0 >= x2a && gx0 + 4 <= x2b;
        // LDS address of middle-level pixel (0, 0): pixel (X, Y) lies at mid[(Y - Y1) * RS2_PITCH + RS2_PAD + X - X1]
        const uint8_t *origin = mid + RS2_PAD - X1 - Y1 * RS2_PITCH;
        resize_rows<RS2_ROWS, true>(origin, RS2_PITCH, 0xffffffffu, gx0, ry0, min(ry0 + RS2_ROWS, y2b), txt, tyt, [&](int dy, uint32_t out) {
            uint8_t *d = tbase + (uint32_t)(__umul24(dy, tpitch) + gx0);
            if (whole) {
                *reinterpret_cast<uint32_t *>(d) = out;
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (gx0 + i >= x2a && gx0 + i < x2b)
                        d[i] = (uint8_t)(out >> (8 * i));
            }
        });
    }
}

hipError_t orbx_launch_resize_pair(int level, const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    const OrbxLevel &S = g.lv[level - 1];
    const OrbxLevel &M = g.lv[level];
    const OrbxLevel &T = g.lv[level + 1];
    dim3 grid((M.w + ORBX_RS2_TW - 1) / ORBX_RS2_TW, (M.h + ORBX_RS2_TH - 1) / ORBX_RS2_TH, g.batch);
    const uint8_t *src = level == 1 ? b.l0.ptr : b.pyr + S.off;
    const long long sfs = level == 1 ? b.l0.fstride : (long long)g.slab;
    const int spitch = level == 1 ? b.l0.pitch : S.pitch;
    const uint32_t src_bytes = (uint32_t)spitch * (uint32_t)S.h + (level == 1 ? 0u : 16u);
    k_resize_pair<<<grid, 256, 0, s>>>(src, sfs, src_bytes, b.pyr, g.slab, spitch, M.off, M.pitch, M.w, M.h,
                                       b.xtab + M.xtab_off, b.ytab + M.ytab_off, T.off, T.pitch, b.xtab + T.xtab_off,
                                       b.ytab + T.ytab_off, b.ptab + T.pcol_off, b.ptab + T.prow_off);
    return hipGetLastError();
}

hipError_t orbx_launch_resize(int level, const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    const OrbxLevel &S = g.lv[level - 1];
    const OrbxLevel &D = g.lv[level];
    // a few frames do not fill the machine: the shorter strips give twice the waves
    const int rows = g.batch >= 32 ? RS_ROWS : RS2_ROWS;
    const int fshift = g.batch >= 8 ? 3 : 0;
    dim3 grid(((D.w + 255) / 256) << fshift, (D.h + 4 * rows - 1) / (4 * rows), (g.batch + (1 << fshift) - 1) >> fshift);
    const uint8_t *src = level == 1 ? b.l0.ptr : b.pyr + S.off;
    const long long sfs = level == 1 ? b.l0.fstride : (long long)g.slab;
    const int spitch = level == 1 ? b.l0.pitch : S.pitch;
    // a frame's own rows at level 0 (the caller's buffer ends there); lower levels sit inside the slab,
    // which has slack behind its last level
    const uint32_t src_bytes = (uint32_t)spitch * (uint32_t)S.h + (level == 1 ? 0u : 16u);
    if (rows == RS_ROWS)
        k_resize<RS_ROWS><<<grid, 256, 0, s>>>(src, sfs, src_bytes, b.pyr, g.slab, spitch, D.off, D.pitch, D.w, D.h,
                                               b.xtab + D.xtab_off, b.ytab + D.ytab_off, g.batch, fshift);
    else
        k_resize<RS2_ROWS><<<grid, 256, 0, s>>>(src, sfs, src_bytes, b.pyr, g.slab, spitch, D.off, D.pitch, D.w, D.h,
                                                b.xtab + D.xtab_off, b.ytab + D.ytab_off, g.batch, fshift);
    return hipGetLastError();
}
