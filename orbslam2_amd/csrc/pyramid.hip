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
#define RS_ROWS 2  // destination rows per thread

// Each thread produces 4 adjacent pixels of RS_ROWS destination rows.  When the source bytes of the
// four pixels fit 12 bytes from an aligned word (any scale factor up to 2), each source row costs
// three word loads and the pixels are picked with byte permutes.
__global__ __launch_bounds__(256) void k_resize(const uint8_t *__restrict__ src_lvl, long long src_fstride,
                                                uint8_t *__restrict__ pyr_out, int slab, int spitch, int doff,
                                                int dpitch, int dw, int dh, const OrbxXCoef *__restrict__ xt,
                                                const OrbxYCoef *__restrict__ yt)
{
    const int f = blockIdx.z;
    const int dy0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * RS_ROWS;
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (dy0 >= dh || dx0 >= dw)
        return;
    // the tables are padded to a multiple of 4 columns / RS_ROWS rows with copies of the last entry
    OrbxXCoef cx[4];
    {
        const uint4 t0 = *reinterpret_cast<const uint4 *>(xt + dx0), t1 = *reinterpret_cast<const uint4 *>(xt + dx0 + 2);
        cx[0] = OrbxXCoef{(int)t0.x, (short)(t0.y & 0xffff), (short)(t0.y >> 16)};
        cx[1] = OrbxXCoef{(int)t0.z, (short)(t0.w & 0xffff), (short)(t0.w >> 16)};
        cx[2] = OrbxXCoef{(int)t1.x, (short)(t1.y & 0xffff), (short)(t1.y >> 16)};
        cx[3] = OrbxXCoef{(int)t1.z, (short)(t1.w & 0xffff), (short)(t1.w >> 16)};
    }
    OrbxYCoef cy[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        const uint4 t = *reinterpret_cast<const uint4 *>(yt + dy0 + r);
        cy[r] = OrbxYCoef{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
    }
    const uint8_t *sbase = src_lvl + (size_t)f * src_fstride;
    uint8_t *dbase = pyr_out + (size_t)f * slab + doff;
    const int wb = cx[0].sx & ~3;               // aligned byte offset of the first word
    const bool wide = cx[3].sx + 1 - wb < 12;   // all needed bytes inside three words
    int h[RS_ROWS][2][4];
    if (wide) {
        uint32_t wds[RS_ROWS][2][3];
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint8_t *row = sbase + (size_t)(k ? cy[r].sy1 : cy[r].sy0) * spitch + wb;
#pragma unroll
                for (int j = 0; j < 3; j++)
                    wds[r][k][j] = wb + 4 * j < spitch ? *reinterpret_cast<const uint32_t *>(row + 4 * j) : 0u;
            }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int off = cx[i].sx - wb;  // 0..10
            const bool up = off >= 4;
            const uint32_t o2 = (uint32_t)(up ? off - 4 : off);       // 0..6 inside the chosen word pair
            const uint32_t sel = 0x0c0c0000u | ((o2 + 1) << 8) | o2;  // bytes o2, o2+1 -> low two bytes
            const int c0 = cx[i].a0, c1 = cx[i].a1;
#pragma unroll
            for (int r = 0; r < RS_ROWS; r++)
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t lo = up ? wds[r][k][1] : wds[r][k][0];
                    const uint32_t hi = up ? wds[r][k][2] : wds[r][k][1];
                    const uint32_t pq = __builtin_amdgcn_perm(hi, lo, sel);
                    h[r][k][i] = (int)(pq & 0xff) * c0 + (int)(pq >> 8) * c1;
                }
        }
    } else {
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint8_t *row = sbase + (size_t)(k ? cy[r].sy1 : cy[r].sy0) * spitch;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    // a1 == 0 marks columns whose right neighbour must not be read
                    const int s1 = cx[i].a1 ? cx[i].sx + 1 : cx[i].sx;
                    h[r][k][i] = row[cx[i].sx] * cx[i].a0 + row[s1] * cx[i].a1;
                }
            }
    }
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        if (dy0 + r >= dh)
            break;
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int v = (((cy[r].b0 * (h[r][0][i] >> 4)) >> 16) + ((cy[r].b1 * (h[r][1][i] >> 4)) >> 16) + 2) >> 2;
            if (dx0 + i < dw)
                out |= (uint32_t)(v & 0xff) << (8 * i);
        }
        *reinterpret_cast<uint32_t *>(dbase + (size_t)(dy0 + r) * dpitch + dx0) = out;
    }
}

hipError_t orbx_launch_resize(int level, const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    const OrbxLevel &S = g.lv[level - 1];
    const OrbxLevel &D = g.lv[level];
    dim3 grid((D.w + 255) / 256, (D.h + 4 * RS_ROWS - 1) / (4 * RS_ROWS), g.batch);
    const uint8_t *src = level == 1 ? b.l0.ptr : b.pyr + S.off;
    const long long sfs = level == 1 ? b.l0.fstride : (long long)g.slab;
    const int spitch = level == 1 ? b.l0.pitch : S.pitch;
    k_resize<<<grid, 256, 0, s>>>(src, sfs, b.pyr, g.slab, spitch, D.off, D.pitch, D.w, D.h, b.xtab + D.xtab_off,
                                  b.ytab + D.ytab_off);
    return hipGetLastError();
}
