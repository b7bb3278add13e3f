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
// ((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2.
// Source positions and coefficients are computed in the kernel with the arithmetic of the host-side
// definition (double position, float fraction, half-to-even rounding), so that a thread's loads
// depend on nothing but its index: fx = (float)((dx+0.5)*scale - 0.5), sx = floor(fx),
// a1 = rint((fx-sx)*2048), a0 = rint((1-(fx-sx))*2048), clamped at the image edges.
struct ResizeCoef {
    int s;        // first source index
    int c0, c1;   // weights of s and s+1 (c1 == 0: s+1 is not read)
};

__device__ __forceinline__ ResizeCoef resize_coef_x(int dx, double scale, int sw)
{
    float fx = (float)(((double)dx + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    bool unweighted = false;
    if (sx < 0) {
        fx = 0.f;
        sx = 0;
    }
    if (sx + 1 >= sw) {
        unweighted = true;  // this and every later column copy the source pixel (weight 2048)
        if (sx >= sw - 1) {
            fx = 0.f;
            sx = sw - 1;
        }
    }
    ResizeCoef c;
    c.s = sx;
    c.c0 = unweighted ? 2048 : (int)rintf((1.f - fx) * 2048.f);
    c.c1 = unweighted ? 0 : (int)rintf(fx * 2048.f);
    return c;
}

__device__ __forceinline__ void resize_coef_y(int dy, double scale, int sh, int &sy0, int &sy1, int &b0, int &b1)
{
    float fy = (float)(((double)dy + 0.5) * scale - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    sy0 = min(max(sy, 0), sh - 1);
    sy1 = min(max(sy + 1, 0), sh - 1);
    b0 = (int)rintf((1.f - fy) * 2048.f);
    b1 = (int)rintf(fy * 2048.f);
}

#define RS_ROWS 2  // destination rows per thread

// Each thread produces 4 adjacent pixels of RS_ROWS destination rows.  When the source bytes of the
// four pixels fit 12 bytes from an aligned word (any scale factor up to 2), each source row costs
// three word loads; all loads of a thread are independent.
__global__ __launch_bounds__(256) void k_resize(const uint8_t *__restrict__ src_lvl, long long src_fstride,
                                                uint8_t *__restrict__ pyr_out, int slab, int spitch, int sw, int sh,
                                                int doff, int dpitch, int dw, int dh, double scale_x, double scale_y)
{
    const int f = blockIdx.z;
    const int dy0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * RS_ROWS;
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (dy0 >= dh || dx0 >= dw)
        return;
    ResizeCoef cx[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        cx[i] = resize_coef_x(min(dx0 + i, dw - 1), scale_x, sw);
    const uint8_t *sbase = src_lvl + (size_t)f * src_fstride;
    uint8_t *dbase = pyr_out + (size_t)f * slab + doff;
    const int wb = cx[0].s & ~3;                       // aligned byte offset of the first word
    const bool wide = cx[3].s + 1 - wb < 12;           // all needed bytes inside three words
    int sy0[RS_ROWS], sy1[RS_ROWS], b0[RS_ROWS], b1[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++)
        resize_coef_y(min(dy0 + r, dh - 1), scale_y, sh, sy0[r], sy1[r], b0[r], b1[r]);
    int h[RS_ROWS][2][4];
    if (wide) {
        uint32_t wds[RS_ROWS][2][3];
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint8_t *row = sbase + (size_t)(k ? sy1[r] : sy0[r]) * spitch + wb;
#pragma unroll
                for (int j = 0; j < 3; j++)
                    wds[r][k][j] = wb + 4 * j < spitch ? *reinterpret_cast<const uint32_t *>(row + 4 * j) : 0u;
            }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int off = cx[i].s - wb;                 // 0..10
            const bool up = off >= 4;
            const uint32_t o2 = (uint32_t)(up ? off - 4 : off);   // 0..6 inside the chosen word pair
            const uint32_t sel = 0x0c0c0000u | ((o2 + 1) << 8) | o2;  // bytes o2, o2+1 -> low two bytes
#pragma unroll
            for (int r = 0; r < RS_ROWS; r++)
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t lo = up ? wds[r][k][1] : wds[r][k][0];
                    const uint32_t hi = up ? wds[r][k][2] : wds[r][k][1];
                    const uint32_t pq = __builtin_amdgcn_perm(hi, lo, sel);
                    h[r][k][i] = (int)(pq & 0xff) * cx[i].c0 + (int)(pq >> 8) * cx[i].c1;
                }
        }
    } else {
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint8_t *row = sbase + (size_t)(k ? sy1[r] : sy0[r]) * spitch;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int s1 = cx[i].c1 ? cx[i].s + 1 : cx[i].s;
                    h[r][k][i] = row[cx[i].s] * cx[i].c0 + row[s1] * cx[i].c1;
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
            const int v = (((b0[r] * (h[r][0][i] >> 4)) >> 16) + ((b1[r] * (h[r][1][i] >> 4)) >> 16) + 2) >> 2;
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
    // scale as cv::resize derives it: 1 / (dsize / ssize), in double
    const double scale_x = 1. / ((double)D.w / S.w), scale_y = 1. / ((double)D.h / S.h);
    dim3 grid((D.w + 255) / 256, (D.h + 4 * RS_ROWS - 1) / (4 * RS_ROWS), g.batch);
    const uint8_t *src = level == 1 ? b.l0.ptr : b.pyr + S.off;
    const long long sfs = level == 1 ? b.l0.fstride : (long long)g.slab;
    const int spitch = level == 1 ? b.l0.pitch : S.pitch;
    k_resize<<<grid, 256, 0, s>>>(src, sfs, b.pyr, g.slab, spitch, S.w, S.h, D.off, D.pitch, D.w, D.h, scale_x, scale_y);
    return hipGetLastError();
}
