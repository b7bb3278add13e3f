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
// ((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2.  The coefficient tables are built on the host.
__global__ __launch_bounds__(256) void k_resize(const uint8_t *__restrict__ pyr_in, uint8_t *__restrict__ pyr_out,
                                                int slab, int soff, int spitch, int doff, int dpitch, int dw, int dh,
                                                const OrbxXCoef *__restrict__ xt, const OrbxYCoef *__restrict__ yt)
{
    const int f = blockIdx.z;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (dy >= dh || dx0 >= dw)
        return;
    const OrbxYCoef yc = yt[dy];
    const uint8_t *r0 = pyr_in + (size_t)f * slab + soff + (size_t)yc.sy0 * spitch;
    const uint8_t *r1 = pyr_in + (size_t)f * slab + soff + (size_t)yc.sy1 * spitch;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = dx0 + i;
        if (dx < dw) {
            const OrbxXCoef xc = xt[dx];
            // a1 == 0 marks columns whose right neighbour must not be read (sx is the last column)
            const int sx1 = xc.a1 ? xc.sx + 1 : xc.sx;
            const int h0 = r0[xc.sx] * xc.a0 + r0[sx1] * xc.a1;
            const int h1 = r1[xc.sx] * xc.a0 + r1[sx1] * xc.a1;
            const int v = (((yc.b0 * (h0 >> 4)) >> 16) + ((yc.b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            out |= (uint32_t)(v & 0xff) << (8 * i);
        }
    }
    *reinterpret_cast<uint32_t *>(pyr_out + (size_t)f * slab + doff + (size_t)dy * dpitch + dx0) = out;
}

hipError_t orbx_launch_resize(int level, const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    const OrbxLevel &S = g.lv[level - 1];
    const OrbxLevel &D = g.lv[level];
    dim3 grid((D.w + 255) / 256, (D.h + 3) / 4, g.batch);
    k_resize<<<grid, 256, 0, s>>>(b.pyr, b.pyr, g.slab, S.off, S.pitch, D.off, D.pitch, D.w, D.h, b.xtab + D.xtab_off,
                                  b.ytab + D.ytab_off);
    return hipGetLastError();
}
