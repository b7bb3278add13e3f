// 7x7 Gaussian blur of every level, then per keypoint: intensity-centroid orientation on the
// unblurred level, 256-bit steered BRIEF on the blurred level, scale-back and level-major output
// (GaussianBlur call src/ORBextractor.cc:1157-1159; IC_Angle :93-124; computeOrbDescriptor :127-169;
// assembly :1147-1181 of the reference).
#include "orbx_internal.h"

// ---------------------------------------------------------------- blur

struct BlurArgs {
    int nlevels, slab, variant;
    int tile_base[ORBX_MAX_LEVELS + 1];
    int tiles_x[ORBX_MAX_LEVELS];
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS], pitch[ORBX_MAX_LEVELS], off[ORBX_MAX_LEVELS];
};

__device__ __forceinline__ int reflect101(int i, int n)
{
    // BORDER_REFLECT_101: ... 2 1 | 0 1 2 ... n-2 n-1 | n-2 n-3 ...
    if (n == 1)
        return 0;
    while (i < 0 || i >= n)
        i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

#define BL_W 64
#define BL_H 16
#define BL_SW (BL_W + 8)  // staged columns tx0-4 .. tx0+67 (word aligned; 3 needed each side)
#define BL_SH (BL_H + 6)

// Separable fixed-point filter: horizontal sums fit 16 bits (kernel sum <= 257), the vertical pass
// rounds with +32768 and shifts by 16; variant 1 (kernel sum 257) saturates to 255.
__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, BlurArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[BL_SH * BL_SW];
    __shared__ uint16_t hs[BL_SH * BL_W];
    const int f = blockIdx.y;
    int lvl = 0;
    while (lvl + 1 < a.nlevels && (int)blockIdx.x >= a.tile_base[lvl + 1])
        lvl++;
    const int t = blockIdx.x - a.tile_base[lvl];
    const int tx0 = (t % a.tiles_x[lvl]) * BL_W, ty0 = (t / a.tiles_x[lvl]) * BL_H;
    const int w = a.w[lvl], h = a.h[lvl], pitch = a.pitch[lvl];
    const uint8_t *src = pyr + (size_t)f * a.slab + a.off[lvl];
    uint8_t *dst = blur + (size_t)f * a.slab + a.off[lvl];
    const int k0 = 18, k1 = 34, k2 = a.variant ? 49 : 48, k3 = a.variant ? 55 : 56;

    const bool interior = tx0 >= 4 && tx0 + BL_W + 4 <= w && ty0 >= 3 && ty0 + BL_H + 3 <= h;
    if (interior) {
        for (int i = threadIdx.x; i < BL_SH * (BL_SW / 4); i += 256) {
            const int r = i / (BL_SW / 4), c4 = i % (BL_SW / 4);
            *reinterpret_cast<uint32_t *>(&img[r * BL_SW + c4 * 4]) =
                *reinterpret_cast<const uint32_t *>(src + (size_t)(ty0 - 3 + r) * pitch + tx0 - 4 + c4 * 4);
        }
    } else {
        for (int i = threadIdx.x; i < BL_SH * BL_SW; i += 256) {
            const int r = i / BL_SW, c = i % BL_SW;
            const int y = reflect101(ty0 - 3 + r, h), x = reflect101(tx0 - 4 + c, w);
            img[i] = src[(size_t)y * pitch + x];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BL_SH * BL_W; i += 256) {
        const int r = i / BL_W, c = i % BL_W;
        const uint8_t *p = &img[r * BL_SW + c + 1];  // p[0..6] = columns x-3 .. x+3
        hs[i] = (uint16_t)(k0 * (p[0] + p[6]) + k1 * (p[1] + p[5]) + k2 * (p[2] + p[4]) + k3 * p[3]);
    }
    __syncthreads();
    const int r = threadIdx.x >> 4, c0 = (threadIdx.x & 15) * 4;
    const int y = ty0 + r;
    if (y < h && tx0 + c0 < pitch) {
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint16_t *q = &hs[r * BL_W + c0 + i];
            const uint32_t acc = k0 * ((uint32_t)q[0] + q[6 * BL_W]) + k1 * ((uint32_t)q[BL_W] + q[5 * BL_W]) +
                                 k2 * ((uint32_t)q[2 * BL_W] + q[4 * BL_W]) + k3 * (uint32_t)q[3 * BL_W];
            uint32_t v = (acc + 32768u) >> 16;
            v = v > 255u ? 255u : v;
            out |= v << (8 * i);
        }
        *reinterpret_cast<uint32_t *>(dst + (size_t)y * pitch + tx0 + c0) = out;
    }
}

hipError_t orbx_launch_blur(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    BlurArgs a;
    a.nlevels = g.nlevels, a.slab = g.slab, a.variant = g.blur_variant;
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &L = g.lv[l];
        a.tile_base[l] = L.tile_base, a.tiles_x[l] = L.tiles_x;
        a.w[l] = L.w, a.h[l] = L.h, a.pitch[l] = L.pitch, a.off[l] = L.off;
    }
    a.tile_base[g.nlevels] = g.tiles_per_frame;
    k_blur<<<dim3(g.tiles_per_frame, g.batch), 256, 0, s>>>(b.pyr, b.blur, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------- orientation + descriptor

// 512 sample points of the rBRIEF test table, (x,y) int8 pairs, point-major
__constant__ __attribute__((aligned(16))) int8_t c_pattern[1024] = {
#include "brief_pattern.inc"
};
// half-width of the circular patch at each |row| (umax, src/ORBextractor.cc:483-505)
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

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
    int nlevels, slab, kp_per_frame, out_cap;
    int pitch[ORBX_MAX_LEVELS], off[ORBX_MAX_LEVELS], kp_base[ORBX_MAX_LEVELS], patch_size[ORBX_MAX_LEVELS];
    float scale[ORBX_MAX_LEVELS];
};

// one wave per keypoint slot of a frame
__global__ __launch_bounds__(256) void k_describe(const uint8_t *__restrict__ pyr, const uint8_t *__restrict__ blur,
                                                  const uint32_t *__restrict__ kept_xy,
                                                  const uint8_t *__restrict__ kept_resp, const int *__restrict__ nkept,
                                                  orbx_keypoint *__restrict__ out_kp, uint8_t *__restrict__ out_desc,
                                                  int *__restrict__ out_n, DescArgs a)
{
    const int f = blockIdx.y;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // level-major concatenation: find the level of this slot
    int lvl = 0, first = 0, total = 0;
    {
        int acc = 0;
        bool found = false;
        for (int l = 0; l < a.nlevels; l++) {
            const int c = nkept[f * a.nlevels + l];
            if (!found && slot < acc + c) {
                lvl = l, first = acc, found = true;
            }
            acc += c;
        }
        total = acc;
        if (blockIdx.x == 0 && threadIdx.x == 0)
            out_n[f] = total;
        if (!found)
            return;
    }
    const int i = slot - first;
    const uint32_t xy = kept_xy[(size_t)f * a.kp_per_frame + a.kp_base[lvl] + i];
    const int x = xy & 0xffff, y = xy >> 16;
    const int pitch = a.pitch[lvl];
    const uint8_t *img = pyr + (size_t)f * a.slab + a.off[lvl] + (size_t)y * pitch + x;
    const uint8_t *bl = blur + (size_t)f * a.slab + a.off[lvl] + (size_t)y * pitch + x;

    // ---- IC_Angle: lanes 0..30 take columns u=-15..15 of rows v = 0,2,4.., lanes 32..62 of rows 1,3,5.. ----
    int m10 = 0, m01 = 0;
    {
        const int u = (lane & 31) - 15;
        const int half = lane >> 5;
        if ((lane & 31) < 31) {
            for (int vv = half; vv <= 15; vv += 2) {
                if (abs(u) <= c_umax[vv]) {
                    if (vv == 0)
                        m10 += u * img[u];
                    else {
                        const int vp = img[u + vv * pitch], vm = img[u - vv * pitch];
                        m10 += u * (vp + vm);
                        m01 += vv * (vp - vm);
                    }
                }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            m10 += __shfl_xor(m10, d, 64);
            m01 += __shfl_xor(m01, d, 64);
        }
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- steered BRIEF: lane L evaluates tests L, L+64, L+128, L+192; a ballot yields 64 descriptor bits ----
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float rad = __fmul_rn(angle, factorPI);
    const float ca = (float)cos((double)rad), sa = (float)sin((double)rad);
    unsigned long long bits[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int tst = lane + 64 * j;
        const char4 p = *reinterpret_cast<const char4 *>(&c_pattern[tst * 4]);
        const float x0 = (float)p.x, y0 = (float)p.y, x1 = (float)p.z, y1 = (float)p.w;
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, sa), __fmul_rn(y0, ca)));
        const int q0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, ca), __fmul_rn(y0, sa)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, sa), __fmul_rn(y1, ca)));
        const int q1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, ca), __fmul_rn(y1, sa)));
        const int t0 = bl[r0 * pitch + q0], t1 = bl[r1 * pitch + q1];
        bits[j] = __ballot(t0 < t1);
    }
    const size_t o = (size_t)f * a.out_cap + slot;
    if (lane < 4)
        *reinterpret_cast<unsigned long long *>(out_desc + o * 32 + lane * 8) =
            lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
    if (lane == 0) {
        orbx_keypoint kp;
        kp.x = (float)x, kp.y = (float)y;
        if (lvl != 0) {
            kp.x = __fmul_rn(kp.x, a.scale[lvl]);
            kp.y = __fmul_rn(kp.y, a.scale[lvl]);
        }
        kp.size = (float)a.patch_size[lvl];
        kp.angle = angle;
        kp.response = (float)kept_resp[(size_t)f * a.kp_per_frame + a.kp_base[lvl] + i];
        kp.octave = lvl;
        kp.class_id = -1;
        out_kp[o] = kp;
    }
}

hipError_t orbx_launch_describe(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    DescArgs a;
    a.nlevels = g.nlevels, a.slab = g.slab, a.kp_per_frame = g.kp_per_frame, a.out_cap = g.out_cap;
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &L = g.lv[l];
        a.pitch[l] = L.pitch, a.off[l] = L.off, a.kp_base[l] = L.kp_base, a.patch_size[l] = L.patch_size;
        a.scale[l] = L.scale;
    }
    dim3 grid((g.out_cap + 3) / 4, g.batch);
    k_describe<<<grid, 256, 0, s>>>(b.pyr, b.blur, b.kept_xy, b.kept_resp, b.nkept, b.out_kp, b.out_desc, b.out_n, a);
    return hipGetLastError();
}
