// Frame::ComputeStereoMatches (src/Frame.cc:534-708 of the reference) for every frame of a batch
// that two extractor handles (left and right image) have just processed: keypoints, descriptors
// and both pyramids are already in device memory.
//
// Every left keypoint is independent up to the final median filter, so one wave takes one left
// keypoint: the right keypoints are bucketed by image row once per frame (k_stereo_rows), the
// wave scans the rows that can hold a candidate (the reference's vRowIndices lists a right
// keypoint in every row of [floor(y - r), ceil(y + r)], r = 2 * scale of its level), keeps the
// smallest  distance<<16 | index  (strict '<' in index order, :611-615), slides the 11x11 window
// over 11 positions at the keypoint's level (:629-660, all integer), fits the parabola (:665-673)
// and writes uRight / depth.  k_stereo_median then drops the matches whose window distance is not
// below 1.5 * 1.4 * median (:694-707).
#include "orbx_internal.h"


struct StereoArgs {
    int nlevels, out_cap, slabL, slabR, band, th_orb;
    OrbxLevel0 l0L, l0R;
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS], pitch[ORBX_MAX_LEVELS], off[ORBX_MAX_LEVELS];
    float sf[ORBX_MAX_LEVELS], inv[ORBX_MAX_LEVELS];
    float mb, mbf;
};

// ---- right keypoints bucketed by floor(y): rowstart[f][0..h0], sorted[f][...] (any order inside a row) ----
__global__ __launch_bounds__(256) void k_stereo_rows(const orbx_keypoint *__restrict__ kpR,
                                                     const uint8_t *__restrict__ dR, const int *__restrict__ nR,
                                                     int *__restrict__ rowstart, uint4 *__restrict__ sorted,
                                                     int out_cap, int h0, StereoArgs a)
{
    extern __shared__ int hist[];  // h0 + 1 counters, then their exclusive prefix
    __shared__ int wsum[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = min(nR[f], out_cap);
    const orbx_keypoint *k = kpR + (size_t)f * out_cap;
    for (int i = tid; i <= h0; i += 256)
        hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int iy = min(max((int)floorf(k[i].y), 0), h0 - 1);
        atomicAdd(&hist[iy], 1);
    }
    __syncthreads();
    // exclusive prefix over the rows: a contiguous run of rows per thread, then a scan of the run sums
    const int per = (h0 + 256) / 256;
    const int lo = min(tid * per, h0 + 1), hi = min(lo + per, h0 + 1);
    int s = 0;
    for (int i = lo; i < hi; i++)
        s += hist[i];
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if ((tid & 63) >= d)
            inc += t;
    }
    if ((tid & 63) == 63)
        wsum[tid >> 6] = inc;
    __syncthreads();
    int base = inc - s;
    for (int i = 0; i < (tid >> 6); i++)
        base += wsum[i];
    int *rs = rowstart + (size_t)f * (h0 + 1);
    for (int i = lo; i < hi; i++) {
        const int c = hist[i];
        hist[i] = base;
        rs[i] = base;
        base += c;
    }
    __syncthreads();
    // In bucket order: what the matcher tests of a right keypoint (x, the rows it is listed in, its level, its
    // index) as one 16-byte record, and its descriptor behind it.  The lanes of a matcher wave then read
    // neighbouring records instead of gathering fields of keypoints scattered over the frame's arrays.
    uint4 *rec = sorted + (size_t)f * out_cap * 3;
    const uint4 *dsrc = reinterpret_cast<const uint4 *>(dR) + (size_t)f * out_cap * 2;
    for (int i = tid; i < n; i += 256) {
        const orbx_keypoint r = k[i];
        const int iy = min(max((int)floorf(r.y), 0), h0 - 1);
        const int p = atomicAdd(&hist[iy], 1);
        const float rad = 2.0f * a.sf[r.octave];
        const int maxr = (int)ceilf(r.y + rad), minr = (int)floorf(r.y - rad);
        rec[p] = make_uint4(__float_as_uint(r.x), ((uint32_t)minr & 0xffffu) | ((uint32_t)maxr << 16),
                            (uint32_t)r.octave | ((uint32_t)i << 16), 0u);
        rec[out_cap + 2 * p] = dsrc[2 * i];
        rec[out_cap + 2 * p + 1] = dsrc[2 * i + 1];
    }
}

// ---- one wave per left keypoint ----
__global__ __launch_bounds__(256) void k_stereo_match(const orbx_keypoint *__restrict__ kpL, const uint8_t *__restrict__ dL,
                                                      const int *__restrict__ nL, const orbx_keypoint *__restrict__ kpR,
                                                      const uint8_t *__restrict__ dR, const int *__restrict__ nR,
                                                      const uint8_t *__restrict__ pyrL, const uint8_t *__restrict__ pyrR,
                                                      const int *__restrict__ rowstart,
                                                      const uint4 *__restrict__ sorted, float *__restrict__ uright,
                                                      float *__restrict__ depth, int *__restrict__ sad, StereoArgs a)
{
    __shared__ int part[4][11][12];  // per wave: window distance of (shift, row)
    __shared__ uint32_t win[4][11][10];  // per wave: rows of the left window (4 words) and the right strip (6 words)
    const int f = blockIdx.y, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + wv;
    if (iL >= a.out_cap)
        return;
    const size_t o = (size_t)f * a.out_cap + iL;
    float out_u = -1.0f, out_d = -1.0f;
    int out_s = -1;
    const int n1 = min(nL[f], a.out_cap);
    if (iL < n1) {
        const orbx_keypoint kl = kpL[o];
        const int levelL = kl.octave, h0 = a.h[0];
        const float vL = kl.y, uL = kl.x;
        const int row = (int)vL;  // vRowIndices[vL]: float index, truncated
        const float minZ = a.mb, minD = 0.0f, maxD = a.mbf / minZ;
        const float minU = uL - maxD, maxU = uL - minD;
        uint32_t best = 0xffffffffu;
        float best_x = 0.f;  // x of this lane's best candidate
        if (row >= 0 && row < h0 && !(maxU < 0)) {
            const int *rs = rowstart + (size_t)f * (h0 + 1);
            const int c_lo = rs[max(row - a.band, 0)], c_hi = rs[min(row + a.band + 1, h0)];
            const uint4 *rec = sorted + (size_t)f * a.out_cap * 3;
            const uint32_t *q = reinterpret_cast<const uint32_t *>(dL) + o * 8;
            uint32_t qd[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                qd[k] = q[k];
            for (int c = c_lo + lane; c < c_hi; c += 64) {
                const uint4 r = rec[c];  // x | rows it is listed in | level, index
                const int minr = (int)(short)(r.y & 0xffffu), maxr = (int)(r.y >> 16);
                const int oct = (int)(r.z & 0xffffu);
                if (row < minr || row > maxr || oct < levelL - 1 || oct > levelL + 1)
                    continue;
                const float rx = __uint_as_float(r.x);
                if (rx >= minU && rx <= maxU) {
                    const uint4 d0 = rec[a.out_cap + 2 * c], d1 = rec[a.out_cap + 2 * c + 1];
                    const uint32_t d = __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) +
                                       __popc(qd[3] ^ d0.w) + __popc(qd[4] ^ d1.x) + __popc(qd[5] ^ d1.y) +
                                       __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
                    const uint32_t key = (d << 16) | (r.z >> 16);
                    if (key < best)
                        best = key, best_x = rx;
                }
            }
        }
        const uint32_t mine = best;
        best = orbx_wave_min_u32(best);
        // bestDist starts at TH_HIGH and must end below (TH_HIGH + TH_LOW) / 2 (:590, :620)
        if (best != 0xffffffffu && (int)(best >> 16) < a.th_orb) {
            // x of the winner from the lane that holds it (keys are distinct: they end in the keypoint's index)
            const unsigned long long owner = __ballot(mine == best);
            const float uR0 = __builtin_bit_cast(
                float, __builtin_amdgcn_readlane(__float_as_int(best_x), (int)__builtin_ctzll(owner)));
            const float scaleFactor = a.inv[levelL];
            const float scaleduL = roundf(kl.x * scaleFactor), scaledvL = roundf(kl.y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (!(iniu < 0 || endu >= (float)a.w[levelL])) {
                const int su = (int)scaleduL, sv = (int)scaledvL, sr = (int)scaleduR0;
                const uint8_t *IL = levelL == 0 ? a.l0L.ptr + (size_t)f * a.l0L.fstride
                                                : pyrL + (size_t)f * a.slabL + a.off[levelL];
                const uint8_t *IR = levelL == 0 ? a.l0R.ptr + (size_t)f * a.l0R.fstride
                                                : pyrR + (size_t)f * a.slabR + a.off[levelL];
                const int pL = levelL == 0 ? a.l0L.pitch : a.pitch[levelL];
                const int pR = levelL == 0 ? a.l0R.pitch : a.pitch[levelL];
                // The 11 x 11 window of the left level and the 11 x 21 strip of the right one, as aligned words
                // (4 and 6 to a row: 110 words, two loads of the wave) through LDS; the 121 window distances
                // then read bytes from there.  (44 byte loads per lane straight from global memory kept the
                // address unit busy for half of the kernel's time.)  The words may reach 3 bytes past the
                // window, which stays inside the row or, at most, in the first bytes of the next one.
                const int xl = (su - w) & ~3, xr = (sr - L - w) & ~3;
                const int oL = (su - w) - xl, oR = (sr - L - w) - xr;
                uint32_t *ws = &win[wv][0][0];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int i = lane + 64 * k;
                    if (i < 11 * 10) {
                        const int r = i / 10, j = i - r * 10;
                        const uint8_t *src = j < 4 ? IL + (size_t)(sv + r - w) * pL + xl + 4 * j
                                                   : IR + (size_t)(sv + r - w) * pR + xr + 4 * (j - 4);
                        ws[i] = *reinterpret_cast<const uint32_t *>(src);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const uint8_t *wb = reinterpret_cast<const uint8_t *>(ws);  // row r: 16 left bytes, 24 right bytes
                const int cL = wb[w * 40 + oL + w];
                // 121 (shift, row) items over the 64 lanes
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int item = lane + 64 * k;
                    if (item < 121) {
                        const int s = item / 11, r = item - s * 11;  // incR = s - 5, window row r - 5
                        const int cR = wb[w * 40 + 16 + oR + s + w];
                        const uint8_t *lp = wb + r * 40 + oL;
                        const uint8_t *rp = wb + r * 40 + 16 + oR + s;
                        int acc = 0;
#pragma unroll
                        for (int x = 0; x < 11; x++)
                            acc += abs(((int)lp[x] - cL) - ((int)rp[x] - cR));
                        part[wv][s][r] = acc;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                // lane s folds the 11 row sums of shift s; the 11 totals then go to every lane
                int tot = 0;
                if (lane < 11) {
#pragma unroll
                    for (int r = 0; r < 11; r++)
                        tot += part[wv][lane][r];
                }
                float vDists[11];
                int bestD = INT_MAX, bestinc = 0;
#pragma unroll
                for (int s = 0; s < 11; s++) {
                    const float dist = (float)__builtin_amdgcn_readlane(tot, s);
                    if (dist < (float)bestD) {
                        bestD = (int)dist;
                        bestinc = s - L;
                    }
                    vDists[s] = dist;
                }
                if (!(bestinc == -L || bestinc == L)) {
                    float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
                    for (int s = 1; s < 10; s++)
                        if (s == L + bestinc)
                            dist1 = vDists[s - 1], dist2 = vDists[s], dist3 = vDists[s + 1];
                    const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
                    if (!(deltaR < -1 || deltaR > 1)) {
                        float bestuR = a.sf[levelL] * ((float)scaleduR0 + (float)bestinc + deltaR);
                        float disparity = uL - bestuR;
                        if (disparity >= minD && disparity < maxD) {
                            if (disparity <= 0) {
                                disparity = 0.01;
                                bestuR = uL - 0.01;
                            }
                            out_d = a.mbf / disparity;
                            out_u = bestuR;
                            out_s = bestD;
                        }
                    }
                }
            }
        }
    }
    if (lane == 0) {
        uright[o] = out_u;
        depth[o] = out_d;
        sad[o] = out_s;
    }
}

// ---- median filter: drops matches whose window distance is >= 1.5 * 1.4 * median ----
__global__ __launch_bounds__(256) void k_stereo_median(const int *__restrict__ nL, const int *__restrict__ sad,
                                                       float *__restrict__ uright, float *__restrict__ depth,
                                                       int *__restrict__ nmatch, int out_cap)
{
    __shared__ int hist[256];
    __shared__ int sel_bin, sel_rank, total;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = min(nL[f], out_cap);
    const int *sd = sad + (size_t)f * out_cap;
    // window distances are below 2^16 (121 * 510): two passes of a 256-bin radix select find the
    // element at index size/2 of the sorted list
    hist[tid] = 0;
    if (tid == 0)
        total = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < n; i += 256)
        if (sd[i] >= 0) {
            atomicAdd(&hist[(sd[i] >> 8) & 255], 1);
            mine++;
        }
    atomicAdd(&total, mine);
    __syncthreads();
    const int size = total;
    if (size == 0) {
        if (tid == 0)
            nmatch[f] = 0;
        return;
    }
    if (tid == 0) {
        int rank = size / 2, b = 0;
        while (rank >= hist[b])
            rank -= hist[b++];
        sel_bin = b, sel_rank = rank;
    }
    __syncthreads();
    const int hb = sel_bin;
    __syncthreads();
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256)
        if (sd[i] >= 0 && ((sd[i] >> 8) & 255) == hb)
            atomicAdd(&hist[sd[i] & 255], 1);
    __syncthreads();
    if (tid == 0) {
        int rank = sel_rank, b = 0;
        while (rank >= hist[b])
            rank -= hist[b++];
        sel_rank = (hb << 8) | b;  // the median
        total = 0;
    }
    __syncthreads();
    const float median = (float)sel_rank;
    const float thDist = 1.5f * 1.4f * median;
    int kept = 0;
    for (int i = tid; i < n; i += 256)
        if (sd[i] >= 0) {
            if ((float)sd[i] < thDist)
                kept++;
            else {
                uright[(size_t)f * out_cap + i] = -1;
                depth[(size_t)f * out_cap + i] = -1;
            }
        }
    atomicAdd(&total, kept);
    __syncthreads();
    if (tid == 0)
        nmatch[f] = total;
}

hipError_t orbx_launch_stereo(const OrbxGeom &g, const OrbxBuffers &bl, const OrbxBuffers &br, const float *sf,
                              const float *inv, float mb, float mbf, int *d_rowstart, uint4 *d_sorted, int *d_sad,
                              float *d_uright, float *d_depth, int *d_nmatch, hipStream_t s)
{
    StereoArgs a;
    a.nlevels = g.nlevels, a.out_cap = g.out_cap, a.slabL = g.slab, a.slabR = g.slab;
    a.l0L = bl.l0, a.l0R = br.l0;
    a.th_orb = (ORBX_TH_HIGH + ORBX_TH_LOW) / 2;
    float rmax = 0.f;
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &L = g.lv[l];
        a.w[l] = L.w, a.h[l] = L.h, a.pitch[l] = L.pitch, a.off[l] = L.off;
        a.sf[l] = sf[l], a.inv[l] = inv[l];
        rmax = fmaxf(rmax, 2.0f * sf[l]);
    }
    a.band = (int)ceilf(rmax) + 1;  // |floor(yR) - row| of any listed right keypoint stays within this
    a.mb = mb, a.mbf = mbf;
    const int h0 = g.lv[0].h;
    k_stereo_rows<<<g.batch, 256, (size_t)(h0 + 1) * sizeof(int), s>>>(br.out_kp, br.out_desc, br.out_n, d_rowstart,
                                                                       d_sorted, g.out_cap, h0, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    k_stereo_match<<<dim3((g.out_cap + 3) / 4, g.batch), 256, 0, s>>>(bl.out_kp, bl.out_desc, bl.out_n, br.out_kp,
                                                                      br.out_desc, br.out_n, bl.pyr, br.pyr, d_rowstart,
                                                                      d_sorted, d_uright, d_depth, d_sad, a);
    if ((e = hipGetLastError()) != hipSuccess)
        return e;
    k_stereo_median<<<g.batch, 256, 0, s>>>(bl.out_n, d_sad, d_uright, d_depth, d_nmatch, g.out_cap);
    return hipGetLastError();
}
