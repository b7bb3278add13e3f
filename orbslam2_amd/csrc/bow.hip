// ORBmatcher::SearchByBoW (src/ORBmatcher.cc:176-311 key frame against frame, :552-686 key frame
// against key frame) for a batch of key frames against one frame (or key frame), on the GPU.
//
// The reference matches only features that share a vocabulary node and walks the nodes in ascending
// id.  A feature belongs to exactly one node, so the order-dependent part (a feature of side 2 that
// is already matched is skipped) never crosses a node: every (key frame, node) entry is independent
// and gets its own wave.  Inside the entry the features of side 1 are taken in list order; the lanes
// share the node's features of side 2, and the two smallest keys  distance<<16 | list position  of
// the wave are bestDist1 / bestDist2 / bestIdx (strict '<' in list order).  The rotation histogram
// only needs counts per bin, so accepted matches add to a per-key-frame histogram with atomics and a
// second kernel applies ComputeThreeMaxima (:1698-1739) and drops the other bins.
#include <algorithm>

#include "orbx_internal.h"


#define BOW_NODE_CAP 4096  // features of side 2 in one vocabulary node that a wave can flag in LDS

struct BowArgs {
    int nb, n2, m2, strict_lt, check_ori, th_low, histo;
    float ratio, factor;
};

__global__ __launch_bounds__(64) void k_bow_match(const uint32_t *__restrict__ d1, const float *__restrict__ ang1,
                                                  const uint8_t *__restrict__ valid1, const int *__restrict__ feat_start,
                                                  const int *__restrict__ node_kf, const int *__restrict__ ids1,
                                                  const int *__restrict__ start1, const int *__restrict__ items1,
                                                  const uint32_t *__restrict__ d2, const float *__restrict__ ang2,
                                                  const uint8_t *__restrict__ valid2, const int *__restrict__ ids2,
                                                  const int *__restrict__ start2, const int *__restrict__ items2,
                                                  int *__restrict__ match12, int *__restrict__ match21,
                                                  int8_t *__restrict__ bin_of, int *__restrict__ hist,
                                                  int *__restrict__ status, BowArgs a)
{
    __shared__ uint8_t taken[BOW_NODE_CAP];
    const int e = blockIdx.x, lane = threadIdx.x;
    const int b = node_kf[e], node = ids1[e];
    // the node in side 2's (ascending) list
    int lo = 0, hi = a.m2;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ids2[mid] < node)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo >= a.m2 || ids2[lo] != node)
        return;
    const int s2 = start2[lo], c2 = start2[lo + 1] - s2;
    if (c2 > BOW_NODE_CAP) {
        if (lane == 0)
            atomicMax(status, c2);
        return;
    }
    for (int c = lane; c < c2; c += 64)
        taken[c] = 0;
    __syncthreads();
    const int f0 = feat_start[b];
    int *m21 = match21 + (size_t)b * a.n2;
    for (int i1 = start1[e]; i1 < start1[e + 1]; i1++) {
        const int idx1 = __builtin_amdgcn_readfirstlane(items1[i1]);
        const int f1 = f0 + idx1;
        if (valid1 && !valid1[f1])
            continue;
        uint32_t q[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            q[k] = d1[(size_t)f1 * 8 + k];  // the same address in every lane: scalar loads
        uint32_t b1 = 0xffffffffu, b2 = 0xffffffffu;
        for (int c = lane; c < c2; c += 64) {
            const int idx2 = items2[s2 + c];
            if (taken[c] || (valid2 && !valid2[idx2]))
                continue;
            const uint32_t *r = d2 + (size_t)idx2 * 8;
            uint32_t d = 0;
#pragma unroll
            for (int k = 0; k < 8; k++)
                d += __popc(q[k] ^ r[k]);
            if (d < 256u) {  // bestDist1 and bestDist2 start at 256 with strict '<'
                const uint32_t key = (d << 16) | (uint32_t)c;
                const uint32_t h = max(b1, key);
                b1 = min(b1, key);
                b2 = min(b2, h);
            }
        }
        const uint32_t w1 = orbx_wave_min_u32(b1);
        if (w1 == 0xffffffffu)
            continue;
        const uint32_t w2 = orbx_wave_min_u32(b1 == w1 ? b2 : b1);
        const int best = (int)(w1 >> 16), best2 = w2 == 0xffffffffu ? 256 : (int)(w2 >> 16);
        if ((a.strict_lt ? best < a.th_low : best <= a.th_low) && (float)best < a.ratio * (float)best2) {
            const int c = (int)(w1 & 0xffffu);
            if (lane == 0) {
                const int idx2 = items2[s2 + c];
                taken[c] = 1;
                match12[f1] = idx2;
                m21[idx2] = idx1;
                if (a.check_ori) {
                    float rot = ang1[f1] - ang2[idx2];
                    if (rot < 0.0f)
                        rot += 360.0f;
                    int bin = (int)roundf(rot * a.factor);
                    if (bin == a.histo)
                        bin = 0;
                    bin_of[f1] = (int8_t)bin;
                    atomicAdd(&hist[b * 32 + (bin & 31)], 1);
                }
            }
            __syncthreads();  // the flag is read by every lane from here on
        }
    }
}

// per key frame: the three largest rotation bins survive
__global__ __launch_bounds__(256) void k_bow_finish(const int *__restrict__ feat_start, const int8_t *__restrict__ bin_of,
                                                    const int *__restrict__ hist, int *__restrict__ match12,
                                                    int *__restrict__ match21, int *__restrict__ nmatches, BowArgs a)
{
    __shared__ int total;
    const int b = blockIdx.x, tid = threadIdx.x;
    int i1m = -1, i2m = -1, i3m = -1;
    if (a.check_ori) {
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < a.histo; i++) {
            const int s = hist[b * 32 + i];
            if (s > max1) {
                max3 = max2, max2 = max1, max1 = s;
                i3m = i2m, i2m = i1m, i1m = i;
            } else if (s > max2) {
                max3 = max2, max2 = s;
                i3m = i2m, i2m = i;
            } else if (s > max3) {
                max3 = s;
                i3m = i;
            }
        }
        if ((float)max2 < 0.1f * (float)max1)
            i2m = -1, i3m = -1;
        else if ((float)max3 < 0.1f * (float)max1)
            i3m = -1;
    }
    if (tid == 0)
        total = 0;
    __syncthreads();
    int kept = 0;
    int *m21 = match21 + (size_t)b * a.n2;
    for (int f = feat_start[b] + tid; f < feat_start[b + 1]; f += 256) {
        const int idx2 = match12[f];
        if (idx2 < 0)
            continue;
        const int bn = bin_of[f];
        if (a.check_ori && !(bn == i1m || bn == i2m || bn == i3m)) {
            match12[f] = -1;
            m21[idx2] = -1;
        } else
            kept++;
    }
    atomicAdd(&total, kept);
    __syncthreads();
    if (tid == 0)
        nmatches[b] = total;
}

hipError_t orbx_launch_bow(const uint8_t *d_desc1, const float *d_ang1, const uint8_t *d_valid1, const int *d_feat_start,
                           int nb, int total_feat1, const int *d_node_kf, const int *d_ids1, const int *d_start1,
                           const int *d_items1, int total_nodes1, const uint8_t *d_desc2, const float *d_ang2,
                           const uint8_t *d_valid2, int n2, const int *d_ids2, const int *d_start2, const int *d_items2,
                           int m2, int strict_lt, float ratio, int check_ori, int *d_match12, int *d_match21,
                           int *d_nmatches, int8_t *d_bin_of, int *d_hist, int *d_status, hipStream_t s)
{
    BowArgs a;
    a.nb = nb, a.n2 = n2, a.m2 = m2, a.strict_lt = strict_lt, a.check_ori = check_ori;
    a.th_low = ORBX_TH_LOW, a.histo = ORBX_HISTO_LENGTH;
    a.ratio = ratio, a.factor = ORBX_HISTO_LENGTH / 360.0f;
    hipError_t e = hipMemsetAsync(d_match12, 0xff, sizeof(int) * (size_t)std::max(total_feat1, 1), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_match21, 0xff, sizeof(int) * (size_t)std::max(nb, 1) * (size_t)std::max(n2, 1), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_hist, 0, sizeof(int) * 32 * (size_t)std::max(nb, 1), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_status, 0, sizeof(int), s);
    if (e != hipSuccess)
        return e;
    if (total_nodes1 > 0 && m2 > 0) {
        k_bow_match<<<total_nodes1, 64, 0, s>>>(reinterpret_cast<const uint32_t *>(d_desc1), d_ang1, d_valid1,
                                                d_feat_start, d_node_kf, d_ids1, d_start1, d_items1,
                                                reinterpret_cast<const uint32_t *>(d_desc2), d_ang2, d_valid2, d_ids2,
                                                d_start2, d_items2, d_match12, d_match21, d_bin_of, d_hist, d_status, a);
        if ((e = hipGetLastError()) != hipSuccess)
            return e;
    }
    if (nb > 0) {
        k_bow_finish<<<nb, 256, 0, s>>>(d_feat_start, d_bin_of, d_hist, d_match12, d_match21, d_nmatches, a);
        e = hipGetLastError();
    }
    return e;
}
