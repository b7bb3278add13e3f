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
#include <cstddef>

#include "orbx_internal.h"


#define BOW_NODE_CAP 4096  // features of side 2 in one vocabulary node that a wave can flag in LDS

struct BowArgs {
    int nb, n2, m2, strict_lt, check_ori, th_low, histo;
    float ratio, factor;
    // angle arrays: bytes between the angles of consecutive features (4: a float array; 28: orbx_keypoint records)
    int ang1_stride, ang2_stride;
    // Strided layout (device-resident frames, as orbx_batch_device_results lays them out): key frame b's
    // features are rows b*cap1 .. of side 1, its node entries b*cap1 .. (ids ascending, padded with INT32_MAX),
    // its entry offsets b*(cap1+1) .. and its item list b*cap1 ..; 0 = the concatenated layout of the host form
    int cap1;
};

#define BOW_PAD_NODE 0x7fffffff  // node id of the padding entries behind a frame's feature vector

__global__ __launch_bounds__(64) void k_bow_match(const uint32_t *__restrict__ d1, const uint8_t *__restrict__ ang1,
                                                  const uint8_t *__restrict__ valid1, const int *__restrict__ feat_start,
                                                  const int *__restrict__ node_kf, const int *__restrict__ ids1,
                                                  const int *__restrict__ start1, const int *__restrict__ items1,
                                                  const uint32_t *__restrict__ d2, const uint8_t *__restrict__ ang2,
                                                  const uint8_t *__restrict__ valid2, const int *__restrict__ ids2,
                                                  const int *__restrict__ start2, const int *__restrict__ items2,
                                                  int *__restrict__ match12, int *__restrict__ match21,
                                                  int8_t *__restrict__ bin_of, int *__restrict__ hist,
                                                  int *__restrict__ status, int *__restrict__ kf_status, BowArgs a)
{
    __shared__ uint8_t taken[BOW_NODE_CAP];
    const int e = blockIdx.x, lane = threadIdx.x;
    int b, f0, it0, it1;
    const int *my_items1;
    const int node = ids1[e];
    if (a.cap1) {
        if (node == BOW_PAD_NODE)
            return;
        b = e / a.cap1;
        const int le = e - b * a.cap1;
        f0 = b * a.cap1;
        it0 = start1[b * (a.cap1 + 1) + le], it1 = start1[b * (a.cap1 + 1) + le + 1];
        my_items1 = items1 + f0;
    } else {
        b = node_kf[e];
        f0 = feat_start[b];
        it0 = start1[e], it1 = start1[e + 1];
        my_items1 = items1;
    }
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
        if (lane == 0) {
            atomicMax(status, c2);
            if (kf_status)
                kf_status[b] = 1;
        }
        return;
    }
    for (int c = lane; c < c2; c += 64)
        taken[c] = 0;
    __syncthreads();
    int *m21 = match21 + (size_t)b * a.n2;
    for (int i1 = it0; i1 < it1; i1++) {
        const int idx1 = __builtin_amdgcn_readfirstlane(my_items1[i1]);
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
                    float rot = *reinterpret_cast<const float *>(ang1 + (size_t)f1 * a.ang1_stride) -
                                *reinterpret_cast<const float *>(ang2 + (size_t)idx2 * a.ang2_stride);
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
__global__ __launch_bounds__(256) void k_bow_finish(const int *__restrict__ feat_start, const int *__restrict__ counts1,
                                                    const int8_t *__restrict__ bin_of, const int *__restrict__ hist,
                                                    const int *__restrict__ kf_status, int *__restrict__ match12,
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
    const int fa = a.cap1 ? b * a.cap1 : feat_start[b];
    const int fb = a.cap1 ? fa + min(max(counts1[b], 0), a.cap1) : feat_start[b + 1];
    for (int f = fa + tid; f < fb; f += 256) {
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
        nmatches[b] = kf_status && kf_status[b] ? ORBX_E_CAPACITY : total;
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
    a.ang1_stride = a.ang2_stride = sizeof(float);
    a.cap1 = 0;
    hipError_t e = hipMemsetAsync(d_match12, 0xff, sizeof(int) * (size_t)std::max(total_feat1, 1), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_match21, 0xff, sizeof(int) * (size_t)std::max(nb, 1) * (size_t)std::max(n2, 1), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_hist, 0, sizeof(int) * 32 * (size_t)std::max(nb, 1), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_status, 0, sizeof(int), s);
    if (e != hipSuccess)
        return e;
    if (total_nodes1 > 0 && m2 > 0) {
        k_bow_match<<<total_nodes1, 64, 0, s>>>(reinterpret_cast<const uint32_t *>(d_desc1),
                                                reinterpret_cast<const uint8_t *>(d_ang1), d_valid1, d_feat_start, d_node_kf,
                                                d_ids1, d_start1, d_items1, reinterpret_cast<const uint32_t *>(d_desc2),
                                                reinterpret_cast<const uint8_t *>(d_ang2), d_valid2, d_ids2, d_start2,
                                                d_items2, d_match12, d_match21, d_bin_of, d_hist, d_status, nullptr, a);
        if ((e = hipGetLastError()) != hipSuccess)
            return e;
    }
    if (nb > 0) {
        k_bow_finish<<<nb, 256, 0, s>>>(d_feat_start, nullptr, d_bin_of, d_hist, nullptr, d_match12, d_match21, d_nmatches,
                                        a);
        e = hipGetLastError();
    }
    return e;
}

// The same search on frames that live in device memory in the layout of orbx_batch_device_results: nb key
// frames of cap1 rows each on side 1, one frame of cap2 rows on side 2, feature vectors as
// orbx_feature_vectors_build_device leaves them (entries padded to the row capacity).  Angles are read from
// the keypoint records.  d_ws: cap1*nb bytes (rotation bins) + 33*nb ints (histograms, per-key-frame status)
// + 1 int, carved by the caller.
hipError_t orbx_launch_bow_strided(const orbx_keypoint *d_kp1, const uint8_t *d_desc1, const int *d_counts1,
                                   const uint8_t *d_valid1, int nb, int cap1, const int *d_ids1, const int *d_start1,
                                   const int *d_items1, const orbx_keypoint *d_kp2, const uint8_t *d_desc2,
                                   const uint8_t *d_valid2, int cap2, const int *d_ids2, const int *d_start2,
                                   const int *d_items2, int strict_lt, float ratio, int check_ori, int *d_match12,
                                   int *d_match21, int *d_nmatches, int8_t *d_bin_of, int *d_hist, int *d_kf_status,
                                   int *d_status, hipStream_t s)
{
    BowArgs a;
    a.nb = nb, a.n2 = cap2, a.m2 = cap2, a.strict_lt = strict_lt, a.check_ori = check_ori;
    a.th_low = ORBX_TH_LOW, a.histo = ORBX_HISTO_LENGTH;
    a.ratio = ratio, a.factor = ORBX_HISTO_LENGTH / 360.0f;
    a.ang1_stride = a.ang2_stride = sizeof(orbx_keypoint);
    a.cap1 = cap1;
    hipError_t e = hipMemsetAsync(d_match12, 0xff, sizeof(int) * (size_t)nb * cap1, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_match21, 0xff, sizeof(int) * (size_t)nb * cap2, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_hist, 0, sizeof(int) * 33 * (size_t)nb + sizeof(int), s);  // + status words
    if (e != hipSuccess)
        return e;
    const uint8_t *ang1 = reinterpret_cast<const uint8_t *>(d_kp1) + offsetof(orbx_keypoint, angle);
    const uint8_t *ang2 = reinterpret_cast<const uint8_t *>(d_kp2) + offsetof(orbx_keypoint, angle);
    k_bow_match<<<nb * cap1, 64, 0, s>>>(reinterpret_cast<const uint32_t *>(d_desc1), ang1, d_valid1, nullptr, nullptr,
                                         d_ids1, d_start1, d_items1, reinterpret_cast<const uint32_t *>(d_desc2), ang2,
                                         d_valid2, d_ids2, d_start2, d_items2, d_match12, d_match21, d_bin_of, d_hist,
                                         d_status, d_kf_status, a);
    if ((e = hipGetLastError()) != hipSuccess)
        return e;
    k_bow_finish<<<nb, 256, 0, s>>>(nullptr, d_counts1, d_bin_of, d_hist, d_kf_status, d_match12, d_match21, d_nmatches, a);
    return hipGetLastError();
}

// ---- DBoW2::FeatureVector of device-resident frames ----
// FeatureVector is a std::map<NodeId, std::vector<unsigned>> filled by addFeature(nid, i_feature) in feature
// order for the features whose word has a positive weight (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1155-1161,
// FeatureVector.cpp:31-45): node ids ascending, a node's features in index order.  That is a sort of the
// features by (node, index).  Frame f's features are rows f*cap ..; node[i] < 0 leaves feature i out.
// k_fv_sort: every workgroup holds all keys of its frame in LDS and ranks 256 of them (keys are distinct).
__global__ __launch_bounds__(256) void k_fv_sort(const int *__restrict__ node, const int *__restrict__ counts, int cap,
                                                 int *__restrict__ sorted_node, int *__restrict__ items)
{
    extern __shared__ unsigned long long fv_keys[];
    const int f = blockIdx.y, tid = threadIdx.x;
    const int n = min(max(counts[f], 0), cap);
    const int *nd = node + (size_t)f * cap;
    for (int i = tid; i < n; i += 256)
        fv_keys[i] = nd[i] >= 0 ? ((unsigned long long)(uint32_t)nd[i] << 16) | (unsigned long long)i : ~0ull;
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i >= n)
        return;
    const unsigned long long k = fv_keys[i];
    if (k == ~0ull)
        return;
    int rank = 0;
    for (int j = 0; j < n; j++)
        rank += fv_keys[j] < k;  // the same address in every lane
    sorted_node[(size_t)f * cap + rank] = (int)(k >> 16);
    items[(size_t)f * cap + rank] = i;
}

// one workgroup per frame: run heads of the sorted node list become the entries (id, first item); entries
// behind the last one are padding (BOW_PAD_NODE, empty range)
__global__ __launch_bounds__(256) void k_fv_entries(const int *__restrict__ node, const int *__restrict__ counts, int cap,
                                                    const int *__restrict__ sorted_node, int *__restrict__ ids,
                                                    int *__restrict__ start, int *__restrict__ num_nodes)
{
    __shared__ int wsum[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = min(max(counts[f], 0), cap);
    const int *nd = node + (size_t)f * cap, *sn = sorted_node + (size_t)f * cap;
    int *my_ids = ids + (size_t)f * cap, *my_start = start + (size_t)f * (cap + 1);
    // features that are in some node
    int v = 0;
    for (int i = tid; i < n; i += 256)
        v += nd[i] >= 0;
    v = __builtin_amdgcn_readlane(orbx_wave_scan_add_i32(v), 63);
    if (lane == 0)
        wsum[wv] = v;
    __syncthreads();
    const int V = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    // a thread takes a run of consecutive positions
    const int per = (cap + 255) / 256, p0 = tid * per, p1 = min(p0 + per, V);
    int heads = 0;
    for (int p = p0; p < p1; p++)
        heads += p == 0 || sn[p] != sn[p - 1];
    const int incl = orbx_wave_scan_add_i32(heads);
    if (lane == 63)
        wsum[wv] = incl;
    __syncthreads();
    int e = incl - heads;
    for (int w = 0; w < wv; w++)
        e += wsum[w];
    const int E = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    for (int p = p0; p < p1; p++)
        if (p == 0 || sn[p] != sn[p - 1]) {
            my_ids[e] = sn[p];
            my_start[e] = p;
            e++;
        }
    for (int q = E + tid; q < cap; q += 256)
        my_ids[q] = BOW_PAD_NODE;
    for (int q = E + tid; q <= cap; q += 256)
        my_start[q] = V;
    if (tid == 0)
        num_nodes[f] = E;
}

hipError_t orbx_launch_feature_vectors(const int *d_node, const int *d_counts, int nframes, int cap, int *d_sorted,
                                       int *d_ids, int *d_start, int *d_items, int *d_num, hipStream_t s)
{
    k_fv_sort<<<dim3((cap + 255) / 256, nframes), 256, (size_t)cap * sizeof(unsigned long long), s>>>(d_node, d_counts, cap,
                                                                                                      d_sorted, d_items);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    k_fv_entries<<<nframes, 256, 0, s>>>(d_node, d_counts, cap, d_sorted, d_ids, d_start, d_num);
    return hipGetLastError();
}
