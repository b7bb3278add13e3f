// ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:433-549 of the reference) for many frame
// pairs at once, entirely on the GPU: one wave per pair (frame t, frame t+1) of a batch whose
// keypoints and descriptors already sit in device memory.
//
// The pass over the keypoints of frame 1 is order dependent (a keypoint of frame 2 that is already
// matched at distance <= d is skipped, :471-472, and matches are stolen, :492-499), so the wave walks
// them in order; the 64 lanes share the candidates of one keypoint.  Only level-0 keypoints take
// part (:450-451 and the level arguments of GetFeaturesInArea :455).  A candidate of
// Frame::GetFeaturesInArea (src/Frame.cc:349-407) is a keypoint whose grid cell (PosInGrid, rounded,
// src/Frame.cc:409-420) lies in the cell window of the query (floor/ceil) and whose coordinates
// differ by less than r; the reference visits them cell column by cell column, each cell in
// keypoint order.  So the wave first sorts the level-0 keypoints of frame 2 that have a cell by
// (cell column, cell row, keypoint index) into LDS, descriptors included: the candidates of a query
// then lie in one contiguous range of positions (its cell columns), the visiting order is the
// position order, and the packed key  distance<<16 | position  orders candidates exactly like the
// strict '<' updates of that visit.  The two smallest keys of the wave are bestDist / bestDist2 /
// bestIdx2.
#include <algorithm>

#include "orbx_internal.h"


#define SFI_COLS 64  // FRAME_GRID_COLS (include/Frame.h:38)
#define SFI_ROWS 48  // FRAME_GRID_ROWS
#define SFI_DSTRIDE 9  // dwords per staged descriptor (8 + 1 pad: lanes hit different LDS banks)
#define SFI_NONE 0xffffu
#define SFI_SLOW 0xffffffffu

struct SfiArgs {
    int max_kp, npairs, cap0, check_ori, th_low, histo;
    int keys_cap;  // candidate keys that fit the LDS behind the per-keypoint arrays
    float min_x, min_y, inv_w, inv_h, r, ratio, factor;
};

// bytes of dynamic LDS for `cap0` level-0 keypoints per frame
size_t orbx_search_init_lds_bytes(int cap0)
{
    return (size_t)cap0 * (SFI_DSTRIDE * 4 + 4 * 7 + 2 * 6 + 4 + 32 + 8);
}

__device__ __forceinline__ int lanes_below64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

#define SFI_NT 1024  // threads that prepare a pair; the first wave then walks it alone

__global__ __launch_bounds__(SFI_NT) void k_search_init(const orbx_keypoint *__restrict__ kp,
                                                    const uint8_t *__restrict__ desc, const int *__restrict__ counts,
                                                    const float *__restrict__ prev_in, float *__restrict__ prev_out,
                                                    int *__restrict__ matches12, int *__restrict__ nmatches,
                                                    int *__restrict__ pair_count, SfiArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int cap = a.cap0;
    // frame 2, level 0, sorted by (cell, keypoint index)
    uint32_t *d2 = lds;                                                   // cap * 9 dwords
    float *x2 = reinterpret_cast<float *>(d2 + (size_t)cap * SFI_DSTRIDE);  // coordinates
    float *y2 = x2 + cap;
    float *ang2 = y2 + cap;
    uint32_t *skey = reinterpret_cast<uint32_t *>(ang2 + cap);  // sort keys in keypoint order: cell << 16 | index
    // frame 1, level 0, in keypoint order
    float *x1 = reinterpret_cast<float *>(skey + cap);  // search centre (vbPrevMatched)
    float *y1 = x1 + cap;
    float *ang1 = y1 + cap;
    uint16_t *cell2 = reinterpret_cast<uint16_t *>(ang1 + cap);  // grid column << 8 | grid row
    uint16_t *idx2 = cell2 + cap;                                // index in frame 2's keypoint array
    uint16_t *mdist = idx2 + cap;                                // vMatchedDistance
    uint16_t *m21 = mdist + cap;                                 // vnMatches21 as position in frame 1's list
    uint16_t *idx1 = m21 + cap;                                  // index in frame 1's keypoint array
    uint16_t *m12 = idx1 + cap;                                  // vnMatches12 as position in frame 2's list
    uint16_t *acc = m12 + cap;                                   // position in frame 2's list a keypoint was accepted for
    uint8_t *bin1 = reinterpret_cast<uint8_t *>(acc + cap);      // rotation bin of an accepted match, 0xff: none
    // descriptors of frame 1's level-0 keypoints in list order: a query read from global memory would set the
    // pace of the walk (one memory latency per step)
    uint32_t *d1 = reinterpret_cast<uint32_t *>(bin1 + cap);     // cap * 8 dwords (cap is a multiple of 4)
    // Candidate keys (distance << 16 | position) of every query, computed by all waves before the walk: the
    // distances do not depend on the walk's state, only the vMatchedDistance filter does.  qoff[j] = first key
    // of query j in `keys` (SFI_SLOW: its keys did not fit, the walk computes them itself), qcnt[j] their number.
    uint32_t *qoff = d1 + (size_t)cap * 8;
    uint32_t *qcnt = qoff + cap;
    uint32_t *keys = qcnt + cap;  // a.keys_cap entries
    __shared__ int s_alloc, s_npairs;
    __shared__ int hist[32];
    __shared__ int colstart[SFI_COLS + 1];  // first sorted position of each cell column
    __shared__ int s_wc[2][2][SFI_NT / 64];  // per chunk parity and frame: the waves' counts

    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n1 = min(counts[p], a.max_kp), n2 = min(counts[p + 1], a.max_kp);
    const orbx_keypoint *k1 = kp + (size_t)p * a.max_kp, *k2 = kp + (size_t)(p + 1) * a.max_kp;
    const uint32_t *dsc1 = reinterpret_cast<const uint32_t *>(desc) + (size_t)p * a.max_kp * 8;
    const uint32_t *dsc2 = reinterpret_cast<const uint32_t *>(desc) + (size_t)(p + 1) * a.max_kp * 8;
    int *out12 = matches12 + (size_t)p * a.max_kp;
    for (int i = tid; i < a.max_kp; i += SFI_NT)
        out12[i] = -1;
    if (tid < 32)
        hist[tid] = 0;
    if (tid == 0)
        s_alloc = 0, s_npairs = 0;

    // ---- set-up by the whole workgroup ----
    // level-0 keypoints of frame 2 that lie in a grid cell (sort keys) and level-0 keypoints of frame 1, both
    // compacted in keypoint order: 256 keypoints of either frame per step, positions from the waves' ballots
    int n20 = 0, n10 = 0;
    const int nmax = max(n1, n2);
    for (int i0 = 0, par = 0; i0 < nmax; i0 += SFI_NT, par ^= 1) {
        const int i = i0 + tid;
        int cell = SFI_NONE;
        if (i < n2 && k2[i].octave == 0) {
            // Frame::PosInGrid: rounded cell; keypoints outside the grid are in no cell
            const int px = (int)roundf((k2[i].x - a.min_x) * a.inv_w), py = (int)roundf((k2[i].y - a.min_y) * a.inv_h);
            if (px >= 0 && px < SFI_COLS && py >= 0 && py < SFI_ROWS)
                cell = (px << 8) | py;
        }
        const bool lvl0 = i < n1 && k1[i].octave <= 0;  // the reference skips octave > 0
        float fx = 0.f, fy = 0.f, fa = 0.f;
        if (lvl0) {
            const size_t pi = ((size_t)p * a.max_kp + i) * 2;
            fx = prev_in ? prev_in[pi] : k1[i].x;
            fy = prev_in ? prev_in[pi + 1] : k1[i].y;
            fa = k1[i].angle;
        }
        const unsigned long long bm2 = __ballot(cell != (int)SFI_NONE), bm1 = __ballot(lvl0);
        if (lane == 0)
            s_wc[par][0][wv] = __popcll(bm2), s_wc[par][1][wv] = __popcll(bm1);
        __syncthreads();  // one barrier per step: the counts of the step before live in the other half
        int before2 = 0, before1 = 0, all2 = 0, all1 = 0;
#pragma unroll
        for (int k = 0; k < SFI_NT / 64; k++) {
            const int c2 = s_wc[par][0][k], c1 = s_wc[par][1][k];
            before2 += k < wv ? c2 : 0, before1 += k < wv ? c1 : 0;
            all2 += c2, all1 += c1;
        }
        const int pos2 = n20 + before2 + lanes_below64(bm2), pos1 = n10 + before1 + lanes_below64(bm1);
        if (cell != (int)SFI_NONE && pos2 < cap)
            skey[pos2] = ((uint32_t)cell << 16) | (uint32_t)i;
        if (lvl0 && pos1 < cap) {
            x1[pos1] = fx, y1[pos1] = fy, ang1[pos1] = fa;
            idx1[pos1] = (uint16_t)i;
            m12[pos1] = (uint16_t)SFI_NONE;
            acc[pos1] = (uint16_t)SFI_NONE;
            bin1[pos1] = 0xff;
        }
        n20 += all2, n10 += all1;
    }
    if (n10 > cap || n20 > cap) {  // more level-0 keypoints than the caller allowed for
        if (tid == 0)
            nmatches[p] = ORBX_E_CAPACITY, pair_count[p] = 0;
        return;
    }
    if (tid < 3 && n20 + tid < ((n20 + 3) & ~3))
        skey[n20 + tid] = 0xffffffffu;  // padding of the last group of four keys (cap is a multiple of 4)
    __syncthreads();
    for (int e = tid; e < n10 * 8; e += SFI_NT)
        d1[e] = dsc1[(size_t)idx1[e >> 3] * 8 + (e & 7)];
    // ---- rank sort (keys are distinct), then everything of frame 2 goes to its sorted position ----
    const uint4 *skey4 = reinterpret_cast<const uint4 *>(skey);
    const int n204 = (n20 + 3) >> 2;
    for (int e0 = 0; e0 < n20; e0 += SFI_NT) {
        const int e = e0 + tid;
        const uint32_t mine = e < n20 ? skey[e] : 0xffffffffu;
        int rank = 0;
#pragma unroll 4
        for (int q = 0; q < n204; q++) {  // the same address in every lane
            const uint4 k4 = skey4[q];
            rank += (k4.x < mine) + (k4.y < mine) + (k4.z < mine) + (k4.w < mine);
        }
        if (e < n20) {
            const int i = mine & 0xffffu;
            x2[rank] = k2[i].x, y2[rank] = k2[i].y, ang2[rank] = k2[i].angle;
            cell2[rank] = (uint16_t)(mine >> 16);
            idx2[rank] = (uint16_t)i;
            mdist[rank] = (uint16_t)SFI_NONE;
            m21[rank] = (uint16_t)SFI_NONE;
#pragma unroll
            for (int k = 0; k < 8; k++)
                d2[rank * SFI_DSTRIDE + k] = dsc2[(size_t)i * 8 + k];
        }
    }
    __syncthreads();
    // colstart[c] = number of sorted entries in cell columns below c
    for (int c = tid; c <= SFI_COLS; c += SFI_NT) {
        int lo = 0, hi = n20;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((cell2[mid] >> 8) < c)
                lo = mid + 1;
            else
                hi = mid;
        }
        colstart[c] = lo;
    }
    __syncthreads();
    // cell window of GetFeaturesInArea as a range of sorted positions and a cell-row interval; an empty range
    // leaves the keypoint unmatched
    auto window = [&](float qx, float qy, int &c_lo, int &c_hi, int &cy0, int &cy1) {
        const int cx0 = max(0, (int)floorf((qx - a.min_x - a.r) * a.inv_w));
        const int cx1 = min(SFI_COLS - 1, (int)ceilf((qx - a.min_x + a.r) * a.inv_w));
        cy0 = max(0, (int)floorf((qy - a.min_y - a.r) * a.inv_h));
        cy1 = min(SFI_ROWS - 1, (int)ceilf((qy - a.min_y + a.r) * a.inv_h));
        c_lo = c_hi = 0;
        if (cx0 >= SFI_COLS || cx1 < 0 || cy0 >= SFI_ROWS || cy1 < 0)
            return;
        c_lo = __builtin_amdgcn_readfirstlane(colstart[cx0]);
        c_hi = cx1 >= cx0 ? __builtin_amdgcn_readfirstlane(colstart[cx1 + 1]) : c_lo;
    };
    auto inside = [&](int c, float qx, float qy, int cy0, int cy1) -> bool {
        const int py = cell2[c] & 0xff;
        return py >= cy0 && py <= cy1 && fabsf(x2[c] - qx) < a.r && fabsf(y2[c] - qy) < a.r;
    };
    auto distance = [&](const uint32_t (&q)[8], int c) -> uint32_t {
        const uint32_t *r = d2 + c * SFI_DSTRIDE;
        uint32_t d = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            d += __popc(q[k] ^ r[k]);
        return d;
    };
    // ---- candidate keys of all queries, a wave per query ----
    for (int j = wv; j < n10; j += SFI_NT / 64) {
        const float qx = x1[j], qy = y1[j];
        int c_lo = 0, c_hi = 0, cy0 = 0, cy1 = 0;
        window(qx, qy, c_lo, c_hi, cy0, cy1);
        // count first: the keys of a query are allocated in one piece
        int cnt = 0;
        for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
            const int c = c0 + lane;
            cnt += __popcll(__ballot(c < c_hi && inside(c, qx, qy, cy0, cy1)));
        }
        uint32_t off = 0;
        if (cnt > 0) {
            int got = 0;
            if (lane == 0)
                got = atomicAdd(&s_alloc, cnt);
            got = __builtin_amdgcn_readfirstlane(got);
            off = got + cnt <= a.keys_cap ? (uint32_t)got : SFI_SLOW;
        }
        if (lane == 0)
            qoff[j] = off, qcnt[j] = (uint32_t)cnt;
        if (cnt == 0 || off == SFI_SLOW)
            continue;
        if (lane == 0)
            atomicAdd(&s_npairs, cnt);
        uint32_t q[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            q[k] = d1[j * 8 + k];
        int at = (int)off;
        for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
            const int c = c0 + lane;
            const bool in = c < c_hi && inside(c, qx, qy, cy0, cy1);
            const unsigned long long bm = __ballot(in);
            if (in)
                keys[at + lanes_below64(bm)] = (distance(q, c) << 16) | (uint32_t)c;
            at += __popcll(bm);
        }
    }
    __syncthreads();
    // The walk is sequential: the first wave does it alone.  From here on there is no workgroup barrier (the
    // other waves are gone); inside one wave LDS operations complete in program order, so a write by lane 0 is
    // seen by the reads of all lanes that follow it once the compiler keeps that order (wave_sync below).
    if (wv != 0)
        return;
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };

    // ---- the order-dependent pass (:446-516) ----
    int npairs = 0;
    // what step j+1 reads first (its key range and its first 64 keys) is fetched while step j decides
    uint32_t noff = 0, nkey = 0xffffffffu;
    int ncnt = 0;
    auto fetch_step = [&](int j) {
        if (j < n10) {
            noff = qoff[j], ncnt = (int)qcnt[j];
            nkey = (noff != SFI_SLOW && lane < ncnt) ? keys[noff + lane] : 0xffffffffu;
        }
    };
    fetch_step(0);
    for (int j = 0; j < n10; j++) {
        const uint32_t off = noff, key0 = nkey;
        const int cnt = ncnt;
        fetch_step(j + 1);
        if (cnt == 0)
            continue;
        uint32_t b1 = 0xffffffffu, b2 = 0xffffffffu;
        auto offer = [&](uint32_t key) {
            if (!((uint32_t)mdist[key & 0xffffu] <= (key >> 16))) {  // vMatchedDistance starts at INT_MAX
                const uint32_t hi = max(b1, key);
                b1 = min(b1, key);
                b2 = min(b2, hi);
            }
        };
        if (off != SFI_SLOW) {
            if (lane < cnt)
                offer(key0);
            for (int e = 64 + lane; e < cnt; e += 64)
                offer(keys[off + e]);
        } else {
            // the keys of this query did not fit: computed here
            const float qx = x1[j], qy = y1[j];
            uint32_t q[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                q[k] = d1[j * 8 + k];
            int c_lo = 0, c_hi = 0, cy0 = 0, cy1 = 0;
            window(qx, qy, c_lo, c_hi, cy0, cy1);
            for (int c = c_lo + lane; c < c_hi; c += 64)
                if (inside(c, qx, qy, cy0, cy1)) {
                    npairs++;
                    offer((distance(q, c) << 16) | (uint32_t)c);
                }
        }
        const uint32_t w1 = orbx_wave_min_u32(b1);
        if (w1 == 0xffffffffu)
            continue;
        // the runner-up is the winner lane's second key or another lane's first (keys are distinct)
        const int old21 = m21[w1 & 0xffffu];  // in flight while the runner-up is found
        const uint32_t w2 = orbx_wave_min_u32(b1 == w1 ? b2 : b1);
        const int best = (int)(w1 >> 16);
        const float best2 = w2 == 0xffffffffu ? 2147483648.0f : (float)(int)(w2 >> 16);  // INT_MAX as float
        if (best <= a.th_low && (float)best < best2 * a.ratio) {
            const int c = (int)(w1 & 0xffffu);
            if (lane == 0) {
                const int old = old21;
                if (old != (int)SFI_NONE)
                    m12[old] = (uint16_t)SFI_NONE;
                m12[j] = (uint16_t)c;
                m21[c] = (uint16_t)j;
                mdist[c] = (uint16_t)best;
                acc[j] = (uint16_t)c;
            }
            wave_sync();  // lane 0's updates are read by every lane from here on
        }
    }
    wave_sync();

    // ---- rotation consistency (:502-541, ComputeThreeMaxima :1698-1739) ----
    // The histogram holds every acceptance of the walk, also those whose feature a later keypoint took over
    // (the reference pushes at acceptance and never removes, :502-513); the bins are worked out here, for all
    // keypoints at once, instead of inside the walk.
    if (a.check_ori) {
        for (int j = lane; j < n10; j += 64) {
            const int c = acc[j];
            if (c != (int)SFI_NONE) {
                float rot = ang1[j] - ang2[c];
                if (rot < 0.0f)
                    rot += 360.0f;
                int bin = (int)roundf(rot * a.factor);
                if (bin == a.histo)
                    bin = 0;
                bin1[j] = (uint8_t)bin;
                atomicAdd(&hist[bin & 31], 1);
            }
        }
        wave_sync();
    }
    int i1m = -1, i2m = -1, i3m = -1;
    if (a.check_ori) {
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < a.histo; i++) {
            const int s = hist[i];
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
    int nm = 0;
    for (int j = lane; j < n10; j += 64) {
        int c = m12[j];
        if (c != (int)SFI_NONE && a.check_ori) {
            const int b = bin1[j];
            if (!(b == i1m || b == i2m || b == i3m))
                c = SFI_NONE;
        }
        if (c != (int)SFI_NONE) {
            nm++;
            const int i1 = idx1[j];
            out12[i1] = idx2[c];
            if (prev_out) {  // :544-546
                const size_t pi = ((size_t)p * a.max_kp + i1) * 2;
                prev_out[pi] = x2[c], prev_out[pi + 1] = y2[c];
            }
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        nm += __shfl_xor(nm, s, 64);
        npairs += __shfl_xor(npairs, s, 64);
    }
    if (lane == 0)
        nmatches[p] = nm, pair_count[p] = npairs + s_npairs;
}

hipError_t orbx_launch_search_init(const orbx_keypoint *d_kp, const uint8_t *d_desc, const int *d_counts, int max_kp,
                                   int npairs, int cap0, float min_x, float max_x, float min_y, float max_y,
                                   const float *d_prev_in, float *d_prev_out, int window, float ratio, int check_ori,
                                   int *d_matches12, int *d_nmatches, int *d_pair_count, hipStream_t s)
{
    if (npairs <= 0)
        return hipSuccess;
    SfiArgs a;
    a.max_kp = max_kp, a.npairs = npairs, a.cap0 = cap0, a.check_ori = check_ori;
    a.th_low = ORBX_TH_LOW, a.histo = ORBX_HISTO_LENGTH;
    a.min_x = min_x, a.min_y = min_y;
    a.inv_w = (float)SFI_COLS / (max_x - min_x), a.inv_h = (float)SFI_ROWS / (max_y - min_y);
    a.r = (float)window, a.ratio = ratio, a.factor = ORBX_HISTO_LENGTH / 360.0f;
    cap0 = (cap0 + 3) & ~3;  // 16-byte reads of the sort keys
    a.cap0 = cap0;
    // the LDS left of 150 KB holds candidate keys (at most 128 per query): a workgroup then has a compute unit
    // to itself, which is what a launch of up to 256 pairs (one per compute unit) gets anyway; larger launches
    // keep two workgroups on a compute unit and fewer keys
    const size_t base = orbx_search_init_lds_bytes(cap0), budget = (npairs <= 256 ? 150 : 76) * 1024;
    a.keys_cap = (int)std::min<size_t>(base < budget ? (budget - base) / 4 : 0, (size_t)cap0 * 128);
    const size_t lds = base + (size_t)a.keys_cap * 4;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_init),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
    }
    k_search_init<<<npairs, SFI_NT, lds, s>>>(d_kp, d_desc, d_counts, d_prev_in, d_prev_out, d_matches12, d_nmatches,
                                          d_pair_count, a);
    return hipGetLastError();
}
