// ORBmatcher::SearchByProjection, the two forms the tracking thread runs on every frame
// (src/ORBmatcher.cc:46-137: local map points against a frame; :1412-1566: the last frame's map points
// against the current frame).  Both walk their sources in order, query the frame grid for a window
// around the projection, take the best (and second best) descriptor distance among the features that
// are still free, and give the feature to the source - which later sources then have to skip.  That
// exclusion makes the walk sequential; everything else is not:
//   1. the windows of all sources are answered at once (k_features_in_area of the frame grid),
//   2. k_proj_dist evaluates the right-coordinate rule and the distance of every (source, candidate),
//   3. k_proj_assign, one workgroup, replays the walk on those numbers: all its waves stage the candidate
//      rows of 64 sources in LDS, then one wave decides them in order: free/taken flags in LDS, the lanes
//      share a source's candidates, best and second best by wave-wide minima of
//      distance << 20 | list position << 4 | level, so ties go to the earlier candidate as the strict '<'
//      of the reference does,
//   4. k_proj_finish builds the rotation histogram of the accepted matches (:1528-1540) and clears
//      those outside its three fullest bins (:1542-1560).
#include <type_traits>

#include "orbx_internal.h"


#define PROJ_SKIP 0xffffffffu
#define PROJ_HISTO 30     // HISTO_LENGTH, src/ORBmatcher.cc:37
#define PROJ_MAX_N 32768  // features of the frame (taken flags in LDS); also the limit of the frame grid
#define PROJ_SLOTS 4      // candidates per lane: lists of up to 256

// entry = distance << 20 | position << 4 | level, or PROJ_SKIP where the right-coordinate rule
// (:89-94, :1488-1494) drops the candidate
__global__ __launch_bounds__(64) void k_proj_dist(const orbx_keypoint *__restrict__ kp, const uint32_t *__restrict__ desc,
                                                  const float *__restrict__ uright, const float *__restrict__ r,
                                                  const float *__restrict__ xr, const uint32_t *__restrict__ src_desc,
                                                  const int32_t *__restrict__ list, const int32_t *__restrict__ count,
                                                  int cap, uint32_t *__restrict__ entry)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const int n = min(count[s], cap);
    if (n <= 0)
        return;
    uint32_t q[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
        q[k] = src_desc[(size_t)s * 8 + k];
    const float rs = r[s], xrs = xr ? xr[s] : 0.f;
    for (int j = lane; j < n; j += 64) {
        const int idx = list[(size_t)s * cap + j];
        bool skip = false;
        if (uright) {
            const float ur = uright[idx];
            if (ur > 0)
                skip = fabsf(xrs - ur) > rs;
        }
        const uint4 d0 = *reinterpret_cast<const uint4 *>(desc + (size_t)idx * 8);
        const uint4 d1 = *reinterpret_cast<const uint4 *>(desc + (size_t)idx * 8 + 4);
        const int d = __popc(q[0] ^ d0.x) + __popc(q[1] ^ d0.y) + __popc(q[2] ^ d0.z) + __popc(q[3] ^ d0.w) +
                      __popc(q[4] ^ d1.x) + __popc(q[5] ^ d1.y) + __popc(q[6] ^ d1.z) + __popc(q[7] ^ d1.w);
        entry[(size_t)s * cap + j] =
            skip ? PROJ_SKIP : ((uint32_t)d << 20) | ((uint32_t)j << 4) | (uint32_t)(kp[idx].octave & 15);
    }
}

struct ProjArgs {
    int n, ns, cap, use_second, th_dist, check_ori;
    float nnratio;
};

#define PROJ_CHUNK 64  // sources staged in LDS at a time
#define PROJ_STAGE 32  // candidates staged per source (longer lists are read from global memory)

// One workgroup: all four waves stage the candidate rows of a chunk of sources in LDS (one round of
// independent loads), then the first wave walks the chunk's sources in order.
__global__ __launch_bounds__(256) void k_proj_assign(const uint8_t *__restrict__ taken_in,
                                                     const uint8_t *__restrict__ valid, const uint8_t *__restrict__ blocks,
                                                     const int32_t *__restrict__ list, const int32_t *__restrict__ count,
                                                     const uint32_t *__restrict__ entry, int32_t *__restrict__ match,
                                                     int32_t *__restrict__ feature_match, int32_t *__restrict__ result,
                                                     ProjArgs a)
{
    __shared__ uint32_t taken[PROJ_MAX_N / 32];  // one bit per feature
    __shared__ uint2 s_ce[PROJ_CHUNK][PROJ_STAGE];  // (feature, entry)
    __shared__ int s_cnt[PROJ_CHUNK], s_blk[PROJ_CHUNK];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < PROJ_MAX_N / 32; i += 256)
        taken[i] = 0;
    __syncthreads();
    for (int i = tid; i < a.n; i += 256) {
        if (taken_in && taken_in[i] != 0)
            atomicOr(&taken[i >> 5], 1u << (i & 31));
        feature_match[i] = -1;
    }
    int nmatches = 0, overflow = 0;
    for (int s0 = 0; s0 < a.ns; s0 += PROJ_CHUNK) {
        __syncthreads();  // the previous chunk is done (and, the first time, the flags are set)
        const int nsrc = min(PROJ_CHUNK, a.ns - s0);
        for (int i = tid; i < nsrc * PROJ_STAGE; i += 256) {
            const int g = i / PROJ_STAGE, j = i % PROJ_STAGE;
            const size_t at = (size_t)(s0 + g) * a.cap + j;  // rows are read whole; the count masks them later
            s_ce[g][j] = make_uint2((uint32_t)list[at], entry[at]);
        }
        if (tid < nsrc) {
            const int s = s0 + tid;
            s_cnt[tid] = (valid && valid[s] == 0) ? 0 : count[s];
            s_blk[tid] = blocks ? (int)(blocks[s] != 0) : 1;
        }
        __syncthreads();
        if (tid >= 64)
            continue;
        // the staged count and row of source g+1 are read while source g is decided
        int ncnt = s_cnt[0];
        uint2 nce = s_ce[0][lane & (PROJ_STAGE - 1)];
        for (int g = 0; g < nsrc; g++) {
            const int s = s0 + g;
            const int cnt = ncnt;
            const uint2 ce0 = nce;
            if (g + 1 < nsrc) {
                ncnt = s_cnt[g + 1];
                nce = s_ce[g + 1][lane & (PROJ_STAGE - 1)];
            }
            // one decision, on SLOTS candidates per lane: nearly every window holds at most 32 candidates (one
            // slot, staged in LDS); the four-slot form serves the crowded ones from global memory
            auto decide = [&](auto slots_tag) -> int {
                constexpr int SLOTS = decltype(slots_tag)::value;
                int idx[SLOTS];
                uint32_t ent[SLOTS];
#pragma unroll
                for (int k = 0; k < SLOTS; k++)
                    idx[k] = -1, ent[k] = PROJ_SKIP;
                if constexpr (SLOTS == 1) {
                    if (lane < cnt)
                        idx[0] = (int)ce0.x, ent[0] = ce0.y;
                } else {
#pragma unroll
                    for (int k = 0; k < SLOTS; k++) {
                        const int j = lane + 64 * k;
                        if (j < min(cnt, a.cap)) {
                            idx[k] = list[(size_t)s * a.cap + j];
                            ent[k] = entry[(size_t)s * a.cap + j];
                        }
                    }
                }
                uint32_t mine = PROJ_SKIP;
#pragma unroll
                for (int k = 0; k < SLOTS; k++) {
                    if (idx[k] >= 0 && ((taken[idx[k] >> 5] >> (idx[k] & 31)) & 1u))
                        ent[k] = PROJ_SKIP;
                    mine = min(mine, ent[k]);
                }
                const uint32_t best = orbx_wave_min_u32(mine);
                if (best == PROJ_SKIP)
                    return -1;
                const int best_d = (int)(best >> 20);
                if (best_d > a.th_dist)
                    return -1;
                if (a.use_second) {
                    uint32_t rest = PROJ_SKIP;
#pragma unroll
                    for (int k = 0; k < SLOTS; k++)
                        rest = min(rest, ent[k] == best ? PROJ_SKIP : ent[k]);
                    const uint32_t second = orbx_wave_min_u32(rest);
                    // no second candidate: bestLevel2 stays -1 and differs from any level (:118-126)
                    if (second != PROJ_SKIP && (best & 15u) == (second & 15u) &&
                        (float)best_d > __fmul_rn(a.nnratio, (float)(int)(second >> 20)))
                        return -1;
                }
                if constexpr (SLOTS == 1) {
                    // an entry carries its position in the list, which here is the lane that holds it
                    return __builtin_amdgcn_readlane(idx[0], (int)((best >> 4) & 63u));
                } else {
                    int cand = -1;
#pragma unroll
                    for (int k = 0; k < SLOTS; k++)
                        if (ent[k] == best)
                            cand = idx[k];
                    return orbx_wave_max_i32(cand);
                }
            };
            int m = -1;
            if (cnt > a.cap)
                overflow = 1;
            if (cnt > 0)
                m = cnt <= PROJ_STAGE ? decide(std::integral_constant<int, 1>{})
                                      : decide(std::integral_constant<int, PROJ_SLOTS>{});
            if (m >= 0) {
                nmatches++;
                if (lane == 0) {
                    // one LDS operation without a return value: nothing waits for it, and the reads of the next
                    // source come behind it in the wave's order
                    const uint32_t bit = 1u << (m & 31);
                    if (s_blk[g])
                        atomicOr(&taken[m >> 5], bit);
                    else
                        atomicAnd(&taken[m >> 5], ~bit);
                    feature_match[m] = s;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0)
                match[s] = m;
        }
    }
    if (tid == 0) {
        result[0] = nmatches;
        result[1] = overflow;
    }
}

// rotation histogram of the accepted matches with factor HISTO_LENGTH/360.0f (:1421, :1528-1540) or, for
// the key-frame form, 1.0f/HISTO_LENGTH (:1581, :1654-1663), ComputeThreeMaxima (:1698-1739) and the
// removal loop (:1546-1557, :1667-1685); one workgroup
__global__ __launch_bounds__(256) void k_proj_finish(const orbx_keypoint *__restrict__ kp, const float *__restrict__ src_angle,
                                                     const int32_t *__restrict__ match, int32_t *__restrict__ feature_match,
                                                     int32_t *__restrict__ result, int ns, float factor)
{
    __shared__ int hist[PROJ_HISTO];
    __shared__ int s_removed;
    const int tid = threadIdx.x;
    if (tid < PROJ_HISTO)
        hist[tid] = 0;
    if (tid == 0)
        s_removed = 0;
    __syncthreads();
    auto bin_of = [&](int s, int m) {
        float rot = __fsub_rn(src_angle[s], kp[m].angle);
        if (rot < 0.0f)
            rot = __fadd_rn(rot, 360.0f);
        const int bin = (int)roundf(__fmul_rn(rot, factor));
        return bin == PROJ_HISTO ? 0 : bin;
    };
    for (int s = tid; s < ns; s += 256) {
        const int m = match[s];
        if (m >= 0)
            atomicAdd(&hist[bin_of(s, m)], 1);
    }
    __syncthreads();
    int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < PROJ_HISTO; i++) {
        const int c = hist[i];
        if (c > max1) {
            max3 = max2, max2 = max1, max1 = c;
            i3 = i2, i2 = i1, i1 = i;
        } else if (c > max2) {
            max3 = max2, max2 = c;
            i3 = i2, i2 = i;
        } else if (c > max3) {
            max3 = c;
            i3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1)
        i2 = -1, i3 = -1;
    else if ((float)max3 < 0.1f * (float)max1)
        i3 = -1;
    int removed = 0;
    for (int s = tid; s < ns; s += 256) {
        const int m = match[s];
        if (m < 0)
            continue;
        const int b = bin_of(s, m);
        if (b != i1 && b != i2 && b != i3) {
            feature_match[m] = -2;
            removed++;
        }
    }
    if (removed)
        atomicAdd(&s_removed, removed);
    __syncthreads();
    if (tid == 0)
        result[0] -= s_removed;
}

// d_list / d_count: the windows answered by orbx_launch_features_in_area (cap entries per source)
hipError_t orbx_launch_search_by_projection(const orbx_keypoint *d_kp, const uint8_t *d_desc, const float *d_uright,
                                            const uint8_t *d_taken, int n, const uint8_t *d_valid, const float *d_r,
                                            const float *d_xr, const uint8_t *d_src_desc, const uint8_t *d_blocks,
                                            const float *d_src_angle, int ns, const int32_t *d_list,
                                            const int32_t *d_count, int cap, uint32_t *d_entry, int use_second,
                                            float nnratio, int th_dist, int check_orientation, int32_t *d_match,
                                            int32_t *d_feature_match, int32_t *d_result, hipStream_t s)
{
    if (n > PROJ_MAX_N || cap > 64 * PROJ_SLOTS)
        return hipErrorInvalidValue;
    if (ns > 0) {
        k_proj_dist<<<ns, 64, 0, s>>>(d_kp, reinterpret_cast<const uint32_t *>(d_desc), d_uright, d_r, d_xr,
                                      reinterpret_cast<const uint32_t *>(d_src_desc), d_list, d_count, cap, d_entry);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
    }
    ProjArgs a{n, ns, cap, use_second, th_dist, check_orientation, nnratio};
    k_proj_assign<<<1, 256, 0, s>>>(d_taken, d_valid, d_blocks, d_list, d_count, d_entry, d_match, d_feature_match,
                                   d_result, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !check_orientation || ns <= 0)
        return e;
    const float factor = check_orientation == 2 ? 1.0f / (float)PROJ_HISTO : (float)PROJ_HISTO / 360.0f;
    k_proj_finish<<<1, 256, 0, s>>>(d_kp, d_src_angle, d_match, d_feature_match, d_result, ns, factor);
    return hipGetLastError();
}
