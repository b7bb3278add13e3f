// Quadtree distribution of the FAST candidates of one pyramid level
// (ORBextractor::DistributeOctTree + ExtractorNode::DivideNode, src/ORBextractor.cc:518-813 of the
// reference), replayed exactly but without its linked list:
//
// The reference keeps the nodes in a std::list.  A round replaces every processed node by its
// non-empty children, pushed to the FRONT in the order n1..n4, and erases the parent.  So after a
// round the list is:  children of the last processed node (n4,n3,n2,n1), ..., children of the first
// processed node, then every unprocessed node in its old order.  That is a pure function of the
// processing order and of the child key counts, so one workgroup computes each round with prefix
// sums over an array that holds the node records in list order:
//   round of the first kind  (:652-711): every node with >1 key is processed, in list order;
//   round of the last kind   (:726-787): the nodes with >1 key are processed in descending
//       (key count, creation order) and the round stops as soon as the list has N nodes.
// The reference sorts on (count, heap address of the node); creation order stands in for the
// address (a node created later counts as larger).  All nodes with >1 key were created in the
// preceding round, so the creation index inside that round is enough.
// Afterwards each node keeps its best-response key, the first such key in candidate order (:791-810).
#include <cstdlib>

#include "orbx_internal.h"

struct OctLevel {
    int cand_base, cand_cap, cell_base, quota, n_ini, node_cap, kp_base, region_h;
    float hx;
};
struct OctArgs {
    int nlevels, cand_per_frame, kp_per_frame, cells_per_frame;
    int t_ini, t_min;  // FAST thresholds clamped to [1,255]
    int level0;        // first level of this launch (the block's unit counts from it)
    int batch;
    OctLevel lv[ORBX_MAX_LEVELS];
};

#ifndef OCT_STAMP
#define OCT_STAMP(id)  // measurement hook of tools/exp_octree.hip
#endif

#define OCT_DROPPED 0xffffu  // node value of a candidate that the per-cell threshold fallback discards

// 16 arrays of `cap` ints carved from dynamic LDS; addresses are computed, not stored, so that
// run-time buffer and quadrant indices cost arithmetic instead of a pointer table in scratch memory
struct OctLds {
    int *base;
    int cap;
    // node records in list order, double buffered (b = 0/1)
    __device__ __forceinline__ int *bx(int b) const { return base + (4 * b + 0) * cap; }
    __device__ __forceinline__ int *by(int b) const { return base + (4 * b + 1) * cap; }
    __device__ __forceinline__ int *cnt(int b) const { return base + (4 * b + 2) * cap; }
    __device__ __forceinline__ int *seq(int b) const { return base + (4 * b + 3) * cap; }
    // child key counts of the node at a list position (q = 0..3)
    __device__ __forceinline__ int *cc(int q) const { return base + (8 + q) * cap; }
    __device__ __forceinline__ int *prank() const { return base + 12 * cap; }  // processing rank or -1
    __device__ __forceinline__ int *order() const { return base + 13 * cap; }  // order[r] = list position
    __device__ __forceinline__ int *nzs() const { return base + 14 * cap; }    // inclusive prefix of non-empty children
    __device__ __forceinline__ int *spos() const { return base + 15 * cap; }   // new position of a surviving node
};
#define OCT_ARRAYS 16
#define OCT_NT 256  // threads of a workgroup (512 shorten nothing and cost a batch 10 % of its throughput)
#define OCT_TAIL_INTS (2 * (OCT_NT / 64) + 4)  // two sets of wave sums of the block scan + the root count, behind the arrays

size_t orbx_octree_lds_bytes(int node_cap) { return ((size_t)OCT_ARRAYS * node_cap + OCT_TAIL_INTS) * sizeof(int); }


// Exclusive prefix of one value per thread over the workgroup; *total = its sum.  wsum holds two sets of
// NT/64 wave sums used in turn (`phase` counts the calls, uniformly): a set is rewritten only after every
// thread has passed the barrier of the call in between, so one barrier per call is enough.
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int *wsum, int &phase, int *total)
{
    const int inc = orbx_wave_scan_add_i32(v);
    const int wv = threadIdx.x >> 6;
    int *ws = wsum + (phase & 1) * (NT / 64);
    phase++;
    if ((threadIdx.x & 63) == 63)
        ws[wv] = inc;
    __syncthreads();
    int base = 0, all = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        const int w = ws[i];
        base += i < wv ? w : 0;
        all += w;
    }
    *total = all;
    return base + inc - v;
}

__device__ __forceinline__ int quadrant_of(uint32_t xy, int bx, int by)
{
    const int x = xy & 0xffff, y = xy >> 16;
    const int ulx = bx & 0xffff, urx = bx >> 16, uly = by & 0xffff, bry = by >> 16;
    const int midx = ulx + ((urx - ulx + 1) >> 1);  // UL.x + ceil((UR.x-UL.x)/2)
    const int midy = uly + ((bry - uly + 1) >> 1);
    return (x < midx ? 0 : 1) + (y < midy ? 0 : 2);  // n1, n2, n3, n4 -> 0..3
}

// Body shared by the batched kernel and the standalone test entry.
// xy/resp/node: the n candidates of this level (coordinates relative to the detection region's corner), in any
// order.  key (may be null: the index is the key) gives the order in which the reference holds them, cell id in
// the bits above 12.  cell_ini (may be null: keep all) = per cell, whether any candidate exceeds iniThFAST: a
// candidate counts iff its score A = resp+1 exceeds t_ini, or its cell has none such and A exceeds t_min
// (src/ORBextractor.cc:868-878).
// out_*: kept keypoints in list order (xy as given, plus `add` on both coordinates); out_idx optional.
// KMAX > 0: every thread keeps its (at most KMAX) candidates in registers, n <= NT*KMAX;
// KMAX == 0: candidates stay in global memory (`node` is the scratch array), any n.
template <int KMAX, int NT>
__device__ __forceinline__ void octree_body(const uint32_t *__restrict__ xy, const uint8_t *__restrict__ resp,
                                            const uint32_t *__restrict__ key, const uint8_t *__restrict__ cell_ini,
                                            int t_ini, int t_min, uint16_t *__restrict__ node, int n, int quota,
                                            int n_ini, float hx, int region_h, int cap, int add,
                                            uint32_t *__restrict__ out_xy, uint8_t *__restrict__ out_resp,
                                            int *__restrict__ out_idx, int *__restrict__ out_n,
                                            int *__restrict__ status)
{
    // all LDS lives in the one dynamic array, so its base alignment is not disturbed
    extern __shared__ __attribute__((aligned(16))) int lds[];
    int *wsum = lds + (size_t)OCT_ARRAYS * cap;
    int &sh_size = wsum[2 * (NT / 64)];
    int phase = 0;  // calls of block_excl_scan so far
    const OctLds L{lds, cap};
    const int tid = threadIdx.x;
    if (n <= 0) {
        if (tid == 0)
            *out_n = 0;
        return;
    }
    const int N = quota;
    OCT_STAMP(0);

    // ---- candidate storage ----
    constexpr int KR = KMAX > 0 ? KMAX : 1;
    uint32_t rxy[KR];
    int rnode[KR];
    unsigned long long rbk[KR];  // response<<32 | ~order key: the maximum is the node's keeper
    // Register form: a thread holds consecutive candidates.  Neighbours in the list are neighbours in the image
    // (the list is cell-major), so they mostly share their node and quadrant, and a thread's LDS atomics
    // collapse into one per run of equal targets.
    const int kper = (n + NT - 1) / NT, kbase = tid * kper;
    const int nmine = max(0, min(kper, n - kbase));
    auto for_keys = [&](auto fn) {
        if constexpr (KMAX > 0) {
#pragma unroll
            for (int j = 0; j < KMAX; j++) {
                if (j < nmine)
                    fn(j, kbase + j);
            }
        } else {
            for (int k = tid; k < n; k += NT)
                fn(0, k);
        }
    };
    auto XY = [&](int j, int k) -> uint32_t {
        if constexpr (KMAX > 0) return rxy[j];
        else return xy[k];
    };
    auto NODE = [&](int j, int k) -> int {
        if constexpr (KMAX > 0) return rnode[j];
        else return node[k];
    };
    auto SET_NODE = [&](int j, int k, int v) {
        if constexpr (KMAX > 0) rnode[j] = v;
        else node[k] = (uint16_t)v;
    };
    auto BK = [&](int j, int k) -> unsigned long long {
        if constexpr (KMAX > 0) return rbk[j];
        else
            return ((unsigned long long)resp[k] << 32) | (unsigned long long)(0xffffffffu - (key ? key[k] : (uint32_t)k));
    };

    // ---- roots (:587-631) ----
    for (int i = tid; i < n_ini; i += NT) {
        const int ulx = (int)(hx * (float)i), urx = (int)(hx * (float)(i + 1));
        L.bx(1)[i] = ulx | (urx << 16);
        L.by(1)[i] = 0 | (region_h << 16);
        L.cnt(1)[i] = 0;
    }
    if constexpr (KMAX > 0) {
        // one round of loads brings everything a thread will ever need of its candidates
        uint8_t rr[KR];
        uint32_t rk[KR];
#pragma unroll
        for (int j = 0; j < KMAX; j++) {
            const int k = kbase + j;
            const bool have = j < nmine;
            rxy[j] = have ? xy[k] : 0u;
            rr[j] = have ? resp[k] : (uint8_t)0;
            rk[j] = (have && key) ? key[k] : (uint32_t)k;
            rnode[j] = OCT_DROPPED;  // slots without a candidate stay dropped
        }
#pragma unroll
        for (int j = 0; j < KMAX; j++)
            rbk[j] = ((unsigned long long)rr[j] << 32) | (unsigned long long)(0xffffffffu - rk[j]);
    }
    __syncthreads();
    for_keys([&](int j, int k) {
        if (cell_ini) {
            const unsigned long long bk = BK(j, k);
            const int A = (int)(bk >> 32) + 1;
            const uint32_t ord = 0xffffffffu - (uint32_t)bk;
            if (!(A > t_ini || (cell_ini[ord >> 12] == 0 && A > t_min))) {
                SET_NODE(j, k, OCT_DROPPED);
                return;
            }
        }
        int r = (int)((float)(XY(j, k) & 0xffff) / hx);
        r = r < n_ini ? r : n_ini - 1;
        SET_NODE(j, k, r);
        if constexpr (KMAX == 0)
            atomicAdd(&L.cnt(1)[r], 1);
    });
    if constexpr (KMAX > 0) {
        int run = OCT_DROPPED, len = 0;
#pragma unroll
        for (int j = 0; j < KMAX; j++) {
            if (rnode[j] != run) {
                if (run != OCT_DROPPED)
                    atomicAdd(&L.cnt(1)[run], len);
                run = rnode[j], len = 0;
            }
            len++;
        }
        if (run != OCT_DROPPED)
            atomicAdd(&L.cnt(1)[run], len);
    }
    __syncthreads();
    if (tid == 0) {
        int s = 0;
        for (int i = 0; i < n_ini; i++) {
            if (L.cnt(1)[i] > 0) {
                L.bx(0)[s] = L.bx(1)[i], L.by(0)[s] = L.by(1)[i], L.cnt(0)[s] = L.cnt(1)[i], L.seq(0)[s] = s;
                L.spos()[i] = s++;
            } else
                L.spos()[i] = 0;
        }
        sh_size = s;
    }
    __syncthreads();
    for_keys([&](int j, int k) {
        const int i = NODE(j, k);
        if (i != OCT_DROPPED)
            SET_NODE(j, k, L.spos()[i]);
    });
    int size = sh_size;
    if (size == 0) {  // every candidate was dropped
        if (tid == 0)
            *out_n = 0;
        return;
    }
    int cur = 0;
    bool last_kind = false;
    __syncthreads();
    OCT_STAMP(1);

    for (int round = 0; round < 64; round++) {
        const int prev_size = size;
        OCT_STAMP(2);
        int *bx = L.bx(cur), *by = L.by(cur), *cnt = L.cnt(cur), *seq = L.seq(cur);
        int *nbx = L.bx(cur ^ 1), *nby = L.by(cur ^ 1), *ncnt = L.cnt(cur ^ 1), *nseq = L.seq(cur ^ 1);
        const int per = (size + NT - 1) / NT;
        const int lo = min(tid * per, size), hi = min(lo + per, size);

        // ---- processing order of the nodes that hold more than one key ----
        // In a round of the last kind the rank of a node is the number of nodes with a larger (count,
        // creation index).  Both fit one word while counts stay below 2^18 (creation indices are below
        // 4 * cap < 2^14): the words of all nodes are laid down first (in nzs, which this round fills later),
        // so that the ranking loop is one broadcast LDS read and one compare per node, unrolled.
        const bool packed_ok = n < (1 << 18);
        int my_e = 0;
        for (int i = lo; i < hi; i++) {
            my_e += cnt[i] >= 2;
            L.cc(0)[i] = L.cc(1)[i] = L.cc(2)[i] = L.cc(3)[i] = 0;
            if (packed_ok)
                L.nzs()[i] = cnt[i] >= 2 ? ((cnt[i] << 14) | seq[i]) : 0;
        }
        if (packed_ok && tid < 3 && size + tid < ((size + 3) & ~3))
            L.nzs()[size + tid] = 0;  // padding of the last group of four
        int m;
        int ebase = block_excl_scan<NT>(my_e, wsum, phase, &m);  // its barriers also publish the packed words
        // what kind of round this is (:716-723): m = the nodes the previous round left with more than one key
        if (!last_kind && round > 0 && size + 3 * m > N)
            last_kind = true;
        const bool packed_rank = last_kind && packed_ok;
        if (!last_kind) {
            for (int i = lo; i < hi; i++)
                L.prank()[i] = cnt[i] >= 2 ? ebase++ : -1;
        } else if (packed_rank) {
            // four words per LDS read (cap is a multiple of 4 and the arrays start 16-byte aligned; the words
            // behind the list's end were cleared above)
            const uint32_t *word = reinterpret_cast<const uint32_t *>(L.nzs());
            const uint4 *word4 = reinterpret_cast<const uint4 *>(L.nzs());
            const int size4 = (size + 3) >> 2;
            for (int i = lo; i < hi; i++) {
                const uint32_t wi = word[i];
                int rk = 0;
#pragma unroll 4
                for (int j = 0; j < size4; j++) {
                    const uint4 w = word4[j];
                    rk += (w.x > wi) + (w.y > wi) + (w.z > wi) + (w.w > wi);
                }
                L.prank()[i] = wi ? rk : -1;
            }
        } else {
            for (int i = lo; i < hi; i++) {
                int rk = -1;
                if (cnt[i] >= 2) {
                    rk = 0;
                    const int ci = cnt[i], si = seq[i];
                    for (int j = 0; j < size; j++) {
                        const int cj = cnt[j];
                        rk += (cj >= 2) && (cj > ci || (cj == ci && seq[j] > si));
                    }
                }
                L.prank()[i] = rk;
            }
        }
        __syncthreads();
        for (int i = lo; i < hi; i++)
            if (L.prank()[i] >= 0)
                L.order()[L.prank()[i]] = i;

        OCT_STAMP(3);
        // ---- key counts of the four children (:551-565) ----
        uint32_t qpack = 0u;  // quadrant of candidate j in bits 2j, 2j+1 (register form)
        if constexpr (KMAX > 0) {
            // all reads first, then all atomics: the reads of different candidates overlap
            int enc[KR];
            qpack = 0u;
#pragma unroll
            for (int j = 0; j < KMAX; j++) {
                const int i = rnode[j];
                const bool live = i != OCT_DROPPED;
                const int ii = live ? i : 0;
                const int c = cnt[ii], b_x = bx[ii], b_y = by[ii];
                const int q = quadrant_of(rxy[j], b_x, b_y);
                qpack |= (uint32_t)q << (2 * j);  // kept for the move at the end of the round
                enc[j] = (live && c >= 2) ? (ii * 4 + q) : -1;
            }
            int run = -1, len = 0;
#pragma unroll
            for (int j = 0; j < KMAX; j++) {
                if (enc[j] != run) {
                    if (run >= 0)
                        atomicAdd(&L.cc(run & 3)[run >> 2], len);
                    run = enc[j], len = 0;
                }
                len++;
            }
            if (run >= 0)
                atomicAdd(&L.cc(run & 3)[run >> 2], len);
        } else {
            for_keys([&](int j, int k) {
                const int i = NODE(j, k);
                if (i != OCT_DROPPED && cnt[i] >= 2)
                    atomicAdd(&L.cc(quadrant_of(XY(j, k), bx[i], by[i]))[i], 1);
            });
        }
        __syncthreads();

        OCT_STAMP(4);
        // ---- how many nodes are processed: all, or up to the one that brings the list to N ----
        const int rper = (m + NT - 1) / NT;
        const int rlo = min(tid * rper, m), rhi = min(rlo + rper, m);
        int my_nz = 0;
        for (int r = rlo; r < rhi; r++) {
            const int i = L.order()[r];
            my_nz += (L.cc(0)[i] > 0) + (L.cc(1)[i] > 0) + (L.cc(2)[i] > 0) + (L.cc(3)[i] > 0);
        }
        int nz_total;
        int run = block_excl_scan<NT>(my_nz, wsum, phase, &nz_total);
        int below = 0;  // processed nodes after which the list is still shorter than N
        for (int r = rlo; r < rhi; r++) {
            const int i = L.order()[r];
            run += (L.cc(0)[i] > 0) + (L.cc(1)[i] > 0) + (L.cc(2)[i] > 0) + (L.cc(3)[i] > 0);
            L.nzs()[r] = run;  // inclusive; list size after r+1 nodes = size + nzs[r] - (r+1)
            below += (size + run - (r + 1)) < N;
        }
        int t = m;
        if (last_kind) {
            int t0;
            block_excl_scan<NT>(below, wsum, phase, &t0);
            t = min(t0 + 1, m);
        }
        __syncthreads();
        const int children = t > 0 ? L.nzs()[t - 1] : 0;

        OCT_STAMP(5);
        // ---- survivors keep their relative order behind the new children ----
        int my_s = 0;
        for (int i = lo; i < hi; i++)
            my_s += !(L.prank()[i] >= 0 && L.prank()[i] < t);
        int nsurv;
        int sbase = block_excl_scan<NT>(my_s, wsum, phase, &nsurv);
        const int new_size = children + nsurv;
        if (new_size > cap) {
            // cannot happen for a correctly sized list; reported instead of writing out of bounds
            if (tid == 0) {
                atomicMax(&status[ORBX_ST_NODE_OVERFLOW], new_size);
                *out_n = 0;
            }
            return;
        }
        for (int i = lo; i < hi; i++) {
            if (!(L.prank()[i] >= 0 && L.prank()[i] < t)) {
                const int p = children + sbase++;
                L.spos()[i] = p;
                nbx[p] = bx[i], nby[p] = by[i], ncnt[p] = cnt[i], nseq[p] = seq[i];
            }
        }
        // ---- children records (:526-548): block of node r starts behind the blocks of r+1..t-1 ----
        for (int r = rlo; r < rhi; r++) {
            if (r >= t)
                break;
            const int i = L.order()[r];
            const int b_x = bx[i], b_y = by[i];
            const int ulx = b_x & 0xffff, urx = b_x >> 16, uly = b_y & 0xffff, bry = b_y >> 16;
            const int midx = ulx + ((urx - ulx + 1) >> 1), midy = uly + ((bry - uly + 1) >> 1);
            const int c[4] = {L.cc(0)[i], L.cc(1)[i], L.cc(2)[i], L.cc(3)[i]};
            const int nz = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0);
            const int blk = children - L.nzs()[r];
            // where this node's keys go: first child position and the mask of non-empty children
            L.spos()[i] = (int)(0x80000000u | ((uint32_t)blk << 4) | (uint32_t)((c[0] > 0) | ((c[1] > 0) << 1) |
                                                                             ((c[2] > 0) << 2) | ((c[3] > 0) << 3)));
            int before = 0;
            for (int q = 0; q < 4; q++) {
                if (c[q] == 0)
                    continue;
                const int p = blk + (nz - 1 - before);  // pushed to the front one after the other
                nbx[p] = (q & 1) ? (midx | (urx << 16)) : (ulx | (midx << 16));
                nby[p] = (q & 2) ? (midy | (bry << 16)) : (uly | (midy << 16));
                ncnt[p] = c[q];
                nseq[p] = L.nzs()[r] - nz + before;  // creation index inside this round
                before++;
            }
        }
        __syncthreads();
        OCT_STAMP(6);
        // ---- move the keys to their new node ----
        // one word per node: a survivor's new position, or (bit 31) the block of a processed node's
        // children, pushed to the front one after the other, so the last non-empty child comes first
        auto destination = [](uint32_t d, int q) -> int {
            const uint32_t mask = d & 15u;
            const int child = (int)((d >> 4) & 0x07ffffffu) + __popc(mask) - 1 - __popc(mask & ((1u << q) - 1u));
            return (d & 0x80000000u) ? child : (int)d;
        };
        if constexpr (KMAX > 0) {
            uint32_t d[KR];
#pragma unroll
            for (int j = 0; j < KMAX; j++)
                d[j] = (uint32_t)L.spos()[rnode[j] != OCT_DROPPED ? rnode[j] : 0];
#pragma unroll
            for (int j = 0; j < KMAX; j++)
                rnode[j] = rnode[j] != OCT_DROPPED ? destination(d[j], (int)((qpack >> (2 * j)) & 3u)) : OCT_DROPPED;
        } else {
            for_keys([&](int j, int k) {
                const int i = NODE(j, k);
                if (i == OCT_DROPPED)
                    return;
                const uint32_t d = (uint32_t)L.spos()[i];
                SET_NODE(j, k, destination(d, (d & 0x80000000u) ? quadrant_of(XY(j, k), bx[i], by[i]) : 0));
            });
        }
        __syncthreads();
        size = new_size;
        cur ^= 1;

        OCT_STAMP(7);
        // ---- finished (:716-718, :784-785)?  The kind of the next round is decided at its start ----
        if (size >= N || size == prev_size)
            break;
    }

    // ---- best key of every node (:791-810): largest response, first in candidate order ----
    __syncthreads();
    OCT_STAMP(8);
    unsigned long long *best = reinterpret_cast<unsigned long long *>(L.cc(0));  // cc[0],cc[1] are adjacent
    for (int i = tid; i < size; i += NT)
        best[i] = 0ull;
    __syncthreads();
    if constexpr (KMAX > 0) {
        int run = OCT_DROPPED;
        unsigned long long top = 0ull;
#pragma unroll
        for (int j = 0; j < KMAX; j++) {
            if (rnode[j] != run) {
                if (run != OCT_DROPPED)
                    atomicMax(&best[run], top);
                run = rnode[j], top = 0ull;
            }
            top = rbk[j] > top ? rbk[j] : top;
        }
        if (run != OCT_DROPPED)
            atomicMax(&best[run], top);
    } else {
        for_keys([&](int j, int k) {
            const int i = NODE(j, k);
            if (i != OCT_DROPPED)
                atomicMax(&best[i], BK(j, k));
        });
    }
    __syncthreads();
    // the winner of a node is the one candidate whose (response, order key) equals the maximum
    for_keys([&](int j, int k) {
        const int i = NODE(j, k);
        if (i == OCT_DROPPED)
            return;
        const unsigned long long bk = BK(j, k);
        if (best[i] == bk) {
            const uint32_t v = XY(j, k);
            out_xy[i] = ((v & 0xffff) + add) | (((v >> 16) + add) << 16);
            out_resp[i] = (uint8_t)(bk >> 32);
            if (out_idx)
                out_idx[i] = (int)k;
        }
    });
    OCT_STAMP(9);
    if (tid == 0)
        *out_n = size;
}

// Candidates per thread held in registers (levels with up to 256 * OCT_KMAX candidates; beyond that they stay
// in global memory)
#ifndef OCT_KMAX
#define OCT_KMAX 16
#endif

template <int NT, int KMAX, typename... Args> __device__ __forceinline__ void octree_dispatch(int n, Args... args)
{
    // n is uniform over the workgroup
    if (n <= NT * KMAX)
        octree_body<KMAX, NT>(args...);
    else
        octree_body<0, NT>(args...);
}

template <int NT, int KMAX>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_octree(const uint32_t *__restrict__ cand_xy,
                                                const uint8_t *__restrict__ cand_resp,
                                                const uint32_t *__restrict__ cand_key,
                                                const uint8_t *__restrict__ cell_ini, uint16_t *__restrict__ cand_node,
                                                const int *__restrict__ ncand, uint32_t *__restrict__ kept_xy,
                                                uint8_t *__restrict__ kept_resp, int *__restrict__ nkept,
                                                int *__restrict__ status, OctArgs a)
{
    // a frame's workgroups run on the XCD whose L2 holds the frame's candidates (written there by the detection kernel)
    const int lvl = ORBX_BLOCK_UNIT + a.level0, f = ORBX_BLOCK_FRAME;
    if (f >= a.batch)
        return;
    const OctLevel &L = a.lv[lvl];
    const size_t cb = (size_t)f * a.cand_per_frame + L.cand_base;
    const size_t kb = (size_t)f * a.kp_per_frame + L.kp_base;
    const int n = min(ncand[f * a.nlevels + lvl], L.cand_cap);  // an overflow is reported by the producer
    octree_dispatch<NT, KMAX>(n, cand_xy + cb, cand_resp + cb, cand_key + cb,
                    cell_ini + (size_t)f * a.cells_per_frame + L.cell_base, a.t_ini, a.t_min, cand_node + cb, n,
                    L.quota, L.n_ini, L.hx, L.region_h, L.node_cap, (int)ORBX_BORDER, kept_xy + kb, kept_resp + kb,
                    (int *)nullptr, nkept + f * a.nlevels + lvl, status);
}

hipError_t orbx_launch_octree(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    OctArgs a;
    a.level0 = 0;
    a.nlevels = g.nlevels, a.cand_per_frame = g.cand_per_frame, a.kp_per_frame = g.kp_per_frame;
    a.cells_per_frame = g.cells_per_frame;
    a.t_ini = min(max(g.ini_th, 1), 255), a.t_min = min(max(g.min_th, 1), 255);
    int cap = 1;
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &L = g.lv[l];
        a.lv[l].cand_base = L.cand_base, a.lv[l].quota = L.quota, a.lv[l].n_ini = L.n_ini;
        a.lv[l].cand_cap = L.cand_cap, a.lv[l].cell_base = L.cell_base;
        a.lv[l].node_cap = L.node_cap, a.lv[l].kp_base = L.kp_base, a.lv[l].region_h = L.max_by - ORBX_BORDER;
        a.lv[l].hx = L.hx;
        cap = L.node_cap > cap ? L.node_cap : cap;
    }
    // every level carves its own node_cap out of an allocation sized for the largest level of its launch
    auto launch = [&](auto kernel, int nt, int l0, int l1) -> hipError_t {
        int c = 1;
        for (int l = l0; l < l1; l++)
            c = max(c, g.lv[l].node_cap);
        const size_t lds = orbx_octree_lds_bytes(c);
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess)
                return e;
        }
        a.level0 = l0;
        a.batch = g.batch;
        kernel<<<orbx_frame_grid(l1 - l0, g.batch), nt, lds, s>>>(b.cand_xy, b.cand_resp, b.cand_key, b.cell_flag, b.cand_node,
                                                       b.ncand, b.kept_xy, b.kept_resp, b.nkept, b.status, a);
        return hipGetLastError();
    };
    (void)cap;
    // Batches: the levels whose arrays need more LDS than a workgroup of the detection kernel holds (20 KB: KITTI
    // levels 0 and 1 with 28 and 23 KB) go in a launch of their own, the others in a second one that reserves only
    // what its largest level needs.  Beside the kernels of the other handles a workgroup of this kernel has to wait
    // for that much free LDS on one compute unit, and the detection kernel frees it 20 KB at a time: with one launch
    // sized for level 0 the kernel takes 0.65 ms instead of 0.105 there.  Two launches are slower alone (0.119 ms) and
    // give 1.5 % more frames/s in the batch; three or more launches give no more.
    int split = 0;
    while (split < g.nlevels && orbx_octree_lds_bytes(g.lv[split].node_cap) > 20 * 1024)
        split++;
    if (g.batch > 4 && split > 0 && split < g.nlevels) {
        const hipError_t e = launch(k_octree<OCT_NT, OCT_KMAX>, OCT_NT, 0, split);
        return e != hipSuccess ? e : launch(k_octree<OCT_NT, OCT_KMAX>, OCT_NT, split, g.nlevels);
    }
    return launch(k_octree<OCT_NT, OCT_KMAX>, OCT_NT, 0, g.nlevels);
}

template <int NT>
__global__ __launch_bounds__(NT) void k_octree_single(const uint32_t *xy, const uint8_t *resp, uint16_t *node, int n,
                                                       int quota, int n_ini, float hx, int region_h, int cap,
                                                       uint32_t *out_xy, uint8_t *out_resp, int *out_idx, int *out_n,
                                                       int *status)
{
    octree_dispatch<NT, OCT_KMAX>(n, xy, resp, (const uint32_t *)nullptr, (const uint8_t *)nullptr, 0, 0, node, n, quota, n_ini, hx,
                    region_h, cap, 0, out_xy, out_resp, out_idx, out_n, status);
}

hipError_t orbx_launch_octree_single(const uint32_t *d_xy, const uint8_t *d_resp, uint16_t *d_node, int n, int region_w,
                                     int region_h, int quota, uint32_t *d_out_xy, uint8_t *d_out_resp, int *d_out_idx,
                                     int *d_nout, int *d_status, int node_cap, hipStream_t s)
{
    const int n_ini = (int)roundf((float)region_w / (float)region_h);
    const float hx = (float)region_w / (float)n_ini;
    const size_t lds = orbx_octree_lds_bytes(node_cap);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree_single<OCT_NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
    }
    k_octree_single<OCT_NT><<<1, OCT_NT, lds, s>>>(d_xy, d_resp, d_node, n, quota, n_ini, hx, region_h, node_cap,
                                                           d_out_xy, d_out_resp, d_out_idx, d_nout, d_status);
    return hipGetLastError();
}
