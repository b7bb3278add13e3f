// 256-bit Hamming matching: brute-force top-2 per query over a descriptor database, and distances
// of an explicit pair list.  ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1744-1760 of the
// reference) is a popcount of a XOR; the top-2 bookkeeping is the inner loop of SearchByBoW
// (src/ORBmatcher.cc:217-242): best and second start at 256, strict '<', first index wins.
#include <algorithm>
#include <cstdlib>

#include "orbx_internal.h"

// A (distance, row) pair packed as distance<<16 | row-in-chunk orders exactly like the reference's
// updates: a smaller distance wins, and among equal distances the earlier row.
#define TOP2_INIT ((256u << 16) | 0xffffu)

// Each lane owns one query (8 dwords in registers); the database rows of the chunk are the same for
// every lane, so they arrive through scalar loads and feed v_xor / v_bcnt (which accumulates) directly:
// 16 vector instructions per pair plus 3 for the packed top-2 update.
__global__ __launch_bounds__(256) void k_top2_partial(const uint32_t *__restrict__ q, int nq,
                                                      const uint32_t *__restrict__ db, long long ndb, int chunk_rows,
                                                      uint2 *__restrict__ partial)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        a[i] = qi < nq ? q[(size_t)qi * 8 + i] : 0u;
    const long long row0 = (long long)blockIdx.y * chunk_rows;
    const int rows = (int)min((long long)chunk_rows, ndb - row0);
    const uint32_t *base = db + (size_t)row0 * 8;
    uint32_t k1 = TOP2_INIT, k2 = TOP2_INIT;
    int j = 0;
    for (; j + 4 <= rows; j += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t *r = base + (size_t)(j + u) * 8;
            uint32_t d = 0;
#pragma unroll
            for (int i = 0; i < 8; i++)
                d += __popc(a[i] ^ r[i]);
            const uint32_t key = (d << 16) | (uint32_t)(j + u);
            const uint32_t lo = min(k1, key);
            k2 = min(k2, max(k1, key));
            k1 = lo;
        }
    }
    for (; j < rows; j++) {
        const uint32_t *r = base + (size_t)j * 8;
        uint32_t d = 0;
#pragma unroll
        for (int i = 0; i < 8; i++)
            d += __popc(a[i] ^ r[i]);
        const uint32_t key = (d << 16) | (uint32_t)j;
        const uint32_t lo = min(k1, key);
        k2 = min(k2, max(k1, key));
        k1 = lo;
    }
    if (qi < nq)
        partial[(size_t)blockIdx.y * nq + qi] = make_uint2(k1, k2);
}

// One wave per query: the lanes fold a strided subset of the chunks each, then a butterfly merges
// the 64 partial top-2 pairs.  Keys are distance<<32 | database row, so equal distances resolve to
// the lower row, as the strict '<' of a single pass over the rows does.
__device__ __forceinline__ void top2_insert(unsigned long long &k1, unsigned long long &k2, unsigned long long v)
{
    const unsigned long long hi = k1 > v ? k1 : v;
    k1 = k1 < v ? k1 : v;
    k2 = k2 < hi ? k2 : hi;
}

__global__ __launch_bounds__(256) void k_top2_merge(const uint2 *__restrict__ partial, int nq, int nchunks,
                                                    int chunk_rows, long long idx_base, int32_t *__restrict__ best,
                                                    uint16_t *__restrict__ d1, uint16_t *__restrict__ d2)
{
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (qi >= nq)
        return;
    const unsigned long long init = (256ull << 32) | 0xffffffffull;
    unsigned long long k1 = init, k2 = init;
    for (int c = lane; c < nchunks; c += 64) {
        const uint2 p = partial[(size_t)c * nq + qi];
        const unsigned long long row0 = (unsigned long long)c * (unsigned long long)chunk_rows;
        top2_insert(k1, k2, ((unsigned long long)(p.x >> 16) << 32) | (row0 + (p.x & 0xffff)));
        top2_insert(k1, k2, ((unsigned long long)(p.y >> 16) << 32) | (row0 + (p.y & 0xffff)));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o1 = __shfl_xor(k1, d, 64), o2 = __shfl_xor(k2, d, 64);
        top2_insert(k1, k2, o1);
        k2 = k2 < o2 ? k2 : o2;
    }
    if (lane == 0) {
        const int b1 = (int)(k1 >> 32), b2 = (int)(k2 >> 32);
        best[qi] = b1 >= 256 ? -1 : (int32_t)((long long)(k1 & 0xffffffffull) + idx_base);
        d1[qi] = (uint16_t)b1;
        d2[qi] = (uint16_t)(b2 > 256 ? 256 : b2);
    }
}

hipError_t orbx_launch_top2(const uint8_t *d_q, int nq, const uint8_t *d_db, int64_t ndb, int64_t idx_base,
                            uint32_t *d_partial, int chunk_rows, int nchunks, int32_t *d_best, uint16_t *d_d1,
                            uint16_t *d_d2, hipStream_t s)
{
    const int qblocks = (nq + 255) / 256;
    if (nchunks > 0) {
        k_top2_partial<<<dim3(qblocks, nchunks), 256, 0, s>>>(reinterpret_cast<const uint32_t *>(d_q), nq,
                                                              reinterpret_cast<const uint32_t *>(d_db), (long long)ndb,
                                                              chunk_rows, reinterpret_cast<uint2 *>(d_partial));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
    }
    k_top2_merge<<<(nq + 3) / 4, 256, 0, s>>>(reinterpret_cast<const uint2 *>(d_partial), nq, nchunks, chunk_rows,
                                         (long long)idx_base, d_best, d_d1, d_d2);
    return hipGetLastError();
}

// distances of an explicit (row of A, row of B) pair list
__global__ __launch_bounds__(256) void k_pair_dist(const uint4 *__restrict__ da, const uint4 *__restrict__ dbs,
                                                   const int2 *__restrict__ pairs, int n, uint16_t *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const int2 p = pairs[i];
    const uint4 a0 = da[(size_t)p.x * 2], a1 = da[(size_t)p.x * 2 + 1];
    const uint4 b0 = dbs[(size_t)p.y * 2], b1 = dbs[(size_t)p.y * 2 + 1];
    const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                  __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
    out[i] = (uint16_t)d;
}

hipError_t orbx_launch_pair_dist(const uint8_t *d_a, const uint8_t *d_b, const int *d_pairs, int n, uint16_t *d_out,
                                 hipStream_t s)
{
    if (n <= 0)
        return hipSuccess;
    k_pair_dist<<<(n + 255) / 256, 256, 0, s>>>(reinterpret_cast<const uint4 *>(d_a),
                                                reinterpret_cast<const uint4 *>(d_b),
                                                reinterpret_cast<const int2 *>(d_pairs), n, d_out);
    return hipGetLastError();
}

// ---- MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:278-318) for many map points ----
// One wave per map point.  Segment p holds the n descriptors observed for the point; lane i owns row i
// of the distance matrix (rows i, i+64, ... for larger segments): the database side of a distance is
// the same row for every lane, so it arrives by scalar loads.  The median of a row is its element
// (int)(0.5*(n-1)) in sorted order = the smallest v with  #(d <= v) > that index, found by a 9-step
// binary search over the possible distances 0..256.  Segments of up to 64 rows keep a lane's row of
// distances in registers; larger ones recompute the distances in every search step.
template <bool SMALL>
__device__ __forceinline__ uint32_t distinctive_row_median(const uint32_t *__restrict__ seg, int n, int i, int k)
{
    uint32_t a[8];
#pragma unroll
    for (int w = 0; w < 8; w++)
        a[w] = seg[(size_t)min(i, n - 1) * 8 + w];
    int lo = 0, hi = 256;  // the answer lies in [lo, hi]
    if constexpr (SMALL) {
        uint32_t d[64];
#pragma unroll
        for (int j = 0; j < 64; j++) {
            d[j] = 0xffffu;
            if (j < n) {  // uniform
                const uint32_t *r = seg + (size_t)j * 8;
                uint32_t s = 0;
#pragma unroll
                for (int w = 0; w < 8; w++)
                    s += __popc(a[w] ^ r[w]);
                d[j] = s;
            }
        }
#pragma unroll 1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
#pragma unroll
            for (int j = 0; j < 64; j++)
                c += d[j] <= (uint32_t)mid;
            if (c > k)
                hi = mid;
            else
                lo = mid + 1;
        }
    } else {
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
            for (int j = 0; j < n; j++) {
                const uint32_t *r = seg + (size_t)j * 8;
                uint32_t s = 0;
#pragma unroll
                for (int w = 0; w < 8; w++)
                    s += __popc(a[w] ^ r[w]);
                c += s <= (uint32_t)mid;
            }
            if (c > k)
                hi = mid;
            else
                lo = mid + 1;
        }
    }
    return (uint32_t)lo;
}


__global__ __launch_bounds__(64) void k_distinctive(const uint32_t *__restrict__ desc, const int *__restrict__ seg_start,
                                                    int npoints, int32_t *__restrict__ best_idx,
                                                    uint32_t *__restrict__ best_desc)
{
    const int p = blockIdx.x, lane = threadIdx.x;
    const int s0 = seg_start[p], n = seg_start[p + 1] - s0;
    if (n <= 0) {
        if (lane == 0)
            best_idx[p] = -1;
        return;
    }
    const uint32_t *seg = desc + (size_t)s0 * 8;
    const int k = (int)(0.5 * (n - 1));
    uint32_t best = 0xffffffffu;  // median << 20 | row: the first row with the least median wins
    if (n <= 64) {
        const uint32_t m = distinctive_row_median<true>(seg, n, lane, k);
        if (lane < n)
            best = (m << 20) | (uint32_t)lane;
    } else {
        for (int i0 = 0; i0 < n; i0 += 64) {  // uniform trips
            const int i = i0 + lane;
            const uint32_t m = distinctive_row_median<false>(seg, n, i, k);
            if (i < n)
                best = min(best, (m << 20) | (uint32_t)i);
        }
    }
    best = orbx_wave_min_u32(best);
    const int bi = (int)(best & 0xfffffu);
    if (lane == 0)
        best_idx[p] = bi;
    if (best_desc && lane < 8)
        best_desc[(size_t)p * 8 + lane] = seg[(size_t)bi * 8 + lane];
}

hipError_t orbx_launch_distinctive(const uint8_t *d_desc, const int *d_seg_start, int npoints, int32_t *d_best_idx,
                                   uint8_t *d_best_desc, hipStream_t s)
{
    if (npoints <= 0)
        return hipSuccess;
    k_distinctive<<<npoints, 64, 0, s>>>(reinterpret_cast<const uint32_t *>(d_desc), d_seg_start, npoints, d_best_idx,
                                         reinterpret_cast<uint32_t *>(d_best_desc));
    return hipGetLastError();
}

// ---- TemplatedVocabulary<FORB>::transform of single features ----
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1259 of the reference; FORB::distance,
// Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101, is a popcount of a XOR.)  One thread per feature descends the
// flat tree: at every node to the child with the smallest distance, the first child on ties, until a
// leaf; the node passed at level depth - levelsup is the feature-vector node.  The distances of one
// node's children are independent loads; the levels are a dependent chain (depth of the tree).
__global__ __launch_bounds__(256) void k_voc_transform(const uint32_t *__restrict__ node_desc,
                                                       const int *__restrict__ child_start,
                                                       const int *__restrict__ children,
                                                       const int *__restrict__ word_id,
                                                       const double *__restrict__ weight, int depth,
                                                       const uint32_t *__restrict__ feat, int n, int levelsup,
                                                       int *__restrict__ out_word, int *__restrict__ out_node,
                                                       double *__restrict__ out_weight)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n)
        return;
    uint32_t a[8];
#pragma unroll
    for (int w = 0; w < 8; w++)
        a[w] = feat[(size_t)f * 8 + w];
    const int nid_level = depth - levelsup;
    int nid = 0, final_id = 0, level = 0;
    int c0 = child_start[0], c1 = child_start[1];
    do {  // the host checked that children have larger ids than their parent: the descent ends
        ++level;
        int best_id = children[c0];
        uint32_t best_d = 0xffffffffu;
        for (int c = c0; c < c1; c++) {
            const int id = children[c];
            const uint32_t *r = node_desc + (size_t)id * 8;
            uint32_t d = 0;
#pragma unroll
            for (int w = 0; w < 8; w++)
                d += __popc(a[w] ^ r[w]);
            if (d < best_d) {
                best_d = d;
                best_id = id;
            }
        }
        final_id = best_id;
        if (level == nid_level)
            nid = final_id;
        c0 = child_start[final_id], c1 = child_start[final_id + 1];
    } while (c0 != c1);
    out_word[f] = word_id[final_id];
    out_node[f] = nid;
    out_weight[f] = weight[final_id];
}

hipError_t orbx_launch_voc_transform(const uint8_t *d_node_desc, const int *d_child_start, const int *d_children,
                                     const int *d_word_id, const double *d_weight, int depth, const uint8_t *d_feat,
                                     int n, int levelsup, int *d_word, int *d_node, double *d_wt, hipStream_t s)
{
    if (n <= 0)
        return hipSuccess;
    k_voc_transform<<<(n + 255) / 256, 256, 0, s>>>(reinterpret_cast<const uint32_t *>(d_node_desc), d_child_start,
                                                    d_children, d_word_id, d_weight, depth,
                                                    reinterpret_cast<const uint32_t *>(d_feat), n, levelsup, d_word,
                                                    d_node, d_wt);
    return hipGetLastError();
}

// ---- brute-force top-2 on the matrix cores ----
// With bits mapped to +-1, the dot product of two 256-bit descriptors is 256 - 2 * distance, so the
// distances of 32 database rows against 32 queries are one 32x32x256 matrix product.  CDNA4's fastest
// exact format for it is FP4 (E2M1: +1 = 0x2, -1 = 0xa): four v_mfma_scale_f32_32x32x64_f8f6f4, 1.74 times
// the rate of the i8 products (tools/probe_mfma_f4.hip, tools/ubench_mfma_i8.hip).  Products and sums
// are small integers, exact in the f32 accumulators.  The block scales (2^7 on either side) multiply
// every product by 16384, which leaves room below a result (dots differ by at least 2, i.e. 32768) for a
// row code that the accumulator input adds for free: results leave the matrix core as sortable keys
//     key = 16384 * (dot + 256) + rowcode,   rowcode = 31 - (row inside the tile) + 32 * (tiles since),
// larger key = smaller distance, and among equal distances the earlier row.  Keys are non-negative
// floats below 2^24, so their bit patterns order like integers and the bookkeeping runs on v_max3_i32 /
// v_med3_i32; before a new tile the two running keys move up by 32 (one float addition), which keeps
// earlier rows ahead.  Every lane holds one query (the column) and 16 rows of a tile.
// A workgroup expands each tile of database rows from bits to FP4 once, into LDS in operand order, for
// its four waves (32 * QT queries each, held as operands in registers for the whole sweep).
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define MF_SCALE 134                      // E8M0 block scale of both operands: 2^(134-127) = 128
#define MF_UNIT 16384                     // 128 * 128: key step per unit of dot
#define MF_STEP (2 * MF_UNIT)             // 32768: key step between neighbouring distances
#define MF_BIAS ((float)(MF_UNIT * 256))  // accumulator input that makes every key non-negative
#define MF_INVALID (-268435456.0f)        // accumulator input of rows behind the end of the database
#define MF_MAX_CHUNK 32736                // rows per chunk: the row code must stay below MF_STEP
#define MF_QT 8                           // 32-query tiles per wave
#define MF_QPB (4 * 32 * MF_QT)           // queries per workgroup of four waves

// median of three; written in the form the compiler selects as one v_med3_i32, so that it also places
// the wait states a vector read of a fresh matrix-core result needs (inline assembly would hide the read)
__device__ __forceinline__ int mf_med3(int a, int b, int c)
{
    return max(min(a, b), min(max(a, b), c));
}

// max of three as one instruction.  The compiler does not see the reads of inline assembly when it
// places the wait states after a matrix-core result, so the statement also takes a value computed
// from the same results by compiler-visible code, which orders it behind that code.
__device__ __forceinline__ int mf_max3_after(int a, int b, int c, int after)
{
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(after));
    return r;
}

// 8 descriptor bits -> 8 FP4 values in a dword, bit i in nibble i: 1 -> 0x2 (+1.0), 0 -> 0xa (-1.0)
__device__ __forceinline__ uint32_t mf_expand8(uint32_t b)
{
    uint32_t x = (b | (b << 12)) & 0x000f000fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return 0xaaaaaaaau ^ (x << 3);
}

// the 128-bit FP4 operand as the 256-bit value the builtin takes (an FP4 operand occupies four registers)
__device__ __forceinline__ v8i mf_operand(const v4i &x)
{
    return v8i{x[0], x[1], x[2], x[3], 0, 0, 0, 0};
}

template <int QT>  // 32-query tiles per wave
__global__ __launch_bounds__(256, 2) void k_top2_mfma(const uint32_t *__restrict__ q, int nq,
                                                   const uint32_t *__restrict__ db, long long ndb, int chunk_rows,
                                                   uint2 *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) v4i tile[2][4 * 64];  // [buffer][k-step * 64 + operand lane]
    __shared__ uint32_t lut8[256];                                // 8 bits -> 8 FP4 values
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    // Operand layout of the 32x32x64 product (tools/probe_mfma_f4.hip): lane = (row or column lane & 31,
    // k-half lane >> 5); the lane's 32 values of k sit in nibbles 0..31 of its four registers.  K-step s
    // covers descriptor bits 64 s .. 64 s + 63, so a lane's operand is dword 2 s + half of the descriptor.
    const int qbase = (blockIdx.x * 4 + wv) * (32 * QT);
    v4i bq[QT][4];
#pragma unroll
    for (int t = 0; t < QT; t++) {
        const int qi = qbase + 32 * t + col;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint32_t w = qi < nq ? q[(size_t)qi * 8 + 2 * s + half] : 0u;
#pragma unroll
            for (int j = 0; j < 4; j++)
                bq[t][s][j] = (int)mf_expand8((w >> (8 * j)) & 0xffu);
        }
    }
    const bool wave_active = qbase < nq;
    const long long row0 = (long long)blockIdx.y * chunk_rows;
    const int rows = (int)min((long long)chunk_rows, ndb - row0);
    const int ntiles = (rows + 31) / 32;
    // accumulator input of a tile: register reg of this lane is row (reg&3) + 8*(reg>>2) + 4*half
    v16f cfull;
#pragma unroll
    for (int reg = 0; reg < 16; reg++)
        cfull[reg] = MF_BIAS + (float)(31 - ((reg & 3) + 8 * (reg >> 2) + 4 * half));
    int k1[QT], k2[QT];  // bit patterns of the two running keys; 0 = no row yet
#pragma unroll
    for (int t = 0; t < QT; t++)
        k1[t] = k2[t] = 0;

    // expansion of one tile of database rows: thread = (row tid & 31, dword tid >> 5), eight bits at a
    // time through a 256-entry table.  While tile t is multiplied, tile t+2 is expanded into the buffer
    // tile t came from and its dword for tile t+3 is requested, and the operands of tile t+1 are read
    // behind the last products that use those of tile t: no step waits for memory.
    lut8[tid] = mf_expand8((uint32_t)tid);
    __syncthreads();
    const int xr = tid & 31, xs = tid >> 5;
    const int xslot = (xs >> 1) * 64 + (xs & 1) * 32 + xr;  // k-step, k-half, row
    // byte offset of this thread's dword of tile t inside the chunk; rows behind the end of the chunk
    // read its last row: whatever they hold never wins (MF_INVALID)
    const char *chunk = reinterpret_cast<const char *>(db + (size_t)row0 * 8);
    const uint32_t off_last = (uint32_t)((rows - 1) * 8 + xs) * 4u;
    auto load_word = [&](int t) -> uint32_t {
        const uint32_t off = min((uint32_t)((32 * t + xr) * 8 + xs) * 4u, off_last);
        return *reinterpret_cast<const uint32_t *>(chunk + off);
    };
    auto expand_word = [&](uint32_t w, int buf) {
        v4i x;
        x[0] = (int)lut8[w & 0xffu], x[1] = (int)lut8[(w >> 8) & 0xffu];
        x[2] = (int)lut8[(w >> 16) & 0xffu], x[3] = (int)lut8[w >> 24];
        tile[buf][xslot] = x;
    };
    expand_word(load_word(0), 0);
    expand_word(load_word(1), 1);
    uint32_t w_next = load_word(2);
    __syncthreads();
    v4i a[4];
#pragma unroll
    for (int s = 0; s < 4; s++)
        a[s] = tile[0][s * 64 + lane];
    __syncthreads();  // buffer 0 is rewritten in the first step

    // One tile against the wave's QT x 32 queries as one basic block of 4 * QT slots, one product each,
    // in exactly this order of issue (nothing moves across a slot's end).  Beside the products of
    // query tile u the vector ALU folds the results of the query tile before into the running top-2
    // (those of the last one in the next step), the database tile after next is expanded (table reads
    // behind the first products, the write one query tile later, the request for the dword of the tile
    // after that in the third), and each operand of the next database tile is read behind the last
    // product that uses the register it lands in.  cin is the accumulator input (bias and row codes;
    // MF_INVALID behind the end).
    // The fold takes 16 results two at a time: with x the running maximum, the second largest of
    // everything seen is the largest of the medians med3(x, a, b) and the old second (which never
    // exceeds x): 20 operations for 16 results.  Earlier rows stay ahead at equal distance because
    // the running keys move up by one tile of row codes first.
    v16f pend;
#pragma unroll
    for (int reg = 0; reg < 16; reg++)
        pend[reg] = 0.f;
    int fx = 0, fy = 0, fm = 0;
    auto fold_piece = [&](int i, int u, const v16f &r) {
        if (i == 0) {
            fx = __float_as_int(__int_as_float(k1[u]) + 32.f);
            fy = __float_as_int(__int_as_float(k2[u]) + 32.f);
        }
        const int ra = __float_as_int(r[2 * i]), rb = __float_as_int(r[2 * i + 1]);
        const int m = mf_med3(fx, ra, rb);
        fx = mf_max3_after(fx, ra, rb, m);
        if (i & 1)
            fy = max(max(fy, fm), m);
        else
            fm = m;
        if (i == 7)
            k1[u] = fx, k2[u] = fy;
    };
    auto step = [&](int t, const v16f &cin) {
        const int buf = t & 1;
        v16f acc[QT];
        uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
#pragma unroll
        for (int u = 0; u < QT; u++) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mf_operand(a[s]), mf_operand(bq[u][s]),
                                                                         s ? acc[u] : cin, 4, 4, 0, MF_SCALE, 0, MF_SCALE);
                // results are read one slot after the product that wrote them: no wait for the matrix pipeline
#pragma unroll
                for (int i = (s == 0 ? 8 : 3 * (s - 1)); i < (s == 3 ? 8 : 3 * s); i++) {
                    if (u == 0)
                        fold_piece(i, QT - 1, pend);
                    else
                        fold_piece(i, u - 1, acc[u - 1]);
                }
                if (u == 0 && s == 1) {
                    e0 = lut8[w_next & 0xffu], e1 = lut8[(w_next >> 8) & 0xffu];
                    e2 = lut8[(w_next >> 16) & 0xffu], e3 = lut8[w_next >> 24];
                }
                if (u == 1 && s == 1)
                    tile[buf][xslot] = v4i{(int)e0, (int)e1, (int)e2, (int)e3};
                if (u == 2 && s == 0)
                    w_next = load_word(t + 3);
                if (u == QT - 1)
                    a[s] = tile[buf ^ 1][s * 64 + lane];  // tile t+1, complete since the last barrier
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        pend = acc[QT - 1];
        __syncthreads();
    };
    const int nfull = rows / 32;
    for (int t = 0; t < nfull; t++)
        step(t, cfull);
    if (nfull < ntiles) {
        // the last, partial tile: rows behind the end enter with a code that never wins
#pragma unroll
        for (int reg = 0; reg < 16; reg++)
            if (32 * nfull + (reg & 3) + 8 * (reg >> 2) + 4 * half >= rows)
                cfull[reg] = MF_INVALID;
        step(nfull, cfull);
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
        fold_piece(i, QT - 1, pend);
    if (!wave_active)
        return;
    // the two lanes of a query (rows 4*half + ...) merge; keys of different rows are distinct
#pragma unroll
    for (int t = 0; t < QT; t++) {
        const int o1 = __shfl_xor(k1[t], 32, 64), o2 = __shfl_xor(k2[t], 32, 64);
        const int hi = max(k1[t], o1), lo = min(k1[t], o1);
        const int second = max(lo, max(k2[t], o2));
        const int qi = qbase + 32 * t + col;
        if (half == 0 && qi < nq) {
            // key = MF_STEP * (256 - distance) + rowcode; row = 32*(ntiles-1) + 31 - rowcode
            auto decode = [&](int bits) -> uint32_t {
                if (bits < __float_as_int((float)MF_STEP))  // distance 256 (never a best match), no row, or invalid
                    return TOP2_INIT;
                const uint32_t v = (uint32_t)__int_as_float(bits);
                const uint32_t u = v / MF_STEP, rc = v - u * MF_STEP;
                return ((256u - u) << 16) | (uint32_t)(32 * (ntiles - 1) + 31 - (int)rc);
            };
            partial[(size_t)blockIdx.y * nq + qi] = make_uint2(decode(hi), decode(second));
        }
    }
}

// rows per chunk of the matrix-core kernel: whole tiles, at most MF_MAX_CHUNK, as many chunks as give
// about 512 workgroups (two per CU are resident: the kernel holds 225 registers per lane)
void orbx_top2_mfma_chunks(int nq, int64_t ndb, int *chunk_rows, int *nchunks)
{
    const int qblocks = (nq + MF_QPB - 1) / MF_QPB;
    const char *env = getenv("ORBX_TOP2_BLOCKS");  // tuning runs only
    const int target = env ? atoi(env) : 512;
    const int want = std::max(1, target / qblocks);
    int64_t rows = (ndb + want - 1) / want;
    rows = std::max<int64_t>(512, std::min<int64_t>(rows, MF_MAX_CHUNK));
    rows = (rows + 31) / 32 * 32;
    *chunk_rows = (int)rows;
    *nchunks = (int)((ndb + rows - 1) / rows);
}

hipError_t orbx_launch_top2_mfma(const uint8_t *d_q, int nq, const uint8_t *d_db, int64_t ndb, int64_t idx_base,
                                 uint32_t *d_partial, int chunk_rows, int nchunks, int32_t *d_best, uint16_t *d_d1,
                                 uint16_t *d_d2, hipStream_t s)
{
    if (nchunks > 0) {
        k_top2_mfma<MF_QT><<<dim3((nq + MF_QPB - 1) / MF_QPB, nchunks), 256, 0, s>>>(reinterpret_cast<const uint32_t *>(d_q), nq,
                                                                    reinterpret_cast<const uint32_t *>(d_db),
                                                                    (long long)ndb, chunk_rows,
                                                                    reinterpret_cast<uint2 *>(d_partial));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
    }
    k_top2_merge<<<(nq + 3) / 4, 256, 0, s>>>(reinterpret_cast<const uint2 *>(d_partial), nq, nchunks, chunk_rows,
                                              (long long)idx_base, d_best, d_d1, d_d2);
    return hipGetLastError();
}

// ---- merge of per-shard results on the device ----
// Each shard's result block is what orbx_match_top2_device wrote for it into one buffer of 8 bytes per
// query: nq best indices (int32), then nq best distances, then nq second distances (uint16 each); the
// blocks of all shards lie shard_stride bytes apart in ascending index order (the layout an all-gather
// of such buffers produces).  One thread per query folds the shards in order with the single-pass rules
// (strict '<': the lower shard wins ties; the second best is the least of the other firsts and all
// seconds), exactly as the host function orbx_top2_merge does.
__global__ __launch_bounds__(256) void k_top2_merge_shards(const uint8_t *__restrict__ blocks, int nshards, int nq,
                                                           long long shard_stride, int32_t *__restrict__ best,
                                                           uint16_t *__restrict__ d1, uint16_t *__restrict__ d2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq)
        return;
    int b1 = 256, b2 = 256;
    int32_t bi = -1;
    for (int s = 0; s < nshards; s++) {
        const uint8_t *blk = blocks + (size_t)s * shard_stride;
        const int a = reinterpret_cast<const uint16_t *>(blk + (size_t)4 * nq)[i];
        const int b = reinterpret_cast<const uint16_t *>(blk + (size_t)6 * nq)[i];
        if (a < b1) {
            b2 = b1;
            b1 = a;
            bi = reinterpret_cast<const int32_t *>(blk)[i];
        } else if (a < b2)
            b2 = a;
        if (b < b2)
            b2 = b;
    }
    best[i] = bi;
    d1[i] = (uint16_t)b1;
    d2[i] = (uint16_t)b2;
}

hipError_t orbx_launch_top2_merge_shards(const uint8_t *d_blocks, int nshards, int nq, int64_t shard_stride,
                                         int32_t *d_best, uint16_t *d_d1, uint16_t *d_d2, hipStream_t s)
{
    k_top2_merge_shards<<<(nq + 255) / 256, 256, 0, s>>>(d_blocks, nshards, nq, (long long)shard_stride, d_best, d_d1,
                                                          d_d2);
    return hipGetLastError();
}

// ---- brute-force top-2 for callers that cannot allocate ----
// The per-chunk partial results of the two kernels above normally live in a buffer the matcher handle
// allocates and grows.  This entry point keeps them in a file-scope device array instead, so a call
// makes no hipMalloc / hipFree at all (code under stream capture, serving loops, harnesses that forbid
// allocation).  Every GPU the library runs on holds its own copy of the array (8 MB) at its own address
// until the process ends.  `scratch` below is only the host-side stand-in of that array: host code never
// passes it anywhere; the device address is asked for once per device with hipGetSymbolAddress and kept
// in g_scratch_ptr.
#define NOALLOC_ENTRIES (1u << 20)  // uint2 entries: chunks * queries of one call must fit
#define NOALLOC_MAX_DEVICES 64
static __device__ __align__(256) uint2 scratch[NOALLOC_ENTRIES];

#include <mutex>
static std::mutex g_scratch_mutex;
static uint2 *g_scratch_ptr[NOALLOC_MAX_DEVICES];  // per device; null until its first call

// device address of this device's copy of the array; `device` must be current
static hipError_t scratch_address(int device, uint2 **out)
{
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess)
            return e;
        g_scratch_ptr[device] = static_cast<uint2 *>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

extern "C" int orbx_match_top2_device_noalloc(int device, void *hip_stream, const uint8_t *d_queries, int nq,
                                              const uint8_t *d_db, int64_t ndb, int64_t idx_base,
                                              int32_t *d_best_idx, uint16_t *d_d1, uint16_t *d_d2)
{
    if (!d_queries || nq < 1 || ndb < 0 || (ndb > 0 && !d_db) || !d_best_idx || !d_d1 || !d_d2 || device < 0 ||
        device >= NOALLOC_MAX_DEVICES || idx_base < 0 || idx_base + ndb > INT32_MAX)
        return ORBX_E_ARG;
    if ((size_t)nq > NOALLOC_ENTRIES)
        return ORBX_E_CAPACITY;
    if (hipSetDevice(device) != hipSuccess)
        return ORBX_E_NODEVICE;
    uint2 *partial = nullptr;
    hipError_t e = scratch_address(device, &partial);
    if (e != hipSuccess) {
        orbx_note_hip_error(e);
        return ORBX_E_HIP;
    }
    // the same kernels and the same choice between them as orbx_match_top2_device; the database is cut
    // into fewer, longer chunks when the usual cut would not fit the array (the merge is exact for any cut)
    const bool mfma = nq >= 32;
    const int max_rows = mfma ? MF_MAX_CHUNK / 32 * 32 : 32768, unit = mfma ? 32 : 4;
    int chunk_rows = 256, nchunks = 0;
    if (ndb > 0) {
        if (mfma)
            orbx_top2_mfma_chunks(nq, ndb, &chunk_rows, &nchunks);
        else {
            const int qblocks = (nq + 255) / 256, want = std::max(1, 4096 / qblocks);
            int64_t rows = std::max<int64_t>(256, std::min<int64_t>((ndb + want - 1) / want, 32768));
            chunk_rows = (int)((rows + 3) / 4 * 4);
            nchunks = (int)((ndb + chunk_rows - 1) / chunk_rows);
        }
        const int64_t fit = (int64_t)(NOALLOC_ENTRIES / (unsigned)nq);  // chunks the array has room for (>= 1)
        if (nchunks > fit) {
            int64_t rows = (ndb + fit - 1) / fit;
            rows = (rows + unit - 1) / unit * unit;
            if (rows > max_rows)
                return ORBX_E_CAPACITY;  // nq * ceil(ndb / longest chunk) partial results exceed the array
            chunk_rows = (int)rows;
            nchunks = (int)((ndb + rows - 1) / rows);
        }
    }
    // Nothing in the array has to start at zero: the first kernel writes entry (chunk, query) for every
    // chunk and query of this call, and the merge reads exactly those.
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    e = mfma ? orbx_launch_top2_mfma(d_queries, nq, d_db, ndb, idx_base, reinterpret_cast<uint32_t *>(partial), chunk_rows,
                                     nchunks, d_best_idx, d_d1, d_d2, s)
             : orbx_launch_top2(d_queries, nq, d_db, ndb, idx_base, reinterpret_cast<uint32_t *>(partial), chunk_rows,
                                nchunks, d_best_idx, d_d1, d_d2, s);
    if (e != hipSuccess) {
        orbx_note_hip_error(e);
        return ORBX_E_HIP;
    }
    return ORBX_OK;
}
