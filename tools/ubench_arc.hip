// Forms of the FAST arc score on a tile staged in LDS (72 rows of 96 bytes), on a queue of 512 pixels per tile:
//   A  one pixel per lane, sixteen byte reads, both polarities packed in the halves of a register
//   B  two pixels per lane (one per half, each with a known polarity), byte reads
//   C  two pixels per lane, the ring from seven unaligned 4- and 8-byte reads per pixel
//   D  one pixel per lane, both polarities, the ring from seven unaligned reads
// Prints the time per tile visit and checks B, C, D against A.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define P 96
#define ROWS 72
#define NQ 512

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t arcs(const uint32_t *d)
{
    uint32_t m2[8], m4[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
        m2[k] = pk_max(d[2 * k], d[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 8; k++)
        m4[k] = pk_max(m2[k], m2[(k + 1) & 7]);
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t m8 = pk_max(m4[k], m4[(k + 2) & 7]);
        best = pk_min(best, pk_max(m8, pk_min(d[(2 * k + 8) & 15], d[(2 * k + 15) & 15])));
    }
    return best;
}
__device__ __forceinline__ void ring_bytes(const uint8_t *t, uint32_t *d)
{
    const int p = P;
    d[0] = t[3 * p], d[1] = t[3 * p + 1], d[2] = t[2 * p + 2], d[3] = t[p + 3], d[4] = t[3], d[5] = t[-p + 3];
    d[6] = t[-2 * p + 2], d[7] = t[-3 * p + 1], d[8] = t[-3 * p], d[9] = t[-3 * p - 1], d[10] = t[-2 * p - 2];
    d[11] = t[-p - 3], d[12] = t[-3], d[13] = t[p - 3], d[14] = t[2 * p - 2], d[15] = t[3 * p - 1];
}
// the ring as words that hold the wanted byte, and the byte's index in the word
struct Wide {
    uint32_t w[12];  // +3, -3, +2 lo/hi, -2 lo/hi, +1 lo/hi, -1 lo/hi, 0 lo/hi
};
__device__ __forceinline__ uint32_t ld4(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint2 ld8(const uint8_t *p)
{
    uint2 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void ring_wide(const uint8_t *t, Wide &r)
{
    const int p = P;
    r.w[0] = ld4(t + 3 * p - 1);
    r.w[1] = ld4(t - 3 * p - 1);
    uint2 v = ld8(t + 2 * p - 2);
    r.w[2] = v.x, r.w[3] = v.y;
    v = ld8(t - 2 * p - 2);
    r.w[4] = v.x, r.w[5] = v.y;
    v = ld8(t + p - 3);
    r.w[6] = v.x, r.w[7] = v.y;
    v = ld8(t - p - 3);
    r.w[8] = v.x, r.w[9] = v.y;
    v = ld8(t - 3);
    r.w[10] = v.x, r.w[11] = v.y;
}
// ring pixel k: word index and byte
__device__ constexpr int RW[16] = {0, 0, 3, 7, 11, 9, 5, 1, 1, 1, 4, 8, 10, 6, 2, 0};
__device__ constexpr int RB[16] = {1, 2, 0, 2, 2, 2, 0, 2, 1, 0, 0, 0, 0, 0, 0, 0};

template <int V> __global__ __launch_bounds__(256) void k_arc(const uint8_t *tile, const uint16_t *q, const uint8_t *pol,
                                                              uint8_t *out, int reps)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[ROWS * P];
    __shared__ uint8_t sc[NQ];
    __shared__ uint16_t queue[NQ];
    __shared__ uint8_t bright[NQ];
    for (int i = threadIdx.x; i < ROWS * P / 16; i += 256)
        reinterpret_cast<uint4 *>(img)[i] = reinterpret_cast<const uint4 *>(tile)[i];
    for (int i = threadIdx.x; i < NQ; i += 256)
        queue[i] = q[i], bright[i] = pol[i], sc[i] = 0;
    __syncthreads();
    for (int rep = 0; rep < reps; rep++) {
        if (V == 0 || V == 3) {
            for (int q0 = 0; q0 < NQ; q0 += 256) {
                const int qi = (q0 + threadIdx.x + 17 * rep) & (NQ - 1);
                const int e = queue[qi];
                const uint8_t *t = &img[(e >> 7) * P + (e & 127)];
                uint32_t d[16];
                int v;
                if (V == 0) {
                    ring_bytes(t, d);
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        d[k] = (uint32_t)(__mul24((int)d[k], -65535) + 0x00ff0000);
                    v = t[0];
                } else {
                    Wide r;
                    ring_wide(t, r);
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        d[k] = __builtin_amdgcn_perm(0u, r.w[RW[k]], 0x0c000c00u | (uint32_t)RB[k] | ((uint32_t)RB[k] << 16)) ^
                               0x00ff0000u;
                    v = (r.w[10] >> 24) & 0xff;
                }
                const uint32_t best = arcs(d);
                const int dark = v - (int)(best & 0xffff), br = 255 - (int)(best >> 16) - v;
                sc[qi] = (uint8_t)max(max(dark, br), 0);
            }
        } else {
            const int qa = (threadIdx.x + 17 * rep) & (NQ - 1), qb = (qa + 256) & (NQ - 1);
            const int ea = queue[qa], eb = queue[qb];
            const uint8_t *ta = &img[(ea >> 7) * P + (ea & 127)], *tb = &img[(eb >> 7) * P + (eb & 127)];
            const uint32_t m = (bright[qa] ? 0xffu : 0u) | (bright[qb] ? 0xff0000u : 0u);
            uint32_t d[16];
            int va, vb;
            if (V == 1) {
                uint32_t da[16], db[16];
                ring_bytes(ta, da);
                ring_bytes(tb, db);
#pragma unroll
                for (int k = 0; k < 16; k++)
                    d[k] = (da[k] | (db[k] << 16)) ^ m;
                va = ta[0], vb = tb[0];
            } else {
                Wide ra, rb;
                ring_wide(ta, ra);
                ring_wide(tb, rb);
#pragma unroll
                for (int k = 0; k < 16; k++)
                    d[k] = __builtin_amdgcn_perm(rb.w[RW[k]], ra.w[RW[k]],
                                                 0x0c000c00u | (uint32_t)RB[k] | ((uint32_t)(4 + RB[k]) << 16)) ^ m;
                va = (ra.w[10] >> 24) & 0xff, vb = (rb.w[10] >> 24) & 0xff;
            }
            const uint32_t best = arcs(d);
            const int ba = best & 0xffff, bb = best >> 16;
            const int sa = bright[qa] ? 255 - ba - va : va - ba, sb = bright[qb] ? 255 - bb - vb : vb - bb;
            sc[qa] = (uint8_t)max(sa, 0);
            sc[qb] = (uint8_t)max(sb, 0);
        }
        __syncthreads();
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < NQ; i += 256)
            out[i] = sc[i];
}

#define OK(x)                                                        \
    do {                                                             \
        hipError_t e_ = (x);                                         \
        if (e_ != hipSuccess) {                                      \
            printf("%s: %s\n", #x, hipGetErrorString(e_));           \
            return 1;                                                \
        }                                                            \
    } while (0)

int main(int argc, char **argv)
{
    const int order = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<uint8_t> tile(ROWS * P), pol(NQ), ref(NQ);
    std::vector<uint16_t> q(NQ);
    srand(7);
    // a smooth image with noise: scores of every size
    for (int y = 0; y < ROWS; y++)
        for (int x = 0; x < P; x++)
            tile[y * P + x] = (uint8_t)(((x / 5 + y / 7) & 1 ? 60 : 170) + rand() % 40);
    static const int dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    static const int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    // about 11 % of the pixels of the tile and its ring pass the compass test; the queue holds them in the
    // order given by the first argument
    struct Px { int r, c; long key; };
    std::vector<Px> px;
    for (int r = 3; r < 69; r++)
        for (int c = 15; c < 81; c++)
            if (rand() % 1000 < 118)
                px.push_back(Px{r, c, 0});
    for (auto &e : px) {
        // the thread that tests the pixel: word column g of 18, rows r0 + 3 + 14 k
        const int g = e.c / 4 - 3, o = e.c & 3, k = (e.r - 3) / 14, r0 = (e.r - 3) % 14;
        const int t = r0 * 18 + g, wave = t / 64, lane = t % 64;
        if (order == 0)
            e.key = rand();
        else if (order == 1)  // by wave, lane, pixel of the word, trip descending
            e.key = (((long)wave * 64 + lane) * 4 + o) * 8 + (7 - k);
        else if (order == 2)  // by wave, trip, lane, pixel of the word
            e.key = (((long)wave * 8 + k) * 64 + lane) * 4 + o;
        else if (order == 3)  // row-major
            e.key = (long)e.r * 128 + e.c;
        else {  // a thread tests five consecutive rows of its word column: by wave, lane, pixel of the word, row descending
            const int kk = (e.r - 3) % 5, rr0 = (e.r - 3) / 5, tt = rr0 * 18 + g;
            e.key = ((long)tt * 4 + o) * 8 + (7 - kk);
        }
    }
    std::sort(px.begin(), px.end(), [](const Px &a, const Px &b) { return a.key < b.key; });
    printf("order %d: %zu pixels pass (the queue holds %d)\n", order, px.size(), NQ);
    for (int i = 0; i < NQ; i++) {
        const int r = px[i % px.size()].r, c = px[i % px.size()].c;
        q[i] = (uint16_t)((r << 7) | c);
        const int v = tile[r * P + c];
        int bd = -999, bb = -999;
        for (int k = 0; k < 16; k++) {
            int md = 999, mb = 999;
            for (int j = 0; j < 9; j++) {
                const int pv = tile[(r + dy[(k + j) & 15]) * P + c + dx[(k + j) & 15]];
                md = md < v - pv ? md : v - pv;
                mb = mb < pv - v ? mb : pv - v;
            }
            bd = bd > md ? bd : md;
            bb = bb > mb ? bb : mb;
        }
        pol[i] = bb > bd;
        const int A = bb > bd ? bb : bd;
        ref[i] = (uint8_t)(A > 0 ? A : 0);
    }
    uint8_t *d_tile, *d_pol, *d_out;
    uint16_t *d_q;
    OK(hipMalloc(&d_tile, tile.size()));
    OK(hipMalloc(&d_pol, NQ));
    OK(hipMalloc(&d_out, NQ));
    OK(hipMalloc(&d_q, NQ * 2));
    OK(hipMemcpy(d_tile, tile.data(), tile.size(), hipMemcpyHostToDevice));
    OK(hipMemcpy(d_pol, pol.data(), NQ, hipMemcpyHostToDevice));
    OK(hipMemcpy(d_q, q.data(), NQ * 2, hipMemcpyHostToDevice));
    const int blocks = 256 * 8, reps = 64;
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    const char *names[4] = {"A one pixel, byte reads      ", "B two pixels, byte reads     ", "C two pixels, unaligned reads",
                            "D one pixel, unaligned reads "};
    for (int v = 0; v < 4; v++) {
        float best = 1e9f;
        for (int it = 0; it < 4; it++) {
            OK(hipEventRecord(e0, 0));
            if (v == 0)
                k_arc<0><<<blocks, 256>>>(d_tile, d_q, d_pol, d_out, reps);
            else if (v == 1)
                k_arc<1><<<blocks, 256>>>(d_tile, d_q, d_pol, d_out, reps);
            else if (v == 2)
                k_arc<2><<<blocks, 256>>>(d_tile, d_q, d_pol, d_out, reps);
            else
                k_arc<3><<<blocks, 256>>>(d_tile, d_q, d_pol, d_out, reps);
            OK(hipEventRecord(e1, 0));
            OK(hipEventSynchronize(e1));
            float ms;
            OK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        std::vector<uint8_t> out(NQ);
        OK(hipMemcpy(out.data(), d_out, NQ, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < NQ; i++)
            bad += out[i] != ref[i];
        printf("%s  %.3f ms for %d tile visits of %d pixels: %.2f ns per visit and compute unit, mismatches %d\n", names[v],
               best, blocks * reps, NQ, best * 1e6 / (blocks * reps) * 256, bad);
    }
    return 0;
}
