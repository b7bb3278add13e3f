// Lane map and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands: builds A (32x64)
// and B (64x32) under the assumed layout (row / column = lane & 31; the lane holds the 32 consecutive k
// of half lane >> 5, element e in nibble e of its first four registers, low nibble first; the scale is an
// E8M0 exponent in byte 0 of the scale register, 127 = 1), multiplies on the GPU and compares with the
// CPU product.  Prints the number of mismatches (0 = the assumptions hold) and a timing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const int *a, const int *b, float *c, int sa, int sb)
{
    const int lane = threadIdx.x;
    v8i va, vb;
    for (int i = 0; i < 8; i++)
        va[i] = a[lane * 8 + i], vb[i] = b[lane * 8 + i];
    v16f acc;
    for (int i = 0; i < 16; i++)
        acc[i] = (float)(31 - ((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)));
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 4, 4, 0, sa, 0, sb);
    for (int i = 0; i < 16; i++)
        c[lane * 16 + i] = acc[i];
}
__global__ __launch_bounds__(256) void rate(int steps, int *out)
{
    v8i a, b;
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
    for (int i = 0; i < 8; i++) {
        unsigned wa = 0, wb = 0;
        for (int j = 0; j < 8; j++) {
            x = x * 1664525u + 1013904223u;
            wa |= ((x >> 16 & 1) ? 0x2u : 0xau) << (4 * j);
            wb |= ((x >> 17 & 1) ? 0x2u : 0xau) << (4 * j);
        }
        a[i] = i < 4 ? (int)wa : 0, b[i] = i < 4 ? (int)wb : 0;
    }
    v16f c0 = {0}, c1 = {0};
    for (int i = 0; i < steps; i++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 134, 0, 134);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 4, 4, 0, 134, 0, 134);
        }
    }
    if (c0[0] + c1[3] == 12345.f)
        out[0] = 1;
}
static const float F4[16] = {0, 0.5f, 1, 1.5f, 2, 3, 4, 6, -0.f, -0.5f, -1, -1.5f, -2, -3, -4, -6};
int main()
{
    unsigned char A[32][64], B[64][32];  // FP4 codes
    srand(1);
    for (int i = 0; i < 32; i++)
        for (int kk = 0; kk < 64; kk++)
            A[i][kk] = rand() & 15, B[kk][i] = rand() & 15;
    int ha[64 * 8] = {0}, hb[64 * 8] = {0};
    for (int lane = 0; lane < 64; lane++)
        for (int e = 0; e < 32; e++) {
            const int kk = 32 * (lane >> 5) + e;
            ha[lane * 8 + e / 8] |= (int)((unsigned)A[lane & 31][kk] << (4 * (e % 8)));
            hb[lane * 8 + e / 8] |= (int)((unsigned)B[kk][lane & 31] << (4 * (e % 8)));
        }
    int *da, *db;
    float *dc;
    hipMalloc(&da, sizeof(ha)), hipMalloc(&db, sizeof(hb)), hipMalloc(&dc, 64 * 16 * 4);
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    for (int sa = 127; sa <= 134; sa += 7) {
        k<<<1, 64>>>(da, db, dc, sa, 134);
        float hc[64 * 16];
        hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; lane++)
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
                double ref = 31 - row;
                for (int kk = 0; kk < 64; kk++)
                    ref += (double)F4[A[row][kk]] * F4[B[kk][col]] * ldexp(1.0, sa - 127) * 128.0;
                if (fabs(ref - hc[lane * 16 + r]) > 1e-3 * fabs(ref) + 1e-3) {
                    if (bad < 4)
                        printf("mismatch lane %d reg %d: gpu %g ref %g\n", lane, r, hc[lane * 16 + r], ref);
                    bad++;
                }
            }
        printf("scale_a %d, scale_b 134: %d mismatches of 1024\n", sa, bad);
    }
    int *out;
    hipMalloc(&out, 4);
    const int steps = 4000, blocks = 512;
    rate<<<blocks, 256>>>(10, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    rate<<<blocks, 256>>>(steps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * steps * 8;
    printf("random +-1 FP4 operands: %.3f ms, %.3e MFMA/s, %.3e MAC/s, %.3e descriptor pairs/s equivalent\n", ms,
           n / (ms * 1e-3), n * 65536 / (ms * 1e-3), n / 4 * 1024 / (ms * 1e-3));
    return 0;
}
