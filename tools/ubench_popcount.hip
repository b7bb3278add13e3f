// Microbenchmark: sustained rate of the two vector instructions the Hamming matcher is made of
// (v_xor_b32 and v_bcnt_u32_b32, which adds its popcount to an accumulator), register operands only.
// Prints lane-operations per second; 16 of them make one 256-bit descriptor pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[8], acc[4] = {0, 0, 0, 0};
    for (int i = 0; i < 8; i++)
        a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;
    uint32_t s = seed;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            s = s * 1664525u + 1013904223u;  // scalar (uniform) value, like a database word
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0)       // xor + bcnt (the matcher's pair of instructions)
                    acc[i & 3] += __popc(a[i] ^ (s + i));
                else if (MODE == 1)  // bcnt only
                    acc[i & 3] += __popc(a[i] + 0) , a[i] += (MODE == 1 ? 0 : 1);
                else                 // xor only
                    acc[i & 3] ^= a[i] ^ (s + i);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main()
{
    uint32_t *d;
    const int blocks = 256 * 8;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4096;
    for (int mode = 0; mode < 3; mode += 2) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0)
                k<0><<<blocks, 256>>>(d, 12345u + rep, iters);
            else
                k<2><<<blocks, 256>>>(d, 12345u + rep, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // lane-ops counted: mode 0 -> 2 per inner step (xor, bcnt); mode 2 -> 2 (xor, xor)
            const double ops = (double)blocks * 256 * iters * 8 * 8 * 2;
            printf("mode %d rep %d: %.3f ms, %.2f Tlane-op/s\n", mode, rep, ms, ops / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
