// How many vector instructions fit beside a v_mfma_i32_32x32x32_i8 at two waves per SIMD: every product
// is followed by F independent v_max3_i32 in a fixed order (nothing moves across a slot's end).
// Prints the product rate per F.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int F>
__global__ __launch_bounds__(256, 2) void k(int steps, int *out)
{
    v4i a = {(int)threadIdx.x * 0x01010101, 0x40c040c0, 0x4040c0c0, 0x40404040}, b = {0x40c0c040, 0x40c040c0, (int)blockIdx.x, 0x40404040};
    v16i c0 = {0};
    int f[8];
    for (int j = 0; j < 8; j++)
        f[j] = threadIdx.x + j;
    for (int i = 0; i < steps; i++) {
#pragma unroll
        for (int s = 0; s < 16; s++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < F; j++)
                asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(f[j % 8]) : "v"(f[(j + 1) % 8]), "v"(i));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int sum = c0[0];
    for (int j = 0; j < 8; j++)
        sum += f[j];
    if (sum == 0x7fffffff)
        out[0] = 1;
}
template <int F> void run(int *out)
{
    const int steps = 2000, blocks = 512;
    k<F><<<blocks, 256>>>(10, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    k<F><<<blocks, 256>>>(steps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * steps * 16;
    printf("%d vector instructions per product: %.3f ms, %.3e MFMA/s, %.1f cycles per product and SIMD at 2.1 GHz\n", F, ms,
           n / (ms * 1e-3), ms * 1e-3 * 2.1e9 / (steps * 16 * 2));
}
int main()
{
    int *out;
    hipMalloc(&out, 4);
    run<0>(out), run<2>(out), run<3>(out), run<4>(out), run<5>(out), run<6>(out), run<8>(out), run<10>(out);
    return 0;
}
