// Sustained rate of v_mfma_i32_32x32x32_i8 with operands in registers: every wave runs two accumulation
// chains of eight products per step, as the matrix-core Hamming kernel does, 768 workgroups of four
// waves.  Prints products per second and the equivalent descriptor pairs per second (8 products = 1024 pairs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(int steps, int *out)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    v16i c0 = {0}, c1 = {0};
    for (int i = 0; i < steps; i++) {
#pragma unroll
        for (int s = 0; s < 8; s++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0);
        }
    }
    if (c0[0] + c1[3] == 0x7fffffff)
        out[0] = 1;
}
int main()
{
    int *out;
    hipMalloc(&out, 4);
    const int steps = 2000, blocks = 768;
    k<<<blocks, 256>>>(10, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    k<<<blocks, 256>>>(steps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * steps * 16;
    printf("%.3f ms, %.3e MFMA/s, %.3e MAC/s, %.3e descriptor pairs/s equivalent\n", ms, n / (ms * 1e-3),
           n * 32768 / (ms * 1e-3), n / 8 * 1024 / (ms * 1e-3));
    return 0;
}
