// Sustained rate of v_mfma_i32_32x32x32_i8 with operands in registers, by the number of independent
// accumulation chains a wave interleaves (1, 2, 4) and the number of waves per SIMD (blocks per CU).
// Prints products per second and the equivalent descriptor pairs per second (8 products = 1024 pairs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(int steps, int *out)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    v16i c[CHAINS];
    for (int j = 0; j < CHAINS; j++)
        c[j] = v16i{0};
    for (int i = 0; i < steps; i++) {
#pragma unroll
        for (int s = 0; s < 16 / CHAINS; s++)
#pragma unroll
            for (int j = 0; j < CHAINS; j++)
                c[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(j & 1 ? b : a, j & 1 ? a : b, c[j], 0, 0, 0);
    }
    int sum = 0;
    for (int j = 0; j < CHAINS; j++)
        sum += c[j][j];
    if (sum == 0x7fffffff)
        out[0] = 1;
}
// the same with eight different operand pairs of random +-C bytes, as real descriptors give: the chip
// lowers its clock with the switching activity of the operands
template <int C>
__global__ __launch_bounds__(256) void k_random(int steps, int *out)
{
    v4i a[8], b[8];
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int s = 0; s < 8; s++)
        for (int e = 0; e < 4; e++) {
            uint32_t wa = 0, wb = 0;
            for (int j = 0; j < 4; j++) {
                x = x * 1664525u + 1013904223u;
                wa |= ((x >> 16 & 1 ? C : -C) & 0xffu) << (8 * j);
                wb |= ((x >> 17 & 1 ? C : -C) & 0xffu) << (8 * j);
            }
            a[s][e] = (int)wa, b[s][e] = (int)wb;
        }
    v16i c0 = {0}, c1 = {0};
    for (int i = 0; i < steps; i++) {
#pragma unroll
        for (int s = 0; s < 8; s++)
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[s], c0, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 8; s++)
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[7 - s], c1, 0, 0, 0);
    }
    if (c0[0] + c1[3] == 0x7fffffff)
        out[0] = 1;
}
template <int C> void run_random(int blocks, int *out)
{
    const int steps = 2000;
    k_random<C><<<blocks, 256>>>(10, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    k_random<C><<<blocks, 256>>>(steps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * steps * 16;
    printf("random +-%d operands, %d workgroups: %.3f ms, %.3e MFMA/s, %.3e MAC/s, %.3e descriptor pairs/s equivalent\n", C,
           blocks, ms, n / (ms * 1e-3), n * 32768 / (ms * 1e-3), n / 8 * 1024 / (ms * 1e-3));
}
template <int CHAINS> void run(int blocks, int *out)
{
    const int steps = 2000;
    k<CHAINS><<<blocks, 256>>>(10, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    k<CHAINS><<<blocks, 256>>>(steps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * steps * 16;
    printf("chains %d, %d workgroups: %.3f ms, %.3e MFMA/s, %.3e MAC/s, %.3e descriptor pairs/s equivalent\n", CHAINS,
           blocks, ms, n / (ms * 1e-3), n * 32768 / (ms * 1e-3), n / 8 * 1024 / (ms * 1e-3));
}
int main()
{
    int *out;
    hipMalloc(&out, 4);
    for (int blocks : {256, 512, 768, 1024}) {
        run<1>(blocks, out);
        run<2>(blocks, out);
        run<4>(blocks, out);
    }
    for (int blocks : {512, 1024}) {
        run_random<64>(blocks, out);
        run_random<127>(blocks, out);
        run_random<1>(blocks, out);
    }
    return 0;
}
