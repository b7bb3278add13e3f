// Cost of dispatching workgroups that do nothing: grids of the size k_fast launches for 256 KITTI frames
// (88 064 workgroups of 256 threads with 20 KB of LDS) against fewer, larger and more, smaller workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int T> __global__ __launch_bounds__(T) void k_empty(int *out)
{
    extern __shared__ int lds[];
    if (out == nullptr)  // never: keeps the LDS allocation and the argument alive
        out[0] = lds[threadIdx.x];
}

#define OK(x)                                                  \
    do {                                                       \
        hipError_t e_ = (x);                                   \
        if (e_ != hipSuccess) {                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));     \
            return 1;                                          \
        }                                                      \
    } while (0)

template <int T> static int run(int blocks, int lds, int *d)
{
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    const dim3 grid(8, blocks / 8 / 32, 32);
    float best = 1e9f;
    for (int it = 0; it < 5; it++) {
        OK(hipEventRecord(e0, 0));
        for (int r = 0; r < 10; r++)
            k_empty<T><<<grid, T, lds>>>(d);
        OK(hipEventRecord(e1, 0));
        OK(hipEventSynchronize(e1));
        float ms;
        OK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    printf("%7d workgroups x %4d threads, %6d B LDS: %.4f ms per launch, %.2f ns per workgroup, %.2f ns per wave\n",
           grid.x * grid.y * grid.z, T, lds, best / 10, best / 10 * 1e6 / (grid.x * grid.y * grid.z),
           best / 10 * 1e6 / (grid.x * grid.y * grid.z) / (T / 64));
    return 0;
}

int main()
{
    int *d;
    OK(hipMalloc(&d, 4));
    const int n = 88064;
    if (run<256>(n, 20388, d) || run<256>(n, 0, d) || run<256>(n, 40776, d) || run<512>(n / 2, 40776, d) ||
        run<512>(n / 2, 0, d) || run<128>(n * 2, 10194, d) || run<1024>(n / 4, 81552, d) || run<64>(n * 4, 0, d) ||
        run<256>(n / 2, 20388, d) || run<256>(n * 2, 20388, d))
        return 1;
    return 0;
}
