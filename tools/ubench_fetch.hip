// Calibration of the TCC FETCH_SIZE / WRITE_SIZE counters for the access widths the extractor
// kernels use (4 bytes per lane) beside the 16-byte-per-lane pattern whose correction is known.
// Each kernel streams a 1 GiB buffer once (larger than the 256 MiB last-level cache), so the true
// HBM byte count is known; run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` separately
// and compare the per-dispatch counter with the bytes printed here.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

__global__ __launch_bounds__(256) void read_b32(const uint32_t *__restrict__ p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc ^= p[i];
    if (acc == 0x12345677u)
        out[0] = acc;
}

__global__ __launch_bounds__(256) void read_b128(const uint4 *__restrict__ p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345677u)
        out[0] = acc;
}

__global__ __launch_bounds__(256) void write_b32(uint32_t *__restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        p[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void write_b128(uint4 *__restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    uint8_t *buf;
    uint32_t *out;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    read_b32<<<grid, 256>>>((const uint32_t *)buf, bytes / 4, out);
    CHECK(hipDeviceSynchronize());
    read_b128<<<grid, 256>>>((const uint4 *)buf, bytes / 16, out);
    CHECK(hipDeviceSynchronize());
    write_b32<<<grid, 256>>>((uint32_t *)buf, bytes / 4);
    CHECK(hipDeviceSynchronize());
    write_b128<<<grid, 256>>>((uint4 *)buf, bytes / 16);
    CHECK(hipDeviceSynchronize());
    printf("each kernel moves %zu bytes (%zu KiB)\n", bytes, bytes / 1024);
    CHECK(hipFree(buf));
    CHECK(hipFree(out));
    return 0;
}
