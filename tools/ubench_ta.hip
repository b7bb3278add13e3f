// Address-unit throughput of row-shaped window loads: how many cycles a compute unit spends per wave
// load instruction when the 64 lanes read `lpr` lanes x `bpl` bytes from each of 64 / lpr image rows
// (cache-resident data).  Prints cycles per wave instruction and bytes per cycle for several shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u1 __attribute__((aligned(1)));
typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(1)));

template <int BPL>
__global__ __launch_bounds__(256) void k_ta(const uint8_t *__restrict__ buf, int lpr, int pitch, int mis, int iters,
                                            uint32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int row = lane / lpr, col = (lane % lpr) * BPL + mis;
    const uint8_t *p = buf + row * pitch + col + (threadIdx.x >> 6) * 64;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint8_t *q = p + ((i + k) & 7) * pitch * 4;
            if (BPL == 4) {
                acc ^= *reinterpret_cast<const u1 *>(q);
            } else {
                const u4 v = *reinterpret_cast<const u4 *>(q);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

int main()
{
    const int pitch = 1280, rows = 128;
    uint8_t *buf;
    uint32_t *out;
    (void)hipMalloc(&buf, pitch * rows + 4096);
    hipMalloc(&out, 4);
    hipMemset(buf, 1, pitch * rows + 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);  // kHz
    const int iters = 2000, blocks = 256 * 8;
    struct Shape { int bpl, lpr, mis; const char *what; };
    const Shape shapes[] = {
        {4, 64, 0, "4B x 64 lanes/row aligned (256 B rows)"}, {4, 16, 0, "4B x 16 lanes/row aligned (64 B rows)"},
        {4, 16, 1, "4B x 16 lanes/row +1"},                  {4, 8, 0, "4B x 8 lanes/row aligned (32 B rows)"},
        {4, 8, 1, "4B x 8 lanes/row +1"},                    {4, 8, 20, "4B x 8 lanes/row +20"},
        {4, 4, 0, "4B x 4 lanes/row aligned (16 B rows)"},   {4, 1, 0, "4B x 1 lane/row"},
        {16, 16, 0, "16B x 16 lanes/row aligned (256 B rows)"}, {16, 4, 0, "16B x 4 lanes/row aligned (64 B rows)"},
        {16, 4, 1, "16B x 4 lanes/row +1"},                  {16, 2, 0, "16B x 2 lanes/row aligned (32 B rows)"},
        {16, 2, 1, "16B x 2 lanes/row +1"},                  {16, 3, 0, "16B x 3 lanes/row aligned (48 B rows)"},
        {16, 3, 2, "16B x 3 lanes/row +2"},                  {16, 1, 0, "16B x 1 lane/row"},
        {4, 9, 0, "4B x 9 lanes/row (36 B rows) from byte 0"},  {4, 9, 44, "4B x 9 lanes/row from byte 44"},
        {4, 11, 0, "4B x 11 lanes/row (44 B rows) from byte 0"}, {4, 11, 36, "4B x 11 lanes/row from byte 36"},
        {16, 8, 0, "16B x 8 lanes/row aligned (128 B rows)"},
    };
    for (const Shape &s : shapes) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (s.bpl == 4)
                k_ta<4><<<blocks, 256>>>(buf, s.lpr, pitch, s.mis, iters, out);
            else
                k_ta<16><<<blocks, 256>>>(buf, s.lpr, pitch, s.mis, iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) {
                const double waves_per_cu = blocks * 4.0 / 256, instr = waves_per_cu * iters * 8.0;
                const double cycles = ms * 1e-3 * clk * 1e3;
                printf("%-45s %7.2f cycles/wave-load  %6.1f B/cycle/CU\n", s.what, cycles / instr,
                       64.0 * s.bpl * instr / cycles);
            }
        }
    }
    if (hipDeviceSynchronize() != hipSuccess)
        return 1;
    return 0;
}
