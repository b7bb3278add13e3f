// Checks the operand and result lane maps of v_mfma_i32_16x16x64_i8 that the matrix-core blur relies on, with
// random int8 data against a host computation:
//   lane l supplies row (A) / column (B) l & 15 and the 16 k-elements of quarter l >> 4, the same k for the
//   same (quarter, element) in A and in B;  result register reg of lane l is C[row][col] with
//   col = l & 15, row = reg + 4 * (l >> 4).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const v4i *a, const v4i *b, v4i *c)
{
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    c[threadIdx.x] = acc;
}
int main()
{
    int8_t ha[64][16], hb[64][16];
    srand(7);
    for (int l = 0; l < 64; l++)
        for (int e = 0; e < 16; e++)
            ha[l][e] = (int8_t)(rand() % 255 - 127), hb[l][e] = (int8_t)(rand() % 255 - 127);
    void *da, *db, *dc;
    hipMalloc(&da, sizeof(ha)), hipMalloc(&db, sizeof(hb)), hipMalloc(&dc, 64 * 4 * 4);
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    k<<<1, 64>>>((const v4i *)da, (const v4i *)db, (v4i *)dc);
    int hc[64][4];
    if (hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost) != hipSuccess) {
        printf("FAIL hip\n");
        return 1;
    }
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int reg = 0; reg < 4; reg++) {
            const int col = l & 15, row = reg + 4 * (l >> 4);
            int want = 0;
            for (int q = 0; q < 4; q++)
                for (int e = 0; e < 16; e++)
                    want += (int)ha[row + 16 * q][e] * (int)hb[col + 16 * q][e];
            bad += want != hc[l][reg];
        }
    printf(bad ? "FAIL %d of 256 results differ\n" : "OK lane maps as assumed (%d differ)\n", bad);
    return bad != 0;
}
