// Timing harness for variants of the matrix-core top-2 kernel (tuning aid): includes the kernel source,
// runs it on random descriptors and prints the average launch time.  Build with -DMF_EXP_... switches.
#include "../orbslam2_amd/csrc/hamming.hip"
#include <cstdio>
#include <vector>
int main()
{
    const int nq = 2000;
    const int64_t ndb = 1000000;
    std::vector<uint32_t> hq((size_t)nq * 8), hdb((size_t)ndb * 8);
    uint32_t x = 12345u;
    for (auto &v : hq) v = (x = x * 1664525u + 1013904223u) ^ (x >> 13);
    for (auto &v : hdb) v = (x = x * 1664525u + 1013904223u) ^ (x >> 13);
    uint32_t *dq, *ddb;
    uint2 *dp;
    int chunk, nchunks;
    orbx_top2_mfma_chunks(nq, ndb, &chunk, &nchunks);
    hipMalloc(&dq, hq.size() * 4), hipMalloc(&ddb, hdb.size() * 4), hipMalloc(&dp, (size_t)nchunks * nq * 8 + nchunks * 8 * 8);
    hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ddb, hdb.data(), hdb.size() * 4, hipMemcpyHostToDevice);
    const dim3 grid((nq + MF_QPB - 1) / MF_QPB, nchunks);
    for (int i = 0; i < 3; i++)
        k_top2_mfma<MF_QT><<<grid, 256>>>(dq, nq, ddb, ndb, chunk, dp);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++)
        k_top2_mfma<MF_QT><<<grid, 256>>>(dq, nq, ddb, ndb, chunk, dp);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.4f ms per launch (chunk %d x %d)\n", hipGetErrorString(hipGetLastError()), ms / reps, chunk, nchunks);
#ifdef MF_EXP_CLOCK
    const int nwg = nchunks * grid.x;
    std::vector<uint2> hc(nwg);
    hipMemcpy(hc.data(), dp + (size_t)nchunks * nq, nwg * 8, hipMemcpyDeviceToHost);
    double sx[16] = {0}; int nx[16] = {0};
    for (int i = 0; i < nwg; i++) {
        const int xcc = hc[i].y >> 16 & 15;
        sx[xcc] += hc[i].x / 100.0, nx[xcc]++;
    }
    for (int x = 0; x < 16; x++) if (nx[x]) printf("xcc %d: %d workgroups, avg %.1f us\n", x, nx[x], sx[x] / nx[x]);
    for (int i = 0; i < nwg; i++)
        printf("wg %d x %d xcc %d se %d cu %d simd %d wave %d: %.1f us\n", i, i % (int)grid.x, hc[i].y >> 16 & 15, hc[i].y >> 13 & 7, hc[i].y >> 8 & 15, hc[i].y >> 4 & 3, hc[i].y & 15, hc[i].x / 100.0);
#endif
    return 0;
}
