// Timing harness of the matrix-core top-2 path (tuning aid): includes the kernel source, runs it on
// random descriptors and prints the average time of a call (top-2 kernel and merge).
#include "../orbslam2_amd/csrc/hamming.hip"
// the error recorder of the library (orbx_api.cpp), which the included source refers to
void orbx_note_hip_error(hipError_t) {}
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
    uint32_t *dq, *ddb, *dp;
    int *dbest;
    uint16_t *dd1, *dd2;
    int chunk, nchunks;
    orbx_top2_mfma_chunks(nq, ndb, &chunk, &nchunks);
    hipMalloc(&dq, hq.size() * 4), hipMalloc(&ddb, hdb.size() * 4), hipMalloc(&dp, (size_t)nchunks * nq * 8);
    hipMalloc(&dbest, nq * 4), hipMalloc(&dd1, nq * 2), hipMalloc(&dd2, nq * 2);
    hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ddb, hdb.data(), hdb.size() * 4, hipMemcpyHostToDevice);
    auto run = [&]() {
        return orbx_launch_top2_mfma((const uint8_t *)dq, nq, (const uint8_t *)ddb, ndb, 0, dp, chunk, nchunks, dbest,
                                     dd1, dd2, 0);
    };
    for (int i = 0; i < 3; i++)
        run();
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++)
        run();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.4f ms per call, top-2 kernel and merge (chunk %d x %d)\n", hipGetErrorString(hipGetLastError()), ms / reps,
           chunk, nchunks);
    return 0;
}
