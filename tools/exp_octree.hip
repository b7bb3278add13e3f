// Where the time of one quadtree distribution goes: runs k_octree_single (the kernel body of the library,
// included as source) on a level like level 0 of the benchmark frames (3077 candidates in 1209 x 344, quota
// 434) with shader-clock stamps at the phase borders of every round.
// usage: bin/exp_octree [n] [quota]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

__device__ unsigned long long g_stamp[512];
__device__ int g_nstamp;
#define OCT_STAMP(id)                                                                                        \
    do {                                                                                                     \
        if (threadIdx.x == 0) {                                                                              \
            const int k_ = g_nstamp++;                                                                       \
            if (k_ < 512)                                                                                    \
                g_stamp[k_] = ((unsigned long long)(id) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); \
        }                                                                                                    \
    } while (0)
#include "octree.hip"

#define CK(x)                                                          \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e_));             \
            return 1;                                                  \
        }                                                              \
    } while (0)

template <int NT> static int run(int n, int quota, const std::vector<uint32_t> &xy, const std::vector<uint8_t> &rp)
{
    const int rw = 1209, rh = 344;
    const int n_ini = (int)roundf((float)rw / rh);
    const float hx = (float)rw / n_ini;
    const int cap = ((quota > 4 * n_ini ? quota : 4 * n_ini) + 4 + 3) & ~3;
    uint32_t *d_xy, *d_oxy;
    uint8_t *d_rp, *d_orp;
    uint16_t *d_node;
    int *d_oidx, *d_n, *d_st;
    CK(hipMalloc(&d_xy, n * 4));
    CK(hipMalloc(&d_rp, n));
    CK(hipMalloc(&d_node, n * 2));
    CK(hipMalloc(&d_oxy, cap * 4));
    CK(hipMalloc(&d_orp, cap));
    CK(hipMalloc(&d_oidx, cap * 4));
    CK(hipMalloc(&d_n, 4));
    CK(hipMalloc(&d_st, 64));
    CK(hipMemcpy(d_xy, xy.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rp, rp.data(), n, hipMemcpyHostToDevice));
    CK(hipMemset(d_st, 0, 64));
    const size_t lds = orbx_octree_lds_bytes(cap);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree_single<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        int zero = 0;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_nstamp), &zero, 4));
        CK(hipEventRecord(e0, 0));
        k_octree_single<NT><<<1, NT, lds, 0>>>(d_xy, d_rp, d_node, n, quota, n_ini, hx, rh, cap, d_oxy, d_orp, d_oidx, d_n,
                                               d_st);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    int ns = 0, kept = 0;
    unsigned long long st[512];
    CK(hipMemcpyFromSymbol(&ns, HIP_SYMBOL(g_nstamp), 4));
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st)));
    CK(hipMemcpy(&kept, d_n, 4, hipMemcpyDeviceToHost));
    printf("%d threads: %d candidates -> %d kept, %.1f us by events, %d stamps (shader clocks):\n", NT, n, kept, ms * 1e3, ns);
    const char *names[] = {"start",        "roots done", "round: order", "round: child counts", "round: prefix of children",
                           "round: lists", "round: move", "round: next", "best keys",           "end"};
    ns = ns < 512 ? ns : 512;
    for (int i = 1; i < ns; i++) {
        const int id = (int)(st[i - 1] >> 56);
        const long long dt = (long long)((st[i] & 0x00ffffffffffffffull) - (st[i - 1] & 0x00ffffffffffffffull));
        printf("  %-28s %8lld\n", names[id], dt);
    }
    printf("  total %lld clocks\n", (long long)((st[ns - 1] & 0x00ffffffffffffffull) - (st[0] & 0x00ffffffffffffffull)));
    return 0;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 3077, quota = argc > 2 ? atoi(argv[2]) : 434;
    std::vector<uint32_t> xy(n);
    std::vector<uint8_t> rp(n);
    unsigned long long s = 12345;
    std::vector<char> used(1209 * 344, 0);
    for (int i = 0; i < n;) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const int x = (int)((s >> 33) % 1209), y = (int)((s >> 13) % 344);
        if (used[y * 1209 + x])
            continue;
        used[y * 1209 + x] = 1;
        xy[i] = (uint32_t)x | ((uint32_t)y << 16);
        rp[i] = (uint8_t)(7 + (s >> 50) % 60);
        i++;
    }
    // the order of the library's candidate lists: by 30-pixel cell, row-major inside a cell
    std::vector<int> idx(n);
    for (int i = 0; i < n; i++)
        idx[i] = i;
    auto cell_key = [&](int i) {
        const unsigned x = xy[i] & 0xffff, y = xy[i] >> 16;
        return ((unsigned long long)((y / 30) * 64 + x / 30) << 32) | (y << 16) | x;
    };
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return cell_key(a) < cell_key(b); });
    std::vector<uint32_t> sxy(n);
    std::vector<uint8_t> srp(n);
    for (int i = 0; i < n; i++)
        sxy[i] = xy[idx[i]], srp[i] = rp[idx[i]];
    return run<256>(n, quota, sxy, srp);
}
