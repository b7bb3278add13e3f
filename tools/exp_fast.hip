// Where the time of the detection kernel goes: runs k_fast (the kernel body of the library, included as
// source) on the pyramids of a batch of benchmark frames, whole and cut short after each of its phases
// (staging, compass test + queue, arc score, non-maximum suppression + candidate records).  The pyramids come from
// the library itself (liborbx.so: extraction of synthetic KITTI-shape frames, levels read back).
// usage: bin/exp_fast [frames] [reps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ int g_fast_cut;
#define FAST_CUT(n)          \
    if (g_fast_cut == (n))   \
        return;
// the library this program links for its pyramids exports kernels of the same names
#define k_fast k_fast_exp
#define orbx_launch_fast orbx_launch_fast_exp
#include "fast.hip"

#define CK(x)                                                          \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e_));             \
            return 1;                                                  \
        }                                                              \
    } while (0)
#define OK(x)                                                \
    do {                                                     \
        int rc_ = (x);                                       \
        if (rc_ < 0) {                                       \
            printf("%s: %s\n", #x, orbx_strerror(rc_));      \
            return 1;                                        \
        }                                                    \
    } while (0)

static int align_up(int v, int a) { return (v + a - 1) / a * a; }

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int W = 1241, H = 376, NL = 8;
    orbx_params p{2000, 1.2f, NL, 20, 7, 0, W, H, 8, 0};
    orbx_extractor *ex;
    OK(orbx_extractor_create(&p, &ex));
    // level geometry as the library lays it out (level sizes from the library, the rest by the same rules)
    FastArgs a{};
    a.nlevels = NL, a.batch = B;
    a.t_ini = 20, a.t_low = 7;
    std::vector<uint8_t> img((size_t)W * H);
    std::vector<const uint8_t *> ptrs(8, img.data());
    OK(orbx_synth_image(1, W, H, 0, img.data(), W));
    OK(orbx_extract_batch_host(ex, ptrs.data(), 8, W, H, W));
    long long off = 0;
    int cells = 0, tiles = 0, cand = 0;
    std::vector<uint32_t> tile_tab;
    for (int l = 0; l < NL; l++) {
        int w, h;
        OK(orbx_pyramid_level(ex, 0, l, 0, nullptr, 0, &w, &h));
        a.w[l] = w, a.h[l] = h, a.pitch[l] = align_up(w, 64), a.off[l] = (int)off;
        off += align_up(a.pitch[l] * h, 256);
        const float width = (float)(w - 32), height = (float)(h - 32);
        const int nc = (int)(width / 30.f), nr = (int)(height / 30.f);
        a.ncols[l] = nc;
        a.wcell[l] = (int)ceilf(width / nc), a.hcell[l] = (int)ceilf(height / nr);
        a.cell_base[l] = cells;
        cells += nc * nr;
        a.tile_base[l] = tiles;
        for (int ty = 0; ty < ORBX_FAST_TILES_Y(h); ty++)
            for (int tx = 0; tx < ORBX_FAST_TILES_X(w); tx++)
                tile_tab.push_back(ORBX_TILE_ENTRY(l, ty, tx));
        tiles = (int)tile_tab.size();
        a.cand_base[l] = cand, a.cand_cap[l] = align_up(w * h / 4 + 4, 4);
        cand += a.cand_cap[l];
        a.wc_magic[l] = ((1u << 20) + a.wcell[l] - 1) / a.wcell[l];
        a.hc_magic[l] = ((1u << 20) + a.hcell[l] - 1) / a.hcell[l];
    }
    a.tile_base[NL] = tiles;
    a.slab = (int)off, a.cells_per_frame = cells, a.tiles_per_frame = tiles, a.cand_per_frame = cand;
    // B frames with their pyramids, eight at a time through the library
    uint8_t *d_pyr, *d_flag, *d_resp;
    int *d_cnt, *d_status;
    uint32_t *d_xy, *d_key;
    CK(hipMalloc(&d_pyr, (size_t)B * a.slab + 256));
    CK(hipMalloc(&d_flag, (size_t)B * cells));
    CK(hipMalloc(&d_cnt, (size_t)B * NL * 4));
    CK(hipMalloc(&d_status, 64));
    CK(hipMalloc(&d_xy, (size_t)B * cand * 4));
    CK(hipMalloc(&d_key, (size_t)B * cand * 4));
    CK(hipMalloc(&d_resp, (size_t)B * cand));
    uint32_t *d_tab;
    CK(hipMalloc(&d_tab, tile_tab.size() * 4));
    CK(hipMemcpy(d_tab, tile_tab.data(), tile_tab.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint8_t> slab((size_t)a.slab);
    std::vector<std::vector<uint8_t>> imgs(8, std::vector<uint8_t>((size_t)W * H));
    for (int f0 = 0; f0 < B; f0 += 8) {
        const int nb = B - f0 < 8 ? B - f0 : 8;
        for (int i = 0; i < nb; i++) {
            OK(orbx_synth_image(0x0B5EED00ull + f0 + i, W, H, 0, imgs[i].data(), W));
            ptrs[i] = imgs[i].data();
        }
        OK(orbx_extract_batch_host(ex, ptrs.data(), nb, W, H, W));
        for (int i = 0; i < nb; i++) {
            for (int l = 0; l < NL; l++)
                OK(orbx_pyramid_level(ex, i, l, 0, slab.data() + a.off[l], a.pitch[l], nullptr, nullptr));
            CK(hipMemcpy(d_pyr + (size_t)(f0 + i) * a.slab, slab.data(), a.slab, hipMemcpyHostToDevice));
        }
    }
    a.l0 = OrbxLevel0{d_pyr, (long long)a.slab, a.pitch[0]};
    const dim3 grid = orbx_frame_grid(tiles, B);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[] = {"cut at the first statement", "cut before staging (tile found)", "cut after staging",
                           "cut after compass test + queue", "cut after arc score", "cut after suppression",
                           "whole kernel"};
    float tt[7];
    long long ff[7];
    float *t = tt + 1;
    long long *found = ff + 1;
    for (int cut = -1; cut <= 5; cut++) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_fast_cut), &cut, 4));
        for (int r = 0; r < 3; r++)
            k_fast<<<grid, 256>>>(d_pyr, d_cnt, d_xy, d_resp, d_key, d_flag, d_status, d_tab, a);
        CK(hipMemset(d_cnt, 0, (size_t)B * NL * 4));
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) {
            CK(hipMemsetAsync(d_cnt, 0, (size_t)B * NL * 4, 0));  // as the library does before every detection
            k_fast<<<grid, 256>>>(d_pyr, d_cnt, d_xy, d_resp, d_key, d_flag, d_status, d_tab, a);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&t[cut], e0, e1));
        t[cut] /= reps;
        std::vector<int> cnt((size_t)B * NL);
        CK(hipMemcpy(cnt.data(), d_cnt, cnt.size() * 4, hipMemcpyDeviceToHost));
        found[cut] = 0;
        for (int c : cnt)
            found[cut] += c;
    }
    printf("%d frames, %d tiles per frame\n", B, tiles);
    for (int cut = -1; cut <= 5; cut++)
        printf("  %-34s %.4f ms   (%.1f candidates per frame)\n", names[cut + 1], t[cut], (double)found[cut] / B);
    orbx_extractor_destroy(ex);
    return 0;
}
