// Native driver of the CPU oracle for tests/test_oracle_sanitizers.py: extraction at five image sizes with both blur
// kernels and a top-2 match, built with -fsanitize=address,undefined.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { int nfeatures; float scale_factor; int nlevels, ini_th_fast, min_th_fast, blur_variant; } orc_params;
typedef struct { float x, y, size, angle, response; int octave, class_id; } orc_keypoint;
int orc_extract(const orc_params *, const uint8_t *, int, int, int, orc_keypoint *, uint8_t *, int, int *, void *);
void orc_match_top2(const uint8_t *, int, const uint8_t *, int64_t, int64_t, int32_t *, uint16_t *, uint16_t *);
int orc_search_for_initialization(const orc_keypoint *, const uint8_t *, int, const orc_keypoint *, const uint8_t *, int, float,
                                  float, float, float, float *, int, float, int, int *, int64_t *);
void orc_grid_queries(const orc_keypoint *, int, float, float, float, float, const float *, const float *, const float *,
                      const int *, const int *, int, int, int *, int *, int *, int *);
void orc_undistort_points(const float *, int, const float *, const float *, int, float *);
int orc_distinctive_descriptor(const uint8_t *, int);
int orc_search_by_projection(const orc_keypoint *, const uint8_t *, const float *, const uint8_t *, int, float, float, float,
                             float, const uint8_t *, const float *, const float *, const float *, const int *, const int *,
                             const float *, const uint8_t *, const uint8_t *, const float *, int, int, float, int, int, int *,
                             int *);
int main(void)
{
    unsigned long long s = 1;
    const int sizes[][2] = {{1241, 376}, {640, 480}, {97, 81}, {333, 217}, {44, 44}};
    for (int k = 0; k < 5; k++) {
        const int w = sizes[k][0], h = sizes[k][1];
        uint8_t *img = malloc((size_t)w * h);
        memset(img, 128, (size_t)w * h);
        for (int r = 0; r < w * h / 300; r++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const int x0 = (s >> 33) % w, y0 = (s >> 13) % h, rw = 6 + (s >> 50) % 59, rh = 6 + (s >> 56) % 59, g = (s >> 5) & 255;
            for (int y = y0; y < h && y < y0 + rh; y++)
                for (int x = x0; x < w && x < x0 + rw; x++)
                    img[y * w + x] = (uint8_t)g;
        }
        for (int i = 0; i < w * h; i++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            int v = img[i] + (int)((s >> 40) % 7) - 3;
            img[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
        for (int variant = 0; variant < 2; variant++) {
            orc_params p = {k == 0 ? 2000 : 1000, 1.2f, 8, 20, 7, variant};
            const int cap = 6000;
            orc_keypoint *kps = malloc(sizeof(orc_keypoint) * cap);
            uint8_t *desc = malloc((size_t)cap * 32);
            int n = 0;
            const int rc = orc_extract(&p, img, w, h, w, kps, desc, cap, &n, NULL);
            printf("%dx%d blur %d: rc %d, %d keypoints\n", w, h, variant, rc, n);
            if (n > 64) {
                int32_t best[64];
                uint16_t d1[64], d2[64];
                orc_match_top2(desc, 64, desc + 64 * 32, n - 64, 0, best, d1, d2);
                // the frame against itself through every frame-level matcher
                float *prev = malloc(sizeof(float) * 2 * n), *xy = malloc(sizeof(float) * 2 * n);
                float *qx = malloc(sizeof(float) * n), *qy = malloc(sizeof(float) * n), *qr = malloc(sizeof(float) * n);
                int *m12 = malloc(sizeof(int) * n), *lo = malloc(sizeof(int) * n), *hi = malloc(sizeof(int) * n);
                int *fm = malloc(sizeof(int) * n), *mt = malloc(sizeof(int) * n), *cnt = malloc(sizeof(int) * n);
                int *lists = malloc(sizeof(int) * (size_t)n * 64), *cs = malloc(sizeof(int) * (64 * 48 + 1));
                int *ci = malloc(sizeof(int) * n);
                float *ang = malloc(sizeof(float) * n);
                for (int i = 0; i < n; i++) {
                    prev[2 * i] = kps[i].x + 2.f, prev[2 * i + 1] = kps[i].y - 1.f;
                    qx[i] = kps[i].x + 1.5f, qy[i] = kps[i].y, qr[i] = 7.f * (1 + kps[i].octave % 3);
                    lo[i] = kps[i].octave - 1, hi[i] = i % 4 ? kps[i].octave + 1 : -1;
                    ang[i] = kps[i].angle;
                }
                int64_t pairs = 0;
                const int nm = orc_search_for_initialization(kps, desc, n, kps, desc, n, 0.f, (float)w, 0.f, (float)h, prev, 30,
                                                             0.9f, 1, m12, &pairs);
                orc_grid_queries(kps, n, 0.f, (float)w, 0.f, (float)h, qx, qy, qr, lo, hi, n, 64, lists, cnt, cs, ci);
                const float cam[4] = {517.3f, 516.5f, 318.6f, 255.3f}, dist[5] = {0.2624f, -0.9531f, -0.0054f, 0.0026f, 1.1633f};
                orc_undistort_points(prev, n, cam, dist, 5, xy);
                const int pick = orc_distinctive_descriptor(desc, n < 90 ? n : 90);
                int np = 0;
                for (int form = 0; form < 3; form++)
                    np += orc_search_by_projection(kps, desc, NULL, NULL, n, 0.f, (float)w, 0.f, (float)h, NULL, qx, qy, qr, lo,
                                                   hi, NULL, desc, NULL, ang, n, form == 0, 0.8f, 100, form, mt, fm);
                printf("  matchers: init %d (%lld pairs), distinctive %d, projection %d\n", nm, (long long)pairs, pick, np);
                free(prev), free(xy), free(qx), free(qy), free(qr), free(m12), free(lo), free(hi), free(fm), free(mt);
                free(cnt), free(lists), free(cs), free(ci), free(ang);
            }
            free(kps), free(desc);
        }
        free(img);
    }
    return 0;
}
