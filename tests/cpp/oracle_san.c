// Native driver of the CPU oracle for tests/test_oracle_sanitizers.py: extraction at five image sizes with both blur
// kernels and a top-2 match, built with -fsanitize=address,undefined.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { int nfeatures; float scale_factor; int nlevels, ini_th_fast, min_th_fast, blur_variant; } orc_params;
typedef struct { float x, y, size, angle, response; int octave, class_id; } orc_keypoint;
int orc_extract(const orc_params *, const uint8_t *, int, int, int, orc_keypoint *, uint8_t *, int, int *, void *);
int orc_match_top2(const uint8_t *, int, const uint8_t *, long long, long long, int *, unsigned short *, unsigned short *);
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
                int best[64];
                unsigned short d1[64], d2[64];
                orc_match_top2(desc, 64, desc + 64 * 32, n - 64, 0, best, d1, d2);
            }
            free(kps), free(desc);
        }
        free(img);
    }
    return 0;
}
