// Exercises the C++ host layer (orbslam2_amd/cpp/orbx.hpp) end to end on the GPU and prints a
// digest that tests/test_gpu_cpp_host.py compares with the Python binding's result.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbx.hpp"

int main(int argc, char **argv)
{
    const int w = argc > 1 ? atoi(argv[1]) : 640, h = argc > 2 ? atoi(argv[2]) : 480;
    std::vector<uint8_t> img((size_t)w * h);
    orbx::check(orbx_synth_image(0x0B5EED00ull, w, h, 0, img.data(), w), "orbx_synth_image");
    try {
        orbx::Extractor ex(1000, 1.2f, 8, 20, 7, w, h);
        std::vector<orbx_keypoint> kps;
        std::vector<uint8_t> desc;
        if (ex(nullptr, 0, 0, 0, kps, desc)) {
            printf("FAIL empty image must return false\n");
            return 1;
        }
        ex(img.data(), w, h, w, kps, desc);
        unsigned long long sum = 1469598103934665603ull;  // FNV-1a over keypoints and descriptors
        auto mix = [&](const void *p, size_t n) {
            const unsigned char *b = (const unsigned char *)p;
            for (size_t i = 0; i < n; i++)
                sum = (sum ^ b[i]) * 1099511628211ull;
        };
        for (const auto &k : kps) {
            mix(&k.x, 4), mix(&k.y, 4), mix(&k.size, 4), mix(&k.response, 4), mix(&k.octave, 4), mix(&k.class_id, 4);
        }
        mix(desc.data(), desc.size());
        int pw, ph;
        std::vector<uint8_t> lvl = ex.pyramid_level(1, 19, pw, ph);
        // a stereo pair: the right image is the left one shifted by 16 pixels
        std::vector<uint8_t> right((size_t)w * h);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                right[(size_t)y * w + x] = img[(size_t)y * w + (x + 16 < w ? x + 16 : w - 1)];
        orbx::Extractor exr(1000, 1.2f, 8, 20, 7, w, h);
        std::vector<orbx_keypoint> kr;
        std::vector<uint8_t> dr;
        exr(right.data(), w, h, w, kr, dr);
        std::vector<float> ur, depth;
        const int nm = orbx::compute_stereo_matches(ex, exr, 0.537f, 386.1448f, ur, depth);
        int near16 = 0;
        for (size_t i = 0; i < ur.size(); i++)
            if (ur[i] >= 0 && kps[i].x - ur[i] > 15.5f && kps[i].x - ur[i] < 16.5f)
                near16++;
        printf("OK n=%zu fnv=%016llx levels=%d scale1=%.9g padded=%dx%d d01=%d stereo=%d near16=%d\n", kps.size(), sum,
               ex.GetLevels(), ex.GetScaleFactors()[1], pw, ph, orbx::descriptor_distance(desc.data(), desc.data() + 32),
               nm, near16);
    } catch (const orbx::Error &e) {
        printf("FAIL %s\n", e.what());
        return 1;
    }
    return 0;
}
