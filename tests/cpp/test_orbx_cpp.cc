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
        printf("OK n=%zu fnv=%016llx levels=%d scale1=%.9g padded=%dx%d d01=%d\n", kps.size(), sum, ex.GetLevels(),
               ex.GetScaleFactors()[1], pw, ph, orbx::descriptor_distance(desc.data(), desc.data() + 32));
    } catch (const orbx::Error &e) {
        printf("FAIL %s\n", e.what());
        return 1;
    }
    return 0;
}
