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
        // orbx::Matcher on the two frames: flat top-2 with the ratio rule, SearchForInitialization,
        // UndistortKeyPoints with the image bounds, GetFeaturesInArea and ComputeDistinctiveDescriptors
        orbx::Matcher mt(0.9f, true);
        std::vector<int32_t> best;
        std::vector<uint16_t> d1, d2;
        mt.match_top2(desc.data(), (int)kps.size(), dr.data(), (int64_t)kr.size(), best, d1, d2);
        std::vector<uint8_t> acc;
        const int nacc = mt.ratio_test(d1, d2, acc);
        unsigned long long tsum = 1469598103934665603ull;
        auto tmix = [&](const void *p, size_t n) {
            const unsigned char *b = (const unsigned char *)p;
            for (size_t i = 0; i < n; i++)
                tsum = (tsum ^ b[i]) * 1099511628211ull;
        };
        tmix(best.data(), best.size() * 4), tmix(d1.data(), d1.size() * 2), tmix(d2.data(), d2.size() * 2);
        const float cam[4] = {517.3f, 516.5f, 318.6f, 255.3f};
        const std::vector<float> dist = {0.2624f, -0.9531f, -0.0054f, 0.0026f, 1.1633f};
        const orbx::Bounds bd = mt.ComputeImageBounds(w, h, cam, dist);
        std::vector<orbx_keypoint> ku = mt.UndistortKeyPoints(kps, cam, dist), kru = mt.UndistortKeyPoints(kr, cam, dist);
        unsigned long long usum = 1469598103934665603ull;
        for (const auto &k : ku)
            for (const float *f : {&k.x, &k.y})
                for (int i = 0; i < 4; i++)
                    usum = (usum ^ ((const unsigned char *)f)[i]) * 1099511628211ull;
        std::vector<float> prev(kps.size() * 2);
        for (size_t i = 0; i < ku.size(); i++)
            prev[2 * i] = ku[i].x, prev[2 * i + 1] = ku[i].y;
        std::vector<int> m12;
        const int nsfi = mt.SearchForInitialization(ku, desc, kru, dr, bd, prev, m12, 30);
        long long m12sum = 0;
        for (size_t i = 0; i < m12.size(); i++)
            m12sum += (long long)(m12[i] + 1) * (long long)(i + 1);
        std::vector<float> qx, qy, qr;
        std::vector<int32_t> lo, hi;
        for (size_t i = 0; i < ku.size(); i += 7)
            qx.push_back(ku[i].x), qy.push_back(ku[i].y), qr.push_back(20.f + (float)(i % 5) * 8.f), lo.push_back(0),
                hi.push_back(i % 3 ? -1 : 2);
        std::vector<std::vector<int32_t>> area;
        mt.GetFeaturesInArea(ku, bd, qx, qy, qr, lo, hi, area);
        long long asum = 0;
        for (size_t q = 0; q < area.size(); q++)
            for (size_t j = 0; j < area[q].size(); j++)
                asum += (long long)(area[q][j] + 1) * (long long)(q + j + 1);
        std::vector<int32_t> seg;
        for (size_t i = 0; i <= kps.size(); i += 9)
            seg.push_back((int32_t)i);
        std::vector<int32_t> bidx;
        std::vector<uint8_t> bdesc;
        mt.ComputeDistinctiveDescriptors(desc, seg, bidx, bdesc);
        long long dsum = 0;
        for (size_t i = 0; i < bidx.size(); i++)
            dsum += (long long)(bidx[i] + 1) * (long long)(i + 1);
        // SearchByProjection, last-frame form: frame 1's features searched in frame 2 around their own positions
        orbx::Matcher::Sources src;
        const std::vector<float> sf = ex.GetScaleFactors();
        for (size_t i = 0; i < ku.size(); i++) {
            src.x.push_back(ku[i].x), src.y.push_back(ku[i].y), src.r.push_back(7.f * sf[ku[i].octave]);
            src.min_level.push_back(ku[i].octave - 1), src.max_level.push_back(ku[i].octave + 1);
            src.angle.push_back(ku[i].angle);
        }
        src.desc = desc;
        std::vector<int32_t> pm, pfm;
        const int nproj = mt.SearchByProjection(kru, dr, {}, {}, bd, src, orbx::Matcher::LastFrame, pm, pfm);
        long long psum = 0;
        for (size_t i = 0; i < pm.size(); i++)
            psum += (long long)(pm[i] + 1) * (long long)(i + 1);
        for (size_t i = 0; i < pfm.size(); i++)
            psum += (long long)(pfm[i] + 3) * (long long)(i + 1);
        printf("OK n=%zu fnv=%016llx levels=%d scale1=%.9g padded=%dx%d d01=%d stereo=%d near16=%d top2=%016llx acc=%d "
               "und=%016llx bounds=%.9g,%.9g,%.9g,%.9g sfi=%d m12=%lld area=%lld dist=%lld proj=%d pm=%lld\n",
               kps.size(), sum, ex.GetLevels(), ex.GetScaleFactors()[1], pw, ph,
               orbx::descriptor_distance(desc.data(), desc.data() + 32), nm, near16, tsum, nacc, usum, bd.min_x, bd.max_x,
               bd.min_y, bd.max_y, nsfi, m12sum, asum, dsum, nproj, psum);
    } catch (const orbx::Error &e) {
        printf("FAIL %s\n", e.what());
        return 1;
    }
    return 0;
}
