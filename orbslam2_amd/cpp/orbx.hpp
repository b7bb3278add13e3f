// orbx.hpp -- OpenCV-free C++ host layer over the C ABI of liborbx (include/orbx.h).
//
// orbx::Extractor has the constructor arguments and the call shape of ORB_SLAM2::ORBextractor
// (include/ORBextractor.h:45-111 of the reference) with plain pointers in place of cv::Mat;
// orbx::descriptor_distance is ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1744-1760) on two
// 32-byte rows.  ORBextractor.h / ORBextractor.cc next to this file put the cv:: types back on top
// of it for builds that have OpenCV.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbx.h"

namespace orbx {

struct Error : std::runtime_error {
    int code;
    Error(int c, const char *where) : std::runtime_error(std::string(where) + ": " + orbx_strerror(c)), code(c) {}
};
inline void check(int rc, const char *where)
{
    if (rc < 0)
        throw Error(rc, where);
}
// data pointer of an optional array: null when the vector is empty
template <typename T> inline const T *opt(const std::vector<T> &v) { return v.empty() ? nullptr : v.data(); }

// Extractor objects of one device that take turns on batches (orbx_pipeline_create): attach each with
// Extractor::set_pipeline; destroy it after them.
class Pipeline {
public:
    explicit Pipeline(int device = 0) { check(orbx_pipeline_create(device, &h_), "orbx_pipeline_create"); }
    ~Pipeline() { orbx_pipeline_destroy(h_); }
    Pipeline(const Pipeline &) = delete;
    Pipeline &operator=(const Pipeline &) = delete;
    orbx_pipeline *handle() const { return h_; }

private:
    orbx_pipeline *h_ = nullptr;
};

class Extractor {
public:
    Extractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int max_width = 1920,
              int max_height = 1088, int max_batch = 1, int device = 0, int blur_variant = 0)
    {
        orbx_params p{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, blur_variant,
                      max_width, max_height,  max_batch, device};
        check(orbx_extractor_create(&p, &h_), "orbx_extractor_create");
        nlevels_ = nlevels;
        scale_.resize(nlevels), inv_scale_.resize(nlevels), sigma2_.resize(nlevels), inv_sigma2_.resize(nlevels);
        check(orbx_extractor_scale_tables(h_, scale_.data(), inv_scale_.data(), sigma2_.data(), inv_sigma2_.data()),
              "orbx_extractor_scale_tables");
        cap_ = orbx_extractor_max_keypoints(h_);
    }
    ~Extractor() { orbx_extractor_destroy(h_); }
    void set_pipeline(const Pipeline *p) { check(orbx_extractor_set_pipeline(h_, p ? p->handle() : nullptr), "set_pipeline"); }
    Extractor(const Extractor &) = delete;
    Extractor &operator=(const Extractor &) = delete;

    // operator(): 8-bit gray image of height rows, stride bytes apart.  Returns false for an empty
    // image and leaves the outputs untouched, as the reference does (src/ORBextractor.cc:1117-1118);
    // zero keypoints clear both outputs.
    bool operator()(const uint8_t *image, int width, int height, int stride, std::vector<orbx_keypoint> &keypoints,
                    std::vector<uint8_t> &descriptors)
    {
        if (!image || width <= 0 || height <= 0)
            return false;
        kp_buf_.resize(cap_);
        desc_buf_.resize((size_t)cap_ * 32);
        int n = 0;
        check(orbx_extract(h_, image, width, height, stride, kp_buf_.data(), desc_buf_.data(), cap_, &n),
              "orbx_extract");
        keypoints.assign(kp_buf_.begin(), kp_buf_.begin() + n);
        descriptors.assign(desc_buf_.begin(), desc_buf_.begin() + (size_t)n * 32);
        return true;
    }

    // The same call leaving the results in the extractor's own buffers (valid until the next call):
    // returns the keypoint count, or -1 for an empty image.  For callers that copy once into their own
    // containers (the cv:: adaptor).
    int extract_raw(const uint8_t *image, int width, int height, int stride, const orbx_keypoint *&keypoints,
                    const uint8_t *&descriptors)
    {
        if (!image || width <= 0 || height <= 0)
            return -1;
        kp_buf_.resize(cap_);
        desc_buf_.resize((size_t)cap_ * 32);
        int n = 0;
        check(orbx_extract(h_, image, width, height, stride, kp_buf_.data(), desc_buf_.data(), cap_, &n),
              "orbx_extract");
        keypoints = kp_buf_.data();
        descriptors = desc_buf_.data();
        return n;
    }

    // level of the pyramid of the last call with a reflect-101 frame of `border` pixels
    // (mvImagePyramid keeps border 19, src/ORBextractor.cc:1199-1219)
    std::vector<uint8_t> pyramid_level(int level, int border, int &padded_w, int &padded_h)
    {
        int w = 0, h = 0;
        check(orbx_pyramid_level(h_, 0, level, 0, nullptr, 0, &w, &h), "orbx_pyramid_level");
        padded_w = w + 2 * border, padded_h = h + 2 * border;
        std::vector<uint8_t> out((size_t)padded_w * padded_h);
        check(orbx_pyramid_level(h_, 0, level, border, out.data(), padded_w, nullptr, nullptr), "orbx_pyramid_level");
        return out;
    }

    // every level of the last call, framed, in one round trip: level l is a dense
    // (width[l]+2*border) x (height[l]+2*border) image at out.data() + offset[l]
    void pyramid(int border, std::vector<uint8_t> &out, std::vector<int64_t> &offset, std::vector<int> &width,
                 std::vector<int> &height)
    {
        offset.resize(nlevels_), width.resize(nlevels_), height.resize(nlevels_);
        int64_t total = 0;
        check(orbx_pyramid_layout(h_, border, offset.data(), width.data(), height.data(), &total),
              "orbx_pyramid_layout");
        out.resize((size_t)total);
        check(orbx_pyramid_download(h_, 0, border, out.data(), total), "orbx_pyramid_download");
    }

    int GetLevels() const { return nlevels_; }
    float GetScaleFactor() const { return nlevels_ > 1 ? scale_[1] : 1.f; }
    std::vector<float> GetScaleFactors() const { return scale_; }
    std::vector<float> GetInverseScaleFactors() const { return inv_scale_; }
    std::vector<float> GetScaleSigmaSquares() const { return sigma2_; }
    std::vector<float> GetInverseScaleSigmaSquares() const { return inv_sigma2_; }
    orbx_extractor *handle() { return h_; }
    int max_keypoints() const { return cap_; }

private:
    orbx_extractor *h_ = nullptr;
    int nlevels_ = 0, cap_ = 0;
    std::vector<float> scale_, inv_scale_, sigma2_, inv_sigma2_;
    std::vector<orbx_keypoint> kp_buf_;
    std::vector<uint8_t> desc_buf_;
};

inline int descriptor_distance(const uint8_t *a, const uint8_t *b) { return orbx_descriptor_distance(a, b); }

// The image bounds of a Frame (mnMinX, mnMaxX, mnMinY, mnMaxY; src/Frame.cc:501-523).
struct Bounds {
    float min_x, max_x, min_y, max_y;
};

// The brute-force top-2 against a database sharded over the GPUs of a node (one thread or process per
// GPU): rank r holds the contiguous rows [idx_base, idx_base + shard_rows) in device memory and all
// queries; match() runs the shard's top-2, one RCCL all-gather of 8 bytes per query and rank and the
// merge, and every rank gets the result of a single pass over the whole database (the bookkeeping of
// src/ORBmatcher.cc:217-242: strict '<', the first row wins ties).  unique_id() on rank 0, handed to
// the other ranks by the caller's own means, then the constructor on every rank (collective).
class ShardedMatcher {
public:
    typedef std::vector<uint8_t> UniqueId;
    static UniqueId unique_id()
    {
        UniqueId id(ORBX_RCCL_UNIQUE_ID_BYTES);
        const int rc = orbx_sharded_unique_id(id.data());
        if (rc < 0)
            throw Error(rc, orbx_sharded_last_error());
        return id;
    }
    ShardedMatcher(int device, int rank, int world, const UniqueId &id, int max_queries = 4096, float nnratio = 0.75f)
        : nnratio_(nnratio), max_queries_(max_queries)
    {
        if (id.size() != ORBX_RCCL_UNIQUE_ID_BYTES)
            throw Error(ORBX_E_ARG, "ShardedMatcher: unique id size");
        const int rc = orbx_sharded_matcher_create(device, rank, world, id.data(), max_queries, &h_);
        if (rc < 0)
            throw Error(rc, rc == ORBX_E_RCCL ? orbx_sharded_last_error() : "orbx_sharded_matcher_create");
    }
    ~ShardedMatcher() { orbx_sharded_matcher_destroy(h_); }
    ShardedMatcher(const ShardedMatcher &) = delete;
    ShardedMatcher &operator=(const ShardedMatcher &) = delete;
    orbx_sharded_matcher *handle() { return h_; }
    void set_stream(void *hip_stream) { check(orbx_sharded_matcher_set_stream(h_, hip_stream), "set_stream"); }
    void synchronize() { check(orbx_sharded_matcher_synchronize(h_), "orbx_sharded_matcher_synchronize"); }
    // device pointers; asynchronous on the handle's stream (collective: every rank calls it)
    void match_device(const uint8_t *d_queries, int nq, const uint8_t *d_shard, int64_t shard_rows, int64_t idx_base,
                      int32_t *d_best, uint16_t *d_d1, uint16_t *d_d2)
    {
        const int rc = orbx_match_top2_sharded_device(h_, d_queries, nq, d_shard, shard_rows, idx_base, d_best, d_d1, d_d2);
        if (rc < 0)
            throw Error(rc, rc == ORBX_E_RCCL ? orbx_sharded_last_error() : "orbx_match_top2_sharded_device");
    }
    // d1 <= th && d1 < nnratio * d2 (src/ORBmatcher.cc:245-249) on downloaded distances
    int ratio_test(const std::vector<uint16_t> &d1, const std::vector<uint16_t> &d2, std::vector<uint8_t> &accept,
                   int th = ORBX_TH_LOW) const
    {
        accept.resize(d1.size());
        return orbx_ratio_test((int)d1.size(), d1.data(), d2.data(), th, nnratio_, accept.data());
    }

private:
    orbx_sharded_matcher *h_ = nullptr;
    float nnratio_;
    int max_queries_;
};

// orbx::Matcher carries the two constructor arguments of ORB_SLAM2::ORBmatcher (nnratio, checkOri;
// include/ORBmatcher.h:41) and the device workspace of the searches.  Methods named like the
// reference's run the same search on plain arrays: frames are (keypoints, descriptors) pairs, map
// points are their projections and descriptors.
class Matcher {
public:
    static constexpr int TH_LOW = ORBX_TH_LOW, TH_HIGH = ORBX_TH_HIGH, HISTO_LENGTH = ORBX_HISTO_LENGTH;

    explicit Matcher(float nnratio = 0.6f, bool checkOri = true, int device = 0, int max_queries = 4096)
        : nnratio_(nnratio), check_ori_(checkOri)
    {
        check(orbx_matcher_create(device, max_queries, &h_), "orbx_matcher_create");
    }
    ~Matcher() { orbx_matcher_destroy(h_); }
    Matcher(const Matcher &) = delete;
    Matcher &operator=(const Matcher &) = delete;
    orbx_matcher *handle() { return h_; }

    static int DescriptorDistance(const uint8_t *a, const uint8_t *b) { return orbx_descriptor_distance(a, b); }

    // Best and second-best database row of every query (the inner loop of SearchByBoW,
    // src/ORBmatcher.cc:217-242, over a flat candidate list).
    void match_top2(const uint8_t *queries, int nq, const uint8_t *db, int64_t ndb, std::vector<int32_t> &best,
                    std::vector<uint16_t> &d1, std::vector<uint16_t> &d2, int64_t idx_base = 0)
    {
        best.resize(nq), d1.resize(nq), d2.resize(nq);
        check(orbx_match_top2(h_, queries, nq, db, ndb, idx_base, best.data(), d1.data(), d2.data()),
              "orbx_match_top2");
    }
    // d1 <= th && d1 < nnratio * d2 (src/ORBmatcher.cc:245-249); returns the number accepted.
    int ratio_test(const std::vector<uint16_t> &d1, const std::vector<uint16_t> &d2, std::vector<uint8_t> &accept,
                   int th = TH_LOW) const
    {
        accept.resize(d1.size());
        const int n = orbx_ratio_test((int)d1.size(), d1.data(), d2.data(), th, nnratio_, accept.data());
        check(n, "orbx_ratio_test");
        return n;
    }

    // ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:433-549).  vbPrevMatched holds one
    // (x, y) pair per keypoint of frame 1 and is updated; vnMatches12 receives indices into frame 2.
    int SearchForInitialization(const std::vector<orbx_keypoint> &k1, const std::vector<uint8_t> &d1,
                                const std::vector<orbx_keypoint> &k2, const std::vector<uint8_t> &d2, const Bounds &b,
                                std::vector<float> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10)
    {
        if (vbPrevMatched.size() != k1.size() * 2)
            throw Error(ORBX_E_ARG, "SearchForInitialization: vbPrevMatched needs 2 floats per keypoint");
        vnMatches12.assign(k1.size(), -1);
        const int n = orbx_search_for_initialization(h_, k1.data(), d1.data(), (int)k1.size(), k2.data(), d2.data(),
                                                     (int)k2.size(), b.min_x, b.max_x, b.min_y, b.max_y,
                                                     vbPrevMatched.data(), windowSize, nnratio_, check_ori_,
                                                     vnMatches12.data(), nullptr);
        check(n, "orbx_search_for_initialization");
        return n;
    }

    // Frame::UndistortKeyPoints (src/Frame.cc:467-499) on host keypoints: positions through
    // cv::undistortPoints with P = K, other fields copied.  cam = fx, fy, cx, cy; dist = k1, k2, p1, p2[, k3].
    std::vector<orbx_keypoint> UndistortKeyPoints(const std::vector<orbx_keypoint> &keys, const float cam[4],
                                                  const std::vector<float> &dist)
    {
        std::vector<orbx_keypoint> out(keys);
        if (keys.empty() || dist.empty() || dist[0] == 0.f)
            return out;
        std::vector<float> xy(keys.size() * 2);
        for (size_t i = 0; i < keys.size(); i++)
            xy[2 * i] = keys[i].x, xy[2 * i + 1] = keys[i].y;
        check(orbx_undistort_points(h_, xy.data(), (int)keys.size(), cam, dist.data(), (int)dist.size(), xy.data()),
              "orbx_undistort_points");
        for (size_t i = 0; i < keys.size(); i++)
            out[i].x = xy[2 * i], out[i].y = xy[2 * i + 1];
        return out;
    }
    // Frame::ComputeImageBounds (src/Frame.cc:501-523)
    Bounds ComputeImageBounds(int cols, int rows, const float cam[4], const std::vector<float> &dist)
    {
        if (dist.empty() || dist[0] == 0.f)
            return Bounds{0.f, (float)cols, 0.f, (float)rows};
        float c[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
        check(orbx_undistort_points(h_, c, 4, cam, dist.data(), (int)dist.size(), c), "orbx_undistort_points");
        auto mn = [](float a, float b) { return a < b ? a : b; };
        auto mx = [](float a, float b) { return a > b ? a : b; };
        return Bounds{mn(c[0], c[4]), mx(c[2], c[6]), mn(c[1], c[3]), mx(c[5], c[7])};
    }

    // Frame::GetFeaturesInArea (src/Frame.cc:349-407) for nq windows at once over the grid of
    // Frame::AssignFeaturesToGrid; window q's indices are out[q], in the reference's order.
    void GetFeaturesInArea(const std::vector<orbx_keypoint> &keys, const Bounds &b, const std::vector<float> &x,
                           const std::vector<float> &y, const std::vector<float> &r,
                           const std::vector<int32_t> &minLevel, const std::vector<int32_t> &maxLevel,
                           std::vector<std::vector<int32_t>> &out, int cap = 512)
    {
        const int nq = (int)x.size();
        std::vector<int32_t> idx((size_t)nq * cap), cnt(nq);
        check(orbx_features_in_area(h_, keys.data(), (int)keys.size(), b.min_x, b.max_x, b.min_y, b.max_y, x.data(),
                                    y.data(), r.data(), minLevel.data(), maxLevel.data(), nq, cap, idx.data(),
                                    cnt.data(), nullptr, nullptr),
              "orbx_features_in_area");
        out.resize(nq);
        for (int q = 0; q < nq; q++) {
            if (cnt[q] > cap)
                throw Error(ORBX_E_CAPACITY, "GetFeaturesInArea: window holds more than cap features");
            out[q].assign(idx.begin() + (size_t)q * cap, idx.begin() + (size_t)q * cap + cnt[q]);
        }
    }

    // ORBmatcher::SearchByProjection from the projections of the map points (x, y, window radius, level
    // window, right coordinate, descriptor per source); returns the number of matches, match[s] = feature
    // or -1, feature_match[i] = source or -1 (-2: cleared by the rotation check).  form selects the rules:
    //   MapPoints  (Frame&, const vector<MapPoint*>&, th), src/ORBmatcher.cc:46-137: best and second best;
    //   LastFrame  (Frame&, const Frame&, th, bMono), :1412-1566: best only, rotation check;
    //   KeyFrame   (Frame&, KeyFrame*, sAlreadyFound, th, ORBdist), :1568-1690: best only against
    //              th_dist = ORBdist, rotation check with the histogram factor 1/HISTO_LENGTH.
    enum Form { MapPoints = 0, LastFrame = 1, KeyFrame = 2 };
    struct Sources {
        std::vector<uint8_t> valid;      // empty: all
        std::vector<float> x, y, r, xr;  // xr empty: monocular
        std::vector<int32_t> min_level, max_level;
        std::vector<uint8_t> desc;     // 32 bytes per source
        std::vector<uint8_t> blocks;   // empty: every source's map point has observations
        std::vector<float> angle;      // last-frame form with the rotation check only
    };
    int SearchByProjection(const std::vector<orbx_keypoint> &keys, const std::vector<uint8_t> &desc,
                           const std::vector<float> &uright, const std::vector<uint8_t> &taken, const Bounds &b,
                           const Sources &s, int form, std::vector<int32_t> &match,
                           std::vector<int32_t> &feature_match, int th_dist = TH_HIGH)
    {
        const int ns = (int)s.x.size();
        match.assign(ns, -1);
        feature_match.assign(keys.size(), -1);
        int32_t nm = 0;
        check(orbx_search_by_projection(h_, keys.data(), desc.data(), opt(uright), opt(taken), (int)keys.size(),
                                        b.min_x, b.max_x, b.min_y, b.max_y, opt(s.valid), s.x.data(), s.y.data(),
                                        s.r.data(), s.min_level.data(), s.max_level.data(), opt(s.xr), s.desc.data(),
                                        opt(s.blocks), opt(s.angle), ns, form == MapPoints ? 1 : 0, nnratio_, th_dist,
                                        (form != MapPoints && check_ori_) ? form : 0, match.data(),
                                        feature_match.data(), &nm),
              "orbx_search_by_projection");
        return nm;
    }

    // MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:248-319) for all segments at once.
    void ComputeDistinctiveDescriptors(const std::vector<uint8_t> &descriptors, const std::vector<int32_t> &seg_start,
                                       std::vector<int32_t> &best_idx, std::vector<uint8_t> &best_desc)
    {
        const int np = (int)seg_start.size() - 1;
        best_idx.resize(np > 0 ? np : 0);
        best_desc.resize((size_t)(np > 0 ? np : 0) * 32);
        if (np <= 0)
            return;
        check(orbx_distinctive_descriptors(h_, descriptors.data(), seg_start.data(), np, best_idx.data(),
                                           best_desc.data()),
              "orbx_distinctive_descriptors");
    }

private:
    orbx_matcher *h_ = nullptr;
    float nnratio_;
    bool check_ori_;
};

// Frame::ComputeStereoMatches (src/Frame.cc:534-708) on the images the two extractors processed last:
// fills mvuRight / mvDepth (one entry per left keypoint, -1 where unmatched), returns the match count.
inline int compute_stereo_matches(Extractor &left, Extractor &right, float mb, float mbf, std::vector<float> &uright,
                                  std::vector<float> &depth)
{
    const int cap = left.max_keypoints();
    uright.assign(cap, -1.0f);
    depth.assign(cap, -1.0f);
    int n = 0;
    const int nm = orbx_compute_stereo_matches(left.handle(), right.handle(), 0, mb, mbf, uright.data(), depth.data(),
                                               cap, &n);
    check(nm, "orbx_compute_stereo_matches");
    uright.resize(n);
    depth.resize(n);
    return nm;
}

}  // namespace orbx
