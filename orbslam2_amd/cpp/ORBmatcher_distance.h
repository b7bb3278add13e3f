// Drop-in body for ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1744-1760 of ORB-SLAM2): stays a
// pure host function (it is called per pair from three threads), now one line over liborbx.
//
//   int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
//   { return orbx_descriptor_distance(a.ptr<uint8_t>(), b.ptr<uint8_t>()); }
//
// Batched matching (brute-force top-2, SearchForInitialization windows) goes through
// orbx_match_top2 / orbx_search_for_initialization of include/orbx.h.
#pragma once
#include "orbx.h"
