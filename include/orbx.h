/*
 * orbx.h -- C ABI of the MI355X ORB front end (extractor + Hamming matcher).
 *
 * Every entry point takes plain pointers and sizes.  Return value: 0 = ok, negative = error
 * (see orbx_strerror).  Citations name the interface of the reference (lianbin/orbslam2) that
 * the entry point replaces, as file:line relative to the reference root.
 *
 * Threading: a handle is not re-entrant (one call at a time per handle, like one
 * ORBextractor instance); different handles may be used concurrently from different threads
 * (the reference runs the left and right extractor in two std::threads, src/Frame.cc:79-82).
 */
#ifndef ORBX_H
#define ORBX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- errors */
enum {
    ORBX_OK = 0,
    ORBX_E_ARG = -1,       /* null pointer, non-positive size, unsupported parameter */
    ORBX_E_HIP = -2,       /* a HIP runtime call failed (orbx_last_hip_error has the code) */
    ORBX_E_CAPACITY = -3,  /* caller buffer or handle workspace too small */
    ORBX_E_GEOMETRY = -4,  /* image too tall for the quadtree: round(width/height) of a level is 0
                              (the reference divides by zero there, src/ORBextractor.cc:584-589) */
    ORBX_E_NODEVICE = -5,  /* no usable GPU */
    ORBX_E_STATE = -6,     /* call order violated (no batch extracted yet, ...) */
    ORBX_E_RCCL = -7       /* RCCL is not installed or one of its calls failed (orbx_sharded_last_error has the text) */
};
const char *orbx_strerror(int code);
int orbx_last_hip_error(void);
/* library version (major*10000 + minor*100 + patch) */
int orbx_version(void);

/* ---------------------------------------------------------------- extractor */

/* Constructor arguments of ORB_SLAM2::ORBextractor (include/ORBextractor.h:51-52,
 * src/ORBextractor.cc:432-506) plus the workspace bounds of this implementation. */
typedef struct orbx_params {
    int nfeatures;      /* ORBextractor.nFeatures */
    float scale_factor; /* ORBextractor.scaleFactor */
    int nlevels;        /* ORBextractor.nLevels (1..16) */
    int ini_th_fast;    /* ORBextractor.iniThFAST */
    int min_th_fast;    /* ORBextractor.minThFAST */
    int blur_variant;   /* 0: 7x7 sigma-2 kernel 18,34,48,56,48,34,18 /256 (OpenCV >= 3.4.1 fixed point);
                           1: 18,34,49,55,49,34,18 /256 saturating (OpenCV 2.4 / <= 3.4.0) */
    int max_width;      /* largest image the handle will see */
    int max_height;
    int max_batch;      /* largest number of frames per batch call (>= 1) */
    int device;         /* HIP device ordinal */
} orbx_params;

/* Same memory layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id):
 * 28 bytes, so a std::vector<cv::KeyPoint> can be filled with one memcpy. */
typedef struct orbx_keypoint {
    float x, y;     /* level-0 pixel coordinates (level coordinate * scale factor of the octave) */
    float size;     /* (int)(31 * scale) */
    float angle;    /* degrees, [0,360) */
    float response; /* FAST score */
    int32_t octave;
    int32_t class_id; /* always -1 */
} orbx_keypoint;

typedef struct orbx_extractor orbx_extractor;

/* replaces ORBextractor::ORBextractor (src/ORBextractor.cc:432-506).  Allocates all device
 * workspaces for max_batch frames of max_width x max_height. */
int orbx_extractor_create(const orbx_params *params, orbx_extractor **out);
void orbx_extractor_destroy(orbx_extractor *ex);

/* Work is enqueued on this stream (a hipStream_t; NULL = the handle's own stream). */
int orbx_extractor_set_stream(orbx_extractor *ex, void *hip_stream);

/* Extractor handles of one device that take turns on batches (the counterpart of several ORBextractor objects
 * running in threads of their own, src/Frame.cc:79-82, for callers that keep a GPU busy): attach them to one
 * pipeline, give each a stream of its own (orbx_extractor_set_stream or the handle's own) and call them round
 * robin.  A handle then enqueues the blur of its batch (more than four frames) behind the work of the handle that
 * was called before it, on that handle's stream, instead of on a second stream of its own; results and the order
 * seen on the handle's own stream are unchanged.  Four handles with 256 frames each reach 258 k KITTI frames/s this
 * way against 247 k with eight independent streams.  The streams must map to different hardware queues: set
 * GPU_MAX_HW_QUEUES to 8 or more before the first HIP call (HIP's default of 4 makes some of them share a queue,
 * and streams that share a queue run in turn).  A handle's stream must stay valid while it is attached.
 * set_pipeline(NULL) detaches.  Calls on handles of one pipeline must not run concurrently on the host. */
typedef struct orbx_pipeline orbx_pipeline;
int orbx_pipeline_create(int device, orbx_pipeline **out);
void orbx_pipeline_destroy(orbx_pipeline *p);
int orbx_extractor_set_pipeline(orbx_extractor *ex, orbx_pipeline *p);

/* replaces GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:63-83); each array has nlevels entries. */
int orbx_extractor_levels(const orbx_extractor *ex);
int orbx_extractor_scale_tables(const orbx_extractor *ex, float *scale, float *inv_scale, float *sigma2,
                                float *inv_sigma2);
/* mnFeaturesPerLevel (src/ORBextractor.cc:460-475), nlevels entries */
int orbx_extractor_features_per_level(const orbx_extractor *ex, int *quota);
/* upper bound of keypoints one frame can produce (sum over levels of quota+3, at least 4*nIni) */
int orbx_extractor_max_keypoints(const orbx_extractor *ex);

/* replaces ORBextractor::operator() (src/ORBextractor.cc:1114-1182) for one host image:
 * uploads, extracts, downloads, synchronises.  image: height rows of stride_bytes, 8-bit gray.
 * keypoints/descriptors: caller buffers for cap entries (cap*28 and cap*32 bytes); *n_out receives
 * the number of keypoints (rows of descriptors), level-major, in-level order = quadtree list order.
 * An empty image (NULL or zero size) returns ORBX_OK with *n_out = -1 and the outputs untouched,
 * as the reference returns early (:1117-1118).  ORBX_E_CAPACITY if cap is too small (nothing
 * written except *n_out = required count). */
int orbx_extract(orbx_extractor *ex, const uint8_t *image, int width, int height, int stride_bytes,
                 orbx_keypoint *keypoints, uint8_t *descriptors, int cap, int *n_out);

/* Throughput form: batch frames already resident in device memory (frame f starts at
 * d_images + f*frame_stride_bytes).  Asynchronous on the handle's stream; results stay in the
 * handle's device buffers until the next batch call.  The images must stay valid and unchanged
 * until the stream has finished the batch.  When d_images, stride_bytes and frame_stride_bytes are
 * multiples of 4 the frames are read in place (no copy into the pyramid); orbx_pyramid_level(.., 0, ..)
 * then reads them again, so they must also stay valid for such a call. */
int orbx_extract_batch_device(orbx_extractor *ex, const uint8_t *d_images, int batch, int width, int height,
                              int stride_bytes, int64_t frame_stride_bytes);
/* Same, frames in host memory (uploaded with one async copy per frame). */
int orbx_extract_batch_host(orbx_extractor *ex, const uint8_t *const *images, int batch, int width, int height,
                            int stride_bytes);

/* Blocks until the last batch is finished and returns the per-frame keypoint counts
 * (batch entries) and, if not NULL, the per-level counts (batch*nlevels).  A negative return
 * reports a device-side failure of the batch (ORBX_E_CAPACITY / ORBX_E_GEOMETRY). */
int orbx_batch_counts(orbx_extractor *ex, int *n_per_frame, int *n_per_level);
/* Downloads the keypoints and descriptors of one frame of the last batch. */
int orbx_batch_download(orbx_extractor *ex, int frame, orbx_keypoint *keypoints, uint8_t *descriptors, int cap,
                        int *n_out);
/* Downloads the whole last batch with one synchronisation: frame f's keypoints start at
 * keypoints + f*cap_per_frame, its descriptors at descriptors + f*cap_per_frame*32, counts[f] of them
 * are valid.  cap_per_frame >= orbx_extractor_max_keypoints. */
int orbx_batch_download_all(orbx_extractor *ex, orbx_keypoint *keypoints, uint8_t *descriptors, int cap_per_frame,
                            int *counts);
/* Device addresses of the result arrays of the last batch: keypoints (frame f at
 * +f*max_keypoints entries), descriptors (32 bytes each, same indexing), per-frame counts. */
int orbx_batch_device_results(orbx_extractor *ex, const orbx_keypoint **d_keypoints, const uint8_t **d_descriptors,
                              const int32_t **d_counts, int *max_keypoints);

/* Per-stage device time of the batches extracted while profiling is enabled, measured with HIP
 * events recorded around each stage.  Normally the blur runs on a second stream beside detection
 * and distribution; while profiling is enabled every stage runs on the main stream, so the stage
 * times are exclusive (and a batch takes a little longer than unprofiled).  The reference has no
 * counterpart (it only wall-clocks whole Track calls, Examples/Stereo/stereo_kitti.cc:80-97).
 * Stages: 0 ingest (copy of the caller's frames into level 0), 1 resize (levels 1..n-1, one kernel
 * per level), 2 fast (the detection kernel alone: score, per-cell non-maximum suppression, tile
 * lists), 3 octree (candidate compaction, threshold fallback and quadtree distribution), 4 blur (one
 * kernel), 5 describe (orientation, rBRIEF, output; one kernel). */
#define ORBX_NSTAGES 6
const char *orbx_stage_name(int stage);
int orbx_extractor_profile(orbx_extractor *ex, int enable); /* enabling clears earlier samples */
int orbx_extractor_stage_times(orbx_extractor *ex, float *ms_sum /* ORBX_NSTAGES */, int *nbatches);

/* replaces reads of the public member mvImagePyramid (include/ORBextractor.h:85; read by
 * src/Frame.cc:541,631,643,648): copies level `level` of frame `frame` of the last batch into
 * dst (dst_stride bytes per row; dst may be NULL to query the size).  border > 0 additionally
 * writes a BORDER_REFLECT_101 frame of that many pixels around it, dst then points at the
 * top-left of the padded buffer of (w+2*border) x (h+2*border) (the reference keeps border 19,
 * src/ORBextractor.cc:1199-1219). */
int orbx_pyramid_level(orbx_extractor *ex, int frame, int level, int border, uint8_t *dst, int dst_stride,
                       int *width, int *height);
/* Lifetime: after orbx_extract_batch_device read the frames in place (4-byte aligned address and strides),
 * level 0 of the pyramid IS the caller's device memory: orbx_pyramid_level(.., 0, ..) and
 * orbx_pyramid_download read it again, so the frames of the last device batch must stay valid and unchanged
 * until the pyramid has been fetched.  Host-frame calls (orbx_extract, orbx_extract_batch_host) keep their own
 * device copy and have no such requirement. */

/* The whole of mvImagePyramid for one frame of the last batch in one round trip (what
 * ORBextractor::ComputePyramid leaves behind, src/ORBextractor.cc:1191-1221): level l is written as a
 * dense (w_l+2*border) x (h_l+2*border) image, BORDER_REFLECT_101 frame included, at dst + offsets[l]
 * with a row stride of w_l+2*border bytes.  orbx_pyramid_layout reports offsets, level sizes (nlevels
 * entries each; any may be NULL) and the bytes dst must hold.  One device-to-host copy of the frame's
 * levels and one synchronisation, against two per level through orbx_pyramid_level. */
int orbx_pyramid_layout(const orbx_extractor *ex, int border, int64_t *offsets, int *widths, int *heights,
                        int64_t *total_bytes);
int orbx_pyramid_download(orbx_extractor *ex, int frame, int border, uint8_t *dst, int64_t dst_bytes);

/* Stage output of the last batch, for parity tests.  what: 1 = blurred level (u8, level-sized). */
int orbx_debug_level_map(orbx_extractor *ex, int what, int frame, int level, uint8_t *dst, int dst_stride);
/* Candidate list of a level before the quadtree (src/ORBextractor.cc:880-890): coordinates
 * relative to (16,16), sorted into the reference's order (cells row-major, row-major inside a cell),
 * after the per-cell iniThFAST -> minThFAST decision.  Returns the count through *n_out. */
int orbx_debug_candidates(orbx_extractor *ex, int frame, int level, int *xs, int *ys, int *resp, int cap,
                          int *n_out);
/* Quadtree distribution alone (ORBextractor::DistributeOctTree, src/ORBextractor.cc:578-813) on n
 * host candidates with coordinates relative to a region_w x region_h detection region and quota N:
 * out_idx receives the kept candidate indices in list order. */
int orbx_debug_octree(int device, const int *xs, const int *ys, const int *resp, int n, int region_w, int region_h,
                      int quota, int *out_idx, int cap, int *n_out);

/* ---------------------------------------------------------------- stereo matching */

/* Frame::ComputeStereoMatches (src/Frame.cc:534-708) for every frame of the batches that a left and
 * a right extractor handle processed last (same image size, settings, batch size and device; frame f
 * of the left batch pairs with frame f of the right batch).  For each left keypoint: the right
 * keypoints listed in its image row (the reference's vRowIndices), octave within +-1, u in
 * [uL - mbf/mb, uL]; best descriptor distance (first index wins ties) below (TH_HIGH+TH_LOW)/2;
 * 11x11 window slid over 11 positions at the keypoint's pyramid level; parabola fit; finally the
 * matches whose window distance reaches 1.5*1.4*median are dropped.  mb = stereo baseline (mb of the
 * Frame), mbf = baseline * fx.  d_uright / d_depth: device arrays of batch x max_keypoints floats
 * (frame f at +f*max_keypoints; mvuRight / mvDepth, -1 where unmatched; entries past a frame's
 * keypoint count are -1); d_nmatches: matches per frame.  Asynchronous on the left handle's stream;
 * the right handle's stream is ordered before and after it.  Keypoints whose row lies outside the
 * image (the reference would index out of range) stay unmatched; a frame without any match is left
 * as it is (the reference reads the median of an empty list). */
int orbx_compute_stereo_matches_device(orbx_extractor *left, orbx_extractor *right, float mb, float mbf,
                                       float *d_uright, float *d_depth, int32_t *d_nmatches);
/* Host form for one frame of the batch: fills uright / depth (cap entries available) for the
 * *n_keypoints left keypoints of `frame`; returns the number of matches or a negative error.  The
 * batch is matched once and kept until either handle processes a new batch. */
int orbx_compute_stereo_matches(orbx_extractor *left, orbx_extractor *right, int frame, float mb, float mbf,
                                float *uright, float *depth, int cap, int *n_keypoints);

/* ---------------------------------------------------------------- matcher */

/* replaces ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1744-1760): Hamming distance of two
 * 256-bit descriptors.  Pure host function, thread-safe, never touches the GPU. */
int orbx_descriptor_distance(const uint8_t *a, const uint8_t *b);

/* ORBmatcher::TH_HIGH / TH_LOW / HISTO_LENGTH (src/ORBmatcher.cc:37-39) */
#define ORBX_TH_HIGH 100
#define ORBX_TH_LOW 50
#define ORBX_HISTO_LENGTH 30

typedef struct orbx_matcher orbx_matcher;
int orbx_matcher_create(int device, int max_queries, orbx_matcher **out);
void orbx_matcher_destroy(orbx_matcher *m);
int orbx_matcher_set_stream(orbx_matcher *m, void *hip_stream);

/* Brute-force top-2 of every query against a descriptor database (shard), the inner loop of
 * ORBmatcher::SearchByBoW (src/ORBmatcher.cc:217-242) over a flat candidate list: best and
 * second-best start at 256, strict '<' updates in database order (first index wins ties).
 * best_idx = idx_base + row of the best entry, or -1 when the shard is empty.
 * Device form: all pointers are device addresses, asynchronous on the matcher's stream. */
int orbx_match_top2_device(orbx_matcher *m, const uint8_t *d_queries, int nq, const uint8_t *d_db, int64_t ndb,
                           int64_t idx_base, int32_t *d_best_idx, uint16_t *d_d1, uint16_t *d_d2);
/* Host form: uploads queries and database, runs, downloads, synchronises. */
int orbx_match_top2(orbx_matcher *m, const uint8_t *queries, int nq, const uint8_t *db, int64_t ndb,
                    int64_t idx_base, int32_t *best_idx, uint16_t *d1, uint16_t *d2);

/* orbx_match_top2_device for callers that cannot allocate.  It has no counterpart in the reference: it
 * is for code under HIP stream capture, tight serving loops and harnesses that forbid hipMalloc.  No
 * handle: the call makes no hipMalloc, hipFree or other allocating call.  The per-chunk partial results,
 * which the handle form keeps in a buffer it allocates and grows, live in an 8 MB array inside the
 * library's device code (every GPU the library is used on holds its own copy until the process ends).
 * Same kernels, same results.  Limit: nq * (number of database chunks) <= 1,048,576 partial results; the
 * database is cut into at most ceil(ndb / 32736) chunks to fit (2000 queries against 1,000,000 rows need
 * 510,000), beyond that the call returns ORBX_E_CAPACITY and launches nothing.  device = HIP ordinal (made
 * current), hip_stream = a hipStream_t (NULL = the default stream); all pointers are device addresses;
 * asynchronous on that stream.  Calls that overlap on one GPU (two streams, two threads) share the array
 * and race: serialise them per GPU.  The first call on a device also asks the runtime for the array's
 * address, so make one call before capturing a stream. */
int orbx_match_top2_device_noalloc(int device, void *hip_stream, const uint8_t *d_queries, int nq,
                                   const uint8_t *d_db, int64_t ndb, int64_t idx_base, int32_t *d_best_idx,
                                   uint16_t *d_d1, uint16_t *d_d2);

/* Merges per-shard top-2 results (nshards arrays of nq entries, shard s at offset s*nq, shards in
 * ascending index order) into the result of one pass over the concatenated database. Host function. */
int orbx_top2_merge(int nshards, int nq, const int32_t *best_idx, const uint16_t *d1, const uint16_t *d2,
                    int32_t *out_idx, uint16_t *out_d1, uint16_t *out_d2);
/* The same merge on the device, for a database sharded over GPUs: shard s's results are the block of
 * 8*nq bytes at d_blocks + s*shard_stride_bytes holding nq best indices (int32), nq best distances and nq
 * second distances (uint16 each) -- what orbx_match_top2_device writes when its three output pointers
 * are block, block + 4*nq and block + 6*nq, and what an all-gather of such blocks lays side by side.
 * Asynchronous on the matcher's stream. */
int orbx_top2_merge_shards_device(orbx_matcher *m, int nshards, int nq, const uint8_t *d_blocks,
                                  int64_t shard_stride_bytes, int32_t *d_best_idx, uint16_t *d_d1, uint16_t *d_d2);
/* The brute-force top-2 against a database whose rows are sharded over the GPUs of a node, for a C++ caller
 * with one thread or process per GPU (the reference has no such step: Tracking::Relocalization,
 * src/Tracking.cc:1554-1590, runs SearchByBoW per candidate key frame on one core; this is that inner loop,
 * src/ORBmatcher.cc:217-249, over a flat database too large or too hot for one GPU).  Rank r holds the
 * contiguous rows [idx_base, idx_base + shard_rows) of the database and all queries.  A call runs the top-2
 * kernel on the shard, one ncclAllGather (RCCL over xGMI) of 8 bytes per query and rank, and the merge kernel
 * of orbx_top2_merge_shards_device, all on one stream: every rank ends up with the result of a single pass
 * over the whole database (ties go to the lower row).  Shards must be in ascending row order by rank.
 * RCCL is looked up when the first of these functions is called (dlopen of librccl.so.1; the environment
 * variable ORBX_RCCL_LIB names another file); without it they return ORBX_E_RCCL and the rest of the
 * library is unaffected.
 *   orbx_sharded_unique_id       rank 0 calls it and hands the ORBX_RCCL_UNIQUE_ID_BYTES bytes to the other
 *                                ranks by its own means (the ncclUniqueId of ncclGetUniqueId)
 *   orbx_sharded_matcher_create  collective: every rank calls it with the same id (ncclCommInitRank)
 *   orbx_match_top2_sharded_device  collective; device pointers; asynchronous on the handle's stream (its own
 *                                unless set): order the inputs before it and wait for that stream, or call
 *                                orbx_sharded_matcher_synchronize, before reading the results.  nq <= max_queries. */
#define ORBX_RCCL_UNIQUE_ID_BYTES 128
typedef struct orbx_sharded_matcher orbx_sharded_matcher;
int orbx_sharded_unique_id(void *id_out);
int orbx_sharded_matcher_create(int device, int rank, int world, const void *unique_id, int max_queries,
                                orbx_sharded_matcher **out);
void orbx_sharded_matcher_destroy(orbx_sharded_matcher *sm);
int orbx_sharded_matcher_set_stream(orbx_sharded_matcher *sm, void *hip_stream);
int orbx_sharded_matcher_synchronize(orbx_sharded_matcher *sm);
int orbx_match_top2_sharded_device(orbx_sharded_matcher *sm, const uint8_t *d_queries, int nq, const uint8_t *d_shard,
                                   int64_t shard_rows, int64_t idx_base, int32_t *d_best_idx, uint16_t *d_d1,
                                   uint16_t *d_d2);
const char *orbx_sharded_last_error(void);

/* Acceptance rule of src/ORBmatcher.cc:245-249: d1 <= th && (float)d1 < ratio*(float)d2.
 * accept[i] = 1/0; returns the number accepted. Host function. */
int orbx_ratio_test(int nq, const uint16_t *d1, const uint16_t *d2, int th, float ratio, uint8_t *accept);

/* replaces ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:433-549) with the candidate
 * windows of Frame::GetFeaturesInArea (src/Frame.cc:349-407) over the 64x48 grid of
 * Frame::AssignFeaturesToGrid (src/Frame.cc:239-253).  Host keypoints/descriptors of both frames;
 * (min_x,max_x,min_y,max_y) are Frame::mnMinX.. (image bounds); prev_matched holds n1 (x,y) pairs,
 * read and updated (vbPrevMatched); matches12 receives n1 indices into frame 2 or -1.
 * The window distances are computed on the GPU, the order-dependent greedy pass is replayed on
 * the host.  Returns the number of matches (>= 0) or a negative error. */
int orbx_search_for_initialization(orbx_matcher *m, const orbx_keypoint *k1, const uint8_t *d1, int n1,
                                   const orbx_keypoint *k2, const uint8_t *d2, int n2, float min_x, float max_x,
                                   float min_y, float max_y, float *prev_matched, int window_size, float nnratio,
                                   int check_orientation, int *matches12, int64_t *pair_count);

/* The same search for every consecutive pair (frame t, frame t+1), t = 0 .. nframes-2, of a batch
 * whose keypoints and descriptors are in device memory in the layout of orbx_batch_device_results
 * (frame f at +f*max_keypoints entries).  Runs entirely on the GPU, asynchronously on the matcher's
 * stream: one wave per pair replays the order-dependent pass of src/ORBmatcher.cc:446-516 and the
 * rotation check (:502-541).  All pointers are device pointers.
 *   d_prev_matched      (nframes-1) x max_keypoints x 2 floats, the search centres (vbPrevMatched); NULL:
 *                       the keypoints' own positions, as Tracking::MonocularInitialization sets them
 *                       (src/Tracking.cc:585-587)
 *   d_prev_matched_out  same shape or NULL; receives the matched positions (:544-546), other entries untouched
 *   max_level0          upper bound of level-0 keypoints per frame (sizes the per-pair workspace in LDS;
 *                       at most 1024).  A pair that exceeds it reports ORBX_E_CAPACITY in d_nmatches.
 *   d_matches12         (nframes-1) x max_keypoints: index into frame t+1 or -1
 *   d_nmatches, d_pair_counts  per pair: matches, and candidates whose distance was evaluated */
int orbx_search_for_initialization_batch_device(orbx_matcher *m, const orbx_keypoint *d_keypoints,
                                                const uint8_t *d_descriptors, const int32_t *d_counts,
                                                int max_keypoints, int nframes, int max_level0, float min_x,
                                                float max_x, float min_y, float max_y, const float *d_prev_matched,
                                                float *d_prev_matched_out, int window_size, float nnratio,
                                                int check_orientation, int32_t *d_matches12, int32_t *d_nmatches,
                                                int32_t *d_pair_counts);

/* The 64x48 lookup grid of Frame (include/Frame.h:38-39).  orbx_frame_grid_build_device is
 * Frame::AssignFeaturesToGrid with Frame::PosInGrid (src/Frame.cc:239-253, 409-420) for n <= 32768
 * keypoints in device memory (bounds = mnMinX, mnMaxX, mnMinY, mnMaxY): d_cell_start receives 64*48+1
 * offsets (cell = column*48 + row), d_cell_items the keypoint indices cell by cell, ascending inside a
 * cell as the reference appends them; keypoints outside the grid are in no cell.
 * orbx_features_in_area_device is Frame::GetFeaturesInArea (:349-407) for nq queries (x, y, r, minLevel,
 * maxLevel) at once: query q's indices go to d_out_idx + q*cap in the reference's order (cell column,
 * cell row, keypoint index), d_out_n[q] is their number (the list is truncated if it exceeds cap).
 * Device pointers, asynchronous on the matcher's stream. */
int orbx_frame_grid_build_device(orbx_matcher *m, const orbx_keypoint *d_keypoints, int n, float min_x, float max_x,
                                 float min_y, float max_y, int32_t *d_cell_start, int32_t *d_cell_items);
int orbx_features_in_area_device(orbx_matcher *m, const orbx_keypoint *d_keypoints, float min_x, float max_x,
                                 float min_y, float max_y, const int32_t *d_cell_start, const int32_t *d_cell_items,
                                 const float *d_x, const float *d_y, const float *d_r, const int32_t *d_min_level,
                                 const int32_t *d_max_level, int nq, int cap, int32_t *d_out_idx, int32_t *d_out_n);
/* ORBmatcher::SearchByProjection in the two forms the tracking thread runs on every frame:
 *   use_second = 1: (Frame &F, const vector<MapPoint*> &vpMapPoints, th), src/ORBmatcher.cc:46-137;
 *   use_second = 0: (Frame &CurrentFrame, const Frame &LastFrame, th, bMono), :1412-1566, with its
 *                   rotation check when check_orientation is 1.  The relocalisation form (Frame &CurrentFrame,
 *                   KeyFrame *pKF, sAlreadyFound, th, ORBdist), :1568-1690, is this search with th_dist =
 *                   ORBdist, taken[i] = mvpMapPoints[i] != NULL (:1636-1637), src_angle = pKF->mvKeysUn[i].angle
 *                   and check_orientation = 2: its histogram bins with 1.0f/HISTO_LENGTH (:1581) where the
 *                   other form uses HISTO_LENGTH/360.0f (:1420).
 * The frame: n <= 32768 keypoints (mvKeysUn) with descriptors, uright = mvuRight or NULL (monocular),
 * taken[i] != 0 where mvpMapPoints[i] has Observations() > 0 (NULL: none), the grid bounds.
 * The ns sources, in the reference's order, after what precedes the window query there: valid[s] = the
 * source is searched at all (mbTrackInView and not bad, :55-59; a last-frame map point that is no outlier
 * and projects inside the image, :1438-1462; NULL: all), (x, y) = projection, r = window radius
 * (r * mvScaleFactors[level] of :67-72; th * mvScaleFactors[octave] of :1465), min_level / max_level =
 * the level arguments of GetFeaturesInArea (:72, :1469-1474; max_level -1 = open), xr = predicted right
 * coordinate (mTrackProjXR; u - mbf * invzc) or NULL with uright NULL, src_desc = the map points'
 * descriptors, blocks[s] != 0 where the source's map point has Observations() > 0, so that the feature
 * it wins is skipped by later sources (NULL: all), src_angle = mvKeysUn[i].angle of the last frame
 * (used with check_orientation only).  th_dist = TH_HIGH; nnratio as in the matcher.
 * Results: match[s] = feature index or -1; feature_match[i] = last source written to feature i, -1 if
 * none, -2 if the rotation check cleared it (:1553); *nmatches as the reference returns it.  A window
 * holding more than 256 candidates returns ORBX_E_CAPACITY. */
int orbx_search_by_projection(orbx_matcher *m, const orbx_keypoint *keypoints, const uint8_t *descriptors,
                              const float *uright, const uint8_t *taken, int n, float min_x, float max_x, float min_y,
                              float max_y, const uint8_t *valid, const float *x, const float *y, const float *r,
                              const int32_t *min_level, const int32_t *max_level, const float *xr,
                              const uint8_t *src_desc, const uint8_t *blocks, const float *src_angle, int ns,
                              int use_second, float nnratio, int th_dist, int check_orientation, int32_t *match,
                              int32_t *feature_match, int32_t *nmatches);

/* Frame::UndistortKeyPoints (src/Frame.cc:467-499): cv::undistortPoints(pts, pts, mK, mDistCoef, Mat(), mK)
 * on the positions of n keypoints; the other fields are copied, so d_out is mvKeysUn.  cam = fx, fy, cx,
 * cy of mK; dist = k1, k2, p1, p2 and, with ndist == 5, k3 (mDistCoef).  As in the reference (:469-473),
 * dist[0] == 0 copies the keypoints unchanged.  d_out may equal d_keypoints.  Device pointers,
 * asynchronous on the matcher's stream. */
int orbx_undistort_keypoints_device(orbx_matcher *m, const orbx_keypoint *d_keypoints, int n, const float *cam,
                                    const float *dist, int ndist, orbx_keypoint *d_out);
/* The same on n (x, y) float pairs in host memory; with the four image corners it is the first half of
 * Frame::ComputeImageBounds (:503-517), whose min/max of the results the caller takes. */
int orbx_undistort_points(orbx_matcher *m, const float *xy, int n, const float *cam, const float *dist, int ndist,
                          float *xy_out);
/* Both steps from host arrays (keypoints and queries in, lists out); cell_start / cell_items may be
 * NULL when the grid itself is not wanted. */
int orbx_features_in_area(orbx_matcher *m, const orbx_keypoint *keypoints, int n, float min_x, float max_x, float min_y,
                          float max_y, const float *x, const float *y, const float *r, const int32_t *min_level,
                          const int32_t *max_level, int nq, int cap, int32_t *out_idx, int32_t *out_n,
                          int32_t *cell_start, int32_t *cell_items);

/* ORBmatcher::SearchByBoW for nb key frames ("side 1") against one frame or key frame ("side 2"):
 * src/ORBmatcher.cc:176-311 (key frame against frame: strict_lt = 0, valid2 = NULL; this is what
 * Tracking::Relocalization runs for every candidate key frame, src/Tracking.cc:1573-1586) and :552-686
 * (key frame against key frame: strict_lt = 1, valid2 given).  Host arrays in, host arrays out.
 * Side 1, all key frames concatenated: desc1 (32 bytes per feature), angle1 (keypoint angles),
 * valid1 (1 = the feature has a usable map point; NULL = all), feat_start[nb+1] (first feature of each
 * key frame).  Its feature vectors (DBoW2::FeatureVector: node id -> feature indices): the node
 * entries of all key frames concatenated, node_kf_start[nb+1] (first entry of each key frame),
 * node_ids1 (ascending inside a key frame), node_start1[entries+1] (offsets into node_items1),
 * node_items1 (feature indices local to the key frame, in the vector's order).
 * Side 2: desc2, angle2, valid2 (or NULL), n2 features; node_ids2 (ascending), node_start2[m2+1],
 * node_items2.  A node may hold at most 4096 features of side 2 (ORBX_E_CAPACITY otherwise).
 * match12 (one entry per feature of side 1): index into side 2 or -1; match21 (nb x n2): feature
 * index local to the key frame or -1; nmatches[nb].  The order-dependent greedy pass, the ratio test
 * and the rotation histogram with its three maxima follow the reference exactly. */
int orbx_search_by_bow(orbx_matcher *m, const uint8_t *desc1, const float *angle1, const uint8_t *valid1,
                       const int32_t *feat_start, int nb, const int32_t *node_kf_start, const int32_t *node_ids1,
                       const int32_t *node_start1, const int32_t *node_items1, const uint8_t *desc2,
                       const float *angle2, const uint8_t *valid2, int n2, const int32_t *node_ids2,
                       const int32_t *node_start2, const int32_t *node_items2, int m2, int strict_lt, float nnratio,
                       int check_orientation, int32_t *match12, int32_t *match21, int32_t *nmatches);

/* The same two searches for frames that stay in device memory: both sides in the layout of
 * orbx_batch_device_results (key frame b of side 1 at rows b*max_keypoints1 .., d_counts1[b] of them valid; side 2
 * is one frame of up to max_keypoints2 rows), angles read from the keypoint records (mvKeysUn[i].angle,
 * src/ORBmatcher.cc:252, :651), feature vectors as orbx_feature_vectors_build_device writes them.  valid1 /
 * valid2: one byte per row as above or NULL.  d_match12: nb x max_keypoints1 (index into side 2 or -1),
 * d_match21: nb x max_keypoints2 (feature of the key frame or -1), d_nmatches[nb]: matches of the key frame, or
 * ORBX_E_CAPACITY when one of its nodes holds more than 4096 features of side 2.  Device pointers; asynchronous
 * on the matcher's stream; no copy to or from the host. */
int orbx_search_by_bow_device(orbx_matcher *m, const orbx_keypoint *d_keypoints1, const uint8_t *d_desc1,
                              const int32_t *d_counts1, const uint8_t *d_valid1, int nb, int max_keypoints1,
                              const int32_t *d_node_ids1, const int32_t *d_node_start1, const int32_t *d_node_items1,
                              const orbx_keypoint *d_keypoints2, const uint8_t *d_desc2, const uint8_t *d_valid2,
                              int max_keypoints2, const int32_t *d_node_ids2, const int32_t *d_node_start2,
                              const int32_t *d_node_items2, int strict_lt, float nnratio, int check_orientation,
                              int32_t *d_match12, int32_t *d_match21, int32_t *d_nmatches);
/* DBoW2::FeatureVector of nframes device-resident frames from the per-feature node ids that
 * orbx_vocabulary_transform_device produced (the fv half of TemplatedVocabulary::transform,
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1145-1193, FeatureVector::addFeature,
 * Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:31-45; Frame::ComputeBoW, src/Frame.cc:422-431).  d_node: nframes x
 * max_keypoints node ids, frame f's first d_counts[f] rows valid; a negative id leaves the feature out (a word
 * of weight 0 is not added, :1157).  Per frame f: d_node_ids + f*max_keypoints receives the distinct node ids in
 * ascending order (the map's order), padded with INT32_MAX; d_node_start + f*(max_keypoints+1) the offset of each
 * entry's features in d_node_items + f*max_keypoints (feature indices local to the frame, ascending inside an
 * entry; padding entries are empty); d_num_nodes[f] the number of entries.  max_keypoints <= 8192.  Device
 * pointers; asynchronous on the matcher's stream. */
int orbx_feature_vectors_build_device(orbx_matcher *m, const int32_t *d_node, const int32_t *d_counts, int nframes,
                                      int max_keypoints, int32_t *d_node_ids, int32_t *d_node_start,
                                      int32_t *d_node_items, int32_t *d_num_nodes);

/* ---------------------------------------------------------------- vocabulary */

/* The descriptor -> word quantisation of DBoW2: TemplatedVocabulary<FORB::TDescriptor, FORB>::transform
 * for single features (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1259; FORB::distance,
 * Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101), which Frame::ComputeBoW / KeyFrame::ComputeBoW run for every
 * descriptor (src/Frame.cc:422-431 with levelsup = 4).  The tree is handed over flat, as the loaders
 * of the reference build it (:1350-1420): node 0 is the root, node_desc holds 32 bytes per node, the
 * children of node i are children[child_start[i] .. child_start[i+1]) in the stored order (every child id
 * is larger than its parent's), word_id and weight are read at leaves, depth = L of the vocabulary.
 * transform: per feature the word id, its weight and the id of the node passed at level
 * depth - levelsup (0 when that level is <= 0); the first child wins distance ties.  The caller folds
 * these into BowVector / FeatureVector (TemplatedVocabulary.h:1145-1193), which is map bookkeeping. */
typedef struct orbx_vocabulary orbx_vocabulary;
int orbx_vocabulary_create(int device, const uint8_t *node_desc, const int32_t *child_start, const int32_t *children,
                           const int32_t *word_id, const double *weight, int nnodes, int depth,
                           orbx_vocabulary **out);
void orbx_vocabulary_destroy(orbx_vocabulary *v);
/* device pointers in and out; asynchronous on `hip_stream` (NULL = the default stream) */
int orbx_vocabulary_transform_device(orbx_vocabulary *v, void *hip_stream, const uint8_t *d_descriptors, int n,
                                     int levelsup, int32_t *d_word, int32_t *d_node, double *d_weight);
int orbx_vocabulary_transform(orbx_vocabulary *v, const uint8_t *descriptors, int n, int levelsup, int32_t *word,
                              int32_t *node, double *weight);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:248-319) for npoints map points at once.
 * The descriptors observed for point p are rows seg_start[p] .. seg_start[p+1]-1 of `descriptors`
 * (32 bytes each, in the order the caller collected them; the reference walks a
 * std::map<KeyFrame*, size_t>, i.e. pointer order).  best_idx[p] receives the index inside the segment
 * of the descriptor with the least median distance to all of the segment (first index on ties), -1 for
 * an empty segment; best_desc (may be NULL) receives that descriptor.  The _device form takes device
 * pointers and is asynchronous on the matcher's stream. */
int orbx_distinctive_descriptors_device(orbx_matcher *m, const uint8_t *d_descriptors, const int32_t *d_seg_start,
                                        int npoints, int32_t *d_best_idx, uint8_t *d_best_desc);
int orbx_distinctive_descriptors(orbx_matcher *m, const uint8_t *descriptors, const int32_t *seg_start, int npoints,
                                 int32_t *best_idx, uint8_t *best_desc);

/* ---------------------------------------------------------------- synthetic input */

/* Deterministic test/benchmark image (splitmix64 stream seeded with `seed`): gray 128 canvas,
 * width*height/300 filled rectangles (variant 0: gray 0..255; variant 2: gray 118..138, low
 * contrast) and uniform noise in [-3,3].  variant 1: canvas and noise only. Host function. */
int orbx_synth_image(uint64_t seed, int width, int height, int variant, uint8_t *dst, int stride_bytes);
/* Deterministic random descriptors (n rows of 32 bytes). Host function. */
int orbx_synth_descriptors(uint64_t seed, int64_t n, uint8_t *dst);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
