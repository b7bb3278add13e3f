// Internal declarations shared by the HIP translation units of liborbx.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx.h"

#define ORBX_MAX_LEVELS 16
#define ORBX_EDGE 19        // EDGE_THRESHOLD of the reference (src/ORBextractor.cc:74)
#define ORBX_BORDER 16      // EDGE_THRESHOLD-3: detection region is [16, cols-16) x [16, rows-16)
#define ORBX_CELL_ORG 19    // first pixel FAST can report: border + 3
#define ORBX_HALF_PATCH 15  // HALF_PATCH_SIZE (src/ORBextractor.cc:73)
#define ORBX_MAX_ROOTS 64   // quadtree roots per level (round(width/height) of the detection region)

// blur tile produced by one workgroup
#define ORBX_FT_W 128
#define ORBX_FT_H 32
// fused resize of two levels: columns / rows of the middle level a workgroup owns (it computes 256 x 32)
#define ORBX_RS2_TW 252
#define ORBX_RS2_TH 31
// FAST tile produced by one workgroup.  The tile grid starts at (16, 19): FAST reports pixels of
// [19, w-19) x [19, h-19) only, and a column origin of 16 keeps the 16-byte pieces of the staging aligned.
#define ORBX_FAST_W 64
#define ORBX_FAST_H 64
#define ORBX_FAST_X0 16
#define ORBX_FAST_Y0 ORBX_CELL_ORG
#define ORBX_FAST_TILES_X(w) ((w) - ORBX_CELL_ORG - ORBX_FAST_X0 > 0 ? ((w) - ORBX_CELL_ORG - ORBX_FAST_X0 + ORBX_FAST_W - 1) / ORBX_FAST_W : 1)
#define ORBX_FAST_TILES_Y(h) ((h) - ORBX_CELL_ORG - ORBX_FAST_Y0 > 0 ? ((h) - ORBX_CELL_ORG - ORBX_FAST_Y0 + ORBX_FAST_H - 1) / ORBX_FAST_H : 1)

// Blurred levels are stored as tiles of 16 x 8 pixels = one 128-byte cache line, tile rows of pitch / 16 tiles
// (pitch * 8 bytes), h / 8 + 2 tile rows per level.  Inside a tile: two halves of 4 rows (64 bytes), each made of
// four groups of 4 pixels (16 bytes) holding their 4 rows one after the other.  A descriptor window (37 x 37
// pixels) then lies in about 20 lines instead of the 50 its rows touch in a row-major level, and the blur
// kernel's 4 x 4-pixel register block is one 16-byte store.  Rows below the level inside its last tile row are
// written as zeros; the padding tile rows behind it and the columns right of the level hold unspecified bytes.
// Byte offset of pixel (x, y) from the level's start:
#define ORBX_BLUR_ADDR(x, y, pitch) \
    (((y) >> 3) * ((pitch) * 8) + ((x) >> 4) * 128 + (((y) >> 2) & 1) * 64 + (((x) >> 2) & 3) * 16 + ((y) & 3) * 4 + ((x) & 3))
#define ORBX_BLUR_LEVEL_BYTES(pitch, h) ((pitch) * 8 * ((h) / 8 + 2) + 512)

struct OrbxLevel {
    int w, h, pitch;  // level size and row pitch (bytes) inside a frame slab
    int off;          // byte offset of the level inside a frame slab
    int boff;         // byte offset of the blurred level inside a frame's blurred slab (tiled layout, below)
    int max_bx, max_by;  // w-16, h-16
    int ncols, nrows, wcell, hcell;  // cell grid of src/ORBextractor.cc:834-840 (0 when the level is too small)
    int cell_base;   // first cell of the level in the per-frame cell arrays
    int tiles_x, tiles_y, tile_base;     // blur tiles of the level in the per-frame tile numbering
    int ftiles_x, ftiles_y, ftile_base;  // FAST tiles, likewise
    int cand_base, cand_cap;          // segment of the per-frame candidate arrays
    int quota;       // mnFeaturesPerLevel[level]
    int n_ini;       // quadtree roots
    float hx;        // root width
    int node_cap;    // capacity of the node list of the level (max(quota, 4*n_ini) + 4, rounded up to a multiple of 4)
    int kp_base;     // segment of the per-frame kept-keypoint arrays
    float scale;     // mvScaleFactor[level]
    int patch_size;  // (int)(31*scale)
    int xtab_off, ytab_off;  // offsets into the resize coefficient tables
    // the level as the upper one of a fused resize pair: offsets (into the pair table) of the first column /
    // row of this level that each tile column / row of the level below produces (one entry more than tiles)
    int pcol_off, prow_off;
};

struct OrbxGeom {
    int nlevels, batch;
    int ini_th, min_th;
    int blur_variant;
    int cells_per_frame, tiles_per_frame, ftiles_per_frame, cand_per_frame, kp_per_frame;
    int out_cap;  // result rows per frame (== kp_per_frame)
    int slab;     // bytes per frame in the level slabs
    int bslab;    // bytes per frame in the blurred slabs
    OrbxLevel lv[ORBX_MAX_LEVELS];
};

// horizontal / vertical bilinear coefficients of one destination column / row
struct OrbxXCoef {
    int sx;
    short a0, a1;
};
struct OrbxYCoef {
    int sy0, sy1;
    int b0, b1;
};

// device status words written by kernels (index into the status array)
enum { ORBX_ST_CAND_OVERFLOW = 0, ORBX_ST_NODE_OVERFLOW = 1, ORBX_ST_COUNT = 4 };

// Level 0 of the current batch: the caller's frames when they can be read in place (4-byte aligned
// rows), else the copy inside the pyramid slabs
struct OrbxLevel0 {
    const uint8_t *ptr;  // first pixel of frame 0
    long long fstride;   // bytes between frames
    int pitch;           // bytes between rows
};

struct OrbxBuffers {
    OrbxLevel0 l0;
    uint8_t *pyr;     // [batch][slab] pyramid levels (level 0 only when the frames had to be copied)
    uint8_t *blur;    // [batch][bslab] blurred levels in 16 x 8-pixel tiles of one 128-byte line (ORBX_BLUR_ADDR)
    uint8_t *cell_flag;  // [batch][cells_per_frame] 1: the cell has a candidate whose score exceeds iniThFAST
    int *ncand;       // [batch][nlevels] candidates appended (before the per-cell threshold decision); cell_flag
                      // lies behind it in the same allocation
    uint32_t *cand_xy;   // [batch][cand_per_frame] x | y<<16, relative to (16,16); arbitrary order
    uint8_t *cand_resp;  // [batch][cand_per_frame] cornerScore = A-1
    uint32_t *cand_key;  // [batch][cand_per_frame] cell<<12 | row-in-cell<<6 | col-in-cell: the reference's order
    uint16_t *cand_node; // [batch][cand_per_frame] quadtree scratch: node position of each candidate
    uint32_t *kept_xy;   // [batch][kp_per_frame] x | y<<16 in level coordinates, quadtree list order
    uint8_t *kept_resp;  // [batch][kp_per_frame]
    int *nkept;          // [batch][nlevels]
    orbx_keypoint *out_kp;  // [batch][out_cap]
    uint8_t *out_desc;      // [batch][out_cap][32]
    int *out_n;             // [batch]
    int *status;            // [ORBX_ST_COUNT]
    const OrbxXCoef *xtab;
    const OrbxYCoef *ytab;
    const int *ptab;  // pair table of the fused resize (OrbxLevel::pcol_off, prow_off)
    // level << 24 | tile row << 12 | tile column of every blur tile of a frame (tiles_per_frame entries), then of
    // every FAST tile (ftiles_per_frame entries): a workgroup finds its tile with one scalar load
    const uint32_t *tile_tab;
    const uint8_t *blur_bands;  // the four constant tap operands of the matrix-core blur
};
#define ORBX_TILE_ENTRY(lvl, ty, tx) (((uint32_t)(lvl) << 24) | ((uint32_t)(ty) << 12) | (uint32_t)(tx))

// kernel launchers (each enqueues on `s` and returns the hipGetLastError code)
// Grids of kernels that work on (unit-of-a-frame, frame) pairs are shaped (gx, units, groups) with
// gx = 8 once the batch has 8 frames: the 8 workgroups dispatched back to back (one to each XCD)
// then belong to 8 different frames, so every workgroup of a frame runs on the same XCD and the
// frame's rows are shared through that XCD's L2.  Smaller batches keep gx = 1 (all XCDs work on
// the same frame).  This is a placement hint only; results do not depend on it.
static inline dim3 orbx_frame_grid(int units_per_frame, int batch)
{
    const int gx = batch >= 8 ? 8 : 1;
    return dim3(gx, units_per_frame, (batch + gx - 1) / gx);
}
#if defined(__HIPCC__)
#define ORBX_BLOCK_FRAME ((int)(blockIdx.z * gridDim.x + blockIdx.x))  // may be >= batch in the last group
#define ORBX_BLOCK_UNIT ((int)blockIdx.y)

// Wave-wide (64 lanes) prefix and reduction steps on the data-parallel-primitive paths of the vector ALU:
// four shifts inside the rows of 16 lanes, then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31 into
// rows 2 and 3.  Six dependent instructions; the device library's __ockl_wfred_* / __ockl_wfscan_* compile to
// six LDS-crossbar permutes here, each a round trip of about a hundred clocks, which sets the pace of every
// kernel that walks a list in order.
#define ORBX_DPP_STEP(OP, T, v, ident, ctrl, rows) \
    v = OP(v, (T)__builtin_amdgcn_update_dpp((int)(ident), (int)(v), ctrl, rows, 0xf, false))
#define ORBX_DPP_SCAN(OP, T, v, ident)            \
    ORBX_DPP_STEP(OP, T, v, ident, 0x111, 0xf);  /* row_shr:1 */  \
    ORBX_DPP_STEP(OP, T, v, ident, 0x112, 0xf);  /* row_shr:2 */  \
    ORBX_DPP_STEP(OP, T, v, ident, 0x114, 0xf);  /* row_shr:4 */  \
    ORBX_DPP_STEP(OP, T, v, ident, 0x118, 0xf);  /* row_shr:8 */  \
    ORBX_DPP_STEP(OP, T, v, ident, 0x142, 0xa);  /* row_bcast:15 */ \
    ORBX_DPP_STEP(OP, T, v, ident, 0x143, 0xc)   /* row_bcast:31 */
__device__ __forceinline__ uint32_t orbx_op_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ int orbx_op_max_i32(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int orbx_op_add_i32(int a, int b) { return a + b; }
// minimum / maximum over the wave, the same value in every lane
__device__ __forceinline__ uint32_t orbx_wave_min_u32(uint32_t v)
{
    ORBX_DPP_SCAN(orbx_op_min_u32, uint32_t, v, 0xffffffffu);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int orbx_wave_max_i32(int v)
{
    ORBX_DPP_SCAN(orbx_op_max_i32, int, v, (int)0x80000000);
    return __builtin_amdgcn_readlane(v, 63);
}
// inclusive prefix sum over the lanes
__device__ __forceinline__ int orbx_wave_scan_add_i32(int v)
{
    ORBX_DPP_SCAN(orbx_op_add_i32, int, v, 0);
    return v;
}
#endif

hipError_t orbx_launch_ingest(const uint8_t *src, int stride, int64_t fstride, const OrbxGeom &g, const OrbxBuffers &b,
                              hipStream_t s);
hipError_t orbx_launch_resize(int level, const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s);
// levels `level` and `level + 1` from level `level - 1` in one launch
hipError_t orbx_launch_resize_pair(int level, const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s);
// part 0: clears the candidate counts and per-cell flags; part 1: detection, which appends to the candidate arrays
hipError_t orbx_launch_fast(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s, int part);
hipError_t orbx_launch_octree(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s);
hipError_t orbx_launch_blur(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s);
hipError_t orbx_launch_describe(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s);
// the blur on the matrix cores (blur_variant 0); its four constant tap operands
#ifdef __cplusplus
#include <vector>
size_t orbx_blur_mfma_band_bytes(void);
void orbx_blur_mfma_tables(std::vector<uint8_t> &bands);
hipError_t orbx_launch_blur_mfma(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s);
#endif
size_t orbx_octree_lds_bytes(int node_cap);

// standalone quadtree run on caller-provided candidates (parity tests)
hipError_t orbx_launch_octree_single(const uint32_t *d_xy, const uint8_t *d_resp, uint16_t *d_node, int n, int region_w,
                                     int region_h, int quota, uint32_t *d_out_xy, uint8_t *d_out_resp, int *d_out_idx,
                                     int *d_nout, int *d_status, int node_cap, hipStream_t s);

// matcher launchers
hipError_t orbx_launch_top2(const uint8_t *d_q, int nq, const uint8_t *d_db, int64_t ndb, int64_t idx_base,
                            uint32_t *d_partial, int chunk_rows, int nchunks, int32_t *d_best, uint16_t *d_d1,
                            uint16_t *d_d2, hipStream_t s);
hipError_t orbx_launch_stereo(const OrbxGeom &g, const OrbxBuffers &bl, const OrbxBuffers &br, const float *sf,
                              const float *inv, float mb, float mbf, int *d_rowstart, uint4 *d_sorted, int *d_sad,
                              float *d_uright, float *d_depth, int *d_nmatch, hipStream_t s);
hipError_t orbx_launch_distinctive(const uint8_t *d_desc, const int *d_seg_start, int npoints, int32_t *d_best_idx,
                                   uint8_t *d_best_desc, hipStream_t s);
hipError_t orbx_launch_bow(const uint8_t *d_desc1, const float *d_ang1, const uint8_t *d_valid1, const int *d_feat_start,
                           int nb, int total_feat1, const int *d_node_kf, const int *d_ids1, const int *d_start1,
                           const int *d_items1, int total_nodes1, const uint8_t *d_desc2, const float *d_ang2,
                           const uint8_t *d_valid2, int n2, const int *d_ids2, const int *d_start2, const int *d_items2,
                           int m2, int strict_lt, float ratio, int check_ori, int *d_match12, int *d_match21,
                           int *d_nmatches, int8_t *d_bin_of, int *d_hist, int *d_status, hipStream_t s);
hipError_t orbx_launch_bow_strided(const orbx_keypoint *d_kp1, const uint8_t *d_desc1, const int *d_counts1,
                                   const uint8_t *d_valid1, int nb, int cap1, const int *d_ids1, const int *d_start1,
                                   const int *d_items1, const orbx_keypoint *d_kp2, const uint8_t *d_desc2,
                                   const uint8_t *d_valid2, int cap2, const int *d_ids2, const int *d_start2,
                                   const int *d_items2, int strict_lt, float ratio, int check_ori, int *d_match12,
                                   int *d_match21, int *d_nmatches, int8_t *d_bin_of, int *d_hist, int *d_kf_status,
                                   int *d_status, hipStream_t s);
hipError_t orbx_launch_feature_vectors(const int *d_node, const int *d_counts, int nframes, int cap, int *d_sorted,
                                       int *d_ids, int *d_start, int *d_items, int *d_num, hipStream_t s);
hipError_t orbx_launch_grid_build(const orbx_keypoint *d_kp, int n, float min_x, float max_x, float min_y, float max_y,
                                  uint32_t *d_keys, int32_t *d_cell_start, int32_t *d_cell_items, hipStream_t s);
hipError_t orbx_launch_features_in_area(const orbx_keypoint *d_kp, float min_x, float max_x, float min_y, float max_y,
                                        const int32_t *d_cell_start, const int32_t *d_cell_items, const float *d_qx,
                                        const float *d_qy, const float *d_qr, const int *d_qmin, const int *d_qmax,
                                        int nq, int cap, int32_t *d_out_idx, int32_t *d_out_n, hipStream_t s);
hipError_t orbx_launch_search_by_projection(const orbx_keypoint *d_kp, const uint8_t *d_desc, const float *d_uright,
                                            const uint8_t *d_taken, int n, const uint8_t *d_valid, const float *d_r,
                                            const float *d_xr, const uint8_t *d_src_desc, const uint8_t *d_blocks,
                                            const float *d_src_angle, int ns, const int32_t *d_list,
                                            const int32_t *d_count, int cap, uint32_t *d_entry, int use_second,
                                            float nnratio, int th_dist, int check_orientation, int32_t *d_match,
                                            int32_t *d_feature_match, int32_t *d_result, hipStream_t s);
hipError_t orbx_launch_undistort(const orbx_keypoint *d_kp, orbx_keypoint *d_kp_out, const float *d_xy, float *d_xy_out,
                                 int n, const float *cam, const float *dist, int ndist, hipStream_t s);
hipError_t orbx_launch_voc_transform(const uint8_t *d_node_desc, const int *d_child_start, const int *d_children,
                                     const int *d_word_id, const double *d_weight, int depth, const uint8_t *d_feat,
                                     int n, int levelsup, int *d_word, int *d_node, double *d_wt, hipStream_t s);
size_t orbx_search_init_lds_bytes(int cap0);
hipError_t orbx_launch_search_init(const orbx_keypoint *d_kp, const uint8_t *d_desc, const int *d_counts, int max_kp,
                                   int npairs, int cap0, float min_x, float max_x, float min_y, float max_y,
                                   const float *d_prev_in, float *d_prev_out, int window, float ratio, int check_ori,
                                   int *d_matches12, int *d_nmatches, int *d_pair_count, hipStream_t s);
void orbx_top2_mfma_chunks(int nq, int64_t ndb, int *chunk_rows, int *nchunks);
hipError_t orbx_launch_top2_mfma(const uint8_t *d_q, int nq, const uint8_t *d_db, int64_t ndb, int64_t idx_base,
                                 uint32_t *d_partial, int chunk_rows, int nchunks, int32_t *d_best, uint16_t *d_d1,
                                 uint16_t *d_d2, hipStream_t s);
hipError_t orbx_launch_pair_dist(const uint8_t *d_a, const uint8_t *d_b, const int *d_pairs, int n, uint16_t *d_out,
                                 hipStream_t s);
hipError_t orbx_launch_top2_merge_shards(const uint8_t *d_blocks, int nshards, int nq, int64_t shard_stride,
                                         int32_t *d_best, uint16_t *d_d1, uint16_t *d_d2, hipStream_t s);
// records a HIP error code for orbx_last_hip_error (calling thread)
void orbx_note_hip_error(hipError_t e);
