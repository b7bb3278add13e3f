// The 64x48 lookup grid of Frame: AssignFeaturesToGrid / PosInGrid (src/Frame.cc:239-253, 409-420 of
// the reference) and GetFeaturesInArea (:349-407), for keypoints that sit in device memory.
//
// The grid is a CSR table over cell = column * 48 + row with the keypoint indices of a cell in
// ascending order (the reference appends them in index order).  Sorting the keypoints that have a
// cell by (cell, index) gives the item list, and the start of a cell is the number of items in
// lower cells.  An area query walks the cell columns of its window; the rows of one column are
// contiguous in the item list, so the lanes of a wave test a column's items side by side and append
// the hits in order, which is exactly the reference's visiting order.
#include "orbx_internal.h"

#define GRID_COLS 64  // FRAME_GRID_COLS (include/Frame.h:38)
#define GRID_ROWS 48  // FRAME_GRID_ROWS
#define GRID_CELLS (GRID_COLS * GRID_ROWS)

struct GridArgs {
    float min_x, min_y, inv_w, inv_h;
};

// Every workgroup computes all sort keys into LDS (key = cell << 16 | index for keypoints inside the
// grid, ~0 for the others, which sort last) and ranks its own 256 of them: keys are distinct, so a
// key's rank is the number of smaller keys.
__global__ __launch_bounds__(256) void k_grid_sort(const orbx_keypoint *__restrict__ kp, int n,
                                                   uint32_t *__restrict__ sorted_keys,
                                                   int32_t *__restrict__ cell_items, GridArgs a)
{
    extern __shared__ uint32_t keys[];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) {
        const int px = (int)roundf((kp[i].x - a.min_x) * a.inv_w), py = (int)roundf((kp[i].y - a.min_y) * a.inv_h);
        const bool in = px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS;
        keys[i] = in ? ((uint32_t)(px * GRID_ROWS + py) << 16) | (uint32_t)i : 0xffffffffu;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i >= n)
        return;
    const uint32_t k = keys[i];
    if (k == 0xffffffffu)
        return;
    int rank = 0;
    for (int j = 0; j < n; j++)
        rank += keys[j] < k;  // the same address in every lane
    sorted_keys[rank] = k;
    cell_items[rank] = (int32_t)(k & 0xffffu);
}

// cell_start[c] = number of sorted keys whose cell is below c (entry 64*48 = all keys inside the grid)
__global__ __launch_bounds__(256) void k_grid_starts(const uint32_t *__restrict__ sorted_keys, int n,
                                                     int32_t *__restrict__ cell_start)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c > GRID_CELLS)
        return;
    int lo = 0, hi = n;  // keys outside the grid were never written: the array was filled with ~0 first
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((sorted_keys[mid] >> 16) < (uint32_t)c)
            lo = mid + 1;
        else
            hi = mid;
    }
    cell_start[c] = lo;
}

// one wave per query
__global__ __launch_bounds__(64) void k_features_in_area(const orbx_keypoint *__restrict__ kp,
                                                         const int32_t *__restrict__ cell_start,
                                                         const int32_t *__restrict__ cell_items,
                                                         const float *__restrict__ qx, const float *__restrict__ qy,
                                                         const float *__restrict__ qr, const int *__restrict__ qmin,
                                                         const int *__restrict__ qmax, int nq, int cap,
                                                         int32_t *__restrict__ out_idx, int32_t *__restrict__ out_n,
                                                         GridArgs a)
{
    const int q = blockIdx.x, lane = threadIdx.x;
    const float x = qx[q], y = qy[q], r = qr[q];
    const int minLevel = qmin[q], maxLevel = qmax[q];
    int n = 0;
    const int x0 = max(0, (int)floorf((x - a.min_x - r) * a.inv_w));
    const int x1 = min(GRID_COLS - 1, (int)ceilf((x - a.min_x + r) * a.inv_w));
    const int y0 = max(0, (int)floorf((y - a.min_y - r) * a.inv_h));
    const int y1 = min(GRID_ROWS - 1, (int)ceilf((y - a.min_y + r) * a.inv_h));
    if (!(x0 >= GRID_COLS || x1 < 0 || y0 >= GRID_ROWS || y1 < 0)) {
        const bool check = (minLevel > 0) || (maxLevel >= 0);
        int32_t *out = out_idx + (size_t)q * cap;
        for (int ix = x0; ix <= x1; ix++) {
            if (y1 < y0)
                break;
            const int lo = cell_start[ix * GRID_ROWS + y0], hi = cell_start[ix * GRID_ROWS + y1 + 1];
            for (int j0 = lo; j0 < hi; j0 += 64) {  // uniform trips
                const int j = j0 + lane;
                bool hit = false;
                int id = 0;
                if (j < hi) {
                    id = cell_items[j];
                    const orbx_keypoint k = kp[id];
                    hit = !(check && (k.octave < minLevel || (maxLevel >= 0 && k.octave > maxLevel))) &&
                          fabsf(k.x - x) < r && fabsf(k.y - y) < r;
                }
                const unsigned long long bm = __ballot(hit);
                const int pos = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                if (hit && pos < cap)
                    out[pos] = id;
                n += __popcll(bm);
            }
        }
    }
    if (lane == 0)
        out_n[q] = n;  // may exceed cap: the list is then truncated
}

size_t orbx_grid_sort_lds_bytes(int n) { return (size_t)(n > 0 ? n : 1) * sizeof(uint32_t); }

hipError_t orbx_launch_grid_build(const orbx_keypoint *d_kp, int n, float min_x, float max_x, float min_y, float max_y,
                                  uint32_t *d_keys, int32_t *d_cell_start, int32_t *d_cell_items, hipStream_t s)
{
    GridArgs a{min_x, min_y, (float)GRID_COLS / (max_x - min_x), (float)GRID_ROWS / (max_y - min_y)};
    hipError_t e = hipMemsetAsync(d_keys, 0xff, sizeof(uint32_t) * (size_t)(n > 0 ? n : 1), s);
    if (e != hipSuccess)
        return e;
    if (n > 0) {
        const size_t lds = orbx_grid_sort_lds_bytes(n);
        if (lds > 48 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_grid_sort),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess)
                return e;
        }
        k_grid_sort<<<(n + 255) / 256, 256, lds, s>>>(d_kp, n, d_keys, d_cell_items, a);
        if ((e = hipGetLastError()) != hipSuccess)
            return e;
    }
    k_grid_starts<<<(GRID_CELLS + 256) / 256, 256, 0, s>>>(d_keys, n, d_cell_start);
    return hipGetLastError();
}

hipError_t orbx_launch_features_in_area(const orbx_keypoint *d_kp, float min_x, float max_x, float min_y, float max_y,
                                        const int32_t *d_cell_start, const int32_t *d_cell_items, const float *d_qx,
                                        const float *d_qy, const float *d_qr, const int *d_qmin, const int *d_qmax,
                                        int nq, int cap, int32_t *d_out_idx, int32_t *d_out_n, hipStream_t s)
{
    if (nq <= 0)
        return hipSuccess;
    GridArgs a{min_x, min_y, (float)GRID_COLS / (max_x - min_x), (float)GRID_ROWS / (max_y - min_y)};
    k_features_in_area<<<nq, 64, 0, s>>>(d_kp, d_cell_start, d_cell_items, d_qx, d_qy, d_qr, d_qmin, d_qmax, nq, cap,
                                         d_out_idx, d_out_n, a);
    return hipGetLastError();
}

// Frame::UndistortKeyPoints (src/Frame.cc:467-499): cv::undistortPoints with the camera matrix as the
// new projection, one thread per point, in double like the library routine: normalise, five fixed-point
// iterations of the radial/tangential model, project back.  The keypoint's other fields are copied.
struct UndistortArgs {
    double fx, fy, cx, cy, k1, k2, p1, p2, k3;
};

__device__ __forceinline__ float2 undistort_point(float px, float py, const UndistortArgs &a)
{
    const double ifx = 1. / a.fx, ify = 1. / a.fy;
    double x = px, y = py;
    const double x0 = x = (x - a.cx) * ifx;
    const double y0 = y = (y - a.cy) * ify;
#pragma unroll 1
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1. / (1 + ((a.k3 * r2 + a.k2) * r2 + a.k1) * r2);
        const double dx = 2 * a.p1 * x * y + a.p2 * (r2 + 2 * x * x);
        const double dy = a.p1 * (r2 + 2 * y * y) + 2 * a.p2 * x * y;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    return make_float2((float)(a.fx * x + a.cx), (float)(a.fy * y + a.cy));
}

__global__ __launch_bounds__(256) void k_undistort_keypoints(const orbx_keypoint *__restrict__ kp, int n,
                                                             orbx_keypoint *__restrict__ out, UndistortArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    orbx_keypoint k = kp[i];
    const float2 u = undistort_point(k.x, k.y, a);
    k.x = u.x, k.y = u.y;
    out[i] = k;
}

__global__ __launch_bounds__(256) void k_undistort_points(const float2 *__restrict__ xy, int n, float2 *__restrict__ out,
                                                          UndistortArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] = undistort_point(xy[i].x, xy[i].y, a);
}

// cam = fx, fy, cx, cy; dist = k1, k2, p1, p2[, k3]; exactly one of (d_kp, d_kp_out) and (d_xy, d_xy_out) is used
hipError_t orbx_launch_undistort(const orbx_keypoint *d_kp, orbx_keypoint *d_kp_out, const float *d_xy, float *d_xy_out,
                                 int n, const float *cam, const float *dist, int ndist, hipStream_t s)
{
    if (n <= 0)
        return hipSuccess;
    UndistortArgs a{cam[0], cam[1], cam[2], cam[3], dist[0], dist[1], dist[2], dist[3], ndist > 4 ? dist[4] : 0.f};
    if (d_kp)
        k_undistort_keypoints<<<(n + 255) / 256, 256, 0, s>>>(d_kp, n, d_kp_out, a);
    else
        k_undistort_points<<<(n + 255) / 256, 256, 0, s>>>(reinterpret_cast<const float2 *>(d_xy), n,
                                                           reinterpret_cast<float2 *>(d_xy_out), a);
    return hipGetLastError();
}
