// Per-cell FAST-9/16 detection (ORBextractor::ComputeKeyPointsOctTree, src/ORBextractor.cc:815-893
// of the reference, which calls cv::FAST(cell, keys, threshold, nonmax=true) per 30-px cell).
//
// The score cv::FAST attaches to a corner, cornerScore = max(t, A) - 1, rests on a quantity that does
// not depend on the threshold t:  A(p) = max over the 16 arcs of 9 contiguous ring pixels of
// min(v - ring) and of min(ring - v).  p is a corner at t  <=>  A(p) > t, and it survives the
// non-maximum suppression of its cell  <=>  A(p) > A(q) for the 8 neighbours q that lie in the same
// cell's detect region (pixels 3 or more inside the cell window).  The detect regions of the cells
// tile [19, w-19) x [19, h-19) exactly, so one map per level serves both thresholds:
//   k_fast_nms   : A for every pixel, kept only where it is a strict local maximum inside its cell
//   k_cell_count : per cell, survivors above iniThFAST, else above minThFAST (the fallback of :874-878)
//   k_cell_scan  : exclusive prefix over the cells of a level (cells row-major)
//   k_cell_emit  : ordered compaction (cell-major, row-major inside a cell), as the reference appends
#include "orbx_internal.h"

#define A_PITCH 68  // LDS row pitch of the score tile (66 used)

__device__ __forceinline__ int min9_max(const int (&d)[16])
{
    // max over the 16 circular windows of length 9 of the window minimum, by doubling:
    // m2[k] = min(d[k],d[k+1]), m4[k] = min(m2[k],m2[k+2]), m8[k] = min(m4[k],m4[k+4]), m9[k] = min(m8[k], d[k+8])
    int m2[16], m4[16], best = -256;
#pragma unroll
    for (int k = 0; k < 16; k++)
        m2[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++)
        m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int m8 = min(m4[k], m4[(k + 4) & 15]);
        best = max(best, min(m8, d[(k + 8) & 15]));
    }
    return best;
}

__device__ __forceinline__ int fast_arc_score(const uint8_t *t, int p /* pitch of the LDS image tile */)
{
    const int v = t[0];
    int d[16];
    d[0] = v - t[3 * p];
    d[1] = v - t[3 * p + 1];
    d[2] = v - t[2 * p + 2];
    d[3] = v - t[p + 3];
    d[4] = v - t[3];
    d[5] = v - t[-p + 3];
    d[6] = v - t[-2 * p + 2];
    d[7] = v - t[-3 * p + 1];
    d[8] = v - t[-3 * p];
    d[9] = v - t[-3 * p - 1];
    d[10] = v - t[-2 * p - 2];
    d[11] = v - t[-p - 3];
    d[12] = v - t[-3];
    d[13] = v - t[p - 3];
    d[14] = v - t[2 * p - 2];
    d[15] = v - t[3 * p - 1];
    const int dark = min9_max(d);
    int e[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        e[k] = -d[k];
    const int bright = min9_max(e);
    return max(dark, bright);
}

struct FastArgs {
    int nlevels;
    int slab;
    int tile_base[ORBX_MAX_LEVELS + 1];
    int tiles_x[ORBX_MAX_LEVELS];
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS], pitch[ORBX_MAX_LEVELS], off[ORBX_MAX_LEVELS];
    int wcell[ORBX_MAX_LEVELS], hcell[ORBX_MAX_LEVELS];
};

#define IMG_W (ORBX_FT_W + 8)  // 72
#define IMG_H (ORBX_FT_H + 8)  // 24

__global__ __launch_bounds__(256) void k_fast_nms(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ nms,
                                                  FastArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[IMG_H * IMG_W];
    __shared__ uint8_t sc[(ORBX_FT_H + 2) * A_PITCH];

    const int f = blockIdx.y;
    int lvl = 0;
    while (lvl + 1 < a.nlevels && (int)blockIdx.x >= a.tile_base[lvl + 1])
        lvl++;
    const int t = blockIdx.x - a.tile_base[lvl];
    const int tx0 = (t % a.tiles_x[lvl]) * ORBX_FT_W;
    const int ty0 = (t / a.tiles_x[lvl]) * ORBX_FT_H;
    const int w = a.w[lvl], h = a.h[lvl], pitch = a.pitch[lvl];
    const uint8_t *src = pyr + (size_t)f * a.slab + a.off[lvl];
    uint8_t *dst = nms + (size_t)f * a.slab + a.off[lvl];

    // stage rows ty0-4 .. ty0+19, columns tx0-4 .. tx0+67 (word aligned: tx0 is a multiple of 64)
    for (int i = threadIdx.x; i < IMG_H * (IMG_W / 4); i += 256) {
        const int r = i / (IMG_W / 4), c4 = i % (IMG_W / 4);
        const int y = ty0 - 4 + r, x = tx0 - 4 + c4 * 4;
        uint32_t v = 0;
        if (y >= 0 && y < h && x >= 0 && x < pitch)
            v = *reinterpret_cast<const uint32_t *>(src + (size_t)y * pitch + x);
        *reinterpret_cast<uint32_t *>(&img[r * IMG_W + c4 * 4]) = v;
    }
    __syncthreads();

    // A for the tile plus a one-pixel ring (66 x 18); zero outside [19, w-19) x [19, h-19)
    const int x_lo = ORBX_CELL_ORG, x_hi = w - ORBX_CELL_ORG, y_lo = ORBX_CELL_ORG, y_hi = h - ORBX_CELL_ORG;
    for (int i = threadIdx.x; i < (ORBX_FT_H + 2) * (ORBX_FT_W + 2); i += 256) {
        const int r = i / (ORBX_FT_W + 2), c = i % (ORBX_FT_W + 2);
        const int y = ty0 - 1 + r, x = tx0 - 1 + c;
        int A = 0;
        if (x >= x_lo && x < x_hi && y >= y_lo && y < y_hi) {
            A = fast_arc_score(&img[(r + 3) * IMG_W + (c + 3)], IMG_W);
            A = A < 0 ? 0 : A;
        }
        sc[r * A_PITCH + c] = (uint8_t)A;
    }
    __syncthreads();

    // strict local maximum among the neighbours that share the pixel's cell
    const int wc = a.wcell[lvl], hc = a.hcell[lvl];
    const int r = threadIdx.x >> 4;        // 0..15
    const int c0 = (threadIdx.x & 15) * 4; // 0..60
    const int y = ty0 + r;
    uint32_t out = 0;
    if (y < h) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = c0 + i, x = tx0 + c;
            const uint8_t *p = &sc[(r + 1) * A_PITCH + (c + 1)];
            const int A = p[0];
            if (A > 0 && wc > 0) {
                // A > 0 implies x in [19, w-19), y in [19, h-19)
                const int cx0 = ORBX_CELL_ORG + ((x - ORBX_CELL_ORG) / wc) * wc;
                const int cy0 = ORBX_CELL_ORG + ((y - ORBX_CELL_ORG) / hc) * hc;
                const bool l = x - 1 >= cx0, rr = (x + 1 < cx0 + wc) && (x + 1 < x_hi);
                const bool u = y - 1 >= cy0, dn = (y + 1 < cy0 + hc) && (y + 1 < y_hi);
                int m = 0;
                if (l) m = max(m, (int)p[-1]);
                if (rr) m = max(m, (int)p[1]);
                if (u) {
                    m = max(m, (int)p[-A_PITCH]);
                    if (l) m = max(m, (int)p[-A_PITCH - 1]);
                    if (rr) m = max(m, (int)p[-A_PITCH + 1]);
                }
                if (dn) {
                    m = max(m, (int)p[A_PITCH]);
                    if (l) m = max(m, (int)p[A_PITCH - 1]);
                    if (rr) m = max(m, (int)p[A_PITCH + 1]);
                }
                if (A > m)
                    out |= (uint32_t)A << (8 * i);
            }
        }
        if (tx0 + c0 < pitch)
            *reinterpret_cast<uint32_t *>(dst + (size_t)y * pitch + tx0 + c0) = out;
    }
}

hipError_t orbx_launch_fast_nms(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    FastArgs a;
    a.nlevels = g.nlevels;
    a.slab = g.slab;
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &L = g.lv[l];
        a.tile_base[l] = L.tile_base;
        a.tiles_x[l] = L.tiles_x;
        a.w[l] = L.w, a.h[l] = L.h, a.pitch[l] = L.pitch, a.off[l] = L.off;
        a.wcell[l] = L.wcell, a.hcell[l] = L.hcell;
    }
    a.tile_base[g.nlevels] = g.tiles_per_frame;
    dim3 grid(g.tiles_per_frame, g.batch);
    k_fast_nms<<<grid, 256, 0, s>>>(b.pyr, b.nms, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------- cells

struct CellArgs {
    int nlevels, slab, cells_per_frame, cand_per_frame;
    int ini_th, min_th;
    int cell_base[ORBX_MAX_LEVELS + 1];
    int ncols[ORBX_MAX_LEVELS];
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS], pitch[ORBX_MAX_LEVELS], off[ORBX_MAX_LEVELS];
    int wcell[ORBX_MAX_LEVELS], hcell[ORBX_MAX_LEVELS];
    int cand_base[ORBX_MAX_LEVELS], cand_cap[ORBX_MAX_LEVELS];
};

struct CellRect {
    int lvl, x0, x1, y0, y1;
};

__device__ __forceinline__ CellRect cell_rect(const CellArgs &a, int cell)
{
    CellRect r;
    int lvl = 0;
    while (lvl + 1 < a.nlevels && cell >= a.cell_base[lvl + 1])
        lvl++;
    const int c = cell - a.cell_base[lvl];
    const int j = c % a.ncols[lvl], i = c / a.ncols[lvl];
    r.lvl = lvl;
    r.x0 = ORBX_CELL_ORG + j * a.wcell[lvl];
    r.y0 = ORBX_CELL_ORG + i * a.hcell[lvl];
    r.x1 = min(r.x0 + a.wcell[lvl], a.w[lvl] - ORBX_CELL_ORG);
    r.y1 = min(r.y0 + a.hcell[lvl], a.h[lvl] - ORBX_CELL_ORG);
    return r;
}

#define CELL_INI_FLAG (1 << 30)

// one wave per cell
__global__ __launch_bounds__(256) void k_cell_count(const uint8_t *__restrict__ nms, int *__restrict__ cell_cnt,
                                                    CellArgs a)
{
    const int f = blockIdx.y;
    const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= a.cells_per_frame)
        return;
    const int lane = threadIdx.x & 63;
    const CellRect r = cell_rect(a, cell);
    const uint8_t *m = nms + (size_t)f * a.slab + a.off[r.lvl];
    const int pitch = a.pitch[r.lvl];
    const int x = r.x0 + lane;
    const int t_ini = max(a.ini_th, 1), t_min = max(a.min_th, 1);
    int n_ini = 0, n_min = 0;
    for (int y = r.y0; y < r.y1; y++) {
        const int v = (x < r.x1) ? m[(size_t)y * pitch + x] : 0;
        n_ini += __popcll(__ballot(v > t_ini));
        n_min += __popcll(__ballot(v > t_min));
    }
    if (lane == 0)
        cell_cnt[(size_t)f * a.cells_per_frame + cell] = n_ini > 0 ? (n_ini | CELL_INI_FLAG) : n_min;
}

// one workgroup per (level, frame): exclusive prefix over the level's cells
__global__ __launch_bounds__(256) void k_cell_scan(const int *__restrict__ cell_cnt, int *__restrict__ cell_off,
                                                   int *__restrict__ ncand, int *__restrict__ status, CellArgs a)
{
    __shared__ int part[256];
    const int lvl = blockIdx.x, f = blockIdx.y;
    const int base = a.cell_base[lvl], n = a.cell_base[lvl + 1] - base;
    const int *cnt = cell_cnt + (size_t)f * a.cells_per_frame + base;
    int *off = cell_off + (size_t)f * a.cells_per_frame + base;
    const int per = (n + 255) / 256;
    const int lo = min((int)threadIdx.x * per, n), hi = min(lo + per, n);
    int s = 0;
    for (int i = lo; i < hi; i++)
        s += cnt[i] & (CELL_INI_FLAG - 1);
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 256 partial sums
    for (int d = 1; d < 256; d <<= 1) {
        int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int i = lo; i < hi; i++) {
        off[i] = run;
        run += cnt[i] & (CELL_INI_FLAG - 1);
    }
    if (threadIdx.x == 255) {
        const int total = part[255];
        ncand[f * a.nlevels + lvl] = total;
        if (total > a.cand_cap[lvl])
            atomicMax(&status[ORBX_ST_CAND_OVERFLOW], total);
    }
}

__global__ __launch_bounds__(256) void k_cell_emit(const uint8_t *__restrict__ nms, const int *__restrict__ cell_cnt,
                                                   const int *__restrict__ cell_off, uint32_t *__restrict__ cand_xy,
                                                   uint8_t *__restrict__ cand_resp, CellArgs a)
{
    const int f = blockIdx.y;
    const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= a.cells_per_frame)
        return;
    const int cc = cell_cnt[(size_t)f * a.cells_per_frame + cell];
    if ((cc & (CELL_INI_FLAG - 1)) == 0)
        return;
    const int lane = threadIdx.x & 63;
    const CellRect r = cell_rect(a, cell);
    const uint8_t *m = nms + (size_t)f * a.slab + a.off[r.lvl];
    const int pitch = a.pitch[r.lvl];
    const int x = r.x0 + lane;
    const int th = max((cc & CELL_INI_FLAG) ? a.ini_th : a.min_th, 1);
    int pos = cell_off[(size_t)f * a.cells_per_frame + cell];
    const int cap = a.cand_cap[r.lvl];
    uint32_t *oxy = cand_xy + (size_t)f * a.cand_per_frame + a.cand_base[r.lvl];
    uint8_t *orp = cand_resp + (size_t)f * a.cand_per_frame + a.cand_base[r.lvl];
    for (int y = r.y0; y < r.y1; y++) {
        const int v = (x < r.x1) ? m[(size_t)y * pitch + x] : 0;
        const bool keep = v > th;
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const int k = pos + __popcll(mask & ((1ull << lane) - 1ull));
            if (k < cap) {
                oxy[k] = (uint32_t)(x - ORBX_BORDER) | ((uint32_t)(y - ORBX_BORDER) << 16);
                orp[k] = (uint8_t)(v - 1);
            }
        }
        pos += __popcll(mask);
    }
}

hipError_t orbx_launch_cells(const OrbxGeom &g, const OrbxBuffers &b, hipStream_t s)
{
    CellArgs a;
    a.nlevels = g.nlevels, a.slab = g.slab, a.cells_per_frame = g.cells_per_frame;
    a.cand_per_frame = g.cand_per_frame;
    a.ini_th = min(max(g.ini_th, 0), 255), a.min_th = min(max(g.min_th, 0), 255);
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &L = g.lv[l];
        a.cell_base[l] = L.cell_base, a.ncols[l] = L.ncols;
        a.w[l] = L.w, a.h[l] = L.h, a.pitch[l] = L.pitch, a.off[l] = L.off;
        a.wcell[l] = L.wcell, a.hcell[l] = L.hcell;
        a.cand_base[l] = L.cand_base, a.cand_cap[l] = L.cand_cap;
    }
    a.cell_base[g.nlevels] = g.cells_per_frame;
    hipError_t e;
    if (g.cells_per_frame > 0) {
        dim3 grid((g.cells_per_frame + 3) / 4, g.batch);
        k_cell_count<<<grid, 256, 0, s>>>(b.nms, b.cell_cnt, a);
        if ((e = hipGetLastError()) != hipSuccess)
            return e;
    }
    k_cell_scan<<<dim3(g.nlevels, g.batch), 256, 0, s>>>(b.cell_cnt, b.cell_off, b.ncand, b.status, a);
    if ((e = hipGetLastError()) != hipSuccess)
        return e;
    if (g.cells_per_frame > 0) {
        dim3 grid((g.cells_per_frame + 3) / 4, g.batch);
        k_cell_emit<<<grid, 256, 0, s>>>(b.nms, b.cell_cnt, b.cell_off, b.cand_xy, b.cand_resp, a);
        if ((e = hipGetLastError()) != hipSuccess)
            return e;
    }
    return hipSuccess;
}
