// C ABI of liborbx: handles, level geometry, workspace management and the launch sequence.
// Host code only; the kernels live in the .hip files next to it.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "orbx_internal.h"

static thread_local int g_last_hip = 0;

#define HIP_TRY(expr)                        \
    do {                                     \
        hipError_t e__ = (expr);             \
        if (e__ != hipSuccess) {             \
            g_last_hip = (int)e__;           \
            return ORBX_E_HIP;               \
        }                                    \
    } while (0)

extern "C" const char *orbx_strerror(int code)
{
    switch (code) {
    case ORBX_OK: return "ok";
    case ORBX_E_ARG: return "invalid argument";
    case ORBX_E_HIP: return "HIP runtime error";
    case ORBX_E_CAPACITY: return "buffer or workspace too small";
    case ORBX_E_GEOMETRY: return "image too tall for the quadtree (round(width/height) is 0)";
    case ORBX_E_NODEVICE: return "no usable GPU";
    case ORBX_E_STATE: return "call order violated";
    case ORBX_E_RCCL: return "RCCL missing or failed";
    default: return "unknown error";
    }
}
extern "C" int orbx_last_hip_error(void) { return g_last_hip; }
void orbx_note_hip_error(hipError_t e) { g_last_hip = (int)e; }
extern "C" int orbx_version(void) { return 100; }

// ------------------------------------------------------------------ extractor handle

// Extractor handles of one device that take turns (orbx_pipeline_create): the stream of the handle that
// was called last, on which the next handle's blur is enqueued.
struct orbx_pipeline {
    int device;
    hipStream_t last_stream;
};

struct orbx_extractor {
    orbx_params p;
    int device;
    hipStream_t own_stream, stream;
    hipStream_t aux_stream;        // the blur runs here, beside detection and distribution
    hipEvent_t ev_pyramid, ev_blur;
    orbx_pipeline *pipe;           // handles that take turns (null: the blur runs on the handle's own second stream)
    uint8_t *h_stage;              // pinned staging for host frames (level-0 layout), allocated on first use
    size_t h_stage_bytes;
    hipEvent_t ev_staged;          // the uploads out of h_stage have finished
    uint8_t *d_stage;              // device copy of page-locked host frames in their own layout
    size_t d_stage_bytes;
    uint8_t *h_out;                // pinned staging of whole-batch downloads into pageable memory
    size_t h_out_bytes;
    uint8_t *h_pyr;                // pinned staging of orbx_pyramid_download (one frame's level slab)
    size_t h_pyr_bytes;
    float sf[ORBX_MAX_LEVELS], inv[ORBX_MAX_LEVELS], sig2[ORBX_MAX_LEVELS], invsig2[ORBX_MAX_LEVELS];
    int quota[ORBX_MAX_LEVELS];
    OrbxGeom cap_geom;  // geometry of max_width x max_height: sizes every buffer
    OrbxGeom geom;      // geometry of the last batch
    int geom_w, geom_h; // image size geom was built for (0 = none)
    bool have_batch;
    unsigned long long batch_serial;  // counts the batches this handle has processed
    OrbxBuffers buf;
    // stereo matching workspace (allocated on first use) and the batch pair it holds results for
    int *st_rowstart;
    uint4 *st_sorted;  // right keypoints in row order: a 16-byte record each, then their descriptors
    int *st_sad, *st_nmatch;
    float *st_uright, *st_depth;
    hipEvent_t ev_stereo;
    const orbx_extractor *st_right;
    unsigned long long st_left_serial, st_right_serial;
    float st_mb, st_mbf;
    OrbxXCoef *d_xtab;
    OrbxYCoef *d_ytab;
    int *d_ptab;
    uint32_t *d_tiletab;
    uint8_t *d_blur_bands;     // the four constant tap operands of the matrix-core blur
    bool blur_mfma;            // ORBX_BLUR=mfma: the matrix-core form of the blur (comparison runs)
    int xtab_cap, ytab_cap, ptab_cap, tiletab_cap;
    std::vector<int> h_counts, h_nkept;
    bool profiling;
    std::vector<hipEvent_t> prof_events;  // 2*ORBX_NSTAGES events (start, end of each stage) per profiled batch
    size_t prof_used;
};

static const char *const k_stage_names[ORBX_NSTAGES] = {"ingest", "resize", "fast", "octree", "blur", "describe"};
extern "C" const char *orbx_stage_name(int i) { return i >= 0 && i < ORBX_NSTAGES ? k_stage_names[i] : ""; }

// records the start (which = 0) or end (which = 1) event of a stage of the current profiled batch
// on the stream the stage runs on
static hipError_t prof_mark(orbx_extractor *ex, int stage, int which, hipStream_t st)
{
    if (!ex->profiling)
        return hipSuccess;
    const size_t at = ex->prof_used * (2 * ORBX_NSTAGES) + 2 * stage + which;
    while (ex->prof_events.size() <= at) {
        hipEvent_t e;
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess)
            return rc;
        ex->prof_events.push_back(e);
    }
    return hipEventRecord(ex->prof_events[at], st);
}

static inline int round_half_even(float v) { return (int)lrintf(v); }
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
static inline short sat_short(int v) { return (short)std::min(32767, std::max(-32768, v)); }

// scale tables as the reference computes them (src/ORBextractor.cc:437-456): cumulative products
// in double (the member scaleFactor is a double), each stored as float
static void scale_tables(orbx_extractor *ex)
{
    const double s = (double)ex->p.scale_factor;
    ex->sf[0] = 1.0f;
    ex->sig2[0] = 1.0f;
    for (int i = 1; i < ex->p.nlevels; i++) {
        ex->sf[i] = (float)((double)ex->sf[i - 1] * s);
        ex->sig2[i] = ex->sf[i] * ex->sf[i];
    }
    for (int i = 0; i < ex->p.nlevels; i++) {
        ex->inv[i] = 1.0f / ex->sf[i];
        ex->invsig2[i] = 1.0f / ex->sig2[i];
    }
    // per-level quotas (src/ORBextractor.cc:460-475)
    const float factor = (float)(1.0 / s);
    float desired = (float)ex->p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)ex->p.nlevels));
    int sum = 0;
    for (int l = 0; l < ex->p.nlevels - 1; l++) {
        ex->quota[l] = round_half_even(desired);
        sum += ex->quota[l];
        desired *= factor;
    }
    ex->quota[ex->p.nlevels - 1] = std::max(ex->p.nfeatures - sum, 0);
}

// Level sizes (src/ORBextractor.cc:1195-1197), cell grids (:826-840), quadtree roots (:587-589)
// and the layout of every per-frame array.  Returns ORBX_E_GEOMETRY for levels the reference cannot
// distribute (nIni == 0).
static int build_geometry(const orbx_extractor *ex, int w, int h, int batch, OrbxGeom *g)
{
    memset(g, 0, sizeof(*g));
    g->nlevels = ex->p.nlevels;
    g->batch = batch;
    g->ini_th = ex->p.ini_th_fast;
    g->min_th = ex->p.min_th_fast;
    g->blur_variant = ex->p.blur_variant;
    long long off = 0, boff = 0;
    int cells = 0, tiles = 0, ftiles = 0, cand = 0, kps = 0, xt = 0, yt = 0, pt = 0;
    for (int l = 0; l < g->nlevels; l++) {
        OrbxLevel &L = g->lv[l];
        L.w = round_half_even((float)w * ex->inv[l]);
        L.h = round_half_even((float)h * ex->inv[l]);
        if (L.w < 1 || L.h < 1)
            return ORBX_E_ARG;
        L.pitch = align_up(L.w, 64);
        L.off = (int)off;
        off += (long long)align_up(L.pitch * L.h, 256);
        L.boff = (int)boff;
        boff += (long long)align_up(ORBX_BLUR_LEVEL_BYTES(L.pitch, L.h), 256);
        if (off > INT_MAX || boff > INT_MAX)
            return ORBX_E_CAPACITY;
        L.max_bx = L.w - ORBX_BORDER;
        L.max_by = L.h - ORBX_BORDER;
        const float width = (float)(L.max_bx - ORBX_BORDER), height = (float)(L.max_by - ORBX_BORDER);
        L.ncols = L.nrows = L.wcell = L.hcell = 0;
        if (width > 0 && height > 0) {
            const int nc = (int)(width / 30.f), nr = (int)(height / 30.f);
            if (nc > 0 && nr > 0) {
                L.ncols = nc, L.nrows = nr;
                L.wcell = (int)ceilf(width / nc);
                L.hcell = (int)ceilf(height / nr);
            }
        }
        L.cell_base = cells;
        cells += L.ncols * L.nrows;
        L.tiles_x = (L.w + ORBX_FT_W - 1) / ORBX_FT_W;
        L.tiles_y = (L.h + ORBX_FT_H - 1) / ORBX_FT_H;
        L.tile_base = tiles;
        tiles += L.tiles_x * L.tiles_y;
        L.ftiles_x = ORBX_FAST_TILES_X(L.w);
        L.ftiles_y = ORBX_FAST_TILES_Y(L.h);
        L.ftile_base = ftiles;
        ftiles += L.ftiles_x * L.ftiles_y;
        L.cand_base = cand;
        // at most one strict local maximum per 2x2 pixels of a cell
        L.cand_cap = L.ncols > 0 ? align_up((L.w * L.h) / 4 + 4, 4) : 0;
        cand += L.cand_cap;
        L.quota = ex->quota[l];
        L.n_ini = 1;
        L.hx = 1.f;
        if (L.ncols > 0) {
            L.n_ini = (int)roundf(width / height);
            if (L.n_ini < 1)
                return ORBX_E_GEOMETRY;
            L.hx = width / L.n_ini;
        }
        L.node_cap = align_up(std::max(L.quota, 4 * L.n_ini) + 4, 4);  // a multiple of 4: 16-byte LDS reads of node words
        if (orbx_octree_lds_bytes(L.node_cap) > 160 * 1024)
            return ORBX_E_ARG;  // the node list of one level must fit the LDS of a compute unit
        L.kp_base = kps;
        kps += L.node_cap;
        L.scale = ex->sf[l];
        L.patch_size = (int)(31 * ex->sf[l]);
        L.xtab_off = xt;
        L.ytab_off = yt;
        L.pcol_off = L.prow_off = 0;
        if (l >= 2) {
            const OrbxLevel &M = g->lv[l - 1];
            L.pcol_off = pt;
            pt += (M.w + ORBX_RS2_TW - 1) / ORBX_RS2_TW + 1;
            L.prow_off = pt;
            pt += (M.h + ORBX_RS2_TH - 1) / ORBX_RS2_TH + 1;
        }
        xt += align_up(L.w, 4);  // a thread reads the coefficients of 4 columns / 2 rows at once
        yt += align_up(L.h, 2);
    }
    g->cells_per_frame = cells;
    g->tiles_per_frame = tiles;
    g->ftiles_per_frame = ftiles;
    g->cand_per_frame = cand;
    g->kp_per_frame = kps;
    g->out_cap = kps;
    g->slab = (int)off;
    g->bslab = (int)boff;
    return ORBX_OK;
}

// cv::resize INTER_LINEAR coefficient tables (8-bit path: 11-bit fixed point, half-to-even rounding)
static void build_resize_tables(const OrbxGeom &g, std::vector<OrbxXCoef> &xt, std::vector<OrbxYCoef> &yt,
                                std::vector<int> &pt)
{
    xt.clear();
    yt.clear();
    pt.clear();
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &D = g.lv[l];
        const size_t x0 = xt.size(), y0 = yt.size();
        xt.resize(x0 + align_up(D.w, 4));
        yt.resize(y0 + align_up(D.h, 2));
        if (l == 0) {
            for (int i = 0; i < align_up(D.w, 4); i++)
                xt[x0 + i] = OrbxXCoef{std::min(i, D.w - 1), 2048, 0};
            for (int i = 0; i < align_up(D.h, 2); i++)
                yt[y0 + i] = OrbxYCoef{std::min(i, D.h - 1), std::min(i, D.h - 1), 2048, 0};
            continue;
        }
        const OrbxLevel &S = g.lv[l - 1];
        const double scale_x = 1. / ((double)D.w / S.w), scale_y = 1. / ((double)D.h / S.h);
        int xmax = D.w;
        for (int dx = 0; dx < D.w; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = (int)floorf(fx);
            fx -= sx;
            if (sx < 0)
                fx = 0, sx = 0;
            if (sx + 1 >= S.w) {
                xmax = std::min(xmax, dx);
                if (sx >= S.w - 1)
                    fx = 0, sx = S.w - 1;
            }
            OrbxXCoef c;
            c.sx = sx;
            c.a0 = sat_short(round_half_even((1.f - fx) * 2048));
            c.a1 = sat_short(round_half_even(fx * 2048));
            xt[x0 + dx] = c;
        }
        for (int dx = xmax; dx < D.w; dx++) {  // columns right of xmax take the source pixel unweighted
            xt[x0 + dx].a0 = 2048;
            xt[x0 + dx].a1 = 0;
        }
        for (int dy = 0; dy < D.h; dy++) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = (int)floorf(fy);
            fy -= sy;
            OrbxYCoef c;
            c.sy0 = std::min(std::max(sy, 0), S.h - 1);
            c.sy1 = std::min(std::max(sy + 1, 0), S.h - 1);
            c.b0 = sat_short(round_half_even((1.f - fy) * 2048));
            c.b1 = sat_short(round_half_even(fy * 2048));
            yt[y0 + dy] = c;
        }
        for (int i = D.w; i < align_up(D.w, 4); i++)  // padding: copies of the last entry
            xt[x0 + i] = xt[x0 + D.w - 1];
        for (int i = D.h; i < align_up(D.h, 2); i++)
            yt[y0 + i] = yt[y0 + D.h - 1];
        if (l >= 2) {
            // fused pairs: the first column (row) of this level whose left (upper) source pixel lies in or
            // behind tile column (row) t of the level below; source positions ascend with the destination
            const int ntx = (S.w + ORBX_RS2_TW - 1) / ORBX_RS2_TW, nty = (S.h + ORBX_RS2_TH - 1) / ORBX_RS2_TH;
            pt.resize(std::max<size_t>(pt.size(), (size_t)D.prow_off + nty + 1));
            int dx = 0;
            for (int t = 0; t <= ntx; t++) {
                while (dx < D.w && xt[x0 + dx].sx < t * ORBX_RS2_TW)
                    dx++;
                pt[D.pcol_off + t] = t == ntx ? D.w : dx;
            }
            int dy = 0;
            for (int t = 0; t <= nty; t++) {
                while (dy < D.h && yt[y0 + dy].sy0 < t * ORBX_RS2_TH)
                    dy++;
                pt[D.prow_off + t] = t == nty ? D.h : dy;
            }
        }
    }
}

template <typename T> static hipError_t dmalloc(T **p, size_t n)
{
    *p = nullptr;
    return hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T));
}

static void free_buffers(orbx_extractor *ex)
{
    OrbxBuffers &b = ex->buf;
    void *ptrs[] = {b.pyr,       b.blur,    b.cand_xy, b.cand_resp, b.cand_key, b.cand_node,
                    b.kept_xy,   b.kept_resp, b.nkept,   b.out_kp, b.out_n /* and status, ncand, cell_flag */, ex->d_ptab, ex->d_xtab,
                    ex->d_ytab,  ex->d_tiletab, ex->d_blur_bands};
    for (void *p : ptrs)
        if (p)
            (void)hipFree(p);
    void *st[] = {ex->st_rowstart, ex->st_sorted, ex->st_sad, ex->st_nmatch, ex->st_uright, ex->st_depth};
    for (void *p : st)
        if (p)
            (void)hipFree(p);
    ex->st_rowstart = ex->st_sad = ex->st_nmatch = nullptr;
    ex->st_sorted = nullptr;
    ex->st_uright = ex->st_depth = nullptr;
    if (ex->ev_stereo)
        (void)hipEventDestroy(ex->ev_stereo);
    ex->ev_stereo = nullptr;
    memset(&ex->buf, 0, sizeof(ex->buf));
    ex->d_xtab = nullptr;
    ex->d_ytab = nullptr;
    ex->d_ptab = nullptr;
    ex->d_tiletab = nullptr;
    ex->d_blur_bands = nullptr;
}

extern "C" int orbx_extractor_create(const orbx_params *params, orbx_extractor **out)
{
    if (!params || !out)
        return ORBX_E_ARG;
    *out = nullptr;
    const orbx_params &p = *params;
    if (p.nfeatures < 1 || p.nlevels < 1 || p.nlevels > ORBX_MAX_LEVELS || !(p.scale_factor > 1.0f) ||
        p.max_width < 1 || p.max_height < 1 || p.max_width > 4096 || p.max_height > 4096 || p.max_batch < 1 ||
        p.blur_variant < 0 || p.blur_variant > 1)
        return ORBX_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || p.device < 0 || p.device >= ndev)
        return ORBX_E_NODEVICE;
    orbx_extractor *ex = new (std::nothrow) orbx_extractor();
    if (!ex)
        return ORBX_E_CAPACITY;
    ex->p = p;
    ex->device = p.device;
    ex->geom_w = ex->geom_h = 0;
    ex->have_batch = false;
    ex->batch_serial = 0;
    ex->st_rowstart = ex->st_sad = ex->st_nmatch = nullptr;
    ex->st_sorted = nullptr;
    ex->st_uright = ex->st_depth = nullptr;
    ex->ev_stereo = nullptr;
    ex->st_right = nullptr;
    ex->st_left_serial = ex->st_right_serial = 0;
    ex->st_mb = ex->st_mbf = 0.f;
    ex->profiling = false;
    ex->prof_used = 0;
    ex->own_stream = ex->stream = ex->aux_stream = nullptr;
    ex->ev_pyramid = ex->ev_blur = ex->ev_staged = nullptr;
    ex->pipe = nullptr;
    ex->h_stage = nullptr;
    ex->h_stage_bytes = 0;
    ex->d_stage = nullptr;
    ex->d_stage_bytes = 0;
    ex->h_out = nullptr;
    ex->h_out_bytes = 0;
    ex->h_pyr = nullptr;
    ex->h_pyr_bytes = 0;
    memset(&ex->buf, 0, sizeof(ex->buf));
    ex->d_xtab = nullptr;
    ex->d_ytab = nullptr;
    ex->d_ptab = nullptr;
    ex->d_tiletab = nullptr;
    ex->d_blur_bands = nullptr;
    scale_tables(ex);
    int rc = build_geometry(ex, p.max_width, p.max_height, p.max_batch, &ex->cap_geom);
    if (rc == ORBX_E_GEOMETRY) {
        // the bound image itself may be too tall; size the workspace from a square of the larger side
        const int side = std::max(p.max_width, p.max_height);
        rc = build_geometry(ex, side, side, p.max_batch, &ex->cap_geom);
    }
    if (rc != ORBX_OK) {
        delete ex;
        return rc;
    }
    hipError_t e = hipSetDevice(p.device);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&ex->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&ex->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&ex->ev_pyramid, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&ex->ev_blur, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&ex->ev_staged, hipEventDisableTiming);
    ex->stream = ex->own_stream;
    const OrbxGeom &g = ex->cap_geom;
    const size_t B = (size_t)p.max_batch;
    OrbxBuffers &b = ex->buf;
    // 256 bytes of slack: window copies read whole words that may run past the last row
    if (e == hipSuccess) e = dmalloc(&b.pyr, B * g.slab + 256);
    if (e == hipSuccess) e = dmalloc(&b.blur, B * g.bslab + 256);
    // One allocation: keypoint counts | status words | candidate counts | per-cell flags.  The counts and the
    // status words are neighbours for the single-frame fetch (one copy); the status words, the candidate counts and
    // the flags are neighbours so that one fill clears them before detection.
    if (e == hipSuccess) {
        uint8_t *all = nullptr;
        e = dmalloc(&all, (B + ORBX_ST_COUNT + B * g.nlevels) * sizeof(int) + B * g.cells_per_frame);
        b.out_n = reinterpret_cast<int *>(all);
        b.status = all ? b.out_n + B : nullptr;
        b.ncand = all ? b.status + ORBX_ST_COUNT : nullptr;
        b.cell_flag = all ? reinterpret_cast<uint8_t *>(b.ncand + B * g.nlevels) : nullptr;
    }
    if (e == hipSuccess) e = dmalloc(&b.cand_xy, B * g.cand_per_frame);
    if (e == hipSuccess) e = dmalloc(&b.cand_resp, B * g.cand_per_frame);
    if (e == hipSuccess) e = dmalloc(&b.cand_key, B * g.cand_per_frame);
    if (e == hipSuccess) e = dmalloc(&b.cand_node, B * g.cand_per_frame);
    if (e == hipSuccess) e = dmalloc(&b.kept_xy, B * g.kp_per_frame);
    if (e == hipSuccess) e = dmalloc(&b.kept_resp, B * g.kp_per_frame);
    if (e == hipSuccess) e = dmalloc(&b.nkept, B * g.nlevels);
    // The descriptors lie behind the keypoint records in one allocation (as the status words behind the counts,
    // above): a handle for single frames (max_batch 1) then fetches a frame's results with two copies, not four.
    // (out_cap is a multiple of 4, so the descriptor rows start 16-byte aligned.)
    if (e == hipSuccess) {
        uint8_t *both = nullptr;
        e = dmalloc(&both, B * g.out_cap * (sizeof(orbx_keypoint) + 32));
        b.out_kp = reinterpret_cast<orbx_keypoint *>(both);
        b.out_desc = both ? both + B * g.out_cap * sizeof(orbx_keypoint) : nullptr;
    }
    int xt = 0, yt = 0;
    for (int l = 0; l < g.nlevels; l++)
        xt += align_up(g.lv[l].w, 4), yt += align_up(g.lv[l].h, 2);
    ex->xtab_cap = xt;
    ex->ytab_cap = yt;
    ex->ptab_cap = 0;
    for (int l = 2; l < g.nlevels; l++)
        ex->ptab_cap = std::max(ex->ptab_cap, g.lv[l].prow_off + (g.lv[l - 1].h + ORBX_RS2_TH - 1) / ORBX_RS2_TH + 1);
    if (e == hipSuccess) e = dmalloc(&ex->d_xtab, (size_t)xt);
    if (e == hipSuccess) e = dmalloc(&ex->d_ytab, (size_t)yt);
    if (e == hipSuccess) e = dmalloc(&ex->d_ptab, (size_t)ex->ptab_cap);
    ex->tiletab_cap = g.tiles_per_frame + g.ftiles_per_frame;
    if (e == hipSuccess) e = dmalloc(&ex->d_tiletab, (size_t)ex->tiletab_cap);
    if (e == hipSuccess) e = dmalloc(&ex->d_blur_bands, orbx_blur_mfma_band_bytes());
    if (e == hipSuccess) {
        std::vector<uint8_t> bands;
        orbx_blur_mfma_tables(bands);
        e = hipMemcpy(ex->d_blur_bands, bands.data(), bands.size(), hipMemcpyHostToDevice);
    }
    {
        const char *mode = getenv("ORBX_BLUR");  // "mfma": the matrix-core kernel (bit-equal, no faster: comparison runs)
        ex->blur_mfma = mode && strcmp(mode, "mfma") == 0;
    }
    if (e != hipSuccess) {
        g_last_hip = (int)e;
        free_buffers(ex);
        if (ex->ev_pyramid) (void)hipEventDestroy(ex->ev_pyramid);
        if (ex->ev_blur) (void)hipEventDestroy(ex->ev_blur);
        if (ex->ev_staged) (void)hipEventDestroy(ex->ev_staged);
        if (ex->aux_stream) (void)hipStreamDestroy(ex->aux_stream);
        if (ex->own_stream) (void)hipStreamDestroy(ex->own_stream);
        delete ex;
        return ORBX_E_HIP;
    }
    b.xtab = ex->d_xtab;
    b.ytab = ex->d_ytab;
    b.ptab = ex->d_ptab;
    b.tile_tab = ex->d_tiletab;
    b.blur_bands = ex->d_blur_bands;
    *out = ex;
    return ORBX_OK;
}

extern "C" void orbx_extractor_destroy(orbx_extractor *ex)
{
    if (!ex)
        return;
    (void)hipSetDevice(ex->device);
    (void)hipStreamSynchronize(ex->stream);
    (void)hipStreamSynchronize(ex->aux_stream);
    if (ex->pipe && ex->pipe->last_stream == ex->stream)
        ex->pipe->last_stream = nullptr;  // (the next handle's blur then runs on its own second stream once)
    for (hipEvent_t e : ex->prof_events)
        (void)hipEventDestroy(e);
    free_buffers(ex);
    if (ex->ev_pyramid) (void)hipEventDestroy(ex->ev_pyramid);
    if (ex->ev_blur) (void)hipEventDestroy(ex->ev_blur);
    if (ex->ev_staged) (void)hipEventDestroy(ex->ev_staged);
    if (ex->h_stage) (void)hipHostFree(ex->h_stage);
    if (ex->d_stage) (void)hipFree(ex->d_stage);
    if (ex->h_out) (void)hipHostFree(ex->h_out);
    if (ex->h_pyr) (void)hipHostFree(ex->h_pyr);
    if (ex->aux_stream) (void)hipStreamDestroy(ex->aux_stream);
    if (ex->own_stream) (void)hipStreamDestroy(ex->own_stream);
    delete ex;
}

extern "C" int orbx_extractor_set_stream(orbx_extractor *ex, void *hip_stream)
{
    if (!ex)
        return ORBX_E_ARG;
    if (ex->pipe && ex->pipe->last_stream == ex->stream) {
        // a neighbour's blur may still be queued on the stream that is given up
        HIP_TRY(hipStreamSynchronize(ex->stream));
        ex->pipe->last_stream = nullptr;
    }
    ex->stream = hip_stream ? (hipStream_t)hip_stream : ex->own_stream;
    return ORBX_OK;
}

// ---- handles that take turns ----

extern "C" int orbx_pipeline_create(int device, orbx_pipeline **out)
{
    if (!out)
        return ORBX_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ORBX_E_NODEVICE;
    if (device < 0 || device >= ndev)
        return ORBX_E_ARG;
    orbx_pipeline *p = new orbx_pipeline();
    p->device = device;
    p->last_stream = nullptr;
    *out = p;
    return ORBX_OK;
}

extern "C" void orbx_pipeline_destroy(orbx_pipeline *p) { delete p; }

extern "C" int orbx_extractor_set_pipeline(orbx_extractor *ex, orbx_pipeline *p)
{
    if (!ex || (p && p->device != ex->device))
        return ORBX_E_ARG;
    if (ex->pipe && ex->pipe != p && ex->pipe->last_stream == ex->stream) {
        HIP_TRY(hipSetDevice(ex->device));
        HIP_TRY(hipStreamSynchronize(ex->stream));
        ex->pipe->last_stream = nullptr;
    }
    ex->pipe = p;
    return ORBX_OK;
}

extern "C" int orbx_extractor_levels(const orbx_extractor *ex) { return ex ? ex->p.nlevels : ORBX_E_ARG; }

extern "C" int orbx_extractor_scale_tables(const orbx_extractor *ex, float *scale, float *inv_scale, float *sigma2,
                                           float *inv_sigma2)
{
    if (!ex)
        return ORBX_E_ARG;
    for (int i = 0; i < ex->p.nlevels; i++) {
        if (scale) scale[i] = ex->sf[i];
        if (inv_scale) inv_scale[i] = ex->inv[i];
        if (sigma2) sigma2[i] = ex->sig2[i];
        if (inv_sigma2) inv_sigma2[i] = ex->invsig2[i];
    }
    return ORBX_OK;
}

extern "C" int orbx_extractor_features_per_level(const orbx_extractor *ex, int *quota)
{
    if (!ex || !quota)
        return ORBX_E_ARG;
    for (int i = 0; i < ex->p.nlevels; i++)
        quota[i] = ex->quota[i];
    return ORBX_OK;
}

extern "C" int orbx_extractor_max_keypoints(const orbx_extractor *ex) { return ex ? ex->cap_geom.out_cap : ORBX_E_ARG; }

// geometry of a (w,h) batch; uploads the resize tables when the image size changed
static int prepare(orbx_extractor *ex, int batch, int w, int h)
{
    if (batch < 1 || batch > ex->p.max_batch || w < 1 || h < 1 || w > ex->p.max_width || h > ex->p.max_height)
        return ORBX_E_CAPACITY;
    HIP_TRY(hipSetDevice(ex->device));
    if (w != ex->geom_w || h != ex->geom_h) {
        OrbxGeom g;
        int rc = build_geometry(ex, w, h, batch, &g);
        if (rc != ORBX_OK)
            return rc;
        const OrbxGeom &c = ex->cap_geom;
        if (g.slab > c.slab || g.bslab > c.bslab || g.cells_per_frame > c.cells_per_frame || g.cand_per_frame > c.cand_per_frame ||
            g.ftiles_per_frame > c.ftiles_per_frame ||
            g.kp_per_frame > c.kp_per_frame)
            return ORBX_E_CAPACITY;
        // per-frame strides of the result arrays stay those of the capacity geometry
        g.kp_per_frame = c.kp_per_frame;
        g.out_cap = c.out_cap;
        std::vector<OrbxXCoef> xt;
        std::vector<OrbxYCoef> yt;
        std::vector<int> pt;
        build_resize_tables(g, xt, yt, pt);
        if ((int)xt.size() > ex->xtab_cap || (int)yt.size() > ex->ytab_cap || (int)pt.size() > ex->ptab_cap)
            return ORBX_E_CAPACITY;
        // the tiles of a frame in the order of the kernels' unit numbers: blur tiles, then FAST tiles
        std::vector<uint32_t> tt;
        for (int l = 0; l < g.nlevels; l++)
            for (int ty = 0; ty < g.lv[l].tiles_y; ty++)
                for (int tx = 0; tx < g.lv[l].tiles_x; tx++)
                    tt.push_back(ORBX_TILE_ENTRY(l, ty, tx));
        for (int l = 0; l < g.nlevels; l++)
            for (int ty = 0; ty < g.lv[l].ftiles_y; ty++)
                for (int tx = 0; tx < g.lv[l].ftiles_x; tx++)
                    tt.push_back(ORBX_TILE_ENTRY(l, ty, tx));
        if ((int)tt.size() != g.tiles_per_frame + g.ftiles_per_frame || (int)tt.size() > ex->tiletab_cap)
            return ORBX_E_CAPACITY;
        // the previous batch may still read the old tables
        HIP_TRY(hipStreamSynchronize(ex->stream));
        HIP_TRY(hipMemcpy(ex->d_xtab, xt.data(), xt.size() * sizeof(OrbxXCoef), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ex->d_ytab, yt.data(), yt.size() * sizeof(OrbxYCoef), hipMemcpyHostToDevice));
        if (!pt.empty())
            HIP_TRY(hipMemcpy(ex->d_ptab, pt.data(), pt.size() * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ex->d_tiletab, tt.data(), tt.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        ex->geom = g;
        ex->geom_w = w;
        ex->geom_h = h;
    }
    ex->geom.batch = batch;
    return ORBX_OK;
}

// everything after level 0 is in place
static int run_pipeline(orbx_extractor *ex)
{
    const OrbxGeom &g = ex->geom;
    const OrbxBuffers &b = ex->buf;
    hipStream_t s = ex->stream;
    // The blur only needs the pyramid: it runs on the second stream beside detection and
    // distribution.  While profiling, every stage runs on the main stream so that the stage times
    // are exclusive; "fast" and "blur" then bracket exactly one kernel each.
    // (one or two frames: the two event records and waits cost the host more than the blur's 10 us beside the
    // detection save, and a single call is bound by the time the host needs to enqueue it)
    const bool overlap = !ex->profiling && g.batch > 2;
    hipStream_t s2 = overlap ? ex->aux_stream : s;
    // Handles that take turns: the blur goes behind the work of the handle that was called before this one, on
    // that handle's stream.  Each stream then carries pyramid, detection, distribution and description of its own
    // batch and the blur of the next handle's batch, one after the other, and the streams of four handles settle
    // into a stagger in which the detection kernels, which all want the same vector units, rarely run side by
    // side.  With the blur on a stream of its own (eight independent streams for four handles) the pairing is left
    // to chance, and to the hardware queues the runtime happens to give the streams: 247 k frames/s against 258 k.
    if (overlap && ex->pipe && g.batch > 4) {
        if (ex->pipe->last_stream)
            s2 = ex->pipe->last_stream;
        ex->pipe->last_stream = s;
    }
    HIP_TRY(orbx_launch_fast(g, b, s, 0));  // clears the status words too
    HIP_TRY(prof_mark(ex, 1, 0, s));
    // A few frames: pairs of levels in one launch (four launches instead of seven; the time of such a call is
    // mostly launches).  Batches: a launch per level, which is faster there (the fused kernel computes 5 % of
    // the middle level twice and is bound by its instruction count).  The fused kernel's margins hold for level
    // ratios up to 2.5.
    for (int l = 1; l < g.nlevels;) {
        if (g.batch <= 4 && l + 1 < g.nlevels && 2 * g.lv[l].w <= 5 * g.lv[l + 1].w &&
            2 * g.lv[l].h <= 5 * g.lv[l + 1].h) {
            HIP_TRY(orbx_launch_resize_pair(l, g, b, s));
            l += 2;
        } else {
            HIP_TRY(orbx_launch_resize(l, g, b, s));
            l += 1;
        }
    }
    HIP_TRY(prof_mark(ex, 1, 1, s));
    if (overlap) {
        HIP_TRY(hipEventRecord(ex->ev_pyramid, s));
        HIP_TRY(hipStreamWaitEvent(s2, ex->ev_pyramid, 0));
    }
    HIP_TRY(prof_mark(ex, 4, 0, s2));
    // (the matrix-core form covers the 256-sum taps; the 257-sum variant saturates and has the vector-ALU form only)
    if (g.blur_variant == 0 && ex->blur_mfma)
        HIP_TRY(orbx_launch_blur_mfma(g, b, s2));
    else
        HIP_TRY(orbx_launch_blur(g, b, s2));
    HIP_TRY(prof_mark(ex, 4, 1, s2));
    if (overlap)
        HIP_TRY(hipEventRecord(ex->ev_blur, s2));
    HIP_TRY(prof_mark(ex, 2, 0, s));
    HIP_TRY(orbx_launch_fast(g, b, s, 1));
    HIP_TRY(prof_mark(ex, 2, 1, s));
    HIP_TRY(prof_mark(ex, 3, 0, s));
    HIP_TRY(orbx_launch_octree(g, b, s));
    HIP_TRY(prof_mark(ex, 3, 1, s));
    if (overlap)
        HIP_TRY(hipStreamWaitEvent(s, ex->ev_blur, 0));
    HIP_TRY(prof_mark(ex, 5, 0, s));
    HIP_TRY(orbx_launch_describe(g, b, s));
    HIP_TRY(prof_mark(ex, 5, 1, s));
    if (ex->profiling)
        ex->prof_used++;
    ex->have_batch = true;
    ex->batch_serial++;
    return ORBX_OK;
}

extern "C" int orbx_extract_batch_device(orbx_extractor *ex, const uint8_t *d_images, int batch, int width, int height,
                                         int stride_bytes, int64_t frame_stride_bytes)
{
    if (!ex || !d_images || stride_bytes < width)
        return ORBX_E_ARG;
    int rc = prepare(ex, batch, width, height);
    if (rc != ORBX_OK)
        return rc;
    // the previous batch's blur (second stream) reads the pyramid this batch overwrites; the describe
    // stage of that batch already waited for it, and runs before this ingest on the main stream
    HIP_TRY(prof_mark(ex, 0, 0, ex->stream));
    const bool in_place = (reinterpret_cast<uintptr_t>(d_images) & 3) == 0 && (stride_bytes & 3) == 0 &&
                          (frame_stride_bytes & 3) == 0;
    if (in_place) {
        // every kernel reads level 0 with aligned word loads: usable where it lies
        ex->buf.l0 = OrbxLevel0{d_images, (long long)frame_stride_bytes, stride_bytes};
    } else {
        HIP_TRY(orbx_launch_ingest(d_images, stride_bytes, frame_stride_bytes, ex->geom, ex->buf, ex->stream));
        ex->buf.l0 = OrbxLevel0{ex->buf.pyr + ex->geom.lv[0].off, (long long)ex->geom.slab, ex->geom.lv[0].pitch};
    }
    HIP_TRY(prof_mark(ex, 0, 1, ex->stream));
    return run_pipeline(ex);
}

extern "C" int orbx_extract_batch_host(orbx_extractor *ex, const uint8_t *const *images, int batch, int width,
                                       int height, int stride_bytes)
{
    if (!ex || !images || stride_bytes < width)
        return ORBX_E_ARG;
    int rc = prepare(ex, batch, width, height);
    if (rc != ORBX_OK)
        return rc;
    const OrbxLevel &L0 = ex->geom.lv[0];
    for (int f = 0; f < batch; f++)
        if (!images[f])
            return ORBX_E_ARG;
    // Frames in page-locked host memory (hipHostMalloc, hipHostRegister, a pinned torch tensor) are read
    // by the copy engine where they lie: no staging pass over them on the CPU.
    bool pinned = true;
    for (int f = 0; f < batch && pinned; f++) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, images[f]) != hipSuccess) {
            (void)hipGetLastError();  // an ordinary pageable pointer is reported as an error: clear it
            pinned = false;
        } else
            pinned = attr.type == hipMemoryTypeHost;
    }
    if (pinned) {
        // Copies in the frames' own row layout (strided copies are slow even from page-locked memory), then
        // the device path: read in place when rows are word aligned, else repacked by the ingest kernel.
        // Frames that lie at equal distances in one allocation (a batch tensor) travel as one copy; else one
        // copy per frame.
        const size_t src_bytes = (size_t)stride_bytes * (height - 1) + width;
        bool even = batch > 1;
        ptrdiff_t dist = batch > 1 ? images[1] - images[0] : 0;
        for (int f = 2; f < batch && even; f++)
            even = images[f] - images[f - 1] == dist;
        even = even && dist >= (ptrdiff_t)src_bytes && (dist & 3) == 0 && dist <= (ptrdiff_t)src_bytes + (1 << 16);
        if (even) {
            // Equal distances do not make one allocation: frames pinned one by one often lie evenly spaced
            // too (consecutive hipHostMalloc calls, the power-of-two blocks of a caching allocator).  One copy
            // may only span memory of a single allocation, so ask the runtime for the allocation around the
            // first frame and require the last frame's end inside it.
            void *base = nullptr;
            size_t size = 0;
            if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)images[0]) != hipSuccess) {
                (void)hipGetLastError();
                even = false;
            } else {
                const uint8_t *lo = static_cast<const uint8_t *>(base);
                even = images[0] >= lo && images[batch - 1] + src_bytes <= lo + size;
            }
        }
        const size_t fstride = even ? (size_t)dist : (src_bytes + 255) / 256 * 256;
        const size_t need = fstride * (size_t)batch;
        if (ex->d_stage_bytes < need) {
            HIP_TRY(hipStreamSynchronize(ex->stream));
            HIP_TRY(hipStreamSynchronize(ex->aux_stream));
            if (ex->d_stage)
                HIP_TRY(hipFree(ex->d_stage));
            ex->d_stage = nullptr;
            ex->d_stage_bytes = 0;
            HIP_TRY(hipMalloc((void **)&ex->d_stage, need + 256));
            ex->d_stage_bytes = need;
        }
        if (even)
            HIP_TRY(hipMemcpyAsync(ex->d_stage, images[0], fstride * (size_t)(batch - 1) + src_bytes,
                                   hipMemcpyHostToDevice, ex->stream));
        else
            for (int f = 0; f < batch; f++)
                HIP_TRY(hipMemcpyAsync(ex->d_stage + (size_t)f * fstride, images[f], src_bytes, hipMemcpyHostToDevice,
                                       ex->stream));
        return orbx_extract_batch_device(ex, ex->d_stage, batch, width, height, stride_bytes, (int64_t)fstride);
    }
    // Pageable rows go through a pinned staging buffer in the level-0 layout: one contiguous
    // asynchronous copy per frame (a strided copy out of pageable memory costs milliseconds).
    const size_t frame_bytes = (size_t)L0.pitch * L0.h;
    const size_t need = frame_bytes * (size_t)ex->p.max_batch;
    if (ex->h_stage_bytes < need) {
        HIP_TRY(hipStreamSynchronize(ex->stream));
        if (ex->h_stage)
            HIP_TRY(hipHostFree(ex->h_stage));
        ex->h_stage = nullptr;
        ex->h_stage_bytes = 0;
        HIP_TRY(hipHostMalloc((void **)&ex->h_stage, need, hipHostMallocDefault));
        ex->h_stage_bytes = need;
    } else {
        HIP_TRY(hipEventSynchronize(ex->ev_staged));  // the previous batch has left the staging buffer
    }
    HIP_TRY(prof_mark(ex, 0, 0, ex->stream));
    // rows into the staging buffer: on the calling thread for a few frames, else on up to four threads,
    // each followed by the asynchronous copies of its frames (in frame order on the one stream)
    auto stage = [&](int f) {
        uint8_t *st = ex->h_stage + (size_t)f * frame_bytes;
        if (stride_bytes == L0.pitch)
            memcpy(st, images[f], frame_bytes);
        else
            for (int y = 0; y < height; y++)
                memcpy(st + (size_t)y * L0.pitch, images[f] + (size_t)y * stride_bytes, width);
    };
    const int nthreads = batch >= 8 ? 4 : 1;
    if (nthreads > 1) {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; t++)
            pool.emplace_back([&, t]() {
                for (int f = t; f < batch; f += nthreads)
                    stage(f);
            });
        for (auto &th : pool)
            th.join();
    }
    for (int f = 0; f < batch; f++) {
        if (nthreads == 1)
            stage(f);
        HIP_TRY(hipMemcpyAsync(ex->buf.pyr + (size_t)f * ex->geom.slab + L0.off, ex->h_stage + (size_t)f * frame_bytes,
                               frame_bytes, hipMemcpyHostToDevice, ex->stream));
    }
    HIP_TRY(hipEventRecord(ex->ev_staged, ex->stream));
    ex->buf.l0 = OrbxLevel0{ex->buf.pyr + L0.off, (long long)ex->geom.slab, L0.pitch};
    HIP_TRY(prof_mark(ex, 0, 1, ex->stream));
    return run_pipeline(ex);
}

extern "C" int orbx_extractor_profile(orbx_extractor *ex, int enable)
{
    if (!ex)
        return ORBX_E_ARG;
    ex->profiling = enable != 0;
    if (enable)
        ex->prof_used = 0;
    return ORBX_OK;
}

extern "C" int orbx_extractor_stage_times(orbx_extractor *ex, float *ms_sum, int *nbatches)
{
    if (!ex || !ms_sum || !nbatches)
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(ex->device));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    HIP_TRY(hipStreamSynchronize(ex->aux_stream));
    for (int i = 0; i < ORBX_NSTAGES; i++)
        ms_sum[i] = 0.f;
    for (size_t b = 0; b < ex->prof_used; b++) {
        const hipEvent_t *ev = &ex->prof_events[b * (2 * ORBX_NSTAGES)];
        for (int i = 0; i < ORBX_NSTAGES; i++) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            ms_sum[i] += ms;
        }
    }
    *nbatches = (int)ex->prof_used;
    return ORBX_OK;
}

extern "C" int orbx_batch_counts(orbx_extractor *ex, int *n_per_frame, int *n_per_level)
{
    if (!ex)
        return ORBX_E_ARG;
    if (!ex->have_batch)
        return ORBX_E_STATE;
    HIP_TRY(hipSetDevice(ex->device));
    const int B = ex->geom.batch, NL = ex->geom.nlevels;
    ex->h_counts.resize(B);
    ex->h_nkept.resize((size_t)B * NL);
    int status[ORBX_ST_COUNT];
    HIP_TRY(hipMemcpyAsync(ex->h_counts.data(), ex->buf.out_n, sizeof(int) * B, hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipMemcpyAsync(ex->h_nkept.data(), ex->buf.nkept, sizeof(int) * B * NL, hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipMemcpyAsync(status, ex->buf.status, sizeof(status), hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    if (status[ORBX_ST_CAND_OVERFLOW] || status[ORBX_ST_NODE_OVERFLOW])
        return ORBX_E_CAPACITY;
    if (n_per_frame)
        memcpy(n_per_frame, ex->h_counts.data(), sizeof(int) * B);
    if (n_per_level)
        memcpy(n_per_level, ex->h_nkept.data(), sizeof(int) * B * NL);
    return ORBX_OK;
}

extern "C" int orbx_batch_download(orbx_extractor *ex, int frame, orbx_keypoint *keypoints, uint8_t *descriptors,
                                   int cap, int *n_out)
{
    if (!ex || !n_out)
        return ORBX_E_ARG;
    if (!ex->have_batch)
        return ORBX_E_STATE;
    if (frame < 0 || frame >= ex->geom.batch)
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(ex->device));
    int n = 0;
    HIP_TRY(hipMemcpyAsync(&n, ex->buf.out_n + frame, sizeof(int), hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    *n_out = n;
    if (n > cap)
        return ORBX_E_CAPACITY;
    if (n > 0) {
        const size_t o = (size_t)frame * ex->geom.out_cap;
        if (keypoints)
            HIP_TRY(hipMemcpyAsync(keypoints, ex->buf.out_kp + o, sizeof(orbx_keypoint) * n, hipMemcpyDeviceToHost,
                                   ex->stream));
        if (descriptors)
            HIP_TRY(hipMemcpyAsync(descriptors, ex->buf.out_desc + o * 32, (size_t)32 * n, hipMemcpyDeviceToHost,
                                   ex->stream));
        HIP_TRY(hipStreamSynchronize(ex->stream));
    }
    return ORBX_OK;
}

extern "C" int orbx_batch_download_all(orbx_extractor *ex, orbx_keypoint *keypoints, uint8_t *descriptors,
                                       int cap_per_frame, int *counts)
{
    if (!ex || !keypoints || !descriptors || !counts)
        return ORBX_E_ARG;
    if (!ex->have_batch)
        return ORBX_E_STATE;
    const int B = ex->geom.batch, oc = ex->geom.out_cap;
    if (cap_per_frame < oc)
        return ORBX_E_CAPACITY;
    HIP_TRY(hipSetDevice(ex->device));
    int status[ORBX_ST_COUNT];
    HIP_TRY(hipMemcpyAsync(counts, ex->buf.out_n, sizeof(int) * B, hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipMemcpyAsync(status, ex->buf.status, sizeof(status), hipMemcpyDeviceToHost, ex->stream));
    // frame f of the device arrays starts at f * out_cap entries; the caller's at f * cap_per_frame.
    // Page-locked destinations are written by the copy engine; pageable ones are filled from a pinned
    // staging buffer on four threads (a direct copy into pageable memory runs at a few GB/s).
    const size_t kp_row = (size_t)oc * sizeof(orbx_keypoint), de_row = (size_t)oc * 32;
    auto is_pinned = [](const void *p) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return attr.type == hipMemoryTypeHost;
    };
    if (is_pinned(keypoints) && is_pinned(descriptors)) {
        if (cap_per_frame == oc) {
            HIP_TRY(hipMemcpyAsync(keypoints, ex->buf.out_kp, kp_row * B, hipMemcpyDeviceToHost, ex->stream));
            HIP_TRY(hipMemcpyAsync(descriptors, ex->buf.out_desc, de_row * B, hipMemcpyDeviceToHost, ex->stream));
        } else {
            for (int f = 0; f < B; f++) {
                HIP_TRY(hipMemcpyAsync(keypoints + (size_t)f * cap_per_frame, ex->buf.out_kp + (size_t)f * oc, kp_row,
                                       hipMemcpyDeviceToHost, ex->stream));
                HIP_TRY(hipMemcpyAsync(descriptors + (size_t)f * cap_per_frame * 32, ex->buf.out_desc + (size_t)f * de_row,
                                       de_row, hipMemcpyDeviceToHost, ex->stream));
            }
        }
        HIP_TRY(hipStreamSynchronize(ex->stream));
    } else {
        const size_t need = (kp_row + de_row) * (size_t)ex->p.max_batch + 256;  // + the header orbx_extract keeps there
        if (ex->h_out_bytes < need) {
            if (ex->h_out)
                HIP_TRY(hipHostFree(ex->h_out));
            ex->h_out = nullptr;
            ex->h_out_bytes = 0;
            HIP_TRY(hipHostMalloc((void **)&ex->h_out, need, hipHostMallocDefault));
            ex->h_out_bytes = need;
        }
        uint8_t *h_kp = ex->h_out, *h_de = ex->h_out + kp_row * B;
        HIP_TRY(hipMemcpyAsync(h_kp, ex->buf.out_kp, kp_row * B, hipMemcpyDeviceToHost, ex->stream));
        HIP_TRY(hipMemcpyAsync(h_de, ex->buf.out_desc, de_row * B, hipMemcpyDeviceToHost, ex->stream));
        HIP_TRY(hipStreamSynchronize(ex->stream));
        // only the rows that hold results
        auto unload = [&](int f) {
            const size_t n = (size_t)std::min(std::max(counts[f], 0), oc);
            memcpy(keypoints + (size_t)f * cap_per_frame, h_kp + (size_t)f * kp_row, n * sizeof(orbx_keypoint));
            memcpy(descriptors + (size_t)f * cap_per_frame * 32, h_de + (size_t)f * de_row, n * 32);
        };
        const int nthreads = B >= 8 ? 4 : 1;
        if (nthreads > 1) {
            std::vector<std::thread> pool;
            for (int t = 0; t < nthreads; t++)
                pool.emplace_back([&, t]() {
                    for (int f = t; f < B; f += nthreads)
                        unload(f);
                });
            for (auto &th : pool)
                th.join();
        } else
            for (int f = 0; f < B; f++)
                unload(f);
    }
    if (status[ORBX_ST_CAND_OVERFLOW] || status[ORBX_ST_NODE_OVERFLOW])
        return ORBX_E_CAPACITY;
    return ORBX_OK;
}

extern "C" int orbx_batch_device_results(orbx_extractor *ex, const orbx_keypoint **d_keypoints,
                                         const uint8_t **d_descriptors, const int32_t **d_counts, int *max_keypoints)
{
    if (!ex)
        return ORBX_E_ARG;
    if (d_keypoints) *d_keypoints = ex->buf.out_kp;
    if (d_descriptors) *d_descriptors = ex->buf.out_desc;
    if (d_counts) *d_counts = ex->buf.out_n;
    if (max_keypoints) *max_keypoints = ex->cap_geom.out_cap;
    return ORBX_OK;
}

extern "C" int orbx_extract(orbx_extractor *ex, const uint8_t *image, int width, int height, int stride_bytes,
                            orbx_keypoint *keypoints, uint8_t *descriptors, int cap, int *n_out)
{
    if (!ex || !n_out)
        return ORBX_E_ARG;
    if (!image || width <= 0 || height <= 0) {
        *n_out = -1;  // empty image: the reference returns without touching its outputs
        return ORBX_OK;
    }
    const uint8_t *one[1] = {image};
    int rc = orbx_extract_batch_host(ex, one, 1, width, height, stride_bytes);
    if (rc != ORBX_OK)
        return rc;
    // One round trip for everything: count, status and the frame's whole result rows go to page-locked
    // staging with asynchronous copies behind the kernels, then one wait.  (Asking for the count first and
    // copying exactly that many rows afterwards costs two more waits than the 150 KB of spare rows cost.)
    const size_t oc = (size_t)ex->geom.out_cap;
    const size_t kp_bytes = oc * sizeof(orbx_keypoint), de_bytes = oc * 32, head_bytes = 256;
    const size_t need = (kp_bytes + de_bytes) * (size_t)ex->p.max_batch + head_bytes;
    if (ex->h_out_bytes < need) {
        if (ex->h_out)
            HIP_TRY(hipHostFree(ex->h_out));
        ex->h_out = nullptr;
        ex->h_out_bytes = 0;
        HIP_TRY(hipHostMalloc((void **)&ex->h_out, need, hipHostMallocDefault));
        ex->h_out_bytes = need;
    }
    // the header sits behind the rows of a whole batch, which orbx_batch_download_all stages in the same buffer
    int *head = reinterpret_cast<int *>(ex->h_out + (kp_bytes + de_bytes) * (size_t)ex->p.max_batch);
    static_assert((1 + ORBX_ST_COUNT) * sizeof(int) <= 256, "header of the staging buffer");
    uint8_t *h_kp = ex->h_out, *h_de = ex->h_out + kp_bytes;
    if (ex->p.max_batch == 1) {
        // count and status words are neighbours, and so are the frame's keypoint and descriptor rows
        HIP_TRY(hipMemcpyAsync(head, ex->buf.out_n, sizeof(int) * (1 + ORBX_ST_COUNT), hipMemcpyDeviceToHost, ex->stream));
        if (keypoints || descriptors)
            HIP_TRY(hipMemcpyAsync(h_kp, ex->buf.out_kp, kp_bytes + (descriptors ? de_bytes : 0), hipMemcpyDeviceToHost,
                                   ex->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(head, ex->buf.out_n, sizeof(int), hipMemcpyDeviceToHost, ex->stream));
        HIP_TRY(hipMemcpyAsync(head + 1, ex->buf.status, sizeof(int) * ORBX_ST_COUNT, hipMemcpyDeviceToHost, ex->stream));
        if (keypoints)
            HIP_TRY(hipMemcpyAsync(h_kp, ex->buf.out_kp, kp_bytes, hipMemcpyDeviceToHost, ex->stream));
        if (descriptors)
            HIP_TRY(hipMemcpyAsync(h_de, ex->buf.out_desc, de_bytes, hipMemcpyDeviceToHost, ex->stream));
    }
    HIP_TRY(hipStreamSynchronize(ex->stream));
    if (head[1 + ORBX_ST_CAND_OVERFLOW] || head[1 + ORBX_ST_NODE_OVERFLOW])
        return ORBX_E_CAPACITY;
    const int n = head[0];
    *n_out = n;
    if (n > cap)
        return ORBX_E_CAPACITY;
    if (n > 0 && keypoints)
        memcpy(keypoints, h_kp, (size_t)n * sizeof(orbx_keypoint));
    if (n > 0 && descriptors)
        memcpy(descriptors, h_de, (size_t)n * 32);
    return ORBX_OK;
}

static inline int reflect101(int i, int n)
{
    if (n == 1)
        return 0;
    while (i < 0 || i >= n)
        i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

static int download_level(orbx_extractor *ex, const uint8_t *slab_buf, int frame, int level, int border, uint8_t *dst,
                          int dst_stride, int *width, int *height)
{
    if (!ex)
        return ORBX_E_ARG;
    if (!ex->have_batch)
        return ORBX_E_STATE;
    if (frame < 0 || frame >= ex->geom.batch || level < 0 || level >= ex->geom.nlevels || border < 0)
        return ORBX_E_ARG;
    const OrbxLevel &L = ex->geom.lv[level];
    if (width) *width = L.w;
    if (height) *height = L.h;
    if (!dst)
        return ORBX_OK;
    if (dst_stride < L.w + 2 * border)
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(ex->device));
    uint8_t *inner = dst + (size_t)border * dst_stride + border;
    const uint8_t *from = slab_buf + (size_t)frame * ex->geom.slab + L.off;
    size_t from_pitch = L.pitch;
    if (level == 0 && slab_buf == ex->buf.pyr) {  // level 0 may still lie in the caller's frames
        from = ex->buf.l0.ptr + (size_t)frame * ex->buf.l0.fstride;
        from_pitch = ex->buf.l0.pitch;
    }
    HIP_TRY(hipMemcpy2DAsync(inner, dst_stride, from, from_pitch, L.w, L.h, hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    if (border > 0) {
        // BORDER_REFLECT_101 frame, as copyMakeBorder leaves it around each level
        for (int y = 0; y < L.h; y++) {
            uint8_t *row = inner + (size_t)y * dst_stride;
            for (int i = 1; i <= border; i++) {
                row[-i] = row[reflect101(-i, L.w)];
                row[L.w - 1 + i] = row[reflect101(L.w - 1 + i, L.w)];
            }
        }
        for (int i = 1; i <= border; i++) {
            memcpy(inner + (ptrdiff_t)(-i) * dst_stride - border,
                   inner + (ptrdiff_t)reflect101(-i, L.h) * dst_stride - border, L.w + 2 * border);
            memcpy(inner + (ptrdiff_t)(L.h - 1 + i) * dst_stride - border,
                   inner + (ptrdiff_t)reflect101(L.h - 1 + i, L.h) * dst_stride - border, L.w + 2 * border);
        }
    }
    return ORBX_OK;
}

extern "C" int orbx_pyramid_level(orbx_extractor *ex, int frame, int level, int border, uint8_t *dst, int dst_stride,
                                  int *width, int *height)
{
    return download_level(ex, ex ? ex->buf.pyr : nullptr, frame, level, border, dst, dst_stride, width, height);
}

extern "C" int orbx_pyramid_layout(const orbx_extractor *ex, int border, int64_t *offsets, int *widths, int *heights,
                                   int64_t *total_bytes)
{
    if (!ex || border < 0)
        return ORBX_E_ARG;
    if (!ex->have_batch)
        return ORBX_E_STATE;
    int64_t off = 0;
    for (int l = 0; l < ex->geom.nlevels; l++) {
        const OrbxLevel &L = ex->geom.lv[l];
        if (offsets) offsets[l] = off;
        if (widths) widths[l] = L.w;
        if (heights) heights[l] = L.h;
        off += (int64_t)(L.w + 2 * border) * (L.h + 2 * border);
    }
    if (total_bytes)
        *total_bytes = off;
    return ORBX_OK;
}

// All levels of one frame with their reflected frames: the frame's part of the level slab comes back in
// one contiguous copy (plus one for a level 0 that still lies in the caller's frame), one wait, and the
// host lays the rows out and mirrors the borders.
extern "C" int orbx_pyramid_download(orbx_extractor *ex, int frame, int border, uint8_t *dst, int64_t dst_bytes)
{
    if (!ex || !dst || border < 0)
        return ORBX_E_ARG;
    if (!ex->have_batch)
        return ORBX_E_STATE;
    if (frame < 0 || frame >= ex->geom.batch)
        return ORBX_E_ARG;
    const OrbxGeom &g = ex->geom;
    int64_t total = 0;
    (void)orbx_pyramid_layout(ex, border, nullptr, nullptr, nullptr, &total);
    if (dst_bytes < total)
        return ORBX_E_CAPACITY;
    HIP_TRY(hipSetDevice(ex->device));
    const OrbxLevel &L0 = g.lv[0];
    const bool l0_outside = ex->buf.l0.ptr != ex->buf.pyr + L0.off;
    const size_t l0_bytes = l0_outside ? (size_t)ex->buf.l0.pitch * (L0.h - 1) + L0.w : 0;
    const size_t need = (size_t)g.slab + l0_bytes;
    if (ex->h_pyr_bytes < need) {
        if (ex->h_pyr)
            HIP_TRY(hipHostFree(ex->h_pyr));
        ex->h_pyr = nullptr;
        ex->h_pyr_bytes = 0;
        HIP_TRY(hipHostMalloc((void **)&ex->h_pyr, need, hipHostMallocDefault));
        ex->h_pyr_bytes = need;
    }
    // with level 0 outside the slab only the levels above it are fetched from the slab
    const size_t skip = l0_outside && g.nlevels > 1 ? (size_t)g.lv[1].off : 0;
    if (!l0_outside || g.nlevels > 1)
        HIP_TRY(hipMemcpyAsync(ex->h_pyr + skip, ex->buf.pyr + (size_t)frame * g.slab + skip, (size_t)g.slab - skip,
                               hipMemcpyDeviceToHost, ex->stream));
    if (l0_outside)
        HIP_TRY(hipMemcpyAsync(ex->h_pyr + g.slab, ex->buf.l0.ptr + (size_t)frame * ex->buf.l0.fstride, l0_bytes,
                               hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    uint8_t *out = dst;
    for (int l = 0; l < g.nlevels; l++) {
        const OrbxLevel &L = g.lv[l];
        const uint8_t *src = ex->h_pyr + L.off;
        size_t src_pitch = L.pitch;
        if (l == 0 && l0_outside)
            src = ex->h_pyr + g.slab, src_pitch = ex->buf.l0.pitch;
        const int pw = L.w + 2 * border, ph = L.h + 2 * border;
        for (int y = 0; y < ph; y++) {
            const uint8_t *srow = src + (size_t)reflect101(y - border, L.h) * src_pitch;
            uint8_t *drow = out + (size_t)y * pw;
            memcpy(drow + border, srow, L.w);
            for (int i = 1; i <= border; i++) {
                drow[border - i] = srow[reflect101(-i, L.w)];
                drow[border + L.w - 1 + i] = srow[reflect101(L.w - 1 + i, L.w)];
            }
        }
        out += (size_t)pw * ph;
    }
    return ORBX_OK;
}

extern "C" int orbx_debug_level_map(orbx_extractor *ex, int what, int frame, int level, uint8_t *dst, int dst_stride)
{
    if (!ex || !dst || what != 1)
        return ORBX_E_ARG;
    if (!ex->have_batch)
        return ORBX_E_STATE;
    if (frame < 0 || frame >= ex->geom.batch || level < 0 || level >= ex->geom.nlevels)
        return ORBX_E_ARG;
    const OrbxLevel &L = ex->geom.lv[level];
    if (dst_stride < L.w)
        return ORBX_E_ARG;
    // the blurred level lies in tiles: fetch its region and put the rows together here
    HIP_TRY(hipSetDevice(ex->device));
    std::vector<uint8_t> tiled((size_t)ORBX_BLUR_LEVEL_BYTES(L.pitch, L.h));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    HIP_TRY(hipStreamSynchronize(ex->aux_stream));
    HIP_TRY(hipMemcpy(tiled.data(), ex->buf.blur + (size_t)frame * ex->geom.bslab + L.boff, tiled.size(),
                      hipMemcpyDeviceToHost));
    for (int y = 0; y < L.h; y++)
        for (int x = 0; x < L.w; x++)
            dst[(size_t)y * dst_stride + x] = tiled[(size_t)ORBX_BLUR_ADDR(x, y, L.pitch)];
    return ORBX_OK;
}

extern "C" int orbx_debug_candidates(orbx_extractor *ex, int frame, int level, int *xs, int *ys, int *resp, int cap,
                                     int *n_out)
{
    if (!ex || !n_out)
        return ORBX_E_ARG;
    if (!ex->have_batch)
        return ORBX_E_STATE;
    if (frame < 0 || frame >= ex->geom.batch || level < 0 || level >= ex->geom.nlevels)
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(ex->device));
    const OrbxGeom &g = ex->geom;
    const OrbxLevel &L = g.lv[level];
    int n = 0;
    HIP_TRY(hipMemcpyAsync(&n, ex->buf.ncand + frame * g.nlevels + level, sizeof(int), hipMemcpyDeviceToHost,
                           ex->stream));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    *n_out = 0;
    if (n > L.cand_cap)
        return ORBX_E_CAPACITY;
    if (n == 0)
        return ORBX_OK;
    std::vector<uint32_t> xy(n), key(n);
    std::vector<uint8_t> rp(n);
    const int ncell = L.ncols * L.nrows;
    std::vector<uint8_t> ini(std::max(ncell, 1));
    const size_t o = (size_t)frame * g.cand_per_frame + L.cand_base;
    HIP_TRY(hipMemcpyAsync(xy.data(), ex->buf.cand_xy + o, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipMemcpyAsync(key.data(), ex->buf.cand_key + o, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipMemcpyAsync(rp.data(), ex->buf.cand_resp + o, (size_t)n, hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipMemcpyAsync(ini.data(), ex->buf.cell_flag + (size_t)frame * g.cells_per_frame + L.cell_base,
                           (size_t)ncell, hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    // the device list is unordered and still holds the candidates that only count when their cell
    // has none above iniThFAST; apply that rule and sort into the reference's order
    const int t_ini = std::min(std::max(g.ini_th, 1), 255), t_min = std::min(std::max(g.min_th, 1), 255);
    std::vector<int> keep;
    for (int i = 0; i < n; i++) {
        const int A = (int)rp[i] + 1;
        if (A > t_ini || (ini[key[i] >> 12] == 0 && A > t_min))
            keep.push_back(i);
    }
    std::sort(keep.begin(), keep.end(), [&](int a, int b) { return key[a] < key[b]; });
    *n_out = (int)keep.size();
    if ((int)keep.size() > cap)
        return ORBX_E_CAPACITY;
    for (size_t j = 0; j < keep.size(); j++) {
        const int i = keep[j];
        if (xs) xs[j] = (int)(xy[i] & 0xffff);
        if (ys) ys[j] = (int)(xy[i] >> 16);
        if (resp) resp[j] = rp[i];
    }
    return ORBX_OK;
}

// Standalone quadtree run (parity tests): n candidates with coordinates relative to the detection
// region of region_w x region_h, quota N.  out_idx receives the kept candidate indices in list order.
extern "C" int orbx_debug_octree(int device, const int *xs, const int *ys, const int *resp, int n, int region_w,
                                 int region_h, int quota, int *out_idx, int cap, int *n_out)
{
    if (!xs || !ys || !resp || !n_out || n < 0 || region_w < 1 || region_h < 1 || quota < 1)
        return ORBX_E_ARG;
    const int n_ini = (int)roundf((float)region_w / (float)region_h);
    if (n_ini < 1)
        return ORBX_E_GEOMETRY;
    const int node_cap = align_up(std::max(quota, 4 * n_ini) + 4, 4);
    if (orbx_octree_lds_bytes(node_cap) > 160 * 1024)
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(device));
    std::vector<uint32_t> xy(std::max(n, 1));
    std::vector<uint8_t> rp(std::max(n, 1));
    for (int i = 0; i < n; i++) {
        xy[i] = (uint32_t)xs[i] | ((uint32_t)ys[i] << 16);
        rp[i] = (uint8_t)resp[i];
    }
    uint32_t *d_xy = nullptr, *d_oxy = nullptr;
    uint8_t *d_rp = nullptr, *d_orp = nullptr;
    uint16_t *d_node = nullptr;
    int *d_oidx = nullptr, *d_n = nullptr, *d_st = nullptr;
    int rc = ORBX_OK;
    hipError_t e = dmalloc(&d_xy, (size_t)n);
    if (e == hipSuccess) e = dmalloc(&d_rp, (size_t)n);
    if (e == hipSuccess) e = dmalloc(&d_node, (size_t)n);
    if (e == hipSuccess) e = dmalloc(&d_oxy, (size_t)node_cap);
    if (e == hipSuccess) e = dmalloc(&d_orp, (size_t)node_cap);
    if (e == hipSuccess) e = dmalloc(&d_oidx, (size_t)node_cap);
    if (e == hipSuccess) e = dmalloc(&d_n, 1);
    if (e == hipSuccess) e = dmalloc(&d_st, (size_t)ORBX_ST_COUNT);
    if (e == hipSuccess) e = hipMemset(d_st, 0, sizeof(int) * ORBX_ST_COUNT);
    if (e == hipSuccess && n > 0) e = hipMemcpy(d_xy, xy.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess && n > 0) e = hipMemcpy(d_rp, rp.data(), (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = orbx_launch_octree_single(d_xy, d_rp, d_node, n, region_w, region_h, quota, d_oxy, d_orp, d_oidx, d_n, d_st,
                                      node_cap, nullptr);
    int m = 0, st[ORBX_ST_COUNT] = {0};
    if (e == hipSuccess) e = hipMemcpy(&m, d_n, sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(st, d_st, sizeof(st), hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        *n_out = m;
        if (st[ORBX_ST_NODE_OVERFLOW] || m > cap)
            rc = ORBX_E_CAPACITY;
        else if (m > 0 && out_idx)
            e = hipMemcpy(out_idx, d_oidx, sizeof(int) * m, hipMemcpyDeviceToHost);
    }
    void *ptrs[] = {d_xy, d_rp, d_node, d_oxy, d_orp, d_oidx, d_n, d_st};
    for (void *p : ptrs)
        if (p)
            (void)hipFree(p);
    if (e != hipSuccess) {
        g_last_hip = (int)e;
        return ORBX_E_HIP;
    }
    return rc;
}

// ------------------------------------------------------------------ matcher

extern "C" int orbx_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) +
           __builtin_popcountll(x[3] ^ y[3]);
}

struct orbx_matcher {
    int device;
    hipStream_t own_stream, stream;
    int max_queries;
    uint32_t *d_partial;
    size_t partial_cap;  // in uint2 entries
    uint8_t *d_q, *d_db;
    size_t q_cap, db_cap;  // bytes
    int32_t *d_best;
    uint16_t *d_d1, *d_d2;
    // pair-list scratch
    uint8_t *d_a, *d_b;
    int *d_pairs;
    uint16_t *d_pd;
    size_t a_cap, b_cap, pairs_cap, pd_cap;
    // segment starts and results of the distinctive-descriptor call
    int *d_seg;
    size_t seg_cap;
    // one allocation carved into the arrays of a SearchByBoW call
    uint8_t *d_ws;
    size_t ws_cap;
    // page-locked mirror of the input and result parts of d_ws: the host-array entry points gather their
    // arguments there and move them with one copy each way instead of one per array
    uint8_t *h_ws;
    size_t h_ws_cap;
    bool force_valu;  // brute-force top-2 on the vector ALUs even for large query sets (ORBX_TOP2=valu)
};

extern "C" int orbx_matcher_create(int device, int max_queries, orbx_matcher **out)
{
    if (!out || max_queries < 1)
        return ORBX_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev)
        return ORBX_E_NODEVICE;
    orbx_matcher *m = new (std::nothrow) orbx_matcher();
    if (!m)
        return ORBX_E_CAPACITY;
    memset(m, 0, sizeof(*m));
    m->device = device;
    m->max_queries = max_queries;
    {
        const char *mode = getenv("ORBX_TOP2");  // "valu": keep the vector-ALU kernel (comparison runs)
        m->force_valu = mode && strcmp(mode, "valu") == 0;
    }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking);
    m->stream = m->own_stream;
    if (e == hipSuccess) e = dmalloc(&m->d_best, (size_t)max_queries);
    if (e == hipSuccess) e = dmalloc(&m->d_d1, (size_t)max_queries);
    if (e == hipSuccess) e = dmalloc(&m->d_d2, (size_t)max_queries);
    if (e != hipSuccess) {
        g_last_hip = (int)e;
        orbx_matcher_destroy(m);
        return ORBX_E_HIP;
    }
    *out = m;
    return ORBX_OK;
}

extern "C" void orbx_matcher_destroy(orbx_matcher *m)
{
    if (!m)
        return;
    (void)hipSetDevice(m->device);
    if (m->stream)
        (void)hipStreamSynchronize(m->stream);
    void *ptrs[] = {m->d_partial, m->d_q, m->d_db, m->d_best, m->d_d1, m->d_d2, m->d_a, m->d_b, m->d_pairs, m->d_pd,
                    m->d_seg, m->d_ws};
    for (void *p : ptrs)
        if (p)
            (void)hipFree(p);
    if (m->h_ws)
        (void)hipHostFree(m->h_ws);
    if (m->own_stream)
        (void)hipStreamDestroy(m->own_stream);
    delete m;
}

extern "C" int orbx_matcher_set_stream(orbx_matcher *m, void *hip_stream)
{
    if (!m)
        return ORBX_E_ARG;
    m->stream = hip_stream ? (hipStream_t)hip_stream : m->own_stream;
    return ORBX_OK;
}

template <typename T> static int grow(T **p, size_t *cap, size_t need, hipStream_t s)
{
    if (need <= *cap)
        return ORBX_OK;
    HIP_TRY(hipStreamSynchronize(s));
    if (*p)
        HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t n = need + need / 4;
    HIP_TRY(hipMalloc((void **)p, n * sizeof(T)));
    *cap = n;
    return ORBX_OK;
}

static int grow_pinned(orbx_matcher *m, size_t need)
{
    if (need <= m->h_ws_cap)
        return ORBX_OK;
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->h_ws)
        HIP_TRY(hipHostFree(m->h_ws));
    m->h_ws = nullptr;
    m->h_ws_cap = 0;
    const size_t n = need + need / 4;
    HIP_TRY(hipHostMalloc((void **)&m->h_ws, n, hipHostMallocDefault));
    m->h_ws_cap = n;
    return ORBX_OK;
}

// rows per database chunk: enough chunks to fill the chip, row index fits 16 bits
static void top2_chunks(int nq, int64_t ndb, int *chunk_rows, int *nchunks)
{
    const int qblocks = (nq + 255) / 256;
    const int want = std::max(1, 4096 / qblocks);  // ~4096 workgroups
    int64_t rows = (ndb + want - 1) / want;
    rows = std::max<int64_t>(256, std::min<int64_t>(rows, 32768));
    rows = (rows + 3) / 4 * 4;
    *chunk_rows = (int)rows;
    *nchunks = (int)((ndb + rows - 1) / rows);
}

extern "C" int orbx_match_top2_device(orbx_matcher *m, const uint8_t *d_queries, int nq, const uint8_t *d_db,
                                      int64_t ndb, int64_t idx_base, int32_t *d_best_idx, uint16_t *d_d1,
                                      uint16_t *d_d2)
{
    if (!m || !d_queries || nq < 1 || ndb < 0 || (ndb > 0 && !d_db) || !d_best_idx || !d_d1 || !d_d2)
        return ORBX_E_ARG;
    if (idx_base + ndb > INT32_MAX)
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(m->device));
    // 32 queries fill a matrix-core tile; below that the vector kernel (one query per lane) does less work
    const bool mfma = nq >= 32 && !m->force_valu;
    int chunk_rows = 256, nchunks = 0;
    if (ndb > 0) {
        if (mfma)
            orbx_top2_mfma_chunks(nq, ndb, &chunk_rows, &nchunks);
        else
            top2_chunks(nq, ndb, &chunk_rows, &nchunks);
    }
    int rc = grow(&m->d_partial, &m->partial_cap, (size_t)std::max(nchunks, 1) * nq * 2, m->stream);
    if (rc != ORBX_OK)
        return rc;
    if (mfma)
        HIP_TRY(orbx_launch_top2_mfma(d_queries, nq, d_db, ndb, idx_base, m->d_partial, chunk_rows, nchunks, d_best_idx,
                                      d_d1, d_d2, m->stream));
    else
        HIP_TRY(orbx_launch_top2(d_queries, nq, d_db, ndb, idx_base, m->d_partial, chunk_rows, nchunks, d_best_idx, d_d1,
                                 d_d2, m->stream));
    return ORBX_OK;
}

extern "C" int orbx_match_top2(orbx_matcher *m, const uint8_t *queries, int nq, const uint8_t *db, int64_t ndb,
                               int64_t idx_base, int32_t *best_idx, uint16_t *d1, uint16_t *d2)
{
    if (!m || !queries || nq < 1 || nq > m->max_queries || ndb < 0 || (ndb > 0 && !db) || !best_idx || !d1 || !d2)
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(m->device));
    int rc = grow(&m->d_q, &m->q_cap, (size_t)nq * 32, m->stream);
    if (rc == ORBX_OK)
        rc = grow(&m->d_db, &m->db_cap, (size_t)std::max<int64_t>(ndb, 1) * 32, m->stream);
    if (rc != ORBX_OK)
        return rc;
    HIP_TRY(hipMemcpyAsync(m->d_q, queries, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
    if (ndb > 0)
        HIP_TRY(hipMemcpyAsync(m->d_db, db, (size_t)ndb * 32, hipMemcpyHostToDevice, m->stream));
    rc = orbx_match_top2_device(m, m->d_q, nq, m->d_db, ndb, idx_base, m->d_best, m->d_d1, m->d_d2);
    if (rc != ORBX_OK)
        return rc;
    HIP_TRY(hipMemcpyAsync(best_idx, m->d_best, sizeof(int32_t) * nq, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(d1, m->d_d1, sizeof(uint16_t) * nq, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(d2, m->d_d2, sizeof(uint16_t) * nq, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

extern "C" int orbx_top2_merge(int nshards, int nq, const int32_t *best_idx, const uint16_t *d1, const uint16_t *d2,
                               int32_t *out_idx, uint16_t *out_d1, uint16_t *out_d2)
{
    if (nshards < 1 || nq < 0 || !best_idx || !d1 || !d2 || !out_idx || !out_d1 || !out_d2)
        return ORBX_E_ARG;
    for (int i = 0; i < nq; i++) {
        int b1 = 256, b2 = 256;
        int32_t bi = -1;
        for (int s = 0; s < nshards; s++) {
            const size_t o = (size_t)s * nq + i;
            const int a = d1[o], b = d2[o];
            if (a < b1) {
                b2 = b1;
                b1 = a;
                bi = best_idx[o];
            } else if (a < b2)
                b2 = a;
            if (b < b2)
                b2 = b;
        }
        out_idx[i] = bi;
        out_d1[i] = (uint16_t)b1;
        out_d2[i] = (uint16_t)b2;
    }
    return ORBX_OK;
}

extern "C" int orbx_top2_merge_shards_device(orbx_matcher *m, int nshards, int nq, const uint8_t *d_blocks,
                                             int64_t shard_stride_bytes, int32_t *d_best_idx, uint16_t *d_d1,
                                             uint16_t *d_d2)
{
    if (!m || nshards < 1 || nq < 1 || !d_blocks || !d_best_idx || !d_d1 || !d_d2 ||
        shard_stride_bytes < (int64_t)8 * nq || (shard_stride_bytes & 3) != 0 ||
        (reinterpret_cast<uintptr_t>(d_blocks) & 3) != 0)
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(orbx_launch_top2_merge_shards(d_blocks, nshards, nq, shard_stride_bytes, d_best_idx, d_d1, d_d2, m->stream));
    return ORBX_OK;
}

extern "C" int orbx_ratio_test(int nq, const uint16_t *d1, const uint16_t *d2, int th, float ratio, uint8_t *accept)
{
    if (nq < 0 || !d1 || !d2)
        return ORBX_E_ARG;
    int n = 0;
    for (int i = 0; i < nq; i++) {
        const int ok = (int)d1[i] <= th && (float)d1[i] < ratio * (float)d2[i];
        if (accept)
            accept[i] = (uint8_t)ok;
        n += ok;
    }
    return n;
}

// ---- Frame::ComputeStereoMatches on the last batches of a left and a right extractor ----

static int stereo_check(const orbx_extractor *left, const orbx_extractor *right)
{
    if (!left || !right || left == right)
        return ORBX_E_ARG;
    if (!left->have_batch || !right->have_batch)
        return ORBX_E_STATE;
    const OrbxGeom &a = left->geom, &b = right->geom;
    if (left->device != right->device || a.batch != b.batch || a.nlevels != b.nlevels || a.out_cap != b.out_cap ||
        a.slab != b.slab || left->geom_w != right->geom_w || left->geom_h != right->geom_h ||
        left->p.scale_factor != right->p.scale_factor)
        return ORBX_E_ARG;  // the two images of a stereo frame share size and extractor settings
    if (a.out_cap > 65535)
        return ORBX_E_CAPACITY;  // the row table and the packed best keys hold right-keypoint indices in 16 bits
    return ORBX_OK;
}

extern "C" int orbx_compute_stereo_matches_device(orbx_extractor *left, orbx_extractor *right, float mb, float mbf,
                                                  float *d_uright, float *d_depth, int32_t *d_nmatches)
{
    int rc = stereo_check(left, right);
    if (rc != ORBX_OK)
        return rc;
    if (!d_uright || !d_depth || !d_nmatches || !(mb > 0.f) || !(mbf > 0.f))
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(left->device));
    {
        const OrbxGeom &c = left->cap_geom;
        const size_t B = (size_t)left->p.max_batch;
        if (!left->st_rowstart)
            HIP_TRY(hipMalloc((void **)&left->st_rowstart, B * (size_t)(left->p.max_height + 1) * sizeof(int)));
        if (!left->st_sorted)
            HIP_TRY(hipMalloc((void **)&left->st_sorted, B * (size_t)c.out_cap * 3 * sizeof(uint4)));
        if (!left->st_sad)
            HIP_TRY(hipMalloc((void **)&left->st_sad, B * (size_t)c.out_cap * sizeof(int)));
        if (!left->ev_stereo)
            HIP_TRY(hipEventCreateWithFlags(&left->ev_stereo, hipEventDisableTiming));
    }
    // the right image's batch runs on the right handle's stream
    HIP_TRY(hipEventRecord(left->ev_stereo, right->stream));
    HIP_TRY(hipStreamWaitEvent(left->stream, left->ev_stereo, 0));
    HIP_TRY(orbx_launch_stereo(left->geom, left->buf, right->buf, left->sf, left->inv, mb, mbf, left->st_rowstart,
                               left->st_sorted, left->st_sad, d_uright, d_depth, d_nmatches, left->stream));
    // the right handle must not start its next batch before these kernels have read its results
    HIP_TRY(hipEventRecord(left->ev_stereo, left->stream));
    HIP_TRY(hipStreamWaitEvent(right->stream, left->ev_stereo, 0));
    return ORBX_OK;
}

extern "C" int orbx_compute_stereo_matches(orbx_extractor *left, orbx_extractor *right, int frame, float mb, float mbf,
                                           float *uright, float *depth, int cap, int *n_keypoints)
{
    int rc = stereo_check(left, right);
    if (rc != ORBX_OK)
        return rc;
    if (frame < 0 || frame >= left->geom.batch || !uright || !depth || !n_keypoints)
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(left->device));
    {
        const size_t n = (size_t)left->p.max_batch * left->cap_geom.out_cap;
        if (!left->st_uright)
            HIP_TRY(hipMalloc((void **)&left->st_uright, n * sizeof(float)));
        if (!left->st_depth)
            HIP_TRY(hipMalloc((void **)&left->st_depth, n * sizeof(float)));
        if (!left->st_nmatch)
            HIP_TRY(hipMalloc((void **)&left->st_nmatch, (size_t)left->p.max_batch * sizeof(int)));
    }
    if (left->st_right != right || left->st_left_serial != left->batch_serial ||
        left->st_right_serial != right->batch_serial || left->st_mb != mb || left->st_mbf != mbf) {
        rc = orbx_compute_stereo_matches_device(left, right, mb, mbf, left->st_uright, left->st_depth, left->st_nmatch);
        if (rc != ORBX_OK)
            return rc;
        left->st_right = right;
        left->st_left_serial = left->batch_serial, left->st_right_serial = right->batch_serial;
        left->st_mb = mb, left->st_mbf = mbf;
    }
    int head[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&head[0], left->buf.out_n + frame, sizeof(int), hipMemcpyDeviceToHost, left->stream));
    HIP_TRY(hipMemcpyAsync(&head[1], left->st_nmatch + frame, sizeof(int), hipMemcpyDeviceToHost, left->stream));
    HIP_TRY(hipStreamSynchronize(left->stream));
    *n_keypoints = head[0];
    if (head[0] > cap)
        return ORBX_E_CAPACITY;
    if (head[0] > 0) {
        const size_t o = (size_t)frame * left->geom.out_cap;
        HIP_TRY(hipMemcpyAsync(uright, left->st_uright + o, sizeof(float) * head[0], hipMemcpyDeviceToHost, left->stream));
        HIP_TRY(hipMemcpyAsync(depth, left->st_depth + o, sizeof(float) * head[0], hipMemcpyDeviceToHost, left->stream));
        HIP_TRY(hipStreamSynchronize(left->stream));
    }
    return head[1];
}

// ---- SearchForInitialization: candidate windows on the host, distances on the GPU, greedy replay ----

namespace {
struct FrameGrid {  // Frame::mGrid[64][48] as CSR (src/Frame.cc:239-253, 409-420)
    static const int COLS = 64, ROWS = 48;
    float min_x, min_y, inv_w, inv_h;
    std::vector<int> start, items;
    void build(const orbx_keypoint *k, int n, float mnx, float mxx, float mny, float mxy)
    {
        min_x = mnx, min_y = mny;
        inv_w = (float)COLS / (mxx - mnx);
        inv_h = (float)ROWS / (mxy - mny);
        start.assign(COLS * ROWS + 1, 0);
        std::vector<int> cell(n);
        for (int i = 0; i < n; i++) {
            const int px = (int)roundf((k[i].x - min_x) * inv_w), py = (int)roundf((k[i].y - min_y) * inv_h);
            cell[i] = (px < 0 || px >= COLS || py < 0 || py >= ROWS) ? -1 : px * ROWS + py;
            if (cell[i] >= 0)
                start[cell[i] + 1]++;
        }
        for (int c = 0; c < COLS * ROWS; c++)
            start[c + 1] += start[c];
        items.resize(n);
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (int i = 0; i < n; i++)
            if (cell[i] >= 0)
                items[fill[cell[i]]++] = i;
    }
    // Frame::GetFeaturesInArea (src/Frame.cc:349-407), appending to out
    void area(const orbx_keypoint *k, float x, float y, float r, int min_level, int max_level,
              std::vector<int> &out) const
    {
        const int x0 = std::max(0, (int)floorf((x - min_x - r) * inv_w));
        if (x0 >= COLS) return;
        const int x1 = std::min(COLS - 1, (int)ceilf((x - min_x + r) * inv_w));
        if (x1 < 0) return;
        const int y0 = std::max(0, (int)floorf((y - min_y - r) * inv_h));
        if (y0 >= ROWS) return;
        const int y1 = std::min(ROWS - 1, (int)ceilf((y - min_y + r) * inv_h));
        if (y1 < 0) return;
        const bool check = (min_level > 0) || (max_level >= 0);
        for (int ix = x0; ix <= x1; ix++)
            for (int iy = y0; iy <= y1; iy++) {
                const int c = ix * ROWS + iy;
                for (int j = start[c]; j < start[c + 1]; j++) {
                    const orbx_keypoint &kp = k[items[j]];
                    if (check) {
                        if (kp.octave < min_level) continue;
                        if (max_level >= 0 && kp.octave > max_level) continue;
                    }
                    if (fabsf(kp.x - x) < r && fabsf(kp.y - y) < r)
                        out.push_back(items[j]);
                }
            }
    }
};
}  // namespace

extern "C" int orbx_search_for_initialization(orbx_matcher *m, const orbx_keypoint *k1, const uint8_t *d1, int n1,
                                              const orbx_keypoint *k2, const uint8_t *d2, int n2, float min_x,
                                              float max_x, float min_y, float max_y, float *prev_matched,
                                              int window_size, float nnratio, int check_orientation, int *matches12,
                                              int64_t *pair_count)
{
    if (!m || n1 < 0 || n2 < 0 || (n1 > 0 && (!k1 || !d1 || !prev_matched || !matches12)) || (n2 > 0 && (!k2 || !d2)) ||
        !(max_x > min_x) || !(max_y > min_y))
        return ORBX_E_ARG;
    for (int i = 0; i < n1; i++)
        matches12[i] = -1;
    if (pair_count)
        *pair_count = 0;
    if (n1 == 0 || n2 == 0)
        return 0;
    FrameGrid grid;
    grid.build(k2, n2, min_x, max_x, min_y, max_y);
    // candidate windows of the level-0 keypoints of frame 1 (:446-455)
    std::vector<int> first(n1 + 1, 0), cand, pairs;
    for (int i1 = 0; i1 < n1; i1++) {
        first[i1] = (int)cand.size();
        if (k1[i1].octave > 0)
            continue;
        grid.area(k2, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window_size, k1[i1].octave, k1[i1].octave,
                  cand);
    }
    first[n1] = (int)cand.size();
    const int np = (int)cand.size();
    if (pair_count)
        *pair_count = np;
    std::vector<uint16_t> dist(std::max(np, 1));
    if (np > 0) {
        pairs.resize((size_t)np * 2);
        for (int i1 = 0; i1 < n1; i1++)
            for (int c = first[i1]; c < first[i1 + 1]; c++) {
                pairs[2 * (size_t)c] = i1;
                pairs[2 * (size_t)c + 1] = cand[c];
            }
        HIP_TRY(hipSetDevice(m->device));
        int rc = grow(&m->d_a, &m->a_cap, (size_t)n1 * 32, m->stream);
        if (rc == ORBX_OK) rc = grow(&m->d_b, &m->b_cap, (size_t)n2 * 32, m->stream);
        if (rc == ORBX_OK) rc = grow(&m->d_pairs, &m->pairs_cap, (size_t)np * 2, m->stream);
        if (rc == ORBX_OK) rc = grow(&m->d_pd, &m->pd_cap, (size_t)np + 8, m->stream);
        if (rc != ORBX_OK)
            return rc;
        HIP_TRY(hipMemcpyAsync(m->d_a, d1, (size_t)n1 * 32, hipMemcpyHostToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(m->d_b, d2, (size_t)n2 * 32, hipMemcpyHostToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(m->d_pairs, pairs.data(), sizeof(int) * 2 * (size_t)np, hipMemcpyHostToDevice,
                               m->stream));
        HIP_TRY(orbx_launch_pair_dist(m->d_a, m->d_b, m->d_pairs, np, m->d_pd, m->stream));
        HIP_TRY(hipMemcpyAsync(dist.data(), m->d_pd, sizeof(uint16_t) * (size_t)np, hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    // order-dependent part (:459-516): sequential over frame-1 keypoints
    int nmatches = 0;
    std::vector<int> matched_dist(n2, INT_MAX), matches21(n2, -1);
    std::vector<int> hist_item, hist_bin;
    const float factor = ORBX_HISTO_LENGTH / 360.0f;
    for (int i1 = 0; i1 < n1; i1++) {
        if (first[i1] == first[i1 + 1])
            continue;
        int best = INT_MAX, best2 = INT_MAX, best_idx = -1;
        for (int c = first[i1]; c < first[i1 + 1]; c++) {
            const int i2 = cand[c], d = dist[c];
            if (matched_dist[i2] <= d)
                continue;
            if (d < best) {
                best2 = best;
                best = d;
                best_idx = i2;
            } else if (d < best2)
                best2 = d;
        }
        if (best <= ORBX_TH_LOW && (float)best < (float)best2 * nnratio) {
            if (matches21[best_idx] >= 0) {
                matches12[matches21[best_idx]] = -1;
                nmatches--;
            }
            matches12[i1] = best_idx;
            matches21[best_idx] = i1;
            matched_dist[best_idx] = best;
            nmatches++;
            if (check_orientation) {
                float rot = k1[i1].angle - k2[best_idx].angle;
                if (rot < 0.0)
                    rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == ORBX_HISTO_LENGTH)
                    bin = 0;
                hist_item.push_back(i1);
                hist_bin.push_back(bin);
            }
        }
    }
    if (check_orientation) {
        // ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1698-1739)
        int sizes[ORBX_HISTO_LENGTH] = {0};
        for (int b : hist_bin)
            sizes[b]++;
        int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < ORBX_HISTO_LENGTH; i++) {
            const int s = sizes[i];
            if (s > max1) {
                max3 = max2, max2 = max1, max1 = s;
                i3 = i2, i2 = i1, i1 = i;
            } else if (s > max2) {
                max3 = max2, max2 = s;
                i3 = i2, i2 = i;
            } else if (s > max3) {
                max3 = s;
                i3 = i;
            }
        }
        if (max2 < 0.1f * (float)max1)
            i2 = -1, i3 = -1;
        else if (max3 < 0.1f * (float)max1)
            i3 = -1;
        for (size_t k = 0; k < hist_item.size(); k++) {
            const int b = hist_bin[k];
            if (b == i1 || b == i2 || b == i3)
                continue;
            if (matches12[hist_item[k]] >= 0) {
                matches12[hist_item[k]] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < n1; i++)
        if (matches12[i] >= 0) {
            prev_matched[2 * i] = k2[matches12[i]].x;
            prev_matched[2 * i + 1] = k2[matches12[i]].y;
        }
    return nmatches;
}

extern "C" int orbx_search_for_initialization_batch_device(orbx_matcher *m, const orbx_keypoint *d_keypoints,
                                                           const uint8_t *d_descriptors, const int32_t *d_counts,
                                                           int max_keypoints, int nframes, int max_level0, float min_x,
                                                           float max_x, float min_y, float max_y,
                                                           const float *d_prev_matched, float *d_prev_matched_out,
                                                           int window_size, float nnratio, int check_orientation,
                                                           int32_t *d_matches12, int32_t *d_nmatches,
                                                           int32_t *d_pair_counts)
{
    if (!m || !d_keypoints || !d_descriptors || !d_counts || !d_matches12 || !d_nmatches || !d_pair_counts ||
        max_keypoints < 1 || max_keypoints > 65535 || nframes < 1 || max_level0 < 1 || max_level0 > 1024 ||
        !(max_x > min_x) || !(max_y > min_y) || window_size < 0)
        return ORBX_E_ARG;
    if (nframes < 2)
        return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(orbx_launch_search_init(d_keypoints, d_descriptors, d_counts, max_keypoints, nframes - 1, max_level0, min_x,
                                    max_x, min_y, max_y, d_prev_matched, d_prev_matched_out, window_size, nnratio,
                                    check_orientation, d_matches12, d_nmatches, d_pair_counts, m->stream));
    return ORBX_OK;
}

extern "C" int orbx_frame_grid_build_device(orbx_matcher *m, const orbx_keypoint *d_keypoints, int n, float min_x,
                                            float max_x, float min_y, float max_y, int32_t *d_cell_start,
                                            int32_t *d_cell_items)
{
    if (!m || n < 0 || n > 32768 || (n > 0 && (!d_keypoints || !d_cell_items)) || !d_cell_start || !(max_x > min_x) ||
        !(max_y > min_y))
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(m->device));
    int rc = grow(&m->d_seg, &m->seg_cap, (size_t)std::max(n, 1), m->stream);  // sort keys
    if (rc != ORBX_OK)
        return rc;
    HIP_TRY(orbx_launch_grid_build(d_keypoints, n, min_x, max_x, min_y, max_y, (uint32_t *)m->d_seg, d_cell_start,
                                   d_cell_items, m->stream));
    return ORBX_OK;
}

extern "C" int orbx_features_in_area_device(orbx_matcher *m, const orbx_keypoint *d_keypoints, float min_x, float max_x,
                                            float min_y, float max_y, const int32_t *d_cell_start,
                                            const int32_t *d_cell_items, const float *d_x, const float *d_y,
                                            const float *d_r, const int32_t *d_min_level, const int32_t *d_max_level,
                                            int nq, int cap, int32_t *d_out_idx, int32_t *d_out_n)
{
    if (!m || nq < 0 || cap < 0 || !d_cell_start || !(max_x > min_x) || !(max_y > min_y) ||
        (nq > 0 && (!d_x || !d_y || !d_r || !d_min_level || !d_max_level || !d_out_n || (cap > 0 && !d_out_idx))))
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(orbx_launch_features_in_area(d_keypoints, min_x, max_x, min_y, max_y, d_cell_start, d_cell_items, d_x, d_y,
                                         d_r, d_min_level, d_max_level, nq, cap, d_out_idx, d_out_n, m->stream));
    return ORBX_OK;
}

extern "C" int orbx_features_in_area(orbx_matcher *m, const orbx_keypoint *keypoints, int n, float min_x, float max_x,
                                     float min_y, float max_y, const float *x, const float *y, const float *r,
                                     const int32_t *min_level, const int32_t *max_level, int nq, int cap,
                                     int32_t *out_idx, int32_t *out_n, int32_t *cell_start, int32_t *cell_items)
{
    if (!m || n < 0 || n > 32768 || nq < 0 || cap < 0 || (n > 0 && !keypoints) ||
        (nq > 0 && (!x || !y || !r || !min_level || !max_level || !out_n || (cap > 0 && !out_idx))))
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(m->device));
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off += (std::max<size_t>(bytes, 4) + 255) / 256 * 256;
        return at;
    };
    // inputs first, results next: both parts are mirrored in page-locked memory and move in one copy each
    const size_t o_kp = carve((size_t)n * sizeof(orbx_keypoint));
    const size_t o_x = carve((size_t)nq * 4), o_y = carve((size_t)nq * 4);
    const size_t o_r = carve((size_t)nq * 4), o_lo = carve((size_t)nq * 4), o_hi = carve((size_t)nq * 4);
    const size_t in_bytes = off;
    const size_t o_n = carve((size_t)nq * 4), o_cs = carve((size_t)(64 * 48 + 1) * 4), o_ci = carve((size_t)n * 4);
    const size_t o_out = carve((size_t)nq * cap * 4);
    const size_t out_bytes = off - in_bytes;
    int rc = grow(&m->d_ws, &m->ws_cap, off, m->stream);
    if (rc == ORBX_OK)
        rc = grow_pinned(m, std::max(in_bytes, out_bytes));
    if (rc != ORBX_OK)
        return rc;
    uint8_t *w = m->d_ws;
    auto put = [&](size_t at, const void *src, size_t bytes) {
        if (bytes && src)
            memcpy(m->h_ws + at, src, bytes);
    };
    put(o_kp, keypoints, (size_t)n * sizeof(orbx_keypoint));
    put(o_x, x, (size_t)nq * 4);
    put(o_y, y, (size_t)nq * 4);
    put(o_r, r, (size_t)nq * 4);
    put(o_lo, min_level, (size_t)nq * 4);
    put(o_hi, max_level, (size_t)nq * 4);
    HIP_TRY(hipMemcpyAsync(w, m->h_ws, in_bytes, hipMemcpyHostToDevice, m->stream));
    rc = orbx_frame_grid_build_device(m, (const orbx_keypoint *)(w + o_kp), n, min_x, max_x, min_y, max_y,
                                      (int32_t *)(w + o_cs), (int32_t *)(w + o_ci));
    if (rc != ORBX_OK)
        return rc;
    rc = orbx_features_in_area_device(m, (const orbx_keypoint *)(w + o_kp), min_x, max_x, min_y, max_y,
                                      (const int32_t *)(w + o_cs), (const int32_t *)(w + o_ci), (const float *)(w + o_x),
                                      (const float *)(w + o_y), (const float *)(w + o_r), (const int32_t *)(w + o_lo),
                                      (const int32_t *)(w + o_hi), nq, cap, (int32_t *)(w + o_out), (int32_t *)(w + o_n));
    if (rc != ORBX_OK)
        return rc;
    // counts and grid always, the index lists only when there are any (they are the bulk)
    const size_t back = (nq > 0 && cap > 0) ? out_bytes : o_out - o_n;
    HIP_TRY(hipMemcpyAsync(m->h_ws, w + o_n, back, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    const uint8_t *h = m->h_ws;
    if (nq > 0) {
        if (cap > 0)
            memcpy(out_idx, h + (o_out - o_n), (size_t)nq * cap * 4);
        memcpy(out_n, h, (size_t)nq * 4);
    }
    if (cell_start)
        memcpy(cell_start, h + (o_cs - o_n), (size_t)(64 * 48 + 1) * 4);
    if (cell_items && n > 0)
        memcpy(cell_items, h + (o_ci - o_n), (size_t)n * 4);
    return ORBX_OK;
}

extern "C" int orbx_search_by_projection(orbx_matcher *m, const orbx_keypoint *keypoints, const uint8_t *descriptors,
                                         const float *uright, const uint8_t *taken, int n, float min_x, float max_x,
                                         float min_y, float max_y, const uint8_t *valid, const float *x, const float *y,
                                         const float *r, const int32_t *min_level, const int32_t *max_level,
                                         const float *xr, const uint8_t *src_desc, const uint8_t *blocks,
                                         const float *src_angle, int ns, int use_second, float nnratio, int th_dist,
                                         int check_orientation, int32_t *match, int32_t *feature_match,
                                         int32_t *nmatches)
{
    const int cap = 256;
    if (!m || n < 0 || n > 32768 || ns < 0 || !nmatches || (n > 0 && (!keypoints || !descriptors || !feature_match)) ||
        (ns > 0 && (!x || !y || !r || !min_level || !max_level || !src_desc || !match)) || (uright && !xr) ||
        (check_orientation && ns > 0 && !src_angle))
        return ORBX_E_ARG;
    *nmatches = 0;
    HIP_TRY(hipSetDevice(m->device));
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off += (std::max<size_t>(bytes, 4) + 255) / 256 * 256;
        return at;
    };
    // inputs first, results next, scratch last: the first two parts are mirrored in page-locked memory
    const size_t o_kp = carve((size_t)n * sizeof(orbx_keypoint)), o_desc = carve((size_t)n * 32);
    const size_t o_ur = carve((size_t)n * 4), o_tk = carve((size_t)n);
    const size_t o_va = carve((size_t)ns), o_x = carve((size_t)ns * 4), o_y = carve((size_t)ns * 4);
    const size_t o_r = carve((size_t)ns * 4), o_lo = carve((size_t)ns * 4), o_hi = carve((size_t)ns * 4);
    const size_t o_xr = carve((size_t)ns * 4), o_sd = carve((size_t)ns * 32), o_bl = carve((size_t)ns);
    const size_t o_an = carve((size_t)ns * 4);
    const size_t in_bytes = off;
    const size_t o_m = carve((size_t)ns * 4), o_fm = carve((size_t)n * 4), o_res = carve(8);
    const size_t out_bytes = off - in_bytes;
    const size_t o_cs = carve((size_t)(64 * 48 + 1) * 4), o_ci = carve((size_t)n * 4);
    const size_t o_list = carve((size_t)ns * cap * 4), o_cnt = carve((size_t)ns * 4), o_ent = carve((size_t)ns * cap * 4);
    int rc = grow(&m->d_ws, &m->ws_cap, off, m->stream);
    if (rc == ORBX_OK)
        rc = grow_pinned(m, std::max(in_bytes, out_bytes));
    if (rc != ORBX_OK)
        return rc;
    uint8_t *w = m->d_ws;
    auto put = [&](size_t at, const void *src, size_t bytes) {
        if (bytes && src)
            memcpy(m->h_ws + at, src, bytes);
    };
    put(o_kp, keypoints, (size_t)n * sizeof(orbx_keypoint));
    put(o_desc, descriptors, (size_t)n * 32);
    put(o_ur, uright, (size_t)n * 4);
    put(o_tk, taken, (size_t)n);
    put(o_va, valid, (size_t)ns);
    put(o_x, x, (size_t)ns * 4);
    put(o_y, y, (size_t)ns * 4);
    put(o_r, r, (size_t)ns * 4);
    put(o_lo, min_level, (size_t)ns * 4);
    put(o_hi, max_level, (size_t)ns * 4);
    put(o_xr, xr, (size_t)ns * 4);
    put(o_sd, src_desc, (size_t)ns * 32);
    put(o_bl, blocks, (size_t)ns);
    put(o_an, src_angle, (size_t)ns * 4);
    HIP_TRY(hipMemcpyAsync(w, m->h_ws, in_bytes, hipMemcpyHostToDevice, m->stream));
    const orbx_keypoint *d_kp = (const orbx_keypoint *)(w + o_kp);
    rc = orbx_frame_grid_build_device(m, d_kp, n, min_x, max_x, min_y, max_y, (int32_t *)(w + o_cs), (int32_t *)(w + o_ci));
    if (rc != ORBX_OK)
        return rc;
    rc = orbx_features_in_area_device(m, d_kp, min_x, max_x, min_y, max_y, (const int32_t *)(w + o_cs),
                                      (const int32_t *)(w + o_ci), (const float *)(w + o_x), (const float *)(w + o_y),
                                      (const float *)(w + o_r), (const int32_t *)(w + o_lo), (const int32_t *)(w + o_hi), ns,
                                      cap, (int32_t *)(w + o_list), (int32_t *)(w + o_cnt));
    if (rc != ORBX_OK)
        return rc;
    HIP_TRY(orbx_launch_search_by_projection(
        d_kp, w + o_desc, uright ? (const float *)(w + o_ur) : nullptr, taken ? w + o_tk : nullptr, n,
        valid ? w + o_va : nullptr, (const float *)(w + o_r), xr ? (const float *)(w + o_xr) : nullptr, w + o_sd,
        blocks ? w + o_bl : nullptr, src_angle ? (const float *)(w + o_an) : nullptr, ns, (const int32_t *)(w + o_list),
        (const int32_t *)(w + o_cnt), cap, (uint32_t *)(w + o_ent), use_second, nnratio, th_dist, check_orientation,
        (int32_t *)(w + o_m), (int32_t *)(w + o_fm), (int32_t *)(w + o_res), m->stream));
    // the results in one copy (the upload out of the mirror has long finished: same stream)
    HIP_TRY(hipMemcpyAsync(m->h_ws, w + o_m, out_bytes, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    const uint8_t *h = m->h_ws;
    int32_t res[2];
    memcpy(res, h + (o_res - o_m), 8);
    if (res[1])
        return ORBX_E_CAPACITY;
    if (ns > 0)
        memcpy(match, h, (size_t)ns * 4);
    if (n > 0)
        memcpy(feature_match, h + (o_fm - o_m), (size_t)n * 4);
    *nmatches = res[0];
    return ORBX_OK;
}

extern "C" int orbx_undistort_keypoints_device(orbx_matcher *m, const orbx_keypoint *d_keypoints, int n, const float *cam,
                                               const float *dist, int ndist, orbx_keypoint *d_out)
{
    if (!m || n < 0 || !cam || !dist || (ndist != 4 && ndist != 5) || (n > 0 && (!d_keypoints || !d_out)))
        return ORBX_E_ARG;
    if (n == 0)
        return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    if (dist[0] == 0.0f) {
        if (d_out != d_keypoints)
            HIP_TRY(hipMemcpyAsync(d_out, d_keypoints, (size_t)n * sizeof(orbx_keypoint), hipMemcpyDeviceToDevice,
                                   m->stream));
        return ORBX_OK;
    }
    HIP_TRY(orbx_launch_undistort(d_keypoints, d_out, nullptr, nullptr, n, cam, dist, ndist, m->stream));
    return ORBX_OK;
}

extern "C" int orbx_undistort_points(orbx_matcher *m, const float *xy, int n, const float *cam, const float *dist,
                                     int ndist, float *xy_out)
{
    if (!m || n < 0 || !cam || !dist || (ndist != 4 && ndist != 5) || (n > 0 && (!xy || !xy_out)))
        return ORBX_E_ARG;
    if (n == 0)
        return ORBX_OK;
    if (dist[0] == 0.0f) {
        if (xy_out != xy)
            memmove(xy_out, xy, (size_t)n * 2 * sizeof(float));
        return ORBX_OK;
    }
    HIP_TRY(hipSetDevice(m->device));
    int rc = grow(&m->d_ws, &m->ws_cap, (size_t)n * 2 * sizeof(float), m->stream);
    if (rc != ORBX_OK)
        return rc;
    float *d = reinterpret_cast<float *>(m->d_ws);
    HIP_TRY(hipMemcpyAsync(d, xy, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(orbx_launch_undistort(nullptr, nullptr, d, d, n, cam, dist, ndist, m->stream));
    HIP_TRY(hipMemcpyAsync(xy_out, d, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

extern "C" int orbx_search_by_bow(orbx_matcher *m, const uint8_t *desc1, const float *angle1, const uint8_t *valid1,
                                  const int32_t *feat_start, int nb, const int32_t *node_kf_start,
                                  const int32_t *node_ids1, const int32_t *node_start1, const int32_t *node_items1,
                                  const uint8_t *desc2, const float *angle2, const uint8_t *valid2, int n2,
                                  const int32_t *node_ids2, const int32_t *node_start2, const int32_t *node_items2,
                                  int m2, int strict_lt, float nnratio, int check_orientation, int32_t *match12,
                                  int32_t *match21, int32_t *nmatches)
{
    if (!m || nb < 0 || n2 < 0 || m2 < 0 || !feat_start || !node_kf_start || !match12 || !match21 || !nmatches)
        return ORBX_E_ARG;
    if (nb == 0)
        return ORBX_OK;
    const int F1 = feat_start[nb], E1 = node_kf_start[nb];
    if (F1 < 0 || E1 < 0 || (F1 > 0 && (!desc1 || !angle1)) || (E1 > 0 && (!node_ids1 || !node_start1 || !node_items1)) ||
        (n2 > 0 && (!desc2 || !angle2)) || (m2 > 0 && (!node_ids2 || !node_start2 || !node_items2)))
        return ORBX_E_ARG;
    const int I1 = E1 > 0 ? node_start1[E1] : 0, I2 = m2 > 0 ? node_start2[m2] : 0;
    // The kernels index with these tables: every offset must ascend from 0 and every item must name a
    // feature of its own key frame (side 1) or of the frame (side 2).
    if (feat_start[0] != 0 || node_kf_start[0] != 0 || (E1 > 0 && node_start1[0] != 0) || (m2 > 0 && node_start2[0] != 0) ||
        I1 < 0 || I2 < 0)
        return ORBX_E_ARG;
    for (int b = 0; b < nb; b++) {
        if (feat_start[b + 1] < feat_start[b] || node_kf_start[b + 1] < node_kf_start[b])
            return ORBX_E_ARG;
        const int nfeat = feat_start[b + 1] - feat_start[b];
        for (int e = node_kf_start[b]; e < node_kf_start[b + 1]; e++) {
            if (node_start1[e + 1] < node_start1[e] || node_start1[e + 1] > I1)
                return ORBX_E_ARG;
            for (int j = node_start1[e]; j < node_start1[e + 1]; j++)
                if (node_items1[j] < 0 || node_items1[j] >= nfeat)
                    return ORBX_E_ARG;
        }
    }
    for (int e = 0; e < m2; e++) {
        if (node_start2[e + 1] < node_start2[e] || node_start2[e + 1] > I2)
            return ORBX_E_ARG;
        for (int j = node_start2[e]; j < node_start2[e + 1]; j++)
            if (node_items2[j] < 0 || node_items2[j] >= n2)
                return ORBX_E_ARG;
    }
    // the key frame of every node entry
    std::vector<int> node_kf((size_t)std::max(E1, 1));
    for (int b = 0; b < nb; b++)
        for (int e = node_kf_start[b]; e < node_kf_start[b + 1]; e++)
            node_kf[e] = b;
    HIP_TRY(hipSetDevice(m->device));
    // carve one workspace: every array starts at a multiple of 256 bytes
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off += (std::max<size_t>(bytes, 4) + 255) / 256 * 256;
        return at;
    };
    const size_t o_d1 = carve((size_t)F1 * 32), o_a1 = carve((size_t)F1 * 4), o_v1 = carve((size_t)F1);
    const size_t o_fs = carve((size_t)(nb + 1) * 4), o_nk = carve((size_t)E1 * 4), o_i1 = carve((size_t)E1 * 4);
    const size_t o_s1 = carve((size_t)(E1 + 1) * 4), o_t1 = carve((size_t)I1 * 4);
    const size_t o_d2 = carve((size_t)n2 * 32), o_a2 = carve((size_t)n2 * 4), o_v2 = carve((size_t)n2);
    const size_t o_i2 = carve((size_t)m2 * 4), o_s2 = carve((size_t)(m2 + 1) * 4), o_t2 = carve((size_t)I2 * 4);
    const size_t in_bytes = off;  // everything above comes from the caller
    const size_t o_m12 = carve((size_t)F1 * 4), o_m21 = carve((size_t)nb * std::max(n2, 1) * 4);
    const size_t o_nm = carve((size_t)nb * 4), o_st = carve(4);
    const size_t out_bytes = off - in_bytes;  // and these go back
    const size_t o_bin = carve((size_t)F1), o_h = carve((size_t)nb * 32 * 4);
    int rc = grow(&m->d_ws, &m->ws_cap, off, m->stream);
    if (rc == ORBX_OK)
        rc = grow_pinned(m, std::max(in_bytes, out_bytes));
    if (rc != ORBX_OK)
        return rc;
    uint8_t *w = m->d_ws;
    auto put = [&](size_t at, const void *src, size_t bytes) {
        if (bytes && src)
            memcpy(m->h_ws + at, src, bytes);
    };
    put(o_d1, desc1, (size_t)F1 * 32);
    put(o_a1, angle1, (size_t)F1 * 4);
    put(o_v1, valid1, (size_t)F1);
    put(o_fs, feat_start, (size_t)(nb + 1) * 4);
    put(o_nk, node_kf.data(), (size_t)E1 * 4);
    put(o_i1, node_ids1, (size_t)E1 * 4);
    if (E1 > 0)
        put(o_s1, node_start1, (size_t)(E1 + 1) * 4);
    put(o_t1, node_items1, (size_t)I1 * 4);
    put(o_d2, desc2, (size_t)n2 * 32);
    put(o_a2, angle2, (size_t)n2 * 4);
    put(o_v2, valid2, (size_t)n2);
    put(o_i2, node_ids2, (size_t)m2 * 4);
    if (m2 > 0)
        put(o_s2, node_start2, (size_t)(m2 + 1) * 4);
    put(o_t2, node_items2, (size_t)I2 * 4);
    HIP_TRY(hipMemcpyAsync(w, m->h_ws, in_bytes, hipMemcpyHostToDevice, m->stream));
    HIP_TRY(orbx_launch_bow(w + o_d1, (const float *)(w + o_a1), valid1 ? w + o_v1 : nullptr, (const int *)(w + o_fs), nb,
                            F1, (const int *)(w + o_nk), (const int *)(w + o_i1), (const int *)(w + o_s1),
                            (const int *)(w + o_t1), E1, w + o_d2, (const float *)(w + o_a2),
                            valid2 ? w + o_v2 : nullptr, n2, (const int *)(w + o_i2), (const int *)(w + o_s2),
                            (const int *)(w + o_t2), m2, strict_lt, nnratio, check_orientation, (int *)(w + o_m12),
                            (int *)(w + o_m21), (int *)(w + o_nm), (int8_t *)(w + o_bin), (int *)(w + o_h),
                            (int *)(w + o_st), m->stream));
    HIP_TRY(hipMemcpyAsync(m->h_ws, w + o_m12, out_bytes, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    const uint8_t *h = m->h_ws;
    int status = 0;
    memcpy(&status, h + (o_st - o_m12), 4);
    if (F1 > 0)
        memcpy(match12, h, (size_t)F1 * 4);
    if (n2 > 0)
        memcpy(match21, h + (o_m21 - o_m12), (size_t)nb * n2 * 4);
    memcpy(nmatches, h + (o_nm - o_m12), (size_t)nb * 4);
    return status ? ORBX_E_CAPACITY : ORBX_OK;
}

extern "C" int orbx_feature_vectors_build_device(orbx_matcher *m, const int32_t *d_node, const int32_t *d_counts,
                                                 int nframes, int max_keypoints, int32_t *d_node_ids,
                                                 int32_t *d_node_start, int32_t *d_node_items, int32_t *d_num_nodes)
{
    if (!m || !d_node || !d_counts || nframes < 1 || max_keypoints < 1 || !d_node_ids || !d_node_start || !d_node_items ||
        !d_num_nodes)
        return ORBX_E_ARG;
    if (max_keypoints > 8192)
        return ORBX_E_CAPACITY;  // the sort keeps a frame's keys (8 bytes each) in 64 KB of LDS
    HIP_TRY(hipSetDevice(m->device));
    int rc = grow(&m->d_seg, &m->seg_cap, (size_t)nframes * max_keypoints, m->stream);  // nodes in sorted order
    if (rc != ORBX_OK)
        return rc;
    HIP_TRY(orbx_launch_feature_vectors(d_node, d_counts, nframes, max_keypoints, m->d_seg, d_node_ids, d_node_start,
                                        d_node_items, d_num_nodes, m->stream));
    return ORBX_OK;
}

extern "C" int orbx_search_by_bow_device(orbx_matcher *m, const orbx_keypoint *d_keypoints1, const uint8_t *d_desc1,
                                         const int32_t *d_counts1, const uint8_t *d_valid1, int nb, int max_keypoints1,
                                         const int32_t *d_node_ids1, const int32_t *d_node_start1,
                                         const int32_t *d_node_items1, const orbx_keypoint *d_keypoints2,
                                         const uint8_t *d_desc2, const uint8_t *d_valid2, int max_keypoints2,
                                         const int32_t *d_node_ids2, const int32_t *d_node_start2,
                                         const int32_t *d_node_items2, int strict_lt, float nnratio,
                                         int check_orientation, int32_t *d_match12, int32_t *d_match21,
                                         int32_t *d_nmatches)
{
    if (!m || !d_keypoints1 || !d_desc1 || !d_counts1 || nb < 1 || max_keypoints1 < 1 || !d_node_ids1 || !d_node_start1 ||
        !d_node_items1 || !d_keypoints2 || !d_desc2 || max_keypoints2 < 1 || !d_node_ids2 || !d_node_start2 ||
        !d_node_items2 || !d_match12 || !d_match21 || !d_nmatches)
        return ORBX_E_ARG;
    if (max_keypoints2 > 65535 || (int64_t)nb * max_keypoints1 > INT32_MAX / 2)
        return ORBX_E_CAPACITY;  // list positions travel in 16 bits of the wave's keys
    HIP_TRY(hipSetDevice(m->device));
    // workspace: rotation bin per feature of side 1, then histograms, per-key-frame status, one status word
    const size_t bins = ((size_t)nb * max_keypoints1 + 255) / 256 * 256;
    int rc = grow(&m->d_ws, &m->ws_cap, bins + sizeof(int) * (33 * (size_t)nb + 1), m->stream);
    if (rc != ORBX_OK)
        return rc;
    int *hist = reinterpret_cast<int *>(m->d_ws + bins);
    HIP_TRY(orbx_launch_bow_strided(d_keypoints1, d_desc1, d_counts1, d_valid1, nb, max_keypoints1, d_node_ids1,
                                    d_node_start1, d_node_items1, d_keypoints2, d_desc2, d_valid2, max_keypoints2,
                                    d_node_ids2, d_node_start2, d_node_items2, strict_lt, nnratio, check_orientation,
                                    d_match12, d_match21, d_nmatches, reinterpret_cast<int8_t *>(m->d_ws), hist,
                                    hist + 32 * (size_t)nb, hist + 33 * (size_t)nb, m->stream));
    return ORBX_OK;
}

extern "C" int orbx_distinctive_descriptors_device(orbx_matcher *m, const uint8_t *d_descriptors,
                                                   const int32_t *d_seg_start, int npoints, int32_t *d_best_idx,
                                                   uint8_t *d_best_desc)
{
    if (!m || npoints < 0 || (npoints > 0 && (!d_descriptors || !d_seg_start || !d_best_idx)))
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(orbx_launch_distinctive(d_descriptors, d_seg_start, npoints, d_best_idx, d_best_desc, m->stream));
    return ORBX_OK;
}

extern "C" int orbx_distinctive_descriptors(orbx_matcher *m, const uint8_t *descriptors, const int32_t *seg_start,
                                            int npoints, int32_t *best_idx, uint8_t *best_desc)
{
    if (!m || npoints < 0 || (npoints > 0 && (!descriptors || !seg_start || !best_idx)))
        return ORBX_E_ARG;
    if (npoints == 0)
        return ORBX_OK;
    const int64_t rows = seg_start[npoints];
    for (int p = 0; p < npoints; p++)
        if (seg_start[p] < 0 || seg_start[p + 1] < seg_start[p])
            return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(m->device));
    // workspace: descriptors in d_db, segment starts and results in d_seg, chosen descriptors in d_a
    int rc = grow(&m->d_db, &m->db_cap, (size_t)std::max<int64_t>(rows, 1) * 32, m->stream);
    if (rc == ORBX_OK) rc = grow(&m->d_seg, &m->seg_cap, (size_t)2 * npoints + 2, m->stream);
    if (rc == ORBX_OK) rc = grow(&m->d_a, &m->a_cap, (size_t)npoints * 32, m->stream);
    if (rc != ORBX_OK)
        return rc;
    int *d_seg = m->d_seg, *d_best = m->d_seg + npoints + 1;
    if (rows > 0)
        HIP_TRY(hipMemcpyAsync(m->d_db, descriptors, (size_t)rows * 32, hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemcpyAsync(d_seg, seg_start, sizeof(int) * (size_t)(npoints + 1), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(orbx_launch_distinctive(m->d_db, d_seg, npoints, d_best, best_desc ? m->d_a : nullptr, m->stream));
    HIP_TRY(hipMemcpyAsync(best_idx, d_best, sizeof(int) * (size_t)npoints, hipMemcpyDeviceToHost, m->stream));
    if (best_desc)
        HIP_TRY(hipMemcpyAsync(best_desc, m->d_a, (size_t)npoints * 32, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

// ------------------------------------------------------------------ vocabulary

struct orbx_vocabulary {
    int device, nnodes, depth;
    uint8_t *d_desc;
    int *d_child_start, *d_children, *d_word;
    double *d_weight;
    // feature and result arrays of the host form, grown on demand
    uint8_t *d_ws;
    size_t ws_cap;
};

extern "C" void orbx_vocabulary_destroy(orbx_vocabulary *v)
{
    if (!v)
        return;
    (void)hipSetDevice(v->device);
    void *ptrs[] = {v->d_desc, v->d_child_start, v->d_children, v->d_word, v->d_weight, v->d_ws};
    for (void *p : ptrs)
        if (p)
            (void)hipFree(p);
    delete v;
}

extern "C" int orbx_vocabulary_create(int device, const uint8_t *node_desc, const int32_t *child_start,
                                      const int32_t *children, const int32_t *word_id, const double *weight, int nnodes,
                                      int depth, orbx_vocabulary **out)
{
    if (!out || !node_desc || !child_start || !children || !word_id || !weight || nnodes < 2 || depth < 1)
        return ORBX_E_ARG;
    *out = nullptr;
    // the root must have children, offsets must ascend, and a child must come after its parent (this
    // also bounds every descent)
    if (child_start[0] != 0 || child_start[1] <= 0)
        return ORBX_E_ARG;
    for (int i = 0; i < nnodes; i++) {
        if (child_start[i + 1] < child_start[i])
            return ORBX_E_ARG;
        for (int c = child_start[i]; c < child_start[i + 1]; c++)
            if (children[c] <= i || children[c] >= nnodes)
                return ORBX_E_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev)
        return ORBX_E_NODEVICE;
    orbx_vocabulary *v = new (std::nothrow) orbx_vocabulary();
    if (!v)
        return ORBX_E_CAPACITY;
    memset(v, 0, sizeof(*v));
    v->device = device, v->nnodes = nnodes, v->depth = depth;
    const size_t nc = (size_t)child_start[nnodes];
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_desc, (size_t)nnodes * 32);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_child_start, ((size_t)nnodes + 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_children, std::max<size_t>(nc, 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_word, (size_t)nnodes * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&v->d_weight, (size_t)nnodes * 8);
    if (e == hipSuccess) e = hipMemcpy(v->d_desc, node_desc, (size_t)nnodes * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_child_start, child_start, ((size_t)nnodes + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_children, children, nc * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_word, word_id, (size_t)nnodes * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_weight, weight, (size_t)nnodes * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        g_last_hip = (int)e;
        orbx_vocabulary_destroy(v);
        return ORBX_E_HIP;
    }
    *out = v;
    return ORBX_OK;
}

extern "C" int orbx_vocabulary_transform_device(orbx_vocabulary *v, void *hip_stream, const uint8_t *d_descriptors,
                                                int n, int levelsup, int32_t *d_word, int32_t *d_node,
                                                double *d_weight)
{
    if (!v || n < 0 || levelsup < 0 || (n > 0 && (!d_descriptors || !d_word || !d_node || !d_weight)))
        return ORBX_E_ARG;
    HIP_TRY(hipSetDevice(v->device));
    HIP_TRY(orbx_launch_voc_transform(v->d_desc, v->d_child_start, v->d_children, v->d_word, v->d_weight, v->depth,
                                      d_descriptors, n, levelsup, d_word, d_node, d_weight, (hipStream_t)hip_stream));
    return ORBX_OK;
}

extern "C" int orbx_vocabulary_transform(orbx_vocabulary *v, const uint8_t *descriptors, int n, int levelsup,
                                         int32_t *word, int32_t *node, double *weight)
{
    if (!v || n < 0 || levelsup < 0 || (n > 0 && (!descriptors || !word || !node || !weight)))
        return ORBX_E_ARG;
    if (n == 0)
        return ORBX_OK;
    HIP_TRY(hipSetDevice(v->device));
    const size_t o_f = 0, o_w = ((size_t)n * 32 + 255) / 256 * 256, o_n = o_w + ((size_t)n * 4 + 255) / 256 * 256;
    const size_t o_wt = o_n + ((size_t)n * 4 + 255) / 256 * 256, total = o_wt + (size_t)n * 8;
    if (total > v->ws_cap) {
        if (v->d_ws)
            HIP_TRY(hipFree(v->d_ws));
        v->d_ws = nullptr, v->ws_cap = 0;
        HIP_TRY(hipMalloc((void **)&v->d_ws, total + total / 4));
        v->ws_cap = total + total / 4;
    }
    uint8_t *w = v->d_ws;
    HIP_TRY(hipMemcpy(w + o_f, descriptors, (size_t)n * 32, hipMemcpyHostToDevice));
    int rc = orbx_vocabulary_transform_device(v, nullptr, w + o_f, n, levelsup, (int32_t *)(w + o_w), (int32_t *)(w + o_n),
                                              (double *)(w + o_wt));
    if (rc != ORBX_OK)
        return rc;
    HIP_TRY(hipMemcpy(word, w + o_w, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(node, w + o_n, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(weight, w + o_wt, (size_t)n * 8, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// ------------------------------------------------------------------ synthetic inputs

static inline uint64_t splitmix64(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

extern "C" int orbx_synth_image(uint64_t seed, int width, int height, int variant, uint8_t *dst, int stride_bytes)
{
    if (!dst || width < 1 || height < 1 || stride_bytes < width || variant < 0 || variant > 2)
        return ORBX_E_ARG;
    uint64_t s = seed;
    for (int y = 0; y < height; y++)
        memset(dst + (size_t)y * stride_bytes, 128, width);
    if (variant != 1) {
        const int nrect = (int)((long long)width * height / 300);
        for (int r = 0; r < nrect; r++) {
            const uint64_t a = splitmix64(s), b = splitmix64(s);
            const int x0 = (int)(a % (uint64_t)width), y0 = (int)((a >> 32) % (uint64_t)height);
            const int rw = 6 + (int)(b % 59), rh = 6 + (int)((b >> 16) % 59);
            int gray = (int)((b >> 32) & 0xff);
            if (variant == 2)
                gray = 118 + gray % 21;
            const int x1 = std::min(width, x0 + rw), y1 = std::min(height, y0 + rh);
            for (int y = y0; y < y1; y++)
                memset(dst + (size_t)y * stride_bytes + x0, gray, x1 - x0);
        }
    }
    for (int y = 0; y < height; y++) {
        uint8_t *row = dst + (size_t)y * stride_bytes;
        for (int x = 0; x < width; x += 16) {
            uint64_t r = splitmix64(s);
            const int n = std::min(16, width - x);
            for (int i = 0; i < n; i++, r >>= 4) {
                const int v = (int)row[x + i] + (int)((r & 15) % 7) - 3;
                row[x + i] = (uint8_t)std::min(255, std::max(0, v));
            }
        }
    }
    return ORBX_OK;
}

extern "C" int orbx_synth_descriptors(uint64_t seed, int64_t n, uint8_t *dst)
{
    if (!dst || n < 0)
        return ORBX_E_ARG;
    uint64_t s = seed;
    uint64_t *p = reinterpret_cast<uint64_t *>(dst);
    if ((reinterpret_cast<uintptr_t>(dst) & 7) == 0) {
        for (int64_t i = 0; i < n * 4; i++)
            p[i] = splitmix64(s);
    } else {
        for (int64_t i = 0; i < n * 4; i++) {
            const uint64_t v = splitmix64(s);
            memcpy(dst + i * 8, &v, 8);
        }
    }
    return ORBX_OK;
}
