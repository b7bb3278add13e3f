/*
 * orb_oracle.c -- CPU restatement of ORB-SLAM2's ORB front end (extractor + Hamming matcher).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED.  The reference keeps no tests, golden vectors or fixtures for this path,
 * and its pixel arithmetic lives in OpenCV (core/imgproc/features2d; find_package(OpenCV 3.0)
 * else 2.4.3, CMakeLists.txt:31-37), which is neither vendored in the reference nor present in
 * the build image, so the reference cannot be built here.  The ORB-SLAM2 logic below follows
 * the cited lines of the reference; the OpenCV primitives (resize INTER_LINEAR 8U, FAST-9/16
 * with non-max suppression, GaussianBlur 7x7 sigma 2, fastAtan2, cvRound) are restated from
 * OpenCV's published generic C++ algorithms.  The hand-computed known-answer tests in
 * tests/test_oracle_kat.py are the only pins.
 *
 * All citations are relative to the reference root.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orb_pattern_data.h"

#define ORC_EDGE 19       /* EDGE_THRESHOLD, src/ORBextractor.cc:74 */
#define ORC_HALF_PATCH 15 /* HALF_PATCH_SIZE, src/ORBextractor.cc:73 */
#define ORC_PATCH 31      /* PATCH_SIZE, src/ORBextractor.cc:72 */
#define ORC_MAX_LEVELS 16

typedef struct {
    int nfeatures;
    float scale_factor;
    int nlevels;
    int ini_th_fast;
    int min_th_fast;
    int blur_variant; /* 0: kernel 18,34,48,56,48,34,18 (sum 256); 1: 18,34,49,55,49,34,18 (sum 257, saturating) */
} orc_params;

/* same memory layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id) */
typedef struct {
    float x, y, size, angle, response;
    int octave, class_id;
} orc_keypoint;

/* cvRound: round half to even, as the SSE cvt instructions OpenCV uses */
static inline int orc_round_f(float v) { return (int)lrintf(v); }
static inline int orc_round_d(double v) { return (int)lrint(v); }

/* ------------------------------------------------------------------ tables */

/* src/ORBextractor.cc:437-456 ; the member scaleFactor is a double (include/ORBextractor.h:98) */
void orc_scale_tables(float scale_factor, int nlevels, float *sf, float *inv, float *sig2, float *invsig2)
{
    const double s = (double)scale_factor;
    sf[0] = 1.0f;
    sig2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        sf[i] = (float)((double)sf[i - 1] * s);
        sig2[i] = sf[i] * sf[i];
    }
    for (int i = 0; i < nlevels; i++) {
        inv[i] = 1.0f / sf[i];
        invsig2[i] = 1.0f / sig2[i];
    }
}

/* src/ORBextractor.cc:460-475 */
void orc_features_per_level(int nfeatures, float scale_factor, int nlevels, int *quota)
{
    const double s = (double)scale_factor;
    const float factor = (float)(1.0 / s);
    float desired = (float)nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        quota[l] = orc_round_f(desired);
        sum += quota[l];
        desired *= factor;
    }
    int last = nfeatures - sum;
    quota[nlevels - 1] = last > 0 ? last : 0;
}

/* src/ORBextractor.cc:483-505 */
void orc_umax(int *umax /* 16 */)
{
    const float half_diag = ORC_HALF_PATCH * sqrtf(2.f) / 2;
    int vmax = (int)floorf(half_diag + 1);
    int vmin = (int)ceilf(half_diag);
    const double hp2 = ORC_HALF_PATCH * ORC_HALF_PATCH;
    int v, v0;
    for (v = 0; v <= vmax; ++v)
        umax[v] = orc_round_d(sqrt(hp2 - v * v));
    for (v = ORC_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1])
            ++v0;
        umax[v] = v0;
        ++v0;
    }
}

/* src/ORBextractor.cc:1195-1197 */
void orc_level_size(int cols, int rows, float inv_scale, int *w, int *h)
{
    *w = orc_round_f((float)cols * inv_scale);
    *h = orc_round_f((float)rows * inv_scale);
}

/* ------------------------------------------------------------------ resize */

/* cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for 8UC1, generic fixed-point path
 * (call site src/ORBextractor.cc:1209).  11-bit coefficients, the vertical pass keeps the
 * ((b*(H>>4))>>16) form of OpenCV's VResizeLinear<uchar,int,short>. */
static inline short orc_sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

int orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh, int dstride)
{
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0)
        return -1;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *alpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) {
            fx = 0;
            sx = 0;
        }
        if (sx + 1 >= sw) {
            if (dx < xmax)
                xmax = dx;
            if (sx >= sw - 1) {
                fx = 0;
                sx = sw - 1;
            }
        }
        xofs[dx] = sx;
        alpha[2 * dx] = orc_sat_short(orc_round_f((1.f - fx) * 2048));
        alpha[2 * dx + 1] = orc_sat_short(orc_round_f(fx * 2048));
    }
    int prev_sy0 = INT_MIN, prev_sy1 = INT_MIN;
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        short b0 = orc_sat_short(orc_round_f((1.f - fy) * 2048));
        short b1 = orc_sat_short(orc_round_f(fy * 2048));
        int sy0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        int *r[2] = {row0, row1};
        int srow[2] = {sy0, sy1};
        (void)prev_sy0;
        (void)prev_sy1;
        for (int k = 0; k < 2; k++) {
            const uint8_t *S = src + (size_t)srow[k] * sstride;
            int *D = r[k];
            int dx = 0;
            for (; dx < xmax; dx++) {
                int sx = xofs[dx];
                D[dx] = S[sx] * alpha[2 * dx] + S[sx + 1] * alpha[2 * dx + 1];
            }
            for (; dx < dw; dx++)
                D[dx] = S[xofs[dx]] * 2048;
        }
        uint8_t *out = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            out[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs);
    free(alpha);
    free(row0);
    free(row1);
    return 0;
}

/* ------------------------------------------------------------------ blur */

static inline int orc_reflect101(int i, int n)
{
    if (n == 1)
        return 0;
    while (i < 0 || i >= n) {
        if (i < 0)
            i = -i;
        else
            i = 2 * (n - 1) - i;
    }
    return i;
}

/* GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) on 8UC1 (call site
 * src/ORBextractor.cc:1157-1159): separable fixed-point filter, 8 fraction bits per pass. */
int orc_gauss7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int variant)
{
    static const int K0[7] = {18, 34, 48, 56, 48, 34, 18};
    static const int K1[7] = {18, 34, 49, 55, 49, 34, 18};
    const int *k = variant ? K1 : K0;
    if (w <= 0 || h <= 0)
        return -1;
    int *hbuf = (int *)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int i = 0; i < 7; i++)
                acc += k[i] * S[orc_reflect101(x + i - 3, w)];
            hbuf[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t *D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int j = 0; j < 7; j++)
                acc += k[j] * hbuf[(size_t)orc_reflect101(y + j - 3, h) * w + x];
            int v = (acc + 32768) >> 16;
            D[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    free(hbuf);
    return 0;
}

/* ------------------------------------------------------------------ FAST */

/* Bresenham ring of radius 3 in cv::FAST's order (makeOffsets, patternSize 16) */
static const int ORC_RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int ORC_RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* Largest threshold at which the pixel is still a FAST-9 corner, plus one: A = max over the 16
 * arcs of 9 contiguous ring pixels of min(v - ring) and of min(ring - v).  cv::FAST's
 * cornerScore<16>(ptr, pixel, t) equals max(t, A) - 1.  A <= 0 means "no corner at any t >= 0". */
int orc_fast_arc_score(const uint8_t *p, int stride)
{
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 16; k++)
        d[k] = v - p[ORC_RING_DY[k] * stride + ORC_RING_DX[k]];
    for (int k = 16; k < 25; k++)
        d[k] = d[k - 16];
    int best = -255;
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int i = 1; i < 9; i++) {
            if (d[k + i] < mn)
                mn = d[k + i];
            if (d[k + i] > mx)
                mx = d[k + i];
        }
        if (mn > best)
            best = mn;
        if (-mx > best)
            best = -mx;
    }
    return best;
}

/* cv::FAST(window, keypoints, threshold, nonmaxSuppression=true), FAST-9/16 (TYPE_9_16), on a
 * w x h window (call sites src/ORBextractor.cc:868-877).  Literal: detection = 9 contiguous of
 * the 25-long unrolled ring strictly darker than v-t or brighter than v+t; score buffer holds
 * cornerScore for detected pixels and 0 elsewhere; a corner is kept iff its score is strictly
 * greater than all 8 neighbours' buffer values.  Output row-major, window-relative.
 * Returns the number of keypoints (xs/ys/resp filled up to cap). */
int orc_fast_detect(const uint8_t *win, int w, int h, int stride, int threshold, int nonmax, int *xs, int *ys,
                    int *resp, int cap)
{
    if (threshold < 0)
        threshold = 0;
    if (threshold > 255)
        threshold = 255;
    if (w < 7 || h < 7)
        return 0;
    uint8_t *score = (uint8_t *)calloc((size_t)w * h, 1);
    uint8_t *is_corner = (uint8_t *)calloc((size_t)w * h, 1);
    for (int y = 3; y < h - 3; y++) {
        for (int x = 3; x < w - 3; x++) {
            const uint8_t *p = win + (size_t)y * stride + x;
            const int v = p[0];
            const int lo = v - threshold, hi = v + threshold;
            /* quick reject: ring pixels 0 and 8 cannot both be inside [lo,hi] for a 9-arc */
            int r0 = p[3 * stride], r8 = p[-3 * stride];
            if (r0 >= lo && r0 <= hi && r8 >= lo && r8 <= hi)
                continue;
            int ring[25];
            for (int k = 0; k < 16; k++)
                ring[k] = p[ORC_RING_DY[k] * stride + ORC_RING_DX[k]];
            for (int k = 16; k < 25; k++)
                ring[k] = ring[k - 16];
            int found = 0, cnt = 0;
            for (int k = 0; k < 25 && !found; k++) {
                if (ring[k] < lo) {
                    if (++cnt > 8)
                        found = 1;
                } else
                    cnt = 0;
            }
            cnt = 0;
            for (int k = 0; k < 25 && !found; k++) {
                if (ring[k] > hi) {
                    if (++cnt > 8)
                        found = 1;
                } else
                    cnt = 0;
            }
            if (found) {
                int a = orc_fast_arc_score(p, stride);
                int s = (a > threshold ? a : threshold) - 1;
                is_corner[(size_t)y * w + x] = 1;
                score[(size_t)y * w + x] = (uint8_t)s;
            }
        }
    }
    int n = 0;
    for (int y = 3; y < h - 3; y++) {
        for (int x = 3; x < w - 3; x++) {
            size_t o = (size_t)y * w + x;
            if (!is_corner[o])
                continue;
            int s = score[o];
            if (nonmax) {
                if (!(s > score[o - 1] && s > score[o + 1] && s > score[o - w - 1] && s > score[o - w] &&
                      s > score[o - w + 1] && s > score[o + w - 1] && s > score[o + w] && s > score[o + w + 1]))
                    continue;
            }
            if (n < cap) {
                xs[n] = x;
                ys[n] = y;
                resp[n] = s;
            }
            n++;
        }
    }
    free(score);
    free(is_corner);
    return n;
}

/* ------------------------------------------------------------------ candidates per level */

typedef struct {
    int min_bx, min_by, max_bx, max_by; /* detection region [16, cols-16) x [16, rows-16) */
    int ncols, nrows, wcell, hcell;
} orc_cellgrid;

/* src/ORBextractor.cc:826-840 */
int orc_cell_grid(int cols, int rows, orc_cellgrid *g)
{
    g->min_bx = ORC_EDGE - 3;
    g->min_by = g->min_bx;
    g->max_bx = cols - ORC_EDGE + 3;
    g->max_by = rows - ORC_EDGE + 3;
    const float W = 30;
    const float width = (float)(g->max_bx - g->min_bx);
    const float height = (float)(g->max_by - g->min_by);
    g->ncols = g->nrows = g->wcell = g->hcell = 0;
    if (width <= 0 || height <= 0)
        return 0;
    g->ncols = (int)(width / W);
    g->nrows = (int)(height / W);
    if (g->ncols <= 0 || g->nrows <= 0) {
        g->ncols = g->nrows = 0; /* the reference's cell loops do not execute */
        return 0;
    }
    g->wcell = (int)ceilf(width / g->ncols);
    g->hcell = (int)ceilf(height / g->nrows);
    return g->ncols * g->nrows;
}

/* src/ORBextractor.cc:843-893: per-cell FAST with the iniThFAST -> minThFAST fallback.
 * Output coordinates are relative to (min_bx, min_by), order = cells row-major, inside a cell
 * row-major.  Returns the candidate count (arrays filled up to cap). */
int orc_level_candidates(const uint8_t *img, int cols, int rows, int stride, int ini_th, int min_th, int *xs, int *ys,
                         int *resp, int cap)
{
    orc_cellgrid g;
    if (!orc_cell_grid(cols, rows, &g))
        return 0;
    int n = 0;
    int cell_cap = (g.wcell + 6) * (g.hcell + 6);
    int *cx = (int *)malloc(sizeof(int) * cell_cap * 3), *cy = cx + cell_cap, *cr = cy + cell_cap;
    for (int i = 0; i < g.nrows; i++) {
        const float iniY = (float)(g.min_by + i * g.hcell);
        float maxY = iniY + g.hcell + 6;
        if (iniY >= g.max_by - 3)
            continue;
        if (maxY > g.max_by)
            maxY = (float)g.max_by;
        for (int j = 0; j < g.ncols; j++) {
            const float iniX = (float)(g.min_bx + j * g.wcell);
            float maxX = iniX + g.wcell + 6;
            if (iniX >= g.max_bx - 6)
                continue;
            if (maxX > g.max_bx)
                maxX = (float)g.max_bx;
            const int x0 = (int)iniX, y0 = (int)iniY, ww = (int)maxX - x0, wh = (int)maxY - y0;
            const uint8_t *win = img + (size_t)y0 * stride + x0;
            int m = orc_fast_detect(win, ww, wh, stride, ini_th, 1, cx, cy, cr, cell_cap);
            if (m == 0)
                m = orc_fast_detect(win, ww, wh, stride, min_th, 1, cx, cy, cr, cell_cap);
            for (int k = 0; k < m; k++) {
                if (n < cap) {
                    xs[n] = cx[k] + j * g.wcell;
                    ys[n] = cy[k] + i * g.hcell;
                    resp[n] = cr[k];
                }
                n++;
            }
        }
    }
    free(cx);
    return n;
}

/* ------------------------------------------------------------------ quadtree distribution */

typedef struct orc_node {
    int ulx, uly, urx, bry; /* UL.x, UL.y, UR.x, BR.y are all DivideNode reads */
    int *keys;              /* indices into the candidate arrays, parent order preserved */
    int nkeys;
    int no_more;
    long seq; /* creation order; stands in for the heap address the reference sorts on */
    struct orc_node *prev, *next;
} orc_node;

typedef struct {
    orc_node *head, *tail;
    int size;
    long next_seq;
} orc_list;

static orc_node *orc_node_new(orc_list *L, int cap)
{
    orc_node *n = (orc_node *)calloc(1, sizeof(orc_node));
    n->keys = (int *)malloc(sizeof(int) * (cap > 0 ? cap : 1));
    n->seq = L->next_seq++;
    return n;
}
static void orc_push_front(orc_list *L, orc_node *n)
{
    n->prev = NULL;
    n->next = L->head;
    if (L->head)
        L->head->prev = n;
    L->head = n;
    if (!L->tail)
        L->tail = n;
    L->size++;
}
static void orc_push_back(orc_list *L, orc_node *n)
{
    n->next = NULL;
    n->prev = L->tail;
    if (L->tail)
        L->tail->next = n;
    L->tail = n;
    if (!L->head)
        L->head = n;
    L->size++;
}
static orc_node *orc_erase(orc_list *L, orc_node *n)
{
    orc_node *nx = n->next;
    if (n->prev)
        n->prev->next = n->next;
    else
        L->head = n->next;
    if (n->next)
        n->next->prev = n->prev;
    else
        L->tail = n->prev;
    L->size--;
    free(n->keys);
    free(n);
    return nx;
}

/* ExtractorNode::DivideNode, src/ORBextractor.cc:518-576 */
static void orc_divide(orc_list *L, const orc_node *p, const int *xs, const int *ys, orc_node *c[4])
{
    const int halfX = (int)ceilf((float)(p->urx - p->ulx) / 2);
    const int halfY = (int)ceilf((float)(p->bry - p->uly) / 2);
    for (int q = 0; q < 4; q++)
        c[q] = orc_node_new(L, p->nkeys);
    const int midx = p->ulx + halfX, midy = p->uly + halfY;
    c[0]->ulx = p->ulx, c[0]->uly = p->uly, c[0]->urx = midx, c[0]->bry = midy;
    c[1]->ulx = midx, c[1]->uly = p->uly, c[1]->urx = p->urx, c[1]->bry = midy;
    c[2]->ulx = p->ulx, c[2]->uly = midy, c[2]->urx = midx, c[2]->bry = p->bry;
    c[3]->ulx = midx, c[3]->uly = midy, c[3]->urx = p->urx, c[3]->bry = p->bry;
    for (int i = 0; i < p->nkeys; i++) {
        const int k = p->keys[i];
        const float kx = (float)xs[k], ky = (float)ys[k];
        orc_node *d;
        if (kx < (float)midx)
            d = (ky < (float)midy) ? c[0] : c[2];
        else
            d = (ky < (float)midy) ? c[1] : c[3];
        d->keys[d->nkeys++] = k;
    }
    for (int q = 0; q < 4; q++)
        if (c[q]->nkeys == 1)
            c[q]->no_more = 1;
}

typedef struct {
    int size;
    orc_node *node;
} orc_size_node;

/* ascending (size, address); the address is replaced by the creation sequence number */
static int orc_cmp_size_node(const void *a, const void *b)
{
    const orc_size_node *A = (const orc_size_node *)a, *B = (const orc_size_node *)b;
    if (A->size != B->size)
        return A->size < B->size ? -1 : 1;
    return A->node->seq < B->node->seq ? -1 : (A->node->seq > B->node->seq ? 1 : 0);
}

/* children pushed to the front in the order n1..n4; returns how many have more than one key */
static int orc_adopt(orc_list *L, orc_node *c[4], orc_size_node *vec, int *nvec)
{
    int expandable = 0;
    for (int q = 0; q < 4; q++) {
        if (c[q]->nkeys > 0) {
            orc_push_front(L, c[q]);
            if (c[q]->nkeys > 1) {
                expandable++;
                vec[*nvec].size = c[q]->nkeys;
                vec[*nvec].node = c[q];
                (*nvec)++;
            }
        } else {
            free(c[q]->keys);
            free(c[q]);
        }
    }
    return expandable;
}

/* ORBextractor::DistributeOctTree, src/ORBextractor.cc:578-813.
 * xs/ys/resp: n candidates (coordinates relative to (minX,minY)).  Writes the index of the kept
 * candidate of every final node, in final list order, to out (capacity cap); returns the count,
 * or -2 when the region is so tall that the reference would divide by zero (nIni == 0).
 * The only departure: equal-size ties in the last-round sort are broken by creation order
 * (later-created node first), where the reference compares heap addresses (:734). */
int orc_distribute_octree(const int *xs, const int *ys, const int *resp, int n, int minX, int maxX, int minY, int maxY,
                          int N, int *out, int cap)
{
    if (n <= 0)
        return 0;
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1)
        return -2;
    const float hX = (float)(maxX - minX) / nIni;
    orc_list L = {0};
    orc_node **roots = (orc_node **)malloc(sizeof(orc_node *) * nIni);
    for (int i = 0; i < nIni; i++) {
        orc_node *r = orc_node_new(&L, n);
        r->ulx = (int)(hX * (float)i);
        r->urx = (int)(hX * (float)(i + 1));
        r->uly = 0;
        r->bry = maxY - minY;
        orc_push_back(&L, r);
        roots[i] = r;
    }
    for (int i = 0; i < n; i++) {
        int r = (int)((float)xs[i] / hX);
        if (r >= nIni)
            r = nIni - 1; /* unreachable for candidates produced by orc_level_candidates */
        roots[r]->keys[roots[r]->nkeys++] = i;
    }
    free(roots);
    for (orc_node *it = L.head; it;) {
        if (it->nkeys == 1) {
            it->no_more = 1;
            it = it->next;
        } else if (it->nkeys == 0)
            it = orc_erase(&L, it);
        else
            it = it->next;
    }

    int finish = 0;
    size_t vec_cap = (size_t)4 * (n + nIni) + 16;
    orc_size_node *vec = (orc_size_node *)malloc(sizeof(orc_size_node) * vec_cap);
    orc_size_node *prev = (orc_size_node *)malloc(sizeof(orc_size_node) * vec_cap);
    int nvec = 0;
    while (!finish) {
        int prevSize = L.size;
        int nToExpand = 0;
        nvec = 0;
        for (orc_node *it = L.head; it;) {
            if (it->no_more) {
                it = it->next;
                continue;
            }
            orc_node *c[4];
            orc_divide(&L, it, xs, ys, c);
            nToExpand += orc_adopt(&L, c, vec, &nvec);
            it = orc_erase(&L, it);
        }
        if (L.size >= N || L.size == prevSize)
            finish = 1;
        else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.size;
                int nprev = nvec;
                memcpy(prev, vec, sizeof(orc_size_node) * nprev);
                nvec = 0;
                qsort(prev, nprev, sizeof(orc_size_node), orc_cmp_size_node);
                for (int j = nprev - 1; j >= 0; j--) {
                    orc_node *c[4];
                    orc_divide(&L, prev[j].node, xs, ys, c);
                    orc_adopt(&L, c, vec, &nvec);
                    orc_erase(&L, prev[j].node);
                    if (L.size >= N)
                        break;
                }
                if (L.size >= N || L.size == prevSize)
                    finish = 1;
            }
        }
    }
    int m = 0;
    for (orc_node *it = L.head; it; it = it->next) {
        int best = it->keys[0];
        int maxr = resp[best];
        for (int k = 1; k < it->nkeys; k++)
            if (resp[it->keys[k]] > maxr) {
                best = it->keys[k];
                maxr = resp[best];
            }
        if (m < cap)
            out[m] = best;
        m++;
    }
    for (orc_node *it = L.head; it;)
        it = orc_erase(&L, it);
    free(vec);
    free(prev);
    return m;
}

/* ------------------------------------------------------------------ orientation */

/* cv::fastAtan2 (scalar form), degrees in [0,360) */
float orc_fast_atan2(float y, float x)
{
    static const float scale = (float)(180.0 / 3.141592653589793238462643383279502884);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0)
        a = 180.f - a;
    if (y < 0)
        a = 360.f - a;
    return a;
}

/* IC_Angle, src/ORBextractor.cc:93-124; (x,y) integer level coordinates of the keypoint */
float orc_ic_angle(const uint8_t *img, int stride, int x, int y, const int *umax, int *m10_out, int *m01_out)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)y * stride + x;
    for (int u = -ORC_HALF_PATCH; u <= ORC_HALF_PATCH; ++u)
        m_10 += u * center[u];
    for (int v = 1; v <= ORC_HALF_PATCH; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    if (m10_out)
        *m10_out = m_10;
    if (m01_out)
        *m01_out = m_01;
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------ descriptor */

/* computeOrbDescriptor, src/ORBextractor.cc:127-169; img is the blurred level */
void orc_brief_descriptor(const uint8_t *img, int stride, int x, int y, float angle_deg, uint8_t *desc /* 32 */)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float angle = angle_deg * factorPI;
    const float a = cosf(angle), b = sinf(angle);
    const uint8_t *center = img + (size_t)y * stride + x;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            const signed char *t = ORC_BRIEF_TESTS[i * 8 + k];
            int r0 = orc_round_f(t[0] * b + t[1] * a), c0 = orc_round_f(t[0] * a - t[1] * b);
            int r1 = orc_round_f(t[2] * b + t[3] * a), c1 = orc_round_f(t[2] * a - t[3] * b);
            int t0 = center[r0 * stride + c0], t1 = center[r1 * stride + c1];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------ whole extractor */

typedef struct {
    int nlevels;
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    uint8_t *img[ORC_MAX_LEVELS];  /* tightly packed levels (stride = w) */
    uint8_t *blur[ORC_MAX_LEVELS]; /* blurred levels (only those with keypoints), else NULL */
    int ncand[ORC_MAX_LEVELS];     /* candidates before distribution */
    int nkeep[ORC_MAX_LEVELS];     /* keypoints after distribution */
} orc_trace;

void orc_trace_free(orc_trace *t)
{
    for (int l = 0; l < ORC_MAX_LEVELS; l++) {
        free(t->img[l]);
        free(t->blur[l]);
        t->img[l] = t->blur[l] = NULL;
    }
}

/* ComputePyramid, src/ORBextractor.cc:1191-1221 (levels only; the 19-px reflected frame around
 * each level is never read by extraction) */
int orc_build_pyramid(const orc_params *p, const uint8_t *image, int cols, int rows, int stride, orc_trace *t)
{
    float sf[ORC_MAX_LEVELS], inv[ORC_MAX_LEVELS], s2[ORC_MAX_LEVELS], is2[ORC_MAX_LEVELS];
    if (p->nlevels < 1 || p->nlevels > ORC_MAX_LEVELS)
        return -1;
    orc_scale_tables(p->scale_factor, p->nlevels, sf, inv, s2, is2);
    memset(t, 0, sizeof(*t));
    t->nlevels = p->nlevels;
    for (int l = 0; l < p->nlevels; l++) {
        orc_level_size(cols, rows, inv[l], &t->w[l], &t->h[l]);
        if (t->w[l] <= 0 || t->h[l] <= 0)
            return -1;
        t->img[l] = (uint8_t *)malloc((size_t)t->w[l] * t->h[l]);
        if (l == 0) {
            for (int y = 0; y < rows; y++)
                memcpy(t->img[0] + (size_t)y * cols, image + (size_t)y * stride, cols);
        } else {
            orc_resize_linear_u8(t->img[l - 1], t->w[l - 1], t->h[l - 1], t->w[l - 1], t->img[l], t->w[l], t->h[l],
                                 t->w[l]);
        }
    }
    return 0;
}

/* ORBextractor::operator(), src/ORBextractor.cc:1114-1182, with ComputeKeyPointsOctTree
 * (:815-921) inlined.  kps/desc hold up to cap keypoints; *n_out receives the true count.
 * trace may be NULL; when given the caller frees it with orc_trace_free.
 * Returns 0, or a negative code (-1 bad arguments, -2 region too tall for the quadtree). */
int orc_extract(const orc_params *p, const uint8_t *image, int cols, int rows, int stride, orc_keypoint *kps,
                uint8_t *desc, int cap, int *n_out, orc_trace *trace)
{
    orc_trace local;
    orc_trace *t = trace ? trace : &local;
    float sf[ORC_MAX_LEVELS], inv[ORC_MAX_LEVELS], s2[ORC_MAX_LEVELS], is2[ORC_MAX_LEVELS];
    int quota[ORC_MAX_LEVELS], umax[ORC_HALF_PATCH + 2];
    *n_out = 0;
    if (!image || cols <= 0 || rows <= 0)
        return 0; /* empty image: outputs untouched (:1117-1118) */
    if (orc_build_pyramid(p, image, cols, rows, stride, t) != 0) {
        orc_trace_free(t);
        return -1;
    }
    orc_scale_tables(p->scale_factor, p->nlevels, sf, inv, s2, is2);
    orc_features_per_level(p->nfeatures, p->scale_factor, p->nlevels, quota);
    orc_umax(umax);

    int total = 0, rc = 0;
    for (int l = 0; l < p->nlevels && rc == 0; l++) {
        const int w = t->w[l], h = t->h[l];
        const uint8_t *img = t->img[l];
        int ccap = (w * h) / 4 + 16;
        int *cx = (int *)malloc(sizeof(int) * ccap * 4), *cy = cx + ccap, *cr = cy + ccap, *keep = cr + ccap;
        int nc = orc_level_candidates(img, w, h, w, p->ini_th_fast, p->min_th_fast, cx, cy, cr, ccap);
        t->ncand[l] = nc;
        const int minB = ORC_EDGE - 3;
        int nk = orc_distribute_octree(cx, cy, cr, nc, minB, w - ORC_EDGE + 3, minB, h - ORC_EDGE + 3, quota[l], keep,
                                       ccap);
        if (nk < 0) {
            rc = nk;
            free(cx);
            break;
        }
        t->nkeep[l] = nk;
        if (nk > 0) {
            t->blur[l] = (uint8_t *)malloc((size_t)w * h);
            orc_gauss7_u8(img, w, h, w, t->blur[l], w, p->blur_variant);
        }
        const int scaledPatch = (int)(ORC_PATCH * sf[l]);
        for (int i = 0; i < nk; i++) {
            const int k = keep[i];
            /* level coordinates are integers: candidate + border (:908-909) */
            const int lx = cx[k] + minB, ly = cy[k] + minB;
            float ang = orc_ic_angle(img, w, lx, ly, umax, NULL, NULL);
            if (total < cap) {
                orc_keypoint *o = &kps[total];
                o->x = (float)lx;
                o->y = (float)ly;
                if (l != 0) {
                    o->x *= sf[l];
                    o->y *= sf[l];
                }
                o->size = (float)scaledPatch;
                o->angle = ang;
                o->response = (float)cr[k];
                o->octave = l;
                o->class_id = -1;
                orc_brief_descriptor(t->blur[l], w, lx, ly, ang, desc + (size_t)total * 32);
            }
            total++;
        }
        free(cx);
    }
    *n_out = total;
    if (!trace)
        orc_trace_free(&local);
    return rc;
}

/* ------------------------------------------------------------------ matcher */

/* ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:1744-1760 (the bit-twiddling popcount) */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t wa, wb;
        memcpy(&wa, a + 4 * i, 4);
        memcpy(&wb, b + 4 * i, 4);
        uint32_t v = wa ^ wb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

static inline int orc_popcnt256(const uint8_t *a, const uint8_t *b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) +
           __builtin_popcountll(x[3] ^ y[3]);
}

/* Brute-force top-2 in the shape of the SearchByBoW inner loop (src/ORBmatcher.cc:217-242):
 * best/second start at 256, strict '<' updates in database order, best index -1 if none.
 * idx_base is added to the reported indices (database shards). */
void orc_match_top2(const uint8_t *q, int nq, const uint8_t *db, int64_t ndb, int64_t idx_base, int32_t *best_idx,
                    uint16_t *d1, uint16_t *d2)
{
    for (int i = 0; i < nq; i++) {
        int b1 = 256, b2 = 256;
        int64_t bi = -1;
        const uint8_t *qi = q + (size_t)i * 32;
        for (int64_t j = 0; j < ndb; j++) {
            int d = orc_popcnt256(qi, db + (size_t)j * 32);
            if (d < b1) {
                b2 = b1;
                b1 = d;
                bi = j;
            } else if (d < b2)
                b2 = d;
        }
        best_idx[i] = bi < 0 ? -1 : (int32_t)(bi + idx_base);
        d1[i] = (uint16_t)b1;
        d2[i] = (uint16_t)b2;
    }
}

/* acceptance rule of src/ORBmatcher.cc:245-249 */
int orc_ratio_accept(int best, int second, int th, float ratio)
{
    return best <= th && (float)best < ratio * (float)second;
}

/* ---- 64x48 lookup grid of Frame (src/Frame.cc:239-253, 349-420) ---- */
#define ORC_GRID_COLS 64
#define ORC_GRID_ROWS 48

typedef struct {
    float min_x, min_y, inv_w, inv_h; /* mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv */
    int *cell_start;                  /* [64*48+1], cell = ix*48+iy */
    int *cell_items;                  /* keypoint indices, insertion order inside a cell */
} orc_grid;

void orc_grid_build(orc_grid *g, const orc_keypoint *kps, int n, float min_x, float max_x, float min_y, float max_y)
{
    g->min_x = min_x;
    g->min_y = min_y;
    g->inv_w = (float)ORC_GRID_COLS / (max_x - min_x);
    g->inv_h = (float)ORC_GRID_ROWS / (max_y - min_y);
    const int ncell = ORC_GRID_COLS * ORC_GRID_ROWS;
    g->cell_start = (int *)calloc(ncell + 1, sizeof(int));
    g->cell_items = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    int *cell_of = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        int px = (int)roundf((kps[i].x - min_x) * g->inv_w);
        int py = (int)roundf((kps[i].y - min_y) * g->inv_h);
        if (px < 0 || px >= ORC_GRID_COLS || py < 0 || py >= ORC_GRID_ROWS)
            cell_of[i] = -1;
        else {
            cell_of[i] = px * ORC_GRID_ROWS + py;
            g->cell_start[cell_of[i] + 1]++;
        }
    }
    for (int c = 0; c < ncell; c++)
        g->cell_start[c + 1] += g->cell_start[c];
    int *fill = (int *)malloc(sizeof(int) * ncell);
    memcpy(fill, g->cell_start, sizeof(int) * ncell);
    for (int i = 0; i < n; i++)
        if (cell_of[i] >= 0)
            g->cell_items[fill[cell_of[i]]++] = i;
    free(fill);
    free(cell_of);
}
void orc_grid_free(orc_grid *g)
{
    free(g->cell_start);
    free(g->cell_items);
}

/* Frame::GetFeaturesInArea, src/Frame.cc:349-407 */
int orc_features_in_area(const orc_grid *g, const orc_keypoint *kps, float x, float y, float r, int minLevel,
                         int maxLevel, int *out)
{
    int n = 0;
    int c;
    c = (int)floorf((x - g->min_x - r) * g->inv_w);
    const int nMinCellX = c > 0 ? c : 0;
    if (nMinCellX >= ORC_GRID_COLS)
        return 0;
    c = (int)ceilf((x - g->min_x + r) * g->inv_w);
    const int nMaxCellX = c < ORC_GRID_COLS - 1 ? c : ORC_GRID_COLS - 1;
    if (nMaxCellX < 0)
        return 0;
    c = (int)floorf((y - g->min_y - r) * g->inv_h);
    const int nMinCellY = c > 0 ? c : 0;
    if (nMinCellY >= ORC_GRID_ROWS)
        return 0;
    c = (int)ceilf((y - g->min_y + r) * g->inv_h);
    const int nMaxCellY = c < ORC_GRID_ROWS - 1 ? c : ORC_GRID_ROWS - 1;
    if (nMaxCellY < 0)
        return 0;
    const int check = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int cell = ix * ORC_GRID_ROWS + iy;
            for (int j = g->cell_start[cell]; j < g->cell_start[cell + 1]; j++) {
                const int id = g->cell_items[j];
                const orc_keypoint *kp = &kps[id];
                if (check) {
                    if (kp->octave < minLevel)
                        continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel)
                        continue;
                }
                const float dx = kp->x - x, dy = kp->y - y;
                if (fabsf(dx) < r && fabsf(dy) < r)
                    out[n++] = id;
            }
        }
    return n;
}

/* Flat form of the two functions above for callers without the struct: builds the grid of n keypoints,
 * copies its CSR arrays out (cell_start: 64*48+1 entries, cell_items: up to n) and answers nq queries;
 * out_idx holds cap entries per query, out_n the (untruncated) counts. */
void orc_grid_queries(const orc_keypoint *kps, int n, float min_x, float max_x, float min_y, float max_y,
                      const float *x, const float *y, const float *r, const int *min_level, const int *max_level,
                      int nq, int cap, int *out_idx, int *out_n, int *cell_start, int *cell_items)
{
    orc_grid g;
    orc_grid_build(&g, kps, n, min_x, max_x, min_y, max_y);
    const int ncell = ORC_GRID_COLS * ORC_GRID_ROWS;
    memcpy(cell_start, g.cell_start, sizeof(int) * (size_t)(ncell + 1));
    memcpy(cell_items, g.cell_items, sizeof(int) * (size_t)g.cell_start[ncell]);
    int *tmp = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int q = 0; q < nq; q++) {
        const int c = orc_features_in_area(&g, kps, x[q], y[q], r[q], min_level[q], max_level[q], tmp);
        out_n[q] = c;
        memcpy(out_idx + (size_t)q * cap, tmp, sizeof(int) * (size_t)(c < cap ? c : cap));
    }
    free(tmp);
    orc_grid_free(&g);
}

/* Frame::UndistortKeyPoints (src/Frame.cc:467-499) and the corner pass of Frame::ComputeImageBounds
 * (:503-517): cv::undistortPoints(points, points, K, distCoef, Mat(), K) on float points.  OpenCV is not
 * part of the reference tree; this restates its published point routine for the call the reference
 * makes (no rectification, new camera matrix = K, default iteration count): the pixel goes to the
 * normalised plane, five fixed-point iterations of the radial (k1, k2, k3) and tangential (p1, p2)
 * model remove the distortion, K maps the result back; all in double, inputs and results float.
 * cam = fx, fy, cx, cy (the float entries of mK); dist = k1, k2, p1, p2[, k3] (mDistCoef, 4 or 5 floats).
 * With dist[0] == 0 the reference copies the points (:469-473). */
void orc_undistort_points(const float *xy, int n, const float *cam, const float *dist, int ndist, float *out)
{
    if (dist[0] == 0.0f) {
        memcpy(out, xy, sizeof(float) * 2 * (size_t)n);
        return;
    }
    const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = ndist > 4 ? dist[4] : 0.;
    for (int i = 0; i < n; i++) {
        double x = xy[2 * i], y = xy[2 * i + 1];
        const double x0 = x = (x - cx) * ifx;
        const double y0 = y = (y - cy) * ify;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            const double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        out[2 * i] = (float)(fx * x + cx);
        out[2 * i + 1] = (float)(fy * y + cy);
    }
}

/* ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1698-1739 */
void orc_three_maxima(const int *hist_sizes, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = hist_sizes[i];
        if (s > max1) {
            max3 = max2;
            max2 = max1;
            max1 = s;
            *ind3 = *ind2;
            *ind2 = *ind1;
            *ind1 = i;
        } else if (s > max2) {
            max3 = max2;
            max2 = s;
            *ind3 = *ind2;
            *ind2 = i;
        } else if (s > max3) {
            max3 = s;
            *ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        *ind2 = -1;
        *ind3 = -1;
    } else if (max3 < 0.1f * (float)max1)
        *ind3 = -1;
}

/* ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:433-549.  TH_LOW = 50, HISTO_LENGTH = 30
 * (:37-39).  prev_matched (n1 x,y pairs) is read and updated; matches12 receives n1 entries.
 * pair_count (may be NULL) receives the number of descriptor distances evaluated. */
int orc_search_for_initialization(const orc_keypoint *k1, const uint8_t *d1, int n1, const orc_keypoint *k2,
                                  const uint8_t *d2, int n2, float min_x, float max_x, float min_y, float max_y,
                                  float *prev_matched, int window, float nnratio, int check_ori, int *matches12,
                                  int64_t *pair_count)
{
    enum { TH_LOW = 50, HISTO = 30 };
    int nmatches = 0;
    int64_t pairs = 0;
    orc_grid g;
    orc_grid_build(&g, k2, n2, min_x, max_x, min_y, max_y);
    int *matched_dist = (int *)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int *matches21 = (int *)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int *cand = (int *)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int *hist_items = (int *)malloc(sizeof(int) * (n1 > 0 ? n1 : 1));
    int *hist_bin = (int *)malloc(sizeof(int) * (n1 > 0 ? n1 : 1));
    int nhist = 0;
    for (int i = 0; i < n1; i++)
        matches12[i] = -1;
    for (int i = 0; i < n2; i++) {
        matched_dist[i] = INT_MAX;
        matches21[i] = -1;
    }
    const float factor = HISTO / 360.0f;
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = k1[i1].octave;
        if (level1 > 0)
            continue;
        int nc = orc_features_in_area(&g, k2, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window, level1,
                                      level1, cand);
        if (nc == 0)
            continue;
        const uint8_t *da = d1 + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            const int dist = orc_descriptor_distance(da, d2 + (size_t)i2 * 32);
            pairs++;
            if (matched_dist[i2] <= dist)
                continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2)
                bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (matches21[bestIdx2] >= 0) {
                    matches12[matches21[bestIdx2]] = -1;
                    nmatches--;
                }
                matches12[i1] = bestIdx2;
                matches21[bestIdx2] = i1;
                matched_dist[bestIdx2] = bestDist;
                nmatches++;
                if (check_ori) {
                    float rot = k1[i1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0)
                        rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO)
                        bin = 0;
                    hist_items[nhist] = i1;
                    hist_bin[nhist] = bin;
                    nhist++;
                }
            }
        }
    }
    if (check_ori) {
        int sizes[HISTO] = {0};
        for (int i = 0; i < nhist; i++)
            sizes[hist_bin[i]]++;
        int i1, i2, i3;
        orc_three_maxima(sizes, HISTO, &i1, &i2, &i3);
        for (int i = 0; i < nhist; i++) {
            const int b = hist_bin[i];
            if (b == i1 || b == i2 || b == i3)
                continue;
            if (matches12[hist_items[i]] >= 0) {
                matches12[hist_items[i]] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < n1; i++)
        if (matches12[i] >= 0) {
            prev_matched[2 * i] = k2[matches12[i]].x;
            prev_matched[2 * i + 1] = k2[matches12[i]].y;
        }
    if (pair_count)
        *pair_count = pairs;
    free(matched_dist);
    free(matches21);
    free(cand);
    free(hist_items);
    free(hist_bin);
    orc_grid_free(&g);
    return nmatches;
}

/* trace accessors for the Python binding */
/* ------------------------------------------------------------------ stereo matching */

typedef struct {
    int dist, idx;
} orc_dist_idx;

static int orc_cmp_dist_idx(const void *a, const void *b)
{
    const orc_dist_idx *A = (const orc_dist_idx *)a, *B = (const orc_dist_idx *)b;
    if (A->dist != B->dist)
        return A->dist < B->dist ? -1 : 1;
    return A->idx < B->idx ? -1 : (A->idx > B->idx ? 1 : 0);
}

/* Frame::ComputeStereoMatches, src/Frame.cc:534-708.  TH_HIGH = 100, TH_LOW = 50 (src/ORBmatcher.cc:37-38).
 * kL/dL, kR/dR: keypoints (level-0 coordinates, octave) and descriptors of the left and right image;
 * pyrL/pyrR: the nlevels pyramid images of both (mvImagePyramid), widths lw, heights lh, row strides
 * lsL/lsR; sf/inv: scale and inverse scale factors; mb, mbf: baseline and baseline * fx.
 * uright/depth receive nL entries (-1 where unmatched); sad (may be NULL) the best window distance
 * of the accepted matches before the median filter, -1 elsewhere.  Returns the number of matches left.
 * Rows outside the image (the reference would index out of range) are skipped. */
int orc_compute_stereo_matches(const orc_keypoint *kL, const uint8_t *dL, int nL, const orc_keypoint *kR,
                               const uint8_t *dR, int nR, const uint8_t *const *pyrL, const uint8_t *const *pyrR,
                               const int *lw, const int *lh, const int *lsL, const int *lsR, int nlevels,
                               const float *sf, const float *inv, float mb, float mbf, float *uright, float *depth,
                               int *sad)
{
    enum { TH_HIGH = 100, TH_LOW = 50 };
    (void)nlevels;
    for (int i = 0; i < nL; i++) {
        uright[i] = -1.0f;
        depth[i] = -1.0f;
        if (sad)
            sad[i] = -1;
    }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = lh[0];
    /* :543-561 row table: right keypoint iR is listed in every row of [floor(y-r), ceil(y+r)] */
    int *cnt = (int *)calloc((size_t)nRows + 1, sizeof(int));
    for (int iR = 0; iR < nR; iR++) {
        const float kpY = kR[iR].y;
        const float r = 2.0f * sf[kR[iR].octave];
        const int maxr = (int)ceil(kpY + r), minr = (int)floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows)
                cnt[yi + 1]++;
    }
    for (int y = 0; y < nRows; y++)
        cnt[y + 1] += cnt[y];
    int *rows = (int *)malloc(sizeof(int) * (size_t)(cnt[nRows] > 0 ? cnt[nRows] : 1));
    int *fill = (int *)malloc(sizeof(int) * (size_t)(nRows + 1));
    memcpy(fill, cnt, sizeof(int) * (size_t)(nRows + 1));
    for (int iR = 0; iR < nR; iR++) {
        const float kpY = kR[iR].y;
        const float r = 2.0f * sf[kR[iR].octave];
        const int maxr = (int)ceil(kpY + r), minr = (int)floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows)
                rows[fill[yi]++] = iR;
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    orc_dist_idx *vd = (orc_dist_idx *)malloc(sizeof(orc_dist_idx) * (size_t)(nL > 0 ? nL : 1));
    int nvd = 0;
    for (int iL = 0; iL < nL; iL++) {
        const orc_keypoint *kpL = &kL[iL];
        const int levelL = kpL->octave;
        const float vL = kpL->y, uL = kpL->x;
        const int row = (int)vL; /* vRowIndices[vL]: float index, truncated */
        if (row < 0 || row >= nRows || cnt[row] == cnt[row + 1])
            continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0)
            continue;
        int bestDist = TH_HIGH;
        int bestIdxR = 0;
        for (int c = cnt[row]; c < cnt[row + 1]; c++) {
            const int iR = rows[c];
            const orc_keypoint *kpR = &kR[iR];
            if (kpR->octave < levelL - 1 || kpR->octave > levelL + 1)
                continue;
            const float uR = kpR->x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_popcnt256(dL + (size_t)iL * 32, dR + (size_t)iR * 32);
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdxR = iR;
                }
            }
        }
        if (!(bestDist < thOrbDist))
            continue;
        /* :619-691 sub-pixel refinement by an 11x11 window slid over 11 positions at the keypoint's level */
        const float uR0 = kR[bestIdxR].x;
        const float scaleFactor = inv[levelL];
        const float scaleduL = roundf(kpL->x * scaleFactor);
        const float scaledvL = roundf(kpL->y * scaleFactor);
        const float scaleduR0 = roundf(uR0 * scaleFactor);
        const int w = 5, L = 5;
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= (float)lw[levelL])
            continue;
        const int su = (int)scaleduL, sv = (int)scaledvL, sr = (int)scaleduR0;
        const uint8_t *IL = pyrL[levelL], *IR = pyrR[levelL];
        const int pL = lsL[levelL], pR = lsR[levelL];
        const int cL = IL[(size_t)sv * pL + su];
        int best = INT_MAX, bestinc = 0;
        float vDists[11];
        for (int incR = -L; incR <= L; incR++) {
            const int cR = IR[(size_t)sv * pR + sr + incR];
            /* cv::norm(IL, IR, NORM_L1) on the centre-subtracted float windows: integer valued, exact */
            int acc = 0;
            for (int y = -w; y <= w; y++)
                for (int x = -w; x <= w; x++) {
                    const int a = IL[(size_t)(sv + y) * pL + su + x] - cL;
                    const int b = IR[(size_t)(sv + y) * pR + sr + incR + x] - cR;
                    acc += abs(a - b);
                }
            const float dist = (float)acc;
            if (dist < (float)best) {
                best = (int)dist;
                bestinc = incR;
            }
            vDists[L + incR] = dist;
        }
        if (bestinc == -L || bestinc == L)
            continue;
        const float dist1 = vDists[L + bestinc - 1], dist2 = vDists[L + bestinc], dist3 = vDists[L + bestinc + 1];
        const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
        if (deltaR < -1 || deltaR > 1)
            continue;
        float bestuR = sf[levelL] * ((float)scaleduR0 + (float)bestinc + deltaR);
        float disparity = uL - bestuR;
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) {
                disparity = 0.01;
                bestuR = uL - 0.01;
            }
            depth[iL] = mbf / disparity;
            uright[iL] = bestuR;
            if (sad)
                sad[iL] = best;
            vd[nvd].dist = best;
            vd[nvd].idx = iL;
            nvd++;
        }
    }
    int left = nvd;
    if (nvd > 0) { /* :694-707 (the reference reads vDistIdx[0] of an empty vector otherwise) */
        qsort(vd, (size_t)nvd, sizeof(orc_dist_idx), orc_cmp_dist_idx);
        const float median = (float)vd[nvd / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nvd - 1; i >= 0; i--) {
            if ((float)vd[i].dist < thDist)
                break;
            uright[vd[i].idx] = -1;
            depth[vd[i].idx] = -1;
            left--;
        }
    }
    free(vd);
    free(fill);
    free(rows);
    free(cnt);
    return left;
}

/* ------------------------------------------------------------------ matching by vocabulary node */

/* ORBmatcher::SearchByBoW, src/ORBmatcher.cc:176-311 (key frame against frame: strict_lt = 0, valid2 =
 * NULL) and :552-686 (key frame against key frame: strict_lt = 1, valid2 given).  The feature vectors
 * (DBoW2::FeatureVector, a map node id -> feature indices) are given as ascending node ids with CSR
 * index lists.  valid1/valid2: the feature has a usable map point.  match12 (n1 entries) / match21
 * (n2 entries) receive the partner's index or -1.  Returns the number of matches. */
int orc_search_by_bow(const uint8_t *d1, const float *ang1, const uint8_t *valid1, int n1, const int *ids1,
                      const int *start1, const int *items1, int m1, const uint8_t *d2, const float *ang2,
                      const uint8_t *valid2, int n2, const int *ids2, const int *start2, const int *items2, int m2,
                      int strict_lt, float ratio, int check_ori, int *match12, int *match21)
{
    enum { TH_LOW = 50, HISTO = 30 };
    for (int i = 0; i < n1; i++)
        match12[i] = -1;
    for (int i = 0; i < n2; i++)
        match21[i] = -1;
    int nmatches = 0;
    int hist[HISTO] = {0};
    int *bin_of = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
    for (int i = 0; i < n1; i++)
        bin_of[i] = -1;
    const float factor = HISTO / 360.0f;
    int a = 0, b = 0;
    while (a < m1 && b < m2) {
        if (ids1[a] == ids2[b]) {
            for (int i1 = start1[a]; i1 < start1[a + 1]; i1++) {
                const int idx1 = items1[i1];
                if (valid1 && !valid1[idx1])
                    continue;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = start2[b]; i2 < start2[b + 1]; i2++) {
                    const int idx2 = items2[i2];
                    if (match21[idx2] >= 0 || (valid2 && !valid2[idx2]))
                        continue;
                    const int dist = orc_popcnt256(d1 + (size_t)idx1 * 32, d2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = idx2;
                    } else if (dist < bestDist2)
                        bestDist2 = dist;
                }
                if (strict_lt ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW) {
                    if ((float)bestDist1 < ratio * (float)bestDist2) {
                        match12[idx1] = bestIdx2;
                        match21[bestIdx2] = idx1;
                        if (check_ori) {
                            float rot = ang1[idx1] - ang2[bestIdx2];
                            if (rot < 0.0)
                                rot += 360.0f;
                            int bin = (int)roundf(rot * factor);
                            if (bin == HISTO)
                                bin = 0;
                            hist[bin]++;
                            bin_of[idx1] = bin;
                        }
                        nmatches++;
                    }
                }
            }
            a++;
            b++;
        } else if (ids1[a] < ids2[b]) {
            while (a < m1 && ids1[a] < ids2[b]) /* lower_bound */
                a++;
        } else {
            while (b < m2 && ids2[b] < ids1[a])
                b++;
        }
    }
    if (check_ori) {
        int i1m, i2m, i3m;
        orc_three_maxima(hist, HISTO, &i1m, &i2m, &i3m);
        for (int i = 0; i < n1; i++) {
            const int bn = bin_of[i];
            if (bn < 0 || bn == i1m || bn == i2m || bn == i3m)
                continue;
            match21[match12[i]] = -1;
            match12[i] = -1;
            nmatches--;
        }
    }
    free(bin_of);
    return nmatches;
}

/* ------------------------------------------------------------------ vocabulary transform */

/* TemplatedVocabulary<FORB>::transform for single features (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:
 * 1217-1259, distance = FORB::distance, Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101, a popcount of a XOR): a
 * feature descends from the root, at every node to the child with the smallest distance (the first child
 * wins ties), until a leaf.  The tree is given flat: node_desc (32 bytes per node), the children of node i
 * are children[child_start[i] .. child_start[i+1]), word_id (leaf) and weight per node; L = depth of the
 * tree.  Outputs per feature: word id, weight and the id of the node reached at level L - levelsup
 * (0 = root when that level is <= 0). */
void orc_vocabulary_transform(const uint8_t *node_desc, const int *child_start, const int *children,
                              const int *word_id, const double *weight, int L, const uint8_t *feat, int n,
                              int levelsup, int *out_word, int *out_node, double *out_weight)
{
    const int nid_level = L - levelsup;
    for (int f = 0; f < n; f++) {
        int nid = 0, final_id = 0, level = 0;
        do {
            ++level;
            const int c0 = child_start[final_id], c1 = child_start[final_id + 1];
            int best_id = children[c0];
            int best_d = orc_popcnt256(feat + (size_t)f * 32, node_desc + (size_t)best_id * 32);
            for (int c = c0 + 1; c < c1; c++) {
                const int id = children[c];
                const int d = orc_popcnt256(feat + (size_t)f * 32, node_desc + (size_t)id * 32);
                if (d < best_d) {
                    best_d = d;
                    best_id = id;
                }
            }
            final_id = best_id;
            if (level == nid_level)
                nid = final_id;
        } while (child_start[final_id] != child_start[final_id + 1]);
        out_word[f] = word_id[final_id];
        out_node[f] = nid;
        out_weight[f] = weight[final_id];
    }
}

/* ------------------------------------------------------------------ distinctive descriptor */

static int orc_cmp_int(const void *a, const void *b)
{
    const int A = *(const int *)a, B = *(const int *)b;
    return A < B ? -1 : (A > B ? 1 : 0);
}

/* MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:278-318: of the n descriptors observed for
 * a map point (rows of 32 bytes, in the order the caller collected them), the index of the one
 * with the least median distance to all of them (the row of distances includes the 0 to itself; the
 * median is element (int)(0.5*(n-1)) of the sorted row; the first index wins ties).  -1 for n == 0. */
int orc_distinctive_descriptor(const uint8_t *desc, int n)
{
    if (n <= 0)
        return -1;
    int *row = (int *)malloc(sizeof(int) * (size_t)n);
    int best_median = INT_MAX, best_idx = 0;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++)
            row[j] = i == j ? 0 : orc_popcnt256(desc + (size_t)i * 32, desc + (size_t)j * 32);
        qsort(row, (size_t)n, sizeof(int), orc_cmp_int);
        const int median = row[(int)(0.5 * (n - 1))];
        if (median < best_median) {
            best_median = median;
            best_idx = i;
        }
    }
    free(row);
    return best_idx;
}

int orc_trace_sizeof(void) { return (int)sizeof(orc_trace); }

/* ORBmatcher::SearchByProjection, the two forms the tracking thread runs on every frame:
 *   use_second = 1: (Frame &F, const vector<MapPoint*>&, th), src/ORBmatcher.cc:46-137 - best and second
 *                   best with their levels, accepted if best <= th_dist unless both lie on one level and
 *                   best > nnratio * second (:118-126);
 *   use_second = 0: (Frame &CurrentFrame, const Frame &LastFrame, th, bMono), :1412-1566 - best only
 *                   (:1500-1520), then the rotation histogram with factor HISTO_LENGTH/360.0f (:1421,
 *                   :1528-1560) when check_orientation is 1.  The relocalisation form (Frame &CurrentFrame,
 *                   KeyFrame *pKF, sAlreadyFound, th, ORBdist), :1568-1690, is the same search with
 *                   th_dist = ORBdist, taken = "holds any map point" (:1636-1637) and check_orientation = 2:
 *                   its histogram uses factor 1.0f/HISTO_LENGTH (:1581).
 * The caller has done what precedes the window query: source s is skipped unless valid[s] (mbTrackInView
 * and not bad; or a last-frame map point that is no outlier and projects into the image), (x, y) is the
 * projection, r the window radius (r * mvScaleFactors[level], resp. th * mvScaleFactors[octave]),
 * min_level / max_level the level window of GetFeaturesInArea, xr the predicted right coordinate that a
 * feature with uright > 0 must meet within r (:89-94, :1488-1494).
 * taken[i] != 0: feature i holds a map point with Observations() > 0 and is skipped (:84-86, :1483-1485).
 * An accepted match writes the feature's map point (:127, :1522): later sources skip that feature iff
 * blocks[s] (the source's map point has observations).  match[s] = feature or -1; feature_match[i] = the
 * last source written to feature i, -1 if none, -2 if the rotation check cleared it (:1553); returns nmatches. */
int orc_search_by_projection(const orc_keypoint *kps, const uint8_t *desc, const float *uright, const uint8_t *taken_in,
                             int n, float min_x, float max_x, float min_y, float max_y, const uint8_t *valid,
                             const float *x, const float *y, const float *r, const int *min_level, const int *max_level,
                             const float *xr, const uint8_t *src_desc, const uint8_t *blocks, const float *src_angle,
                             int ns, int use_second, float nnratio, int th_dist, int check_orientation, int *match,
                             int *feature_match)
{
    enum { HISTO_LENGTH = 30 };
    orc_grid g;
    orc_grid_build(&g, kps, n, min_x, max_x, min_y, max_y);
    uint8_t *taken = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    int *cand = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int *bin_of = (int *)malloc(sizeof(int) * (size_t)(ns > 0 ? ns : 1));
    int hist[HISTO_LENGTH] = {0};
    if (taken_in)
        memcpy(taken, taken_in, (size_t)n);
    for (int i = 0; i < n; i++)
        feature_match[i] = -1;
    const float factor = check_orientation == 2 ? 1.0f / HISTO_LENGTH : HISTO_LENGTH / 360.0f;
    int nmatches = 0;
    for (int s = 0; s < ns; s++) {
        match[s] = -1;
        bin_of[s] = -1;
        if (valid && !valid[s])
            continue;
        const int nc = orc_features_in_area(&g, kps, x[s], y[s], r[s], min_level[s], max_level[s], cand);
        if (nc == 0)
            continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (taken[idx])
                continue;
            if (uright && uright[idx] > 0) {
                const float er = fabsf(xr[s] - uright[idx]);
                if (er > r[s])
                    continue;
            }
            const int dist = orc_descriptor_distance(src_desc + (size_t)s * 32, desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = kps[idx].octave;
                bestIdx = idx;
            } else if (use_second && dist < bestDist2) {
                bestLevel2 = kps[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= th_dist) {
            if (use_second && bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2)
                continue;
            match[s] = bestIdx;
            feature_match[bestIdx] = s;
            taken[bestIdx] = blocks ? (blocks[s] != 0) : 1;
            nmatches++;
            if (check_orientation) {
                float rot = src_angle[s] - kps[bestIdx].angle;
                if (rot < 0.0)
                    rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH)
                    bin = 0;
                bin_of[s] = bin;
                hist[bin]++;
            }
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        orc_three_maxima(hist, HISTO_LENGTH, &i1, &i2, &i3);
        for (int s = 0; s < ns; s++) {
            const int b = bin_of[s];
            if (b >= 0 && b != i1 && b != i2 && b != i3) {
                feature_match[match[s]] = -2;
                nmatches--;
            }
        }
    }
    free(taken), free(cand), free(bin_of);
    orc_grid_free(&g);
    return nmatches;
}
