/*
 * oracle/lk.c -- CPU restatement of cv::calcOpticalFlowPyrLK with default arguments
 * (TEST INFRASTRUCTURE; see svo_oracle.h.  PARITY UNPINNED.)
 *
 * Reference call sites: /root/reference/src/tracking.cpp:18 (denseLKtracking, L->R) and
 * src/tracking.cpp:52 (PyrLKtrackFrame2Frame, t-1 -> t); both pass no optional arguments,
 * so winSize 21x21, maxLevel 3, criteria (COUNT+EPS, 30, 0.01), flags 0,
 * minEigThreshold 1e-4 apply.  Images arrive as 3-channel BGR uint8
 * (src/keyFrameManagement.cpp:52-54: imread default, grey conversion commented out).
 *
 * The arithmetic is OpenCV's video/lkpyramid.cpp, restated from its published algorithm
 * (SURVEY.md appendix A.1): 5-tap pyrDown with reflect-101, Scharr derivative with
 * reflect-101 inside the image and ZERO outside, 14-bit fixed-point bilinear weights,
 * int16 patches (intensity x32), 2^-20 scaling of the normal equations, minimum
 * eigenvalue normalised by the window area but not by the channel count, two stop rules.
 *
 * One stated deviation: OpenCV accumulates A11,A12,A22,b1,b2 in float32 in an order that
 * depends on its SIMD build (scalar, SSE2, NEON and universal-intrinsic paths all differ
 * in the last bits).  Here those sums of integer products are accumulated EXACTLY (int64)
 * and rounded to float32 once.  That is inside OpenCV's own cross-build spread and makes
 * the result independent of summation order, so a wavefront-parallel reduction on the GPU
 * can be compared bit-for-bit.
 */
#include "svo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
#else
int orc_num_threads(void) { return 1; }
void orc_set_num_threads(int n) { (void)n; }
#endif

void orc_lk_default_params(orc_lk_params *p)
{
    p->win = ORC_LK_WIN;
    p->max_level = 3;
    p->max_count = 30;
    p->epsilon = 0.01;
    p->min_eig_thr = 1e-4;
}

void orc_pyr_sizes(int w, int h, int levels, int *ws, int *hs)
{
    ws[0] = w;
    hs[0] = h;
    for (int l = 1; l < levels; l++) {
        ws[l] = (ws[l - 1] + 1) / 2;
        hs[l] = (hs[l - 1] + 1) / 2;
    }
}

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * (len - 1) - p;
    }
    return p;
}

void orc_pyr_down(const uint8_t *src, int w, int h, int c, uint8_t *dst)
{
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    int *rows = (int *)malloc(sizeof(int) * 5 * dw * c);
    for (int y = 0; y < dh; y++) {
        /* horizontal pass of the 5 source rows 2y-2 .. 2y+2 */
        for (int k = 0; k < 5; k++) {
            int sy = reflect101(2 * y - 2 + k, h);
            const uint8_t *s = src + (size_t)sy * w * c;
            int *r = rows + k * dw * c;
            for (int x = 0; x < dw; x++) {
                int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = 2 * x,
                    x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
                for (int ch = 0; ch < c; ch++)
                    r[x * c + ch] = s[x0 * c + ch] + 4 * s[x1 * c + ch] + 6 * s[x2 * c + ch] +
                                    4 * s[x3 * c + ch] + s[x4 * c + ch];
            }
        }
        uint8_t *d = dst + (size_t)y * dw * c;
        for (int i = 0; i < dw * c; i++) {
            int v = rows[i] + 4 * rows[dw * c + i] + 6 * rows[2 * dw * c + i] +
                    4 * rows[3 * dw * c + i] + rows[4 * dw * c + i];
            d[i] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
}

/* calcSharrDeriv: dx = [-1 0 1] applied to the [3 10 3]^T-smoothed column, dy likewise
 * transposed; reflect-101 at the image border. */
void orc_scharr(const uint8_t *src, int w, int h, int c, int16_t *dst)
{
    for (int y = 0; y < h; y++) {
        int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
        const uint8_t *r0 = src + (size_t)ym * w * c, *r1 = src + (size_t)y * w * c,
                      *r2 = src + (size_t)yp * w * c;
        for (int x = 0; x < w; x++) {
            int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            for (int ch = 0; ch < c; ch++) {
                int t0m = (r0[xm * c + ch] + r2[xm * c + ch]) * 3 + r1[xm * c + ch] * 10;
                int t0p = (r0[xp * c + ch] + r2[xp * c + ch]) * 3 + r1[xp * c + ch] * 10;
                int t1m = r2[xm * c + ch] - r0[xm * c + ch];
                int t1c = r2[x * c + ch] - r0[x * c + ch];
                int t1p = r2[xp * c + ch] - r0[xp * c + ch];
                int16_t *d = dst + ((size_t)y * w + x) * c * 2 + ch * 2;
                d[0] = (int16_t)(t0p - t0m);
                d[1] = (int16_t)((t1p + t1m) * 3 + t1c * 10);
            }
        }
    }
}

/* Padded level buffers, as buildOpticalFlowPyramid lays them out: the image carries a
 * reflect-101 border of `win` pixels (pyrBorder default), the derivative a zero border
 * (BORDER_CONSTANT).  pix()/der() index them with image coordinates in [-win, size+win). */
static inline int pix(const uint8_t *img, int w, int h, int c, int x, int y, int ch)
{
    (void)h;
    return img[((size_t)(y + ORC_LK_WIN) * (w + 2 * ORC_LK_WIN) + (x + ORC_LK_WIN)) * c + ch];
}
static inline int der(const int16_t *d, int w, int h, int c, int x, int y, int ch, int which)
{
    (void)h;
    return d[((size_t)(y + ORC_LK_WIN) * (w + 2 * ORC_LK_WIN) + (x + ORC_LK_WIN)) * c * 2 +
             ch * 2 + which];
}

static uint8_t *pad_image(const uint8_t *src, int w, int h, int c)
{
    const int B = ORC_LK_WIN, pw = w + 2 * B, ph = h + 2 * B;
    uint8_t *out = (uint8_t *)malloc((size_t)pw * ph * c);
    for (int y = 0; y < ph; y++) {
        int sy = reflect101(y - B, h);
        for (int x = 0; x < pw; x++) {
            int sx = reflect101(x - B, w);
            for (int ch = 0; ch < c; ch++)
                out[((size_t)y * pw + x) * c + ch] = src[((size_t)sy * w + sx) * c + ch];
        }
    }
    return out;
}
static int16_t *pad_deriv(const int16_t *src, int w, int h, int c)
{
    const int B = ORC_LK_WIN, pw = w + 2 * B, ph = h + 2 * B;
    int16_t *out = (int16_t *)calloc((size_t)pw * ph * c * 2, sizeof(int16_t));
    for (int y = 0; y < h; y++)
        memcpy(out + ((size_t)(y + B) * pw + B) * c * 2, src + (size_t)y * w * c * 2,
               sizeof(int16_t) * w * c * 2);
    return out;
}

#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static inline int cv_round_f(float v) { return (int)lrintf(v); } /* round-half-even */
static inline int cv_floor_f(float v) { return (int)floorf(v); }

/* optional instrumentation: per-point iteration count summed over levels (tests/bench) */
static int *g_iter_counter = 0;
void orc_lk_set_iter_counter(int *buf) { g_iter_counter = buf; }

typedef struct {
    int w, h;
    uint8_t *img;   /* unpadded level */
    uint8_t *pimg;  /* reflect-101 padded by win */
    int16_t *deriv; /* zero padded by win; only for the previous image */
} level_t;

int orc_lk_track(const uint8_t *prev, const uint8_t *next, int w, int h, int c,
                 const float *prev_pts, int n, float *next_pts, uint8_t *status,
                 float *err, float *min_eig, const orc_lk_params *params)
{
    orc_lk_params P;
    if (params)
        P = *params;
    else
        orc_lk_default_params(&P);
    if (P.win != ORC_LK_WIN)
        return -1; /* the padded layout is compiled for the 21x21 default window */
    if (!prev || !next || w <= 0 || h <= 0 || c <= 0 || c > 4 || n < 0 ||
        P.max_level >= ORC_LK_MAX_LEVELS)
        return -1;
    const int win = P.win, nl = P.max_level + 1;
    /* criteria clamp as in calcOpticalFlowPyrLK */
    int max_count = P.max_count < 0 ? 0 : (P.max_count > 100 ? 100 : P.max_count);
    double eps = P.epsilon < 0 ? 0 : (P.epsilon > 10 ? 10 : P.epsilon);
    eps *= eps;

    int ws[ORC_LK_MAX_LEVELS], hs[ORC_LK_MAX_LEVELS];
    orc_pyr_sizes(w, h, nl, ws, hs);
    level_t pl[ORC_LK_MAX_LEVELS], nx[ORC_LK_MAX_LEVELS];
    for (int l = 0; l < nl; l++) {
        size_t sz = (size_t)ws[l] * hs[l] * c;
        pl[l].w = nx[l].w = ws[l];
        pl[l].h = nx[l].h = hs[l];
        pl[l].img = (uint8_t *)malloc(sz);
        nx[l].img = (uint8_t *)malloc(sz);
        nx[l].deriv = NULL;
        if (l == 0) {
            memcpy(pl[l].img, prev, sz);
            memcpy(nx[l].img, next, sz);
        } else {
            orc_pyr_down(pl[l - 1].img, ws[l - 1], hs[l - 1], c, pl[l].img);
            orc_pyr_down(nx[l - 1].img, ws[l - 1], hs[l - 1], c, nx[l].img);
        }
        int16_t *dtmp = (int16_t *)malloc(sz * 2 * sizeof(int16_t));
        orc_scharr(pl[l].img, ws[l], hs[l], c, dtmp);
        pl[l].deriv = pad_deriv(dtmp, ws[l], hs[l], c);
        free(dtmp);
        pl[l].pimg = pad_image(pl[l].img, ws[l], hs[l], c);
        nx[l].pimg = pad_image(nx[l].img, ws[l], hs[l], c);
    }

    for (int i = 0; i < n; i++) {
        status[i] = 1;
        if (err)
            err[i] = 0.f;
        if (min_eig)
            min_eig[i] = 0.f;
    }

    const int wn = win * win * c;
    const float half = (win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int W_BITS = 14;

    /* Points are independent of each other (only the levels of ONE point depend on each other),
     * so the point loop is shared among OpenMP threads; every sum is exact integer arithmetic,
     * the result does not depend on the thread count (OMP_NUM_THREADS=1 gives the scalar port). */
#pragma omp parallel
    {
    int16_t *Iw = (int16_t *)malloc(sizeof(int16_t) * wn);
    int16_t *dIw = (int16_t *)malloc(sizeof(int16_t) * wn * 2);
    for (int level = P.max_level; level >= 0; level--) {
        const int lw = ws[level], lh = hs[level];
        const uint8_t *I = pl[level].pimg, *J = nx[level].pimg;
        const int16_t *dI = pl[level].deriv;
        const float scale = (float)(1. / (1 << level));
#pragma omp for schedule(dynamic, 16)
        for (int p = 0; p < n; p++) {
            float px = prev_pts[2 * p] * scale, py = prev_pts[2 * p + 1] * scale;
            float nxp, nyp;
            if (level == P.max_level) {
                nxp = px;
                nyp = py;
            } else {
                nxp = next_pts[2 * p] * 2.f;
                nyp = next_pts[2 * p + 1] * 2.f;
            }
            next_pts[2 * p] = nxp;
            next_pts[2 * p + 1] = nyp;

            px -= half;
            py -= half;
            int ipx = cv_floor_f(px), ipy = cv_floor_f(py);
            if (ipx < -win || ipx >= lw || ipy < -win || ipy >= lh) {
                if (level == 0) {
                    status[p] = 0;
                    if (err)
                        err[p] = 0.f;
                }
                continue;
            }
            float a = px - ipx, b = py - ipy;
            int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

            int64_t sA11 = 0, sA12 = 0, sA22 = 0;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++)
                    for (int ch = 0; ch < c; ch++) {
                        int X = ipx + x, Y = ipy + y;
                        int ival = DESCALE(pix(I, lw, lh, c, X, Y, ch) * iw00 +
                                               pix(I, lw, lh, c, X + 1, Y, ch) * iw01 +
                                               pix(I, lw, lh, c, X, Y + 1, ch) * iw10 +
                                               pix(I, lw, lh, c, X + 1, Y + 1, ch) * iw11,
                                           W_BITS - 5);
                        int ixval = DESCALE(der(dI, lw, lh, c, X, Y, ch, 0) * iw00 +
                                                der(dI, lw, lh, c, X + 1, Y, ch, 0) * iw01 +
                                                der(dI, lw, lh, c, X, Y + 1, ch, 0) * iw10 +
                                                der(dI, lw, lh, c, X + 1, Y + 1, ch, 0) * iw11,
                                            W_BITS);
                        int iyval = DESCALE(der(dI, lw, lh, c, X, Y, ch, 1) * iw00 +
                                                der(dI, lw, lh, c, X + 1, Y, ch, 1) * iw01 +
                                                der(dI, lw, lh, c, X, Y + 1, ch, 1) * iw10 +
                                                der(dI, lw, lh, c, X + 1, Y + 1, ch, 1) * iw11,
                                            W_BITS);
                        int e = (y * win + x) * c + ch;
                        Iw[e] = (int16_t)ival;
                        dIw[2 * e] = (int16_t)ixval;
                        dIw[2 * e + 1] = (int16_t)iyval;
                        sA11 += (int64_t)ixval * ixval;
                        sA12 += (int64_t)ixval * iyval;
                        sA22 += (int64_t)iyval * iyval;
                    }
            float A11 = (float)(double)sA11 * FLT_SCALE;
            float A12 = (float)(double)sA12 * FLT_SCALE;
            float A22 = (float)(double)sA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                           (float)(2 * win * win);
            if (level == 0 && min_eig)
                min_eig[p] = minEig;
            if (minEig < (float)P.min_eig_thr || D < 1.1920928955078125e-7f) {
                if (level == 0)
                    status[p] = 0;
                continue;
            }
            D = 1.f / D;
            nxp -= half;
            nyp -= half;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_count; j++) {
                int inx = cv_floor_f(nxp), iny = cv_floor_f(nyp);
                if (inx < -win || inx >= lw || iny < -win || iny >= lh) {
                    if (level == 0)
                        status[p] = 0;
                    break;
                }
                a = nxp - inx;
                b = nyp - iny;
                iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t sb1 = 0, sb2 = 0;
                if (g_iter_counter)
                    g_iter_counter[p]++;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++)
                        for (int ch = 0; ch < c; ch++) {
                            int X = inx + x, Y = iny + y;
                            int e = (y * win + x) * c + ch;
                            int diff = DESCALE(pix(J, lw, lh, c, X, Y, ch) * iw00 +
                                                   pix(J, lw, lh, c, X + 1, Y, ch) * iw01 +
                                                   pix(J, lw, lh, c, X, Y + 1, ch) * iw10 +
                                                   pix(J, lw, lh, c, X + 1, Y + 1, ch) * iw11,
                                               W_BITS - 5) -
                                       Iw[e];
                            sb1 += (int64_t)diff * dIw[2 * e];
                            sb2 += (int64_t)diff * dIw[2 * e + 1];
                        }
                float b1 = (float)(double)sb1 * FLT_SCALE;
                float b2 = (float)(double)sb2 * FLT_SCALE;
                float dx = (float)((A12 * b2 - A22 * b1) * D);
                float dy = (float)((A12 * b1 - A11 * b2) * D);
                nxp += dx;
                nyp += dy;
                next_pts[2 * p] = nxp + half;
                next_pts[2 * p + 1] = nyp + half;
                if ((double)dx * dx + (double)dy * dy <= eps)
                    break;
                if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                    next_pts[2 * p] -= dx * 0.5f;
                    next_pts[2 * p + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx;
                pdy = dy;
            }
            /* the reference always asks for err (src/tracking.cpp:17,50), so the final
             * out-of-image check below always runs; err itself is optional here */
            if (status[p] && level == 0) {
                float qx = next_pts[2 * p] - half, qy = next_pts[2 * p + 1] - half;
                int iqx = cv_floor_f(qx), iqy = cv_floor_f(qy);
                if (iqx < -win || iqx >= lw || iqy < -win || iqy >= lh) {
                    status[p] = 0;
                    continue;
                }
                float aa = qx - iqx, bb = qy - iqy;
                iw00 = cv_round_f((1.f - aa) * (1.f - bb) * (1 << W_BITS));
                iw01 = cv_round_f(aa * (1.f - bb) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - aa) * bb * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t sabs = 0; /* < 2^24, exact in float as OpenCV sums it */
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++)
                        for (int ch = 0; ch < c; ch++) {
                            int X = iqx + x, Y = iqy + y;
                            int e = (y * win + x) * c + ch;
                            int diff = DESCALE(pix(J, lw, lh, c, X, Y, ch) * iw00 +
                                                   pix(J, lw, lh, c, X + 1, Y, ch) * iw01 +
                                                   pix(J, lw, lh, c, X, Y + 1, ch) * iw10 +
                                                   pix(J, lw, lh, c, X + 1, Y + 1, ch) * iw11,
                                               W_BITS - 5) -
                                       Iw[e];
                            sabs += diff < 0 ? -diff : diff;
                        }
                if (err)
                    err[p] = (float)sabs / (float)(32 * win * c * win);
            }
        }
    }
    free(Iw);
    free(dIw);
    }
    for (int l = 0; l < nl; l++) {
        free(pl[l].img);
        free(nx[l].img);
        free(pl[l].pimg);
        free(nx[l].pimg);
        free(pl[l].deriv);
    }
    return 0;
}

/* /root/reference/src/tracking.cpp:4-12 (denseKeypointExtractor) */
int orc_grid_keypoints(int rows, int cols, int step, float *out_xy, int cap)
{
    int k = 0;
    if (step <= 0)
        return 0;
    for (int y = step; y < rows - step; y += step)
        for (int x = step; x < cols - step; x += step) {
            if (out_xy && k < cap) {
                out_xy[2 * k] = (float)x;
                out_xy[2 * k + 1] = (float)y;
            }
            k++;
        }
    return k;
}
