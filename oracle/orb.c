/*
 * orb.c -- CPU restatement of the feature side of visualSLAM::checkLoopDetectorStatus
 * (src/optimizationStuff.cpp:49-64: cv::ORB::create()->detectAndCompute, 500 features).
 *
 * TEST INFRASTRUCTURE (see svo_oracle.h).  PARITY UNPINNED: cv::ORB is un-vendored and its learned
 * 256-pair sampling pattern is part of the OpenCV sources, which are absent; what follows is the
 * published ORB recipe (oFAST + steered BRIEF: FAST-9 corners, Harris ranking, intensity-centroid
 * orientation, binary tests on a smoothed patch) with these stated choices:
 *   - grey = (1868 B + 9617 G + 4899 R + 8192) >> 14 (OpenCV's fixed-point BGR2GRAY);
 *   - 3 octaves of the factor-2 Gaussian pyramid the tracker already builds (upstream: 8 levels of
 *     factor 1.2), 500 features split 286 / 143 / 71;
 *   - FAST-9 threshold 20, Harris response (7x7 block, k = 0.04, integer gradient sums, the float
 *     formula below), 3x3 non-maximum suppression on that response, 19-pixel image margin;
 *   - orientation from the integer moments m10, m01 of the radius-15 disc, used as the unit vector
 *     (m10, m01) / |.| (no angle, no trigonometric library call);
 *   - 256 tests on the 5x5-binomial-smoothed level, sample offsets from the seeded generator
 *     orb_pattern() below (within +-13), rotated by the unit vector and rounded to the pixel grid.
 * All of it is integer arithmetic plus a handful of individually rounded float operations, so the
 * GPU implementation is compared bit for bit.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"

#define ORB_EDGE 19
#define ORB_HALF_PATCH 15

void orc_orb_pattern(int8_t *pat /* 256 * 4 */)
{
    /* sum of three uniforms in [-13, 13] scaled down: a bell-shaped spread like BRIEF's G II */
    uint32_t s = 0x9E3779B9u;
    for (int i = 0; i < 256 * 4; i++) {
        int acc = 0;
        for (int k = 0; k < 3; k++) {
            s = s * 1664525u + 1013904223u;
            acc += (int)((s >> 16) % 27u) - 13;
        }
        int v = acc / 2;
        if (v > 13)
            v = 13;
        if (v < -13)
            v = -13;
        pat[i] = (int8_t)v;
    }
    for (int i = 0; i < 256; i++) /* a test must compare two different pixels */
        if (pat[4 * i] == pat[4 * i + 2] && pat[4 * i + 1] == pat[4 * i + 3])
            pat[4 * i + 2] = (int8_t)(pat[4 * i + 2] >= 0 ? pat[4 * i + 2] - 1 : pat[4 * i + 2] + 1);
}

void orc_bgr_to_gray(const uint8_t *img, int w, int h, int c, uint8_t *gray)
{
    for (int i = 0; i < w * h; i++)
        gray[i] = c == 1 ? img[i]
                         : (uint8_t)((1868 * img[3 * i] + 9617 * img[3 * i + 1] + 4899 * img[3 * i + 2] + 8192) >> 14);
}

static inline int refl(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

/* 5x5 binomial smoothing at the same size, reflect-101 border, (sum + 128) >> 8 */
void orc_blur5(const uint8_t *src, int w, int h, uint8_t *dst)
{
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int j = -2; j <= 2; j++)
                for (int i = -2; i <= 2; i++)
                    s += k[j + 2] * k[i + 2] * src[refl(y + j, h) * w + refl(x + i, w)];
            dst[y * w + x] = (uint8_t)((s + 128) >> 8);
        }
}

static const int CIRC[16][2] = {{0, -3}, {1, -3}, {2, -2}, {3, -1}, {3, 0},  {3, 1},   {2, 2},   {1, 3},
                                {0, 3},  {-1, 3}, {-2, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}};

static int nine_contiguous(unsigned m)
{
    unsigned d = m | (m << 16);
    unsigned r = d;
    for (int k = 1; k < 9; k++)
        r &= d >> k;
    return (r & 0xffffu) != 0;
}

int orc_fast9(const uint8_t *g, int w, int x, int y, int t)
{
    const int p = g[y * w + x];
    unsigned br = 0, dk = 0;
    for (int k = 0; k < 16; k++) {
        const int q = g[(y + CIRC[k][1]) * w + x + CIRC[k][0]];
        if (q > p + t)
            br |= 1u << k;
        if (q < p - t)
            dk |= 1u << k;
    }
    return nine_contiguous(br) || nine_contiguous(dk);
}

float orc_harris(const uint8_t *g, int w, int x, int y)
{
    int a = 0, b = 0, c = 0;
    for (int j = -3; j <= 3; j++)
        for (int i = -3; i <= 3; i++) {
            const uint8_t *p = g + (y + j) * w + x + i;
            const int ix = (p[1] - p[-1]) * 2 + (p[-w + 1] - p[-w - 1]) + (p[w + 1] - p[w - 1]);
            const int iy = (p[w] - p[-w]) * 2 + (p[w - 1] - p[-w - 1]) + (p[w + 1] - p[-w + 1]);
            a += ix * ix;
            b += iy * iy;
            c += ix * iy;
        }
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    const float sc = 1.f / (4 * 7 * 255.f);
    const float s4 = sc * sc * sc * sc;
    return (fa * fb - fc * fc - 0.04f * (fa + fb) * (fa + fb)) * s4;
}

typedef struct {
    float resp;
    int idx;
} cand_t;

static int cmp_cand(const void *A, const void *B)
{
    const cand_t *a = (const cand_t *)A, *b = (const cand_t *)B;
    if (a->resp != b->resp)
        return a->resp > b->resp ? -1 : 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/* Features of one pyramid level (grey, w x h) and its smoothed copy.  Outputs for at most `want`
 * keypoints in RASTER order: xy (level coordinates, ints), response, unit orientation vector,
 * 8-word descriptor.  Returns the number written. */
int orc_orb_level(const uint8_t *g, const uint8_t *blur, int w, int h, int want, int fast_t, int *xy, float *resp,
                  float *dir, uint32_t *desc)
{
    if (w <= 2 * ORB_EDGE || h <= 2 * ORB_EDGE || want <= 0)
        return 0;
    float *R = (float *)calloc((size_t)w * h, sizeof(float));
    uint8_t *is_c = (uint8_t *)calloc((size_t)w * h, 1);
    for (int y = ORB_EDGE; y < h - ORB_EDGE; y++)
        for (int x = ORB_EDGE; x < w - ORB_EDGE; x++)
            if (orc_fast9(g, w, x, y, fast_t)) {
                is_c[y * w + x] = 1;
                R[y * w + x] = orc_harris(g, w, x, y);
            }
    /* 3x3 non-maximum suppression among corners: strictly above the neighbours that come earlier in
     * raster order, at least equal to those that come later (a plateau keeps its first pixel) */
    cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)w * h);
    int nc = 0;
    for (int y = ORB_EDGE; y < h - ORB_EDGE; y++)
        for (int x = ORB_EDGE; x < w - ORB_EDGE; x++) {
            if (!is_c[y * w + x])
                continue;
            const float r = R[y * w + x];
            int keep = 1;
            for (int j = -1; j <= 1 && keep; j++)
                for (int i = -1; i <= 1; i++) {
                    if (!i && !j)
                        continue;
                    const int n = (y + j) * w + x + i;
                    if (!is_c[n])
                        continue;
                    const int earlier = j < 0 || (j == 0 && i < 0);
                    if (earlier ? !(r > R[n]) : !(r >= R[n])) {
                        keep = 0;
                        break;
                    }
                }
            if (keep) {
                cand[nc].resp = r;
                cand[nc].idx = y * w + x;
                nc++;
            }
        }
    /* the `want` strongest; ties at the cut go to the earlier pixel */
    int n = nc < want ? nc : want;
    if (nc > want)
        qsort(cand, nc, sizeof(cand_t), cmp_cand);
    /* back to raster order */
    int *sel = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++)
        sel[i] = cand[i].idx;
    for (int i = 1; i < n; i++) { /* insertion sort: n <= 500 */
        int v = sel[i], j = i - 1;
        while (j >= 0 && sel[j] > v) {
            sel[j + 1] = sel[j];
            j--;
        }
        sel[j + 1] = v;
    }
    int8_t pat[1024];
    orc_orb_pattern(pat);
    int umax[ORB_HALF_PATCH + 1];
    for (int v = 0; v <= ORB_HALF_PATCH; v++)
        umax[v] = (int)floor(sqrt((double)(ORB_HALF_PATCH * ORB_HALF_PATCH - v * v)));
    for (int i = 0; i < n; i++) {
        const int x = sel[i] % w, y = sel[i] / w;
        int m10 = 0, m01 = 0;
        for (int v = -ORB_HALF_PATCH; v <= ORB_HALF_PATCH; v++) {
            const int um = umax[v < 0 ? -v : v];
            for (int u = -um; u <= um; u++) {
                const int p = g[(y + v) * w + x + u];
                m10 += u * p;
                m01 += v * p;
            }
        }
        const float f10 = (float)m10, f01 = (float)m01;
        const float nrm = sqrtf(f10 * f10 + f01 * f01);
        const float cs = nrm > 0.f ? f10 / nrm : 1.f, sn = nrm > 0.f ? f01 / nrm : 0.f;
        uint32_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < 256; b++) {
            const float x1 = (float)pat[4 * b], y1 = (float)pat[4 * b + 1], x2 = (float)pat[4 * b + 2],
                        y2 = (float)pat[4 * b + 3];
            const int ax = (int)rintf(cs * x1 - sn * y1), ay = (int)rintf(sn * x1 + cs * y1);
            const int bx = (int)rintf(cs * x2 - sn * y2), by = (int)rintf(sn * x2 + cs * y2);
            const int pa = blur[(y + ay) * w + x + ax], pb = blur[(y + by) * w + x + bx];
            if (pa < pb)
                d[b >> 5] |= 1u << (b & 31);
        }
        xy[2 * i] = x;
        xy[2 * i + 1] = y;
        resp[i] = R[sel[i]];
        dir[2 * i] = cs;
        dir[2 * i + 1] = sn;
        memcpy(desc + 8 * i, d, sizeof(d));
    }
    free(R);
    free(is_c);
    free(cand);
    free(sel);
    return n;
}

/* The whole extractor: image (h x w x c, BGR or grey) -> up to n_features keypoints over 3 octaves.
 * xy in level-0 pixels (x * 2^octave), octave, response, dir (unit vector), desc (8 words each). */
int orc_orb_extract(const uint8_t *img, int w, int h, int c, int n_features, int fast_t, float *xy, int *octave,
                    float *resp, float *dir, uint32_t *desc)
{
    uint8_t *lvl[3], *blur;
    int ws[3], hs[3];
    orc_pyr_sizes(w, h, 3, ws, hs);
    lvl[0] = (uint8_t *)malloc((size_t)w * h);
    orc_bgr_to_gray(img, w, h, c, lvl[0]);
    for (int l = 1; l < 3; l++) {
        lvl[l] = (uint8_t *)malloc((size_t)ws[l] * hs[l]);
        orc_pyr_down(lvl[l - 1], ws[l - 1], hs[l - 1], 1, lvl[l]);
    }
    blur = (uint8_t *)malloc((size_t)w * h);
    /* budget per octave: proportional to 1 / 2^l as upstream splits by 1 / scale (286 / 143 / 71 of 500) */
    int want[3];
    want[0] = (int)(n_features * 4.0 / 7.0 + 0.5);
    want[1] = (int)(n_features * 2.0 / 7.0 + 0.5);
    want[2] = n_features - want[0] - want[1];
    int total = 0;
    int *lxy = (int *)malloc(sizeof(int) * 2 * (n_features + 1));
    for (int l = 0; l < 3; l++) {
        orc_blur5(lvl[l], ws[l], hs[l], blur);
        const int k = orc_orb_level(lvl[l], blur, ws[l], hs[l], want[l], fast_t, lxy, resp + total, dir + 2 * total,
                                    desc + 8 * total);
        for (int i = 0; i < k; i++) {
            xy[2 * (total + i)] = (float)(lxy[2 * i] << l);
            xy[2 * (total + i) + 1] = (float)(lxy[2 * i + 1] << l);
            octave[total + i] = l;
        }
        total += k;
    }
    for (int l = 0; l < 3; l++)
        free(lvl[l]);
    free(blur);
    free(lxy);
    return total;
}
