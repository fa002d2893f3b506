/*
 * oracle/anms.c -- CPU restatement of adaptiveNonMaximalSuppresion
 * (TEST INFRASTRUCTURE; see svo_oracle.h.)
 *
 * Reference: /root/reference/src/ANMS.cpp:18-67.  This function is fully specified in the
 * reference tree (no third-party arithmetic): sort by response descending; radius_i =
 * min over the earlier keypoints whose response exceeds 1.11f * response_i of the
 * distance to them (double, cv::norm of a Point2f difference); decision radius =
 * radiiSorted[numToKeep] (0-based, descending); keep every keypoint with radius >=
 * decision radius, in response order.
 *
 * Deviations: std::sort is unstable, so ties in response have no
 * defined order upstream -- here ties keep input order (stable); the reference early-outs
 * on size < numToKeep and reads radiiSorted[numToKeep] out of bounds when size ==
 * numToKeep -- here size <= numToKeep returns everything (in sorted order).
 */
#include "svo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

typedef struct {
    float resp;
    int idx;
} kp_t;

static int cmp_resp_desc(const void *a, const void *b)
{
    const kp_t *x = (const kp_t *)a, *y = (const kp_t *)b;
    if (x->resp > y->resp)
        return -1;
    if (x->resp < y->resp)
        return 1;
    return x->idx - y->idx; /* stable */
}
static int cmp_double_desc(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return x > y ? -1 : (x < y ? 1 : 0);
}

int orc_anms(const float *xy, const float *response, int n, int num_to_keep, int *out_idx, double *out_radii)
{
    if (n <= 0)
        return 0;
    kp_t *kp = (kp_t *)malloc(sizeof(kp_t) * n);
    for (int i = 0; i < n; i++) {
        kp[i].resp = response[i];
        kp[i].idx = i;
    }
    qsort(kp, n, sizeof(kp_t), cmp_resp_desc);
    if (n <= num_to_keep) {
        for (int i = 0; i < n; i++)
            out_idx[i] = kp[i].idx;
        free(kp);
        return n;
    }
    double *radii = (double *)malloc(sizeof(double) * n), *sorted = (double *)malloc(sizeof(double) * n);
    const float robust = 1.11f;
    /* every keypoint's suppression radius is independent (a minimum: order-free), so the loop is
     * shared among OpenMP threads */
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        const float r = kp[i].resp * robust;
        double radius = DBL_MAX;
        for (int j = 0; j < i && kp[j].resp > r; j++) {
            float dx = xy[2 * kp[i].idx] - xy[2 * kp[j].idx], dy = xy[2 * kp[i].idx + 1] - xy[2 * kp[j].idx + 1];
            double d = sqrt((double)dx * dx + (double)dy * dy);
            if (d < radius)
                radius = d;
        }
        radii[i] = radius;
        sorted[i] = radius;
    }
    qsort(sorted, n, sizeof(double), cmp_double_desc);
    const double decision = sorted[num_to_keep];
    int k = 0;
    for (int i = 0; i < n; i++) {
        if (out_radii)
            out_radii[i] = radii[i];
        if (radii[i] >= decision)
            out_idx[k++] = kp[i].idx;
    }
    free(kp);
    free(radii);
    free(sorted);
    return k;
}
