/*
 * sor.c -- CPU restatement of visualSLAM::SORcloud (src/rosFuncs.cpp:9-39).
 *
 * TEST INFRASTRUCTURE (see svo_oracle.h).  PARITY UNPINNED: the filter itself is
 * pcl::StatisticalOutlierRemoval<PointXYZRGB> (PCL is un-vendored and absent); what follows is
 * the published algorithm of its applyFilterIndices:
 *   1. for every point, the mean_k nearest OTHER points (PCL asks the kd-tree for mean_k + 1
 *      neighbours, sorted by distance, and skips the first, the query point itself);
 *      distance_i = float( sum_k sqrt(float squared distance) / mean_k ), the sum in double;
 *   2. mean and standard deviation of the distance_i over all points:
 *         sum += d; sq_sum += d * d (the product in float);  mean = sum / n;
 *         variance = (sq_sum - sum * sum / n) / (n - 1);  stddev = sqrt(variance);
 *   3. keep point i iff distance_i <= mean + stddev_mul * stddev, order preserved.
 * The reference drops points with -z > 500 first (src/rosFuncs.cpp:12) and uses mean_k = 200,
 * stddev_mul = 0.01 (:21-22).
 * Stated deviation: with fewer than mean_k + 1 points PCL reads neighbour slots the search did
 * not fill; here the mean runs over the n - 1 points that exist (0 for a single point).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"

static int cmp_float_asc(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

int orc_sor_filter(const float *xyz, const float *color, int n, int mean_k, double stddev_mul, float z_limit,
                   float *xyz_out, float *color_out, float *mean_dist_out)
{
    if (n <= 0 || mean_k <= 0)
        return 0;
    /* src/rosFuncs.cpp:11-14: the far-point pre-filter */
    int *src = (int *)malloc(sizeof(int) * n);
    int m = 0;
    for (int i = 0; i < n; i++)
        if (!(z_limit > 0 && -1.f * xyz[3 * i + 2] > z_limit))
            src[m++] = i;
    float *dist = (float *)malloc(sizeof(float) * (m > 0 ? m : 1));
    const int kk = mean_k < m - 1 ? mean_k : m - 1;
#pragma omp parallel
    {
        float *d2 = (float *)malloc(sizeof(float) * (m > 0 ? m : 1));
#pragma omp for schedule(dynamic, 16)
        for (int a = 0; a < m; a++) {
            const float *p = xyz + 3 * src[a];
            int c = 0;
            for (int b = 0; b < m; b++) {
                if (b == a)
                    continue; /* the query point itself: PCL's skipped first neighbour */
                const float *q = xyz + 3 * src[b];
                const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
                d2[c++] = dx * dx + dy * dy + dz * dz;
            }
            qsort(d2, c, sizeof(float), cmp_float_asc);
            double sum = 0;
            for (int k = 0; k < kk; k++)
                sum += sqrtf(d2[k]);
            dist[a] = kk > 0 ? (float)(sum / kk) : 0.f;
        }
        free(d2);
    }
    double sum = 0, sq_sum = 0;
    for (int a = 0; a < m; a++) {
        sum += dist[a];
        sq_sum += dist[a] * dist[a];
    }
    double thr = DBL_MAX;
    if (m > 1) {
        const double mean = sum / m;
        double variance = (sq_sum - sum * sum / m) / (m - 1);
        if (variance < 0)
            variance = 0;
        thr = mean + stddev_mul * sqrt(variance);
    }
    int k = 0;
    for (int a = 0; a < m; a++) {
        if (mean_dist_out)
            mean_dist_out[a] = dist[a];
        if (dist[a] <= thr) {
            memcpy(xyz_out + 3 * k, xyz + 3 * src[a], 3 * sizeof(float));
            if (color && color_out)
                memcpy(color_out + 3 * k, color + 3 * src[a], 3 * sizeof(float));
            k++;
        }
    }
    free(src);
    free(dist);
    return k;
}
