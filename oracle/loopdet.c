/*
 * loopdet.c -- descriptor matching behind the loop-closure detector of
 * visualSLAM::checkLoopDetectorStatus (src/optimizationStuff.cpp:49-64 ->
 * DLoopDetector::detectLoop, include/TemplatedLoopDetector.h:696-861).
 *
 * TEST INFRASTRUCTURE (see svo_oracle.h).  PARITY UNPINNED.  Stated deviation: upstream scores
 * database entries with DBoW2's bag-of-words L1 score over the ORB vocabulary orb_voc00.yml.gz,
 * which was stripped from the checkout together with DBoW2/DLib.  Here the similarity of the query
 * to an entry is the fraction of query descriptors whose nearest descriptor in the entry lies within
 * a Hamming radius (a direct, vocabulary-free measure on the same 256-bit descriptors); the detector
 * logic on top of the scores (normalisation by the previous frame, alpha cut, islands, temporal
 * window, geometric check) follows the vendored header line by line (oracle/loop_detector.py).
 */
#include <stddef.h>
#include <stdint.h>

#include "svo_oracle.h"

int orc_hamming256(const uint32_t *a, const uint32_t *b)
{
    int d = 0;
    for (int k = 0; k < 8; k++)
        d += __builtin_popcount(a[k] ^ b[k]);
    return d;
}

/* counts[e] = number of query descriptors whose nearest descriptor in entry e is within
 * hamming_thr.  db: n_entries blocks of `stride` descriptors (8 words each), db_n[e] of them valid. */
void orc_lc_scores(const uint32_t *q, int nq, const uint32_t *db, const int *db_n, int stride, int n_entries,
                   int hamming_thr, int *counts)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int e = 0; e < n_entries; e++) {
        const uint32_t *E = db + (size_t)e * stride * 8;
        int c = 0;
        for (int i = 0; i < nq; i++) {
            int best = 1 << 30;
            for (int j = 0; j < db_n[e]; j++) {
                const int d = orc_hamming256(q + 8 * i, E + 8 * j);
                if (d < best)
                    best = d;
            }
            c += best <= hamming_thr;
        }
        counts[e] = c;
    }
}

/* getMatches_neighratio's search (include/TemplatedLoopDetector.h:1255-1291): for every A[i] the
 * nearest B (first one on ties), its distance d1 and the second-best distance d2 (1e9 when absent) */
void orc_lc_nearest2(const uint32_t *A, int na, const uint32_t *B, int nb, int *best_j, int *d1, int *d2)
{
    for (int i = 0; i < na; i++) {
        int bj = -1, b1 = 1000000000, b2 = 1000000000;
        for (int j = 0; j < nb; j++) {
            const int d = orc_hamming256(A + 8 * i, B + 8 * j);
            if (d < b1) {
                bj = j;
                b2 = b1;
                b1 = d;
            } else if (d < b2)
                b2 = d;
        }
        best_j[i] = bj;
        d1[i] = b1;
        d2[i] = b2;
    }
}
