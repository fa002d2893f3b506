/*
 * bow.c -- the bag-of-words side of the loop detector: DBoW2's TemplatedVocabulary<FORB> / TemplatedDatabase as
 * visualSLAM uses them through DLoopDetector (include/visualSLAM.h:115-137: OrbVocabulary loaded from orb_voc00.yml.gz,
 * OrbLoopDetector(voc, params) with use_nss, alpha 0.9, k 1, GEOM_DI, di_levels 2; detectLoop,
 * include/TemplatedLoopDetector.h:696-861) and as the reference's own trainer builds one
 * (src/bagOfWordsDetector.cpp:46-56: OrbVocabulary(k = 9, L = 6, TF_IDF, L1_NORM).create(features)).
 *
 * TEST INFRASTRUCTURE (see svo_oracle.h).  PARITY UNPINNED: DBoW2 / DLib are un-vendored third-party code, absent from
 * /root/reference (stripped together with the vocabulary files, .MISSING_LARGE_BLOBS) and from this image.  What is
 * restated here is DBoW2's published algorithm (Galvez-Lopez & Tardos, "Bags of binary words for fast place recognition
 * in image sequences", T-RO 2012, and the library's TemplatedVocabulary.h / TemplatedDatabase.h / ScoringObject.cpp /
 * FORB.cpp as recalled -- none of it is in the checkout):
 *   create            hierarchical k-medians++ in Hamming space: HKmeansStep per node (kmeans++ seeding with probability
 *                     proportional to F::distance to the nearest centre -- for FORB the Hamming distance itself, upstream
 *                     does not square it), Lloyd steps until the associations stop changing, the mean of a cluster = the bitwise majority (FORB::meanValue: bit set iff
 *                     at least ceil(n / 2) members have it), a node with at most k descriptors gets one child per
 *                     descriptor, recursion while level < L and the child holds more than one descriptor; node ids in
 *                     creation order (all children of a node consecutively, then depth first), words = leaves in id order
 *   setNodeWeights    TF_IDF: idf = log(N_images / N_images_with_the_word)
 *   transform         per feature: descend by the nearest child (first minimum), word id + weight + the ancestor `levelsup`
 *                     levels above the leaf; BowVector: weights added per word in feature order, features with weight 0
 *                     skipped, then L1-normalised (L1_NORM scoring "must normalise"); map order = ascending word id
 *   score / query     L1: sum over the common words, ascending, of |v - w| - |v| - |w|; score = -sum / 2
 *   direct index      FeatureVector: node id at level L - di_levels -> feature indices; isGeometricallyConsistent_DI matches
 *                     only features under a common node (include/TemplatedLoopDetector.h:1005-1087)
 * Stated deviations: (1) DUtils::Random is replaced by the counter-based generator of the RANSAC stages, keyed by the node's
 * path in the tree, so that a GPU training level by level draws what this depth-first recursion draws; the kmeans++ cut is
 * an integer in [1, sum of distances] (upstream: a double in (0, sum]); (2) a cluster that loses all its members keeps its
 * centre (upstream would take the mean of an empty set); (3) Lloyd steps are capped at ORC_VOC_MAX_LLOYD (upstream: no cap);
 * (4) log comes from include/svo_math.h (shared with the HIP library, ~1 ulp).
 */
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"

#define ORC_VOC_MAX_LLOYD 64
#define ORC_VOC_MAX_K 12

struct orc_voc {
    int k, L;
    int n_nodes, cap_nodes, n_words;
    int *parent, *first_child, *n_children, *word_id, *level;
    uint32_t *desc; /* 8 words per node */
    double *weight;
    int *word_node; /* word id -> node id */
};

static int hamming8(const uint32_t *a, const uint32_t *b)
{
    int d = 0;
    for (int k = 0; k < 8; k++)
        d += __builtin_popcount(a[k] ^ b[k]);
    return d;
}

static int voc_new_node(orc_voc *v, int parent, int level, const uint32_t *d)
{
    if (v->n_nodes == v->cap_nodes) {
        const int cap = v->cap_nodes ? 2 * v->cap_nodes : 1024;
        v->parent = realloc(v->parent, sizeof(int) * cap);
        v->first_child = realloc(v->first_child, sizeof(int) * cap);
        v->n_children = realloc(v->n_children, sizeof(int) * cap);
        v->word_id = realloc(v->word_id, sizeof(int) * cap);
        v->level = realloc(v->level, sizeof(int) * cap);
        v->desc = realloc(v->desc, sizeof(uint32_t) * 8 * cap);
        v->weight = realloc(v->weight, sizeof(double) * cap);
        v->cap_nodes = cap;
    }
    const int id = v->n_nodes++;
    v->parent[id] = parent;
    v->first_child[id] = -1;
    v->n_children[id] = 0;
    v->word_id[id] = -1;
    v->level[id] = level;
    v->weight[id] = 0;
    if (d)
        memcpy(v->desc + 8 * (size_t)id, d, 32);
    else
        memset(v->desc + 8 * (size_t)id, 0, 32);
    if (parent >= 0) {
        if (v->n_children[parent] == 0)
            v->first_child[parent] = id;
        v->n_children[parent]++;
    }
    return id;
}

/* FORB::meanValue: bit set iff at least ceil(n / 2) of the members have it */
static void majority(const uint32_t *D, const int *members, int n, uint32_t *out)
{
    int cnt[256];
    memset(cnt, 0, sizeof(cnt));
    for (int m = 0; m < n; m++) {
        const uint32_t *d = D + 8 * (size_t)members[m];
        for (int b = 0; b < 256; b++)
            cnt[b] += (d[b >> 5] >> (b & 31)) & 1u;
    }
    const int need = n / 2 + (n & 1);
    memset(out, 0, 32);
    for (int b = 0; b < 256; b++)
        if (cnt[b] >= need)
            out[b >> 5] |= 1u << (b & 31);
}

/* One node's clustering (HKmeansStep's body).  idx: the node's n descriptors in order.  Fills centres (nc x 8 words) and
 * assoc[i] in 0..nc-1; returns nc.  key: the node's path key (root 1, child c of K: 16 K + c + 1).                        */
int orc_voc_cluster(const uint32_t *D, const int *idx, int n, int k, uint64_t seed, uint64_t key, uint32_t *centres,
                    int *assoc, int *lloyd_steps)
{
    if (lloyd_steps)
        *lloyd_steps = 0;
    if (n <= k) { /* trivial case: one cluster per feature */
        for (int i = 0; i < n; i++) {
            memcpy(centres + 8 * i, D + 8 * (size_t)idx[i], 32);
            assoc[i] = i;
        }
        return n;
    }
    /* initiateClustersKMpp */
    int *min_d = malloc(sizeof(int) * (size_t)n);
    uint32_t draw = 0;
    int nc = 0;
    {
        const int f = (int)(orc_rng_u32(seed ^ (key * 0x9E3779B97F4A7C15ull), 0, draw++) % (uint32_t)n);
        memcpy(centres, D + 8 * (size_t)idx[f], 32);
        nc = 1;
        for (int i = 0; i < n; i++)
            min_d[i] = hamming8(D + 8 * (size_t)idx[i], centres);
    }
    while (nc < k) {
        long long sum = 0;
        for (int i = 0; i < n; i++)
            sum += min_d[i];
        if (sum <= 0)
            break; /* every descriptor coincides with a centre: fewer than k clusters */
        /* two 32-bit draws make one 64-bit number: sum < 2^31 * 256 */
        const uint64_t hi = orc_rng_u32(seed ^ (key * 0x9E3779B97F4A7C15ull), 0, draw++);
        const uint64_t lo = orc_rng_u32(seed ^ (key * 0x9E3779B97F4A7C15ull), 0, draw++);
        const long long cut = 1 + (long long)(((hi << 32) | lo) % (uint64_t)sum);
        long long up = 0;
        int f = n - 1;
        for (int i = 0; i < n; i++) {
            up += min_d[i];
            if (up >= cut) {
                f = i;
                break;
            }
        }
        memcpy(centres + 8 * nc, D + 8 * (size_t)idx[f], 32);
        for (int i = 0; i < n; i++) {
            const int d = hamming8(D + 8 * (size_t)idx[i], centres + 8 * nc);
            if (d < min_d[i])
                min_d[i] = d;
        }
        nc++;
    }
    free(min_d);
    /* Lloyd: associate, recompute the centres, until the associations stop changing */
    int *members = malloc(sizeof(int) * (size_t)n);
    int first = 1, steps = 0;
    for (;;) {
        if (!first) {
            for (int c = 0; c < nc; c++) {
                int m = 0;
                for (int i = 0; i < n; i++)
                    if (assoc[i] == c)
                        members[m++] = idx[i];
                if (m > 0) /* deviation (2): an empty cluster keeps its centre */
                    majority(D, members, m, centres + 8 * c);
            }
        }
        int changed = 0;
        for (int i = 0; i < n; i++) {
            const uint32_t *d = D + 8 * (size_t)idx[i];
            int best = hamming8(d, centres), bc = 0;
            for (int c = 1; c < nc; c++) {
                const int dd = hamming8(d, centres + 8 * c);
                if (dd < best) {
                    best = dd;
                    bc = c;
                }
            }
            if (first || assoc[i] != bc)
                changed = 1;
            assoc[i] = bc;
        }
        if (first) {
            first = 0;
            steps = 1;
            continue; /* upstream: the first association is never "converged" */
        }
        if (!changed || steps >= ORC_VOC_MAX_LLOYD)
            break;
        steps++;
    }
    free(members);
    if (lloyd_steps)
        *lloyd_steps = steps;
    return nc;
}

static void hkmeans_step(orc_voc *v, const uint32_t *D, int parent, const int *idx, int n, int level, uint64_t seed,
                         uint64_t key)
{
    if (n <= 0)
        return;
    uint32_t centres[8 * ORC_VOC_MAX_K];
    int *assoc = malloc(sizeof(int) * (size_t)n);
    const int nc = orc_voc_cluster(D, idx, n, v->k, seed, key, centres, assoc, 0);
    int child[ORC_VOC_MAX_K];
    for (int c = 0; c < nc; c++)
        child[c] = voc_new_node(v, parent, level, centres + 8 * c);
    if (level < v->L) {
        int *sub = malloc(sizeof(int) * (size_t)n);
        for (int c = 0; c < nc; c++) {
            int m = 0;
            for (int i = 0; i < n; i++)
                if (assoc[i] == c)
                    sub[m++] = idx[i];
            if (m > 1)
                hkmeans_step(v, D, child[c], sub, m, level + 1, seed, key * 16 + (uint64_t)c + 1);
        }
        free(sub);
    }
    free(assoc);
}

static void voc_finish_words(orc_voc *v)
{
    v->n_words = 0;
    free(v->word_node);
    v->word_node = malloc(sizeof(int) * (size_t)(v->n_nodes > 0 ? v->n_nodes : 1));
    for (int i = 1; i < v->n_nodes; i++) /* createWords: the leaves in node order */
        if (v->n_children[i] == 0) {
            v->word_id[i] = v->n_words;
            v->word_node[v->n_words++] = i;
        }
}

/* descend by the nearest child, first minimum (TemplatedVocabulary::transform) */
static int voc_descend(const orc_voc *v, const uint32_t *d, int nid_level, int *nid)
{
    int cur = 0, level = 0;
    if (nid)
        *nid = 0;
    while (v->n_children[cur] > 0) {
        level++;
        const int c0 = v->first_child[cur];
        int best = hamming8(d, v->desc + 8 * (size_t)c0), bi = c0;
        for (int c = 1; c < v->n_children[cur]; c++) {
            const int dd = hamming8(d, v->desc + 8 * (size_t)(c0 + c));
            if (dd < best) {
                best = dd;
                bi = c0 + c;
            }
        }
        cur = bi;
        if (nid && level == nid_level)
            *nid = cur;
    }
    return cur;
}

orc_voc *orc_voc_train(const uint32_t *desc, const int *img_off, int n_images, int k, int L, uint64_t seed)
{
    if (k < 2 || k > ORC_VOC_MAX_K || L < 1 || L > 10 || n_images < 1)
        return 0;
    orc_voc *v = calloc(1, sizeof(*v));
    v->k = k;
    v->L = L;
    const int n = img_off[n_images];
    voc_new_node(v, -1, 0, 0); /* root */
    int *idx = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++)
        idx[i] = i;
    hkmeans_step(v, desc, 0, idx, n, 1, seed, 1);
    free(idx);
    voc_finish_words(v);
    /* setNodeWeights, TF_IDF: idf = log(N / Ni), Ni = images that contain the word */
    int *ni = calloc((size_t)(v->n_words > 0 ? v->n_words : 1), sizeof(int));
    int *last = malloc(sizeof(int) * (size_t)(v->n_words > 0 ? v->n_words : 1));
    for (int w = 0; w < v->n_words; w++)
        last[w] = -1;
    for (int im = 0; im < n_images; im++)
        for (int f = img_off[im]; f < img_off[im + 1]; f++) {
            const int w = v->word_id[voc_descend(v, desc + 8 * (size_t)f, 0, 0)];
            if (last[w] != im) {
                last[w] = im;
                ni[w]++;
            }
        }
    for (int w = 0; w < v->n_words; w++)
        if (ni[w] > 0)
            v->weight[v->word_node[w]] = svo_log((double)n_images / (double)ni[w]);
    free(ni);
    free(last);
    return v;
}

/* a vocabulary from its arrays (a file that was loaded, or the GPU's training result): nodes in id order, node 0 = root,
 * the children of a node consecutive */
orc_voc *orc_voc_import(int k, int L, int n_nodes, const int *parent, const uint32_t *desc, const double *weight)
{
    orc_voc *v = calloc(1, sizeof(*v));
    v->k = k;
    v->L = L;
    voc_new_node(v, -1, 0, 0);
    for (int i = 1; i < n_nodes; i++) {
        if (parent[i] < 0 || parent[i] >= i) {
            orc_voc_free(v);
            return 0;
        }
        const int id = voc_new_node(v, parent[i], v->level[parent[i]] + 1, desc + 8 * (size_t)i);
        v->weight[id] = weight[i];
        if (v->first_child[parent[i]] + v->n_children[parent[i]] - 1 != id) { /* children must be consecutive */
            orc_voc_free(v);
            return 0;
        }
    }
    voc_finish_words(v);
    return v;
}

void orc_voc_free(orc_voc *v)
{
    if (!v)
        return;
    free(v->parent);
    free(v->first_child);
    free(v->n_children);
    free(v->word_id);
    free(v->level);
    free(v->desc);
    free(v->weight);
    free(v->word_node);
    free(v);
}

int orc_voc_nodes(const orc_voc *v) { return v->n_nodes; }
int orc_voc_words(const orc_voc *v) { return v->n_words; }
int orc_voc_k(const orc_voc *v) { return v->k; }
int orc_voc_levels(const orc_voc *v) { return v->L; }

void orc_voc_export(const orc_voc *v, int *parent, int *first_child, int *n_children, uint32_t *desc, double *weight,
                    int *word_id)
{
    memcpy(parent, v->parent, sizeof(int) * (size_t)v->n_nodes);
    memcpy(first_child, v->first_child, sizeof(int) * (size_t)v->n_nodes);
    memcpy(n_children, v->n_children, sizeof(int) * (size_t)v->n_nodes);
    memcpy(desc, v->desc, 32 * (size_t)v->n_nodes);
    memcpy(weight, v->weight, sizeof(double) * (size_t)v->n_nodes);
    memcpy(word_id, v->word_id, sizeof(int) * (size_t)v->n_nodes);
}

/* per feature: word id, the word's weight, the ancestor at level L - levelsup (the direct index's node; 0 = root when
 * L - levelsup <= 0).  A leaf that sits above that level (a short branch) is its own ancestor there... upstream records the
 * node only when the descent passes level L - levelsup, so a shorter branch leaves the root (0): restated as that.        */
void orc_voc_transform(const orc_voc *v, const uint32_t *desc, int n, int levelsup, int *word, double *weight, int *node)
{
    const int nid_level = v->L - levelsup;
    for (int i = 0; i < n; i++) {
        int nid = 0;
        const int leaf = voc_descend(v, desc + 8 * (size_t)i, nid_level, &nid);
        word[i] = v->word_id[leaf];
        weight[i] = v->weight[leaf];
        if (node)
            node[i] = nid_level <= 0 ? 0 : nid;
    }
}

/* BowVector of an image (TF_IDF weighting, L1_NORM scoring): weights added per word in feature order, zero-weight words
 * skipped, L1-normalised; out in ascending word order.  Returns the number of words.                                      */
int orc_bow_vector(const int *word, const double *weight, int n, int *out_words, double *out_vals)
{
    /* stable insertion into a sorted list: n is a few hundred */
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (!(weight[i] > 0))
            continue;
        int lo = 0, hi = m;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (out_words[mid] < word[i])
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < m && out_words[lo] == word[i])
            out_vals[lo] += weight[i];
        else {
            memmove(out_words + lo + 1, out_words + lo, sizeof(int) * (size_t)(m - lo));
            memmove(out_vals + lo + 1, out_vals + lo, sizeof(double) * (size_t)(m - lo));
            out_words[lo] = word[i];
            out_vals[lo] = weight[i];
            m++;
        }
    }
    double norm = 0;
    for (int i = 0; i < m; i++)
        norm += out_vals[i] < 0 ? -out_vals[i] : out_vals[i];
    if (norm > 0)
        for (int i = 0; i < m; i++)
            out_vals[i] /= norm;
    return m;
}

/* L1Scoring: the raw sum over the common words (ascending) of |v - w| - |v| - |w|; *common = their number.
 * score = -sum / 2 (TemplatedDatabase::queryL1 accumulates exactly these terms per entry, word by word).                  */
double orc_bow_l1_sum(const int *w1, const double *v1, int n1, const int *w2, const double *v2, int n2, int *common)
{
    double s = 0;
    int i = 0, j = 0, c = 0;
    while (i < n1 && j < n2) {
        if (w1[i] == w2[j]) {
            const double a = v1[i], b = v2[j];
            const double d = a - b;
            s += (d < 0 ? -d : d) - (a < 0 ? -a : a) - (b < 0 ? -b : b);
            c++;
            i++;
            j++;
        } else if (w1[i] < w2[j])
            i++;
        else
            j++;
    }
    if (common)
        *common = c;
    return s;
}

/* the query against every entry of a database stored as rows of `stride` (word, value) pairs */
void orc_bow_query(const int *qw, const double *qv, int nq, const int *db_w, const double *db_v, const int *db_n, int stride,
                   int n_entries, double *sums, int *common)
{
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n_entries; e++)
        sums[e] = orc_bow_l1_sum(qw, qv, nq, db_w + (size_t)e * stride, db_v + (size_t)e * stride, db_n[e], common + e);
}

/* isGeometricallyConsistent_DI's matching (include/TemplatedLoopDetector.h:1005-1054 + getMatches_neighratio :1255-1316):
 * for every direct-index node both images have, ascending, the neighbour-ratio matches between the OLD image's features
 * under the node (A) and the current image's (B).  node_*[i] < 0: the feature is not in the direct index (weight 0).
 * Returns the number of pairs (i_old[], i_cur[]).                                                                        */
int orc_di_matches(const uint32_t *A, const int *node_a, int na, const uint32_t *B, const int *node_b, int nb,
                   double max_ratio, int *i_old, int *i_cur)
{
    int n_out = 0;
    /* the distinct nodes of A, ascending */
    int *nodes = malloc(sizeof(int) * (size_t)(na > 0 ? na : 1));
    int nn = 0;
    for (int i = 0; i < na; i++) {
        if (node_a[i] < 0)
            continue;
        int lo = 0;
        while (lo < nn && nodes[lo] < node_a[i])
            lo++;
        if (lo < nn && nodes[lo] == node_a[i])
            continue;
        memmove(nodes + lo + 1, nodes + lo, sizeof(int) * (size_t)(nn - lo));
        nodes[lo] = node_a[i];
        nn++;
    }
    for (int t = 0; t < nn; t++) {
        const int node = nodes[t];
        const int base = n_out; /* matches of THIS call of getMatches_neighratio */
        for (int i = 0; i < na; i++) {
            if (node_a[i] != node)
                continue;
            int best_j = -1;
            double b1 = 1e9, b2 = 1e9;
            for (int j = 0; j < nb; j++) {
                if (node_b[j] != node)
                    continue;
                const double d = hamming8(A + 8 * (size_t)i, B + 8 * (size_t)j);
                if (d < b1) {
                    best_j = j;
                    b2 = b1;
                    b1 = d;
                } else if (d < b2)
                    b2 = d;
            }
            if (best_j < 0)
                continue; /* the node is not common */
            if (b1 / b2 <= max_ratio) {
                int at = -1;
                for (int m = base; m < n_out; m++)
                    if (i_cur[m] == best_j) {
                        at = m;
                        break;
                    }
                if (at < 0) {
                    i_cur[n_out] = best_j;
                    i_old[n_out] = i;
                    n_out++;
                } else {
                    const double d = hamming8(A + 8 * (size_t)i_old[at], B + 8 * (size_t)best_j);
                    if (b1 < d)
                        i_old[at] = i;
                }
            }
        }
    }
    free(nodes);
    return n_out;
}
