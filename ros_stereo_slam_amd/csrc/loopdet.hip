// loopdet.hip -- the loop-closure detector on gfx950.
//
// Replaces visualSLAM::checkLoopDetectorStatus (src/optimizationStuff.cpp:49-64): features by
// orb.hip, then DLoopDetector::detectLoop (include/TemplatedLoopDetector.h:696-861) with the
// parameters visualSLAM sets (include/visualSLAM.h:120-127) on top of Parameters::set(1)
// (:552-568).  The control flow -- database query below `dislocal`, normalisation by the score
// against the previous frame, alpha cut, islands (:875-951), temporal window (:966-1003), geometric
// check by neighbour-ratio matches + a RANSAC fundamental matrix (:1101-1160, :1255-1316) -- follows
// the vendored header; it is host code, a few dozen integers per frame.
//
// Stated deviation (also oracle/loopdet.c): DBoW2, DLib and the vocabulary orb_voc00.yml.gz were
// stripped from the reference checkout, so the bag-of-words score is replaced by a vocabulary-free
// similarity on the same 256-bit descriptors -- the fraction of query descriptors with a neighbour
// within a Hamming radius in the entry -- and GEOM_DI's direct index by the header's own exhaustive
// neighbour-ratio matching.  The data-parallel part is that similarity: every database entry
// against the query = N_db x 500 x 500 Hamming distances per frame (popcounts of XORs, entry
// descriptors staged in LDS and read as broadcasts), one workgroup per entry.
#include <algorithm>
#include <vector>

#include "svo_internal.h"

namespace {

__device__ __forceinline__ int hamming256(const uint32_t (&a)[8], const uint32_t *b)
{
    int d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        d += __popc(a[k] ^ b[k]);
    return d;
}

// counts[e] = number of query descriptors whose nearest descriptor of entry e is within thr
__global__ __launch_bounds__(256) void lc_score_kernel(const uint32_t *__restrict__ q, const int *__restrict__ d_nq,
                                                       const uint32_t *__restrict__ db, const int *__restrict__ db_n,
                                                       int stride, int thr, int *__restrict__ counts)
{
    extern __shared__ uint32_t s_e[];  // stride * 8 words
    __shared__ int s_red[4];
    const int e = blockIdx.x, t = threadIdx.x;
    const int ne = db_n[e], nq = *d_nq;
    const uint32_t *E = db + (size_t)e * stride * 8;
    for (int i = t; i < ne * 8; i += 256)
        s_e[i] = E[i];
    __syncthreads();
    int c = 0;
    for (int i = t; i < nq; i += 256) {
        uint32_t a[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            a[k] = q[(size_t)8 * i + k];
        int best = 1 << 30;
        for (int j = 0; j < ne; j++) {
            const int d = hamming256(a, s_e + 8 * j);
            best = d < best ? d : best;
        }
        c += best <= thr ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        c += __shfl_xor(c, off, 64);
    if ((t & 63) == 0)
        s_red[t >> 6] = c;
    __syncthreads();
    if (t == 0)
        counts[e] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// getMatches_neighratio's search (:1268-1291): nearest B (first on ties), its distance, the second best
__global__ __launch_bounds__(256) void lc_nearest2_kernel(const uint32_t *__restrict__ A, int na,
                                                          const uint32_t *__restrict__ B, const int *__restrict__ d_nb,
                                                          int *__restrict__ best_j, int *__restrict__ d1,
                                                          int *__restrict__ d2)
{
    extern __shared__ uint32_t s_b[];
    const int nb = *d_nb;
    for (int i = threadIdx.x; i < nb * 8; i += 256)
        s_b[i] = B[i];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= na)
        return;
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
        a[k] = A[(size_t)8 * i + k];
    int bj = -1, b1 = 1000000000, b2 = 1000000000;
    for (int j = 0; j < nb; j++) {
        const int d = hamming256(a, s_b + 8 * j);
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

struct Island {  // tIsland, :268-330
    int first, last;
    double score;
    int best_entry;
    double best_score;
};
struct Result {
    int id;
    double score;
};

}  // namespace

struct svo_lc {
    svo_ctx *ctx = nullptr;
    svo_lc_params prm;
    svo_orb *orb = nullptr;
    int w = 0, h = 0, c = 0, nf = 0, capacity = 0;
    DevBuf db_desc, db_xy, db_n, q, counts, nn, img;
    std::vector<int> n_host;                 // descriptors per entry
    std::vector<std::vector<float>> xy_host; // m_image_keys (positions only)
    int window_n = 0, window_first = 0, window_last = 0, window_query = -1;
    bool have_last = false;                  // m_last_bowvec's stand-in: the previous entry is the reference
};

extern "C" {

void svo_lc_default_params(svo_lc_params *p)
{
    if (!p)
        return;
    p->n_features = 500;          // cv::ORB::create() default
    p->fast_threshold = 20;
    p->hamming_threshold = 64;
    p->max_entries = 8192;
    p->use_nss = 1;               // include/visualSLAM.h:122
    p->alpha = 0.9f;              // :123
    p->k = 1;                     // :124
    p->dislocal = 20;             // Parameters::set(1), include/TemplatedLoopDetector.h:552-568
    p->max_db_results = 50;
    p->min_nss_factor = 0.005f;
    p->min_matches_per_group = 1;
    p->max_intragroup_gap = 3;
    p->max_distance_between_groups = 3;
    p->max_distance_between_queries = 2;
    p->min_Fpoints = 12;
    p->max_ransac_iterations = 500;
    p->ransac_probability = 0.99;
    p->max_reprojection_error = 2.0;
    p->max_neighbor_ratio = 0.6;
    p->seed = 0;
}

int svo_lc_create(svo_ctx *ctx, const svo_lc_params *params, int width, int height, int channels, svo_lc **out)
{
    SVO_CHECK_ARG(ctx && out);
    svo_lc *l = new svo_lc();
    l->ctx = ctx;
    if (params)
        l->prm = *params;
    else
        svo_lc_default_params(&l->prm);
    const svo_lc_params &p = l->prm;
    if (p.n_features < 8 || p.n_features > 2048 || p.max_entries < 2 || p.dislocal < 0) {
        delete l;
        svo_set_error("svo_lc_create: n_features in 8..2048, max_entries >= 2");
        return SVO_ERR_ARG;
    }
    l->w = width;
    l->h = height;
    l->c = channels;
    l->nf = p.n_features;
    l->capacity = p.max_entries;
    const size_t nf = (size_t)l->nf, cap = (size_t)l->capacity;
    int rc = svo_orb_create(ctx, width, height, channels, p.n_features, p.fast_threshold, &l->orb);
    if (rc || (rc = l->db_desc.ensure(cap * nf * 32)) || (rc = l->db_xy.ensure(cap * nf * 8)) ||
        (rc = l->db_n.ensure(cap * 4)) || (rc = l->q.ensure(nf * (8 + 4 + 4 + 8 + 32) + 64)) ||
        (rc = l->counts.ensure(cap * 4)) || (rc = l->nn.ensure(nf * 12)) ||
        (rc = l->img.ensure((size_t)width * height * channels))) {
        svo_lc_destroy(l);
        return rc;
    }
    *out = l;
    return SVO_OK;
}

int svo_lc_destroy(svo_lc *l)
{
    if (!l)
        return SVO_OK;
    (void)hipStreamSynchronize(l->ctx->stream);
    if (l->orb)
        svo_orb_destroy(l->orb);
    DevBuf *bufs[] = {&l->db_desc, &l->db_xy, &l->db_n, &l->q, &l->counts, &l->nn, &l->img};
    for (DevBuf *b : bufs)
        b->release();
    delete l;
    return SVO_OK;
}

int svo_lc_size(const svo_lc *l) { return l ? (int)l->n_host.size() : 0; }

int svo_lc_detect(svo_lc *l, const uint8_t *image, int mem, int *status, int *query, int *match)
{
    SVO_CHECK_ARG(l && image && status);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    svo_ctx *ctx = l->ctx;
    const svo_lc_params &p = l->prm;
    hipStream_t st = ctx->stream;
    const int entry_id = (int)l->n_host.size();
    if (entry_id >= l->capacity) {
        svo_set_error("loop detector database is full (%d entries)", l->capacity);
        return SVO_ERR_STATE;
    }
    const size_t nf = (size_t)l->nf;
    // ---- features of this frame ----
    const uint8_t *d_img = image;
    if (mem == SVO_MEM_HOST) {
        SVO_HIP(hipMemcpyAsync(l->img.p, image, (size_t)l->w * l->h * l->c, hipMemcpyHostToDevice, st));
        d_img = l->img.as<uint8_t>();
    }
    float *qxy = l->q.as<float>();
    int *qoct = reinterpret_cast<int *>(qxy + 2 * nf);
    float *qresp = reinterpret_cast<float *>(qoct + nf), *qdir = qresp + nf;
    uint32_t *qdesc = reinterpret_cast<uint32_t *>(qdir + 2 * nf);
    int *d_nq = reinterpret_cast<int *>(qdesc + 8 * nf);
    int rc = svo_orb_launch(l->orb, d_img, qxy, qoct, qresp, qdir, qdesc, d_nq);
    if (rc)
        return rc;
    // ---- similarity of the query to every stored entry (one workgroup per entry) ----
    std::vector<int> counts((size_t)entry_id, 0);
    int nq = 0;
    if (entry_id > 0)
        hipLaunchKernelGGL(lc_score_kernel, dim3(entry_id), dim3(256), nf * 32, st, qdesc, d_nq,
                           l->db_desc.as<uint32_t>(), l->db_n.as<int>(), l->nf, p.hamming_threshold,
                           l->counts.as<int>());
    // ---- the query becomes entry `entry_id` (m_database->add + m_image_keys/descriptors, :728,:842-851) ----
    SVO_HIP(hipMemcpyAsync(l->db_desc.as<uint32_t>() + (size_t)entry_id * nf * 8, qdesc, nf * 32,
                           hipMemcpyDeviceToDevice, st));
    SVO_HIP(hipMemcpyAsync(l->db_xy.as<float>() + (size_t)entry_id * nf * 2, qxy, nf * 8, hipMemcpyDeviceToDevice, st));
    SVO_HIP(hipMemcpyAsync(l->db_n.as<int>() + entry_id, d_nq, 4, hipMemcpyDeviceToDevice, st));
    SVO_HIP(hipMemcpyAsync(&nq, d_nq, 4, hipMemcpyDeviceToHost, st));
    if (entry_id > 0)
        SVO_HIP(hipMemcpyAsync(counts.data(), l->counts.p, (size_t)entry_id * 4, hipMemcpyDeviceToHost, st));
    std::vector<float> kxy(nf * 2, 0.f);
    SVO_HIP(hipMemcpyAsync(kxy.data(), qxy, nf * 8, hipMemcpyDeviceToHost, st));
    SVO_HIP(hipStreamSynchronize(st));
    kxy.resize((size_t)nq * 2);
    auto score = [&](int e) { return nq > 0 ? (double)counts[e] / (double)nq : 0.; };

    int st_out = SVO_LC_CLOSE_MATCHES_ONLY, match_out = -1;
    if (entry_id > p.dislocal) {  // :714-722
        const int max_id = entry_id - p.dislocal;
        std::vector<Result> qret;
        for (int e = 0; e < max_id; e++)
            if (score(e) > 0)
                qret.push_back({e, score(e)});
        std::stable_sort(qret.begin(), qret.end(), [](const Result &a, const Result &b) { return a.score > b.score; });
        if ((int)qret.size() > p.max_db_results)
            qret.resize(p.max_db_results);
        if (!qret.empty()) {
            double ns = 1.0;
            if (p.use_nss)
                ns = l->have_last ? score(entry_id - 1) : 0.;  // :736-739
            if (!p.use_nss || ns >= p.min_nss_factor) {
                const double cut = (double)p.alpha * ns;  // removeLowScores, :1320-1338
                size_t keep = 0;
                while (keep < qret.size() && qret[keep].score >= cut)
                    keep++;
                qret.resize(keep);
                if (!qret.empty()) {
                    match_out = qret[0].id;
                    // ---- computeIslands, :875-951 ----
                    std::vector<Island> islands;
                    if (qret.size() == 1) {
                        islands.push_back({qret[0].id, qret[0].id, qret[0].score, qret[0].id, qret[0].score});
                    } else {
                        std::stable_sort(qret.begin(), qret.end(), [](const Result &a, const Result &b) { return a.id < b.id; });
                        int first = qret[0].id, last = qret[0].id;
                        size_t i_first = 0, i_last = 0;
                        double best_score = qret[0].score;
                        int best_entry = qret[0].id;
                        auto close = [&]() {
                            if (last - first + 1 >= p.min_matches_per_group) {
                                double sum = 0;
                                for (size_t i = i_first; i <= i_last; i++)
                                    sum += qret[i].score;
                                islands.push_back({first, last, sum, best_entry, best_score});
                            }
                        };
                        for (size_t idx = 1; idx < qret.size(); idx++) {
                            if (qret[idx].id - last < p.max_intragroup_gap) {
                                last = qret[idx].id;
                                i_last = idx;
                                if (qret[idx].score > best_score) {
                                    best_score = qret[idx].score;
                                    best_entry = qret[idx].id;
                                }
                            } else {
                                close();
                                first = last = qret[idx].id;
                                i_first = i_last = idx;
                                best_score = qret[idx].score;
                                best_entry = qret[idx].id;
                            }
                        }
                        close();
                    }
                    if (!islands.empty()) {
                        size_t bi = 0;  // std::max_element: the first maximum
                        for (size_t i = 1; i < islands.size(); i++)
                            if (islands[bi].score < islands[i].score)
                                bi = i;
                        const Island &isl = islands[bi];
                        // ---- updateTemporalWindow, :966-1003 ----
                        if (l->window_n == 0 || entry_id - l->window_query > p.max_distance_between_queries) {
                            l->window_n = 1;
                        } else {
                            const int a1 = l->window_first, a2 = l->window_last, b1 = isl.first, b2 = isl.last;
                            bool fit = (b1 <= a1 && a1 <= b2) || (a1 <= b1 && b1 <= a2);
                            if (!fit) {
                                const int d1 = a1 - b2, d2 = b1 - a2;
                                fit = (d1 > d2 ? d1 : d2) <= p.max_distance_between_groups;
                            }
                            l->window_n = fit ? l->window_n + 1 : 1;
                        }
                        l->window_first = isl.first;
                        l->window_last = isl.last;
                        l->window_query = entry_id;
                        match_out = isl.best_entry;
                        if (l->window_n > p.k) {
                            // ---- geometric check: neighbour-ratio matches + RANSAC F (:1101-1160) ----
                            bool detection = false;
                            const int old = isl.best_entry, na = l->n_host[old];
                            if (na > 0 && nq > 0) {
                                int *bj = l->nn.as<int>(), *dd1 = bj + nf, *dd2 = dd1 + nf;
                                hipLaunchKernelGGL(lc_nearest2_kernel, dim3((na + 255) / 256), dim3(256), nf * 32, st,
                                                   l->db_desc.as<uint32_t>() + (size_t)old * nf * 8, na, qdesc, d_nq, bj,
                                                   dd1, dd2);
                                std::vector<int> h((size_t)3 * nf);
                                SVO_HIP(hipMemcpyAsync(h.data(), bj, nf * 12, hipMemcpyDeviceToHost, st));
                                SVO_HIP(hipStreamSynchronize(st));
                                const int *hbj = h.data(), *hd1 = hbj + nf, *hd2 = hd1 + nf;
                                std::vector<int> mA, mB;
                                for (int i = 0; i < na; i++) {
                                    if ((double)hd1[i] / (double)hd2[i] <= p.max_neighbor_ratio) {  // :1293
                                        const int jb = hbj[i];
                                        auto it = std::find(mB.begin(), mB.end(), jb);
                                        if (it == mB.end()) {
                                            mB.push_back(jb);
                                            mA.push_back(i);
                                        } else {
                                            const size_t k2 = (size_t)(it - mB.begin());
                                            if (hd1[i] < hd1[mA[k2]])
                                                mA[k2] = i;
                                        }
                                    }
                                }
                                if ((int)mA.size() >= p.min_Fpoints) {
                                    std::vector<float> po(mA.size() * 2), pc(mA.size() * 2);
                                    const std::vector<float> &ko = l->xy_host[old];
                                    for (size_t i = 0; i < mA.size(); i++) {
                                        po[2 * i] = ko[2 * mA[i]];
                                        po[2 * i + 1] = ko[2 * mA[i] + 1];
                                        pc[2 * i] = kxy[2 * mB[i]];
                                        pc[2 * i + 1] = kxy[2 * mB[i] + 1];
                                    }
                                    std::vector<uint8_t> mask(mA.size());
                                    int cnt = 0;
                                    rc = svo_fransac(ctx, po.data(), pc.data(), (int)mA.size(), p.max_reprojection_error,
                                                     p.ransac_probability, p.max_ransac_iterations,
                                                     p.seed + (uint64_t)entry_id, mask.data(), nullptr, &cnt, nullptr,
                                                     SVO_MEM_HOST);
                                    if (rc)
                                        return rc;
                                    detection = cnt >= p.min_Fpoints;
                                }
                            }
                            st_out = detection ? SVO_LC_LOOP_DETECTED : SVO_LC_NO_GEOMETRICAL_CONSISTENCY;
                        } else
                            st_out = SVO_LC_NO_TEMPORAL_CONSISTENCY;
                    } else
                        st_out = SVO_LC_NO_GROUPS;
                } else
                    st_out = SVO_LC_LOW_SCORES;
            } else
                st_out = SVO_LC_LOW_NSS_FACTOR;
        } else
            st_out = SVO_LC_NO_DB_RESULTS;
    }
    l->n_host.push_back(nq);
    l->xy_host.push_back(kxy);
    if (p.use_nss && entry_id + 1 > p.dislocal)  // :855-858
        l->have_last = true;
    *status = st_out;
    if (query)
        *query = entry_id;
    if (match)
        *match = match_out;
    return SVO_OK;
}

}  // extern "C"
