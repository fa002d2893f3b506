// orb.hip -- binary features for the loop-closure detector on gfx950.
//
// Replaces the cv::ORB::create()->detectAndCompute(img, Mat(), kp, desc) call of
// visualSLAM::checkLoopDetectorStatus (src/optimizationStuff.cpp:49-56): oriented FAST corners
// ranked by their Harris response + rotation-steered binary tests on a smoothed patch.  The
// recipe (stated in full in oracle/orb.c, which this file matches bit for bit): OpenCV's
// fixed-point grey conversion, 3 octaves of the tracker's factor-2 pyramid, FAST-9 (t = 20),
// Harris 7x7 / k = 0.04 on integer gradient sums, 3x3 non-maximum suppression, the strongest
// 286 / 143 / 71 corners per octave, orientation = unit vector of the radius-15 intensity moments,
// 256 tests from a seeded pattern on the 5x5-binomial-smoothed octave.
//
// Kernels per octave (all byte / integer work, HBM-bound, row-contiguous accesses on the padded
// pyramid level so no border arithmetic):
//   blur5            5x5 binomial, thread per pixel
//   fast_harris      thread per pixel: 16-pixel ring -> two 16-bit masks -> nine-contiguous test by
//                    shifts; corners get their Harris response
//   nms + strip count, strip scan, candidate write: order-preserving multi-workgroup compaction
//   select           one workgroup: k-th largest response by a 32-step bitwise search over the
//                    candidates, ties at the cut in raster order, ballot-scan write
//   describe         ONE WAVEFRONT PER KEYPOINT: disc moments reduced with DPP, the 256 tests 64 at
//                    a time, descriptor words straight from __ballot
#include "svo_internal.h"

namespace {

constexpr int EDGE = 19, HALF_PATCH = 15, STRIP = 1024;

__global__ __launch_bounds__(256) void gray_kernel(const uint8_t *__restrict__ img, int n, int c,
                                                   uint8_t *__restrict__ gray)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    gray[i] = c == 1 ? img[i]
                     : (uint8_t)((1868 * img[3 * i] + 9617 * img[3 * i + 1] + 4899 * img[3 * i + 2] + 8192) >> 14);
}

__global__ __launch_bounds__(256) void blur5_kernel(const uint8_t *__restrict__ lvl, int pitch, int w, int h,
                                                    uint8_t *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h)
        return;
    int s = 0;
#pragma unroll
    for (int j = -2; j <= 2; j++) {
        const uint8_t *r = lvl + (ptrdiff_t)(y + j) * pitch + x;
        const int kj = j == 0 ? 6 : ((j == 1 || j == -1) ? 4 : 1);
        s += kj * (r[-2] + 4 * r[-1] + 6 * r[0] + 4 * r[1] + r[2]);
    }
    out[(size_t)y * w + x] = (uint8_t)((s + 128) >> 8);
}

__device__ __forceinline__ bool nine_contiguous(unsigned m)
{
    const unsigned d = m | (m << 16);
    unsigned r = d;
#pragma unroll
    for (int k = 1; k < 9; k++)
        r &= d >> k;
    return (r & 0xffffu) != 0;
}

__global__ __launch_bounds__(256) void fast_harris_kernel(const uint8_t *__restrict__ lvl, int pitch, int w, int h,
                                                          int t, float *__restrict__ R, uint8_t *__restrict__ corner)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h)
        return;
    const size_t idx = (size_t)y * w + x;
    float resp = 0.f;
    uint8_t is_c = 0;
    if (x >= EDGE && x < w - EDGE && y >= EDGE && y < h - EDGE) {
        const uint8_t *p0 = lvl + (ptrdiff_t)y * pitch + x;
        const int p = p0[0];
        constexpr int CX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
        constexpr int CY[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
        unsigned br = 0, dk = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int q = p0[CY[k] * pitch + CX[k]];
            br |= (q > p + t ? 1u : 0u) << k;
            dk |= (q < p - t ? 1u : 0u) << k;
        }
        if (nine_contiguous(br) || nine_contiguous(dk)) {
            is_c = 1;
            int a = 0, b = 0, c = 0;
            for (int j = -3; j <= 3; j++) {
                const uint8_t *r = p0 + (ptrdiff_t)j * pitch;
#pragma unroll
                for (int i = -3; i <= 3; i++) {
                    const uint8_t *q = r + i;
                    const int ix = (q[1] - q[-1]) * 2 + (q[-pitch + 1] - q[-pitch - 1]) + (q[pitch + 1] - q[pitch - 1]);
                    const int iy = (q[pitch] - q[-pitch]) * 2 + (q[pitch - 1] - q[-pitch - 1]) + (q[pitch + 1] - q[-pitch + 1]);
                    a += ix * ix;
                    b += iy * iy;
                    c += ix * iy;
                }
            }
            const float fa = (float)a, fb = (float)b, fc = (float)c;
            const float sc = 1.f / (4 * 7 * 255.f);
            const float s4 = sc * sc * sc * sc;
            resp = (fa * fb - fc * fc - 0.04f * (fa + fb) * (fa + fb)) * s4;
        }
    }
    R[idx] = resp;
    corner[idx] = is_c;
}

// 3x3 non-maximum suppression among corners (strictly above the raster-earlier neighbours, at
// least equal to the later ones) + the number of survivors per 1024-pixel strip
__global__ __launch_bounds__(STRIP) void nms_count_kernel(const float *__restrict__ R, const uint8_t *__restrict__ corner,
                                                          int w, int h, uint8_t *__restrict__ keep,
                                                          int *__restrict__ strip_count)
{
    __shared__ int s_w[16];
    const int idx = blockIdx.x * STRIP + threadIdx.x;
    bool k = false;
    if (idx < w * h && corner[idx]) {
        const int x = idx % w, y = idx / w;
        const float r = R[idx];
        k = true;
#pragma unroll
        for (int j = -1; j <= 1; j++)
#pragma unroll
            for (int i = -1; i <= 1; i++) {
                if (!i && !j)
                    continue;
                const int n = (y + j) * w + x + i;  // corners keep a 19-pixel margin: always inside
                if (!corner[n])
                    continue;
                const bool earlier = j < 0 || (j == 0 && i < 0);
                if (earlier ? !(r > R[n]) : !(r >= R[n]))
                    k = false;
            }
    }
    if (idx < w * h)
        keep[idx] = k ? 1 : 0;
    const unsigned long long bal = __ballot(k);
    if ((threadIdx.x & 63) == 0)
        s_w[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < 16; i++)
            s += s_w[i];
        strip_count[blockIdx.x] = s;
    }
}

// exclusive scan of the strip counts (one workgroup; n_strips <= a few thousand)
__global__ __launch_bounds__(1024) void strip_scan_kernel(const int *__restrict__ cnt, int n, int *__restrict__ off,
                                                          int *__restrict__ total)
{
    __shared__ int s_part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    int s = 0;
    for (int i = t * per; i < (t + 1) * per && i < n; i++)
        s += cnt[i];
    s_part[t] = s;
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int i = 0; i < 1024; i++) {
            const int v = s_part[i];
            s_part[i] = acc;
            acc += v;
        }
        *total = acc;
    }
    __syncthreads();
    int acc = s_part[t];
    for (int i = t * per; i < (t + 1) * per && i < n; i++) {
        off[i] = acc;
        acc += cnt[i];
    }
}

__global__ __launch_bounds__(STRIP) void cand_write_kernel(const uint8_t *__restrict__ keep, const float *__restrict__ R,
                                                           int n_pix, const int *__restrict__ off, int cap,
                                                           int *__restrict__ cand_idx, float *__restrict__ cand_resp)
{
    __shared__ int s_w[16];
    const int idx = blockIdx.x * STRIP + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool k = idx < n_pix && keep[idx];
    const unsigned long long bal = __ballot(k);
    if (lane == 0)
        s_w[wave] = __popcll(bal);
    __syncthreads();
    int base = off[blockIdx.x];
    for (int i = 0; i < wave; i++)
        base += s_w[i];
    if (k) {
        const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (pos < cap) {
            cand_idx[pos] = idx;
            cand_resp[pos] = R[idx];
        }
    }
}

__device__ __forceinline__ unsigned sortable(float f)
{
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// the `want` strongest candidates (ties at the cut: earlier pixel first), written in raster order
__global__ __launch_bounds__(1024) void select_kernel(const int *__restrict__ cand_idx, const float *__restrict__ cand_resp,
                                                      const int *__restrict__ d_nc, int cap, int want,
                                                      int *__restrict__ sel_idx, float *__restrict__ sel_resp,
                                                      int *__restrict__ d_nsel)
{
    __shared__ int s_red[16], s_bcast, s_base, s_ties;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nc = min(*d_nc, cap);
    auto block_sum = [&](int v) -> int {
        v += __shfl_xor(v, 32, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        __syncthreads();
        if (lane == 0)
            s_red[wave] = v;
        __syncthreads();
        if (t == 0) {
            int s = 0;
            for (int i = 0; i < 16; i++)
                s += s_red[i];
            s_bcast = s;
        }
        __syncthreads();
        return s_bcast;
    };
    unsigned thr = 0;  // everything is >= 0
    int n_above = 0;
    if (nc > want) {
        for (int bit = 31; bit >= 0; bit--) {
            const unsigned test = thr | (1u << bit);
            int c = 0;
            for (int i = t; i < nc; i += 1024)
                c += sortable(cand_resp[i]) >= test ? 1 : 0;
            if (block_sum(c) >= want)
                thr = test;
        }
        int c = 0;
        for (int i = t; i < nc; i += 1024)
            c += sortable(cand_resp[i]) > thr ? 1 : 0;
        n_above = block_sum(c);
    }
    const int ties_allowed = nc > want ? want - n_above : 0;
    if (t == 0) {
        s_base = 0;
        s_ties = 0;
    }
    __syncthreads();
    for (int start = 0; start < nc; start += 1024) {
        const int i = start + t;
        const unsigned key = i < nc ? sortable(cand_resp[i]) : 0u;
        const bool above = i < nc && (nc <= want || key > thr);
        const bool tie = i < nc && nc > want && key == thr;
        // ties first get their running rank (raster order), the first `ties_allowed` are kept
        const unsigned long long tb = __ballot(tie);
        if (lane == 0)
            s_red[wave] = __popcll(tb);
        __syncthreads();
        int tie_rank = s_ties + __popcll(tb & ((1ull << lane) - 1ull));
        int tie_total = 0;
        for (int w2 = 0; w2 < 16; w2++) {
            tie_rank += w2 < wave ? s_red[w2] : 0;
            tie_total += s_red[w2];
        }
        const bool keep = above || (tie && tie_rank < ties_allowed);
        __syncthreads();
        const unsigned long long kb = __ballot(keep);
        if (lane == 0)
            s_red[wave] = __popcll(kb);
        __syncthreads();
        int pos = s_base + __popcll(kb & ((1ull << lane) - 1ull));
        int kept_total = 0;
        for (int w2 = 0; w2 < 16; w2++) {
            pos += w2 < wave ? s_red[w2] : 0;
            kept_total += s_red[w2];
        }
        if (keep) {
            sel_idx[pos] = cand_idx[i];
            sel_resp[pos] = cand_resp[i];
        }
        __syncthreads();
        if (t == 0) {
            s_base += kept_total;
            s_ties += tie_total;
        }
        __syncthreads();
    }
    if (t == 0)
        *d_nsel = s_base;
}

__device__ __forceinline__ int wave_sum_int(int v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// one wavefront per selected keypoint: orientation + descriptor, appended at out_base
__global__ __launch_bounds__(256) void describe_kernel(const uint8_t *__restrict__ lvl, int pitch, int w,
                                                       const uint8_t *__restrict__ blur, const int *__restrict__ sel_idx,
                                                       const float *__restrict__ sel_resp, const int *__restrict__ d_nsel,
                                                       const int8_t *__restrict__ pat, int octave,
                                                       const int *__restrict__ d_out_base, float *__restrict__ xy,
                                                       int *__restrict__ oct, float *__restrict__ resp,
                                                       float *__restrict__ dir, uint32_t *__restrict__ desc)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= *d_nsel)
        return;
    const int idx = sel_idx[i], x = idx % w, y = idx / w;
    int m10 = 0, m01 = 0;
    // the radius-15 disc, row by row; lanes take its 709 pixels in raster order
    for (int e = lane; e < 31 * 31; e += 64) {
        const int v = e / 31 - HALF_PATCH, u = e % 31 - HALF_PATCH;
        if (u * u + v * v <= HALF_PATCH * HALF_PATCH) {  // == |u| <= floor(sqrt(225 - v*v))
            const int p = lvl[(ptrdiff_t)(y + v) * pitch + x + u];
            m10 += u * p;
            m01 += v * p;
        }
    }
    m10 = wave_sum_int(m10);
    m01 = wave_sum_int(m01);
    const float f10 = (float)m10, f01 = (float)m01;
    const float nrm = sqrtf(f10 * f10 + f01 * f01);
    const float cs = nrm > 0.f ? f10 / nrm : 1.f, sn = nrm > 0.f ? f01 / nrm : 0.f;
    const int o = *d_out_base + i;
    uint32_t *d = desc + (size_t)8 * o;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const int b = pass * 64 + lane;
        const float x1 = (float)pat[4 * b], y1 = (float)pat[4 * b + 1], x2 = (float)pat[4 * b + 2],
                    y2 = (float)pat[4 * b + 3];
        const int ax = (int)rintf(cs * x1 - sn * y1), ay = (int)rintf(sn * x1 + cs * y1);
        const int bx = (int)rintf(cs * x2 - sn * y2), by = (int)rintf(sn * x2 + cs * y2);
        const int pa = blur[(size_t)(y + ay) * w + x + ax], pb = blur[(size_t)(y + by) * w + x + bx];
        const unsigned long long bits = __ballot(pa < pb);
        if (lane == 0) {
            d[2 * pass] = (uint32_t)bits;
            d[2 * pass + 1] = (uint32_t)(bits >> 32);
        }
    }
    if (lane == 0) {
        xy[2 * o] = (float)(x << octave);
        xy[2 * o + 1] = (float)(y << octave);
        oct[o] = octave;
        resp[o] = sel_resp[i];
        dir[2 * o] = cs;
        dir[2 * o + 1] = sn;
    }
}

__global__ void add_count_kernel(int *total, const int *add) { *total += *add; }
__global__ void zero_int_kernel(int *p) { *p = 0; }

void orb_pattern_host(int8_t *pat)
{
    // the seeded generator stated in oracle/orb.c (three uniforms in [-13, 13], halved)
    uint32_t s = 0x9E3779B9u;
    for (int i = 0; i < 256 * 4; i++) {
        int acc = 0;
        for (int k = 0; k < 3; k++) {
            s = s * 1664525u + 1013904223u;
            acc += (int)((s >> 16) % 27u) - 13;
        }
        int v = acc / 2;
        v = v > 13 ? 13 : (v < -13 ? -13 : v);
        pat[i] = (int8_t)v;
    }
    for (int i = 0; i < 256; i++)
        if (pat[4 * i] == pat[4 * i + 2] && pat[4 * i + 1] == pat[4 * i + 3])
            pat[4 * i + 2] = (int8_t)(pat[4 * i + 2] >= 0 ? pat[4 * i + 2] - 1 : pat[4 * i + 2] + 1);
}

}  // namespace

struct svo_orb {
    svo_ctx *ctx = nullptr;
    int w = 0, h = 0, c = 0, n_features = 500, fast_t = 20;
    svo_pyramid *pyr = nullptr;  // grey, 3 octaves
    DevBuf gray, blur, R, corner, keep, strip_cnt, strip_off, cand_idx, cand_resp, sel_idx, sel_resp, pat, counts;
    int cand_cap = 0;
};

int svo_orb_create(svo_ctx *ctx, int w, int h, int c, int n_features, int fast_t, svo_orb **out)
{
    SVO_CHECK_ARG(ctx && out && w > 4 * EDGE && h > 4 * EDGE && (c == 1 || c == 3) && n_features > 0 && fast_t > 0);
    svo_orb *o = new svo_orb();
    o->ctx = ctx;
    o->w = w;
    o->h = h;
    o->c = c;
    o->n_features = n_features;
    o->fast_t = fast_t;
    int rc = svo_pyramid_create(ctx, w, h, 1, 3, &o->pyr);
    const size_t npix = (size_t)w * h;
    o->cand_cap = (int)(npix / 4 + 1024);  // non-maximum suppression leaves at most one corner per 2x2
    const int strips = (int)((npix + STRIP - 1) / STRIP);
    if (rc || (rc = o->gray.ensure(npix)) || (rc = o->blur.ensure(npix)) || (rc = o->R.ensure(npix * 4)) ||
        (rc = o->corner.ensure(npix)) || (rc = o->keep.ensure(npix)) || (rc = o->strip_cnt.ensure((size_t)strips * 4)) ||
        (rc = o->strip_off.ensure((size_t)strips * 4)) || (rc = o->cand_idx.ensure((size_t)o->cand_cap * 4)) ||
        (rc = o->cand_resp.ensure((size_t)o->cand_cap * 4)) || (rc = o->sel_idx.ensure((size_t)n_features * 4 + 64)) ||
        (rc = o->sel_resp.ensure((size_t)n_features * 4 + 64)) || (rc = o->pat.ensure(1024)) ||
        (rc = o->counts.ensure(64))) {
        svo_orb_destroy(o);
        return rc;
    }
    int8_t pat[1024];
    orb_pattern_host(pat);
    if (hipMemcpy(o->pat.p, pat, sizeof(pat), hipMemcpyHostToDevice) != hipSuccess) {
        svo_set_error("orb: pattern upload failed");
        svo_orb_destroy(o);
        return SVO_ERR_HIP;
    }
    *out = o;
    return SVO_OK;
}

int svo_orb_destroy(svo_orb *o)
{
    if (!o)
        return SVO_OK;
    (void)hipStreamSynchronize(o->ctx->stream);
    if (o->pyr)
        svo_pyramid_destroy(o->ctx, o->pyr);
    DevBuf *bufs[] = {&o->gray, &o->blur, &o->R, &o->corner, &o->keep, &o->strip_cnt, &o->strip_off,
                      &o->cand_idx, &o->cand_resp, &o->sel_idx, &o->sel_resp, &o->pat, &o->counts};
    for (DevBuf *b : bufs)
        b->release();
    delete o;
    return SVO_OK;
}

// d_image: device image (h x w x c).  Device outputs with n_features capacity; *d_n (device int) =
// number of features.  Asynchronous on the context's stream.
int svo_orb_launch(svo_orb *o, const uint8_t *d_image, float *d_xy, int *d_oct, float *d_resp, float *d_dir,
                   uint32_t *d_desc, int *d_n)
{
    svo_ctx *ctx = o->ctx;
    hipStream_t st = ctx->stream;
    const int npix0 = o->w * o->h;
    hipLaunchKernelGGL(gray_kernel, dim3((npix0 + 255) / 256), dim3(256), 0, st, d_image, npix0, o->c,
                       o->gray.as<uint8_t>());
    int rc = svo_build_pyramid_from_device(ctx, o->pyr, o->gray.as<uint8_t>());
    if (rc)
        return rc;
    int *d_nc = o->counts.as<int>(), *d_nsel = d_nc + 1;
    hipLaunchKernelGGL(zero_int_kernel, dim3(1), dim3(1), 0, st, d_n);
    int want[3];
    want[0] = (int)(o->n_features * 4.0 / 7.0 + 0.5);
    want[1] = (int)(o->n_features * 2.0 / 7.0 + 0.5);
    want[2] = o->n_features - want[0] - want[1];
    const PyrDev &pd = o->pyr->dev;
    for (int l = 0; l < 3; l++) {
        const int w = pd.w[l], h = pd.h[l], pitch = pd.pitch[l], npix = w * h;
        if (w <= 2 * EDGE || h <= 2 * EDGE || want[l] <= 0)
            continue;
        const uint8_t *lvl = pd.lvl[l];
        const dim3 g2((w + 255) / 256, h), b2(256);
        const int strips = (npix + STRIP - 1) / STRIP;
        hipLaunchKernelGGL(blur5_kernel, g2, b2, 0, st, lvl, pitch, w, h, o->blur.as<uint8_t>());
        hipLaunchKernelGGL(fast_harris_kernel, g2, b2, 0, st, lvl, pitch, w, h, o->fast_t, o->R.as<float>(),
                           o->corner.as<uint8_t>());
        hipLaunchKernelGGL(nms_count_kernel, dim3(strips), dim3(STRIP), 0, st, o->R.as<float>(), o->corner.as<uint8_t>(),
                           w, h, o->keep.as<uint8_t>(), o->strip_cnt.as<int>());
        hipLaunchKernelGGL(strip_scan_kernel, dim3(1), dim3(1024), 0, st, o->strip_cnt.as<int>(), strips,
                           o->strip_off.as<int>(), d_nc);
        hipLaunchKernelGGL(cand_write_kernel, dim3(strips), dim3(STRIP), 0, st, o->keep.as<uint8_t>(), o->R.as<float>(),
                           npix, o->strip_off.as<int>(), o->cand_cap, o->cand_idx.as<int>(), o->cand_resp.as<float>());
        hipLaunchKernelGGL(select_kernel, dim3(1), dim3(1024), 0, st, o->cand_idx.as<int>(), o->cand_resp.as<float>(),
                           d_nc, o->cand_cap, want[l], o->sel_idx.as<int>(), o->sel_resp.as<float>(), d_nsel);
        hipLaunchKernelGGL(describe_kernel, dim3((want[l] + 3) / 4), dim3(256), 0, st, lvl, pitch, w,
                           o->blur.as<uint8_t>(), o->sel_idx.as<int>(), o->sel_resp.as<float>(), d_nsel,
                           o->pat.as<int8_t>(), l, d_n, d_xy, d_oct, d_resp, d_dir, d_desc);
        hipLaunchKernelGGL(add_count_kernel, dim3(1), dim3(1), 0, st, d_n, d_nsel);
    }
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

extern "C" int svo_orb_extract(svo_ctx *ctx, const uint8_t *image, int w, int h, int c, int n_features,
                               int fast_threshold, float *xy, int *octave, float *response, float *dir,
                               uint32_t *desc, int *n, int mem)
{
    SVO_CHECK_ARG(ctx && image && xy && desc && n && n_features > 0);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    SVO_HIP(hipSetDevice(ctx->device));
    int rc;
    const int key[5] = {w, h, c, n_features, fast_threshold};
    if (!ctx->orb_cache || memcmp(key, ctx->orb_key, sizeof(key)) != 0) {  // the context keeps one extractor: rebuilt on a new shape
        if (ctx->orb_cache)
            svo_orb_destroy(ctx->orb_cache);
        ctx->orb_cache = nullptr;
        if ((rc = svo_orb_create(ctx, w, h, c, n_features, fast_threshold, &ctx->orb_cache)))
            return rc;
        memcpy(ctx->orb_key, key, sizeof(key));
    }
    svo_orb *o = ctx->orb_cache;
    const size_t nf = (size_t)n_features;
    DevBuf &out = ctx->orb_out;
    const uint8_t *d_img = image;
    if ((rc = out.ensure(nf * (8 + 4 + 4 + 8 + 32) + 64)) ||
        (mem == SVO_MEM_HOST && (rc = ctx->s_img.ensure((size_t)w * h * c))))
        return rc;
    float *dxy = out.as<float>();
    int *doct = reinterpret_cast<int *>(dxy + 2 * nf);
    float *dresp = reinterpret_cast<float *>(doct + nf), *ddir = dresp + nf;
    uint32_t *ddesc = reinterpret_cast<uint32_t *>(ddir + 2 * nf);
    int *dn = reinterpret_cast<int *>(ddesc + 8 * nf);
    hipError_t e = hipSuccess;
    if (mem == SVO_MEM_HOST) {
        e = hipMemcpyAsync(ctx->s_img.p, image, (size_t)w * h * c, hipMemcpyHostToDevice, ctx->stream);
        d_img = ctx->s_img.as<uint8_t>();
    }
    if (e == hipSuccess)
        rc = svo_orb_launch(o, d_img, dxy, doct, dresp, ddir, ddesc, dn);
    int hn = 0;
    if (e == hipSuccess && !rc) {
        e = hipMemcpyAsync(&hn, dn, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess && !rc && hn > 0) {
        const hipMemcpyKind k = mem == SVO_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
        (void)hipMemcpyAsync(xy, dxy, (size_t)hn * 8, k, ctx->stream);
        if (octave)
            (void)hipMemcpyAsync(octave, doct, (size_t)hn * 4, k, ctx->stream);
        if (response)
            (void)hipMemcpyAsync(response, dresp, (size_t)hn * 4, k, ctx->stream);
        if (dir)
            (void)hipMemcpyAsync(dir, ddir, (size_t)hn * 8, k, ctx->stream);
        (void)hipMemcpyAsync(desc, ddesc, (size_t)hn * 32, k, ctx->stream);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e != hipSuccess) {
        svo_set_error("svo_orb_extract -> %s", hipGetErrorString(e));
        return SVO_ERR_HIP;
    }
    if (rc)
        return rc;
    *n = hn;
    return SVO_OK;
}
