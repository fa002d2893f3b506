// lk.hip -- pyramidal Lucas-Kanade tracker for gfx950 (wave64).
//
// Replaces cv::calcOpticalFlowPyrLK(prev, next, pts, ...) with default arguments as the
// reference calls it at src/tracking.cpp:18 (left->right) and src/tracking.cpp:52
// (t-1 -> t): 21x21 window, 4 pyramid levels, <=30 iterations, eps 0.01, minEig 1e-4,
// 14-bit fixed-point bilinear weights, int16 patches, Scharr derivatives.
//
// Mapping: ONE WAVEFRONT PER KEYPOINT, all pyramid levels inside one launch (the levels
// of one point depend on each other, points never do).  Lane l owns window row l/3 and a
// 7-pixel segment (l%3) of it, i.e. 7*C patch elements that stay in VGPRs for the whole
// level (template patch I and both derivative patches).  Per level a wave
//   1. stages the 24x24 neighbourhood of the previous image in LDS (row-contiguous loads,
//      reflect-101 border), derives the 22x22 Scharr tile LDS->LDS (zero outside the
//      image, as OpenCV pads the derivative buffer), builds its patch registers and the
//      2x2 normal matrix;
//   2. stages a (22+2*JR)^2 tile of the next image around the current guess and iterates
//      out of LDS; the tile is re-staged only when the guess drifts more than JR pixels.
// All sums of integer products are exact (int32 per lane, int64 across the wave via
// cross-lane shuffles) and are rounded to float once, so the result does not depend on
// the reduction order; scalar float math is compiled with -ffp-contract=off.
//
// Roofline: the kernel's algorithmic HBM traffic is both pyramids once plus 21 B/point
// (SURVEY.md section 8d); its time is VALU/LDS work, see DESIGN.md.
#include "svo_internal.h"

namespace {

constexpr int WIN = SVO_LK_WIN;
constexpr int JR = 5;                  // drift radius one staged J tile tolerates
constexpr int TS = WIN + 1 + 2 * JR;   // 32: J tile side
constexpr int PT = WIN + 3;            // 24: previous-image tile side (Scharr + bilinear halo)
constexpr int DT = WIN + 1;            // 22: derivative tile side
constexpr int SEG = 7;                 // pixels per lane; 3 lanes per window row
constexpr int WAVES = 4;               // waves (= keypoints) per workgroup
constexpr int W_BITS = 14;

struct LkParams {
    int max_level;
    int max_count;
    double eps_sq;
    float min_eig_thr;
};

template <int C> struct Lds {
    static constexpr int T_BYTES = ((PT * PT * C + 15) / 16) * 16;
    static constexpr int D_BYTES = DT * DT * C * 4;
    static constexpr int J_BYTES = TS * TS * C;
    static constexpr int A = T_BYTES + D_BYTES;
    static constexpr int WAVE_BYTES = (((A > J_BYTES ? A : J_BYTES) + 15) / 16) * 16;
};

__device__ __forceinline__ int reflect101(int p, int len)
{
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// LDS produced by some lanes of a wave and consumed by others of the SAME wave: LDS
// instructions of one wave execute in order; this only stops the compiler from moving
// accesses across the hand-off.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int lo = __shfl_xor((int)(v & 0xffffffffll), off);
        int hi = __shfl_xor((int)(v >> 32), off);
        v += ((long long)hi << 32) | (long long)(unsigned)lo;
    }
    return v;
}

__device__ __forceinline__ void bilinear_weights(float a, float b, int &w00, int &w01, int &w10,
                                                 int &w11)
{
    w00 = (int)rintf((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    w01 = (int)rintf(a * (1.f - b) * (float)(1 << W_BITS));
    w10 = (int)rintf((1.f - a) * b * (float)(1 << W_BITS));
    w11 = (1 << W_BITS) - w00 - w01 - w10;
}

template <int C>
__device__ __forceinline__ void stage_tile(uint8_t *tile, int side, const uint8_t *__restrict__ img,
                                           int lw, int lh, int ox, int oy, int lane)
{
    const int rowb = side * C;
    for (int i = lane; i < side * rowb; i += 64) {
        int r = i / rowb, cc = i - r * rowb;
        int px = cc / C, ch = cc - px * C;
        int X = reflect101(ox + px, lw), Y = reflect101(oy + r, lh);
        tile[i] = img[((size_t)Y * lw + X) * C + ch];
    }
}

// sum over this lane's 7*C elements of (J - I) * {Ix, Iy}   (or |J - I| when ABS)
template <int C, bool ABS>
__device__ __forceinline__ void lane_residual(const uint8_t *tj, int tx, int ty, int w00, int w01,
                                              int w10, int w11, const int (&Iv)[SEG * C],
                                              const int (&Ix)[SEG * C], const int (&Iy)[SEG * C],
                                              int &s1, int &s2)
{
    const uint8_t *q0 = tj + (ty * TS + tx) * C;
    const uint8_t *q1 = q0 + TS * C;
    int r0[(SEG + 1) * C], r1[(SEG + 1) * C];
#pragma unroll
    for (int k = 0; k < (SEG + 1) * C; k++) {
        r0[k] = q0[k];
        r1[k] = q1[k];
    }
    s1 = 0;
    s2 = 0;
#pragma unroll
    for (int k = 0; k < SEG * C; k++) {
        int v = r0[k] * w00 + r0[k + C] * w01 + r1[k] * w10 + r1[k + C] * w11;
        int diff = ((v + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5)) - Iv[k];
        if (ABS) {
            s1 += diff < 0 ? -diff : diff;
        } else {
            s1 += diff * Ix[k];
            s2 += diff * Iy[k];
        }
    }
}

template <int C>
__global__ __launch_bounds__(64 * WAVES) void lk_track_kernel(
    PyrDev prev, PyrDev next, const float *__restrict__ prev_pts, int n,
    float *__restrict__ next_pts, uint8_t *__restrict__ status, float *__restrict__ err,
    float *__restrict__ min_eig_out, LkParams prm)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * WAVES + wave;
    if (p >= n)
        return;  // whole wave leaves; no workgroup barrier is used below
    uint8_t *lds = smem + wave * Lds<C>::WAVE_BYTES;
    uint8_t *T = lds;                                           // PT x PT x C bytes
    int *D = reinterpret_cast<int *>(lds + Lds<C>::T_BYTES);    // DT x DT x C packed (dx | dy<<16)
    uint8_t *TJ = lds;                                          // TS x TS x C bytes (reuses T/D)

    const bool active = lane < 3 * WIN;
    const int wy = active ? lane / 3 : WIN - 1;  // window row of this lane
    const int ws = active ? lane - 3 * wy : 0;   // segment
    const int wx = ws * SEG;

    const float ptx = prev_pts[2 * p], pty = prev_pts[2 * p + 1];
    const float half = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);

    float outx = 0.f, outy = 0.f;  // == nextPts[ptidx] of the reference implementation
    int st = 1;
    float errv = 0.f, mineig0 = 0.f;

    for (int level = prm.max_level; level >= 0; level--) {
        const int lw = prev.w[level], lh = prev.h[level];
        const uint8_t *I = prev.lvl[level];
        const uint8_t *J = next.lvl[level];
        const float scale = 1.f / (float)(1 << level);
        float px = ptx * scale, py = pty * scale;
        float nxp, nyp;
        if (level == prm.max_level) {
            nxp = px;
            nyp = py;
        } else {
            nxp = outx * 2.f;
            nyp = outy * 2.f;
        }
        outx = nxp;
        outy = nyp;
        px -= half;
        py -= half;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -WIN || ipx >= lw || ipy < -WIN || ipy >= lh) {
            if (level == 0) {
                st = 0;
                errv = 0.f;
            }
            continue;
        }
        int w00, w01, w10, w11;
        bilinear_weights(px - (float)ipx, py - (float)ipy, w00, w01, w10, w11);

        // ---- 1. previous-image tile, Scharr tile, patch registers, normal matrix ----
        wave_lds_sync();
        stage_tile<C>(T, PT, I, lw, lh, ipx - 1, ipy - 1, lane);
        wave_lds_sync();
        for (int i = lane; i < DT * DT * C; i += 64) {
            int yy = i / (DT * C), cc = i - yy * (DT * C);
            int xx = cc / C, ch = cc - xx * C;
            int X = ipx + xx, Y = ipy + yy;
            int packed = 0;
            if (X >= 0 && X < lw && Y >= 0 && Y < lh) {
                const uint8_t *t = T + (yy * PT + xx) * C + ch;  // top-left of the 3x3
                int a0 = t[0], a1 = t[C], a2 = t[2 * C];
                int b0 = t[PT * C], b2 = t[PT * C + 2 * C];
                int c0 = t[2 * PT * C], c1 = t[2 * PT * C + C], c2 = t[2 * PT * C + 2 * C];
                int dx = 3 * (a2 - a0) + 10 * (b2 - b0) + 3 * (c2 - c0);
                int dy = 3 * (c0 - a0) + 10 * (c1 - a1) + 3 * (c2 - a2);
                packed = (dx & 0xffff) | (dy << 16);
            }
            D[i] = packed;
        }
        wave_lds_sync();

        int Iv[SEG * C], Ix[SEG * C], Iy[SEG * C];
        int a11 = 0, a12 = 0, a22 = 0;
        {
            const uint8_t *t0 = T + ((wy + 1) * PT + (wx + 1)) * C;
            const uint8_t *t1 = t0 + PT * C;
            const int *d0 = D + (wy * DT + wx) * C;
            const int *d1 = d0 + DT * C;
#pragma unroll
            for (int k = 0; k < SEG * C; k++) {
                int v = t0[k] * w00 + t0[k + C] * w01 + t1[k] * w10 + t1[k + C] * w11;
                Iv[k] = (v + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                int p00 = d0[k], p01 = d0[k + C], p10 = d1[k], p11 = d1[k + C];
                int gx = (int)(short)(p00 & 0xffff) * w00 + (int)(short)(p01 & 0xffff) * w01 +
                         (int)(short)(p10 & 0xffff) * w10 + (int)(short)(p11 & 0xffff) * w11;
                int gy = (p00 >> 16) * w00 + (p01 >> 16) * w01 + (p10 >> 16) * w10 + (p11 >> 16) * w11;
                int ix = (gx + (1 << (W_BITS - 1))) >> W_BITS;
                int iy = (gy + (1 << (W_BITS - 1))) >> W_BITS;
                Ix[k] = ix;
                Iy[k] = iy;
                a11 += ix * ix;
                a12 += ix * iy;
                a22 += iy * iy;
            }
        }
        if (!active) {
            a11 = 0;
            a12 = 0;
            a22 = 0;
        }
        const long long sA11 = wave_sum_i64(a11), sA12 = wave_sum_i64(a12), sA22 = wave_sum_i64(a22);
        const float A11 = (float)(double)sA11 * FLT_SCALE;
        const float A12 = (float)(double)sA12 * FLT_SCALE;
        const float A22 = (float)(double)sA22 * FLT_SCALE;
        float Dd = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                             (float)(2 * WIN * WIN);
        if (level == 0)
            mineig0 = minEig;
        if (minEig < prm.min_eig_thr || Dd < 1.1920928955078125e-7f) {
            if (level == 0)
                st = 0;
            continue;
        }
        Dd = 1.f / Dd;

        // ---- 2. iterate on the next image out of an LDS tile ----
        nxp -= half;
        nyp -= half;
        float pdx = 0.f, pdy = 0.f;
        int ox = 0, oy = 0;
        bool have_tile = false;
        for (int j = 0; j < prm.max_count; j++) {
            const int inx = (int)floorf(nxp), iny = (int)floorf(nyp);
            if (inx < -WIN || inx >= lw || iny < -WIN || iny >= lh) {
                if (level == 0)
                    st = 0;
                break;
            }
            if (!have_tile || inx < ox || inx > ox + 2 * JR || iny < oy || iny > oy + 2 * JR) {
                ox = inx - JR;
                oy = iny - JR;
                wave_lds_sync();
                stage_tile<C>(TJ, TS, J, lw, lh, ox, oy, lane);
                wave_lds_sync();
                have_tile = true;
            }
            bilinear_weights(nxp - (float)inx, nyp - (float)iny, w00, w01, w10, w11);
            int s1, s2;
            lane_residual<C, false>(TJ, inx - ox + wx, iny - oy + wy, w00, w01, w10, w11, Iv, Ix, Iy,
                                    s1, s2);
            if (!active) {
                s1 = 0;
                s2 = 0;
            }
            const long long sb1 = wave_sum_i64(s1), sb2 = wave_sum_i64(s2);
            const float b1 = (float)(double)sb1 * FLT_SCALE;
            const float b2 = (float)(double)sb2 * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * Dd;
            const float dy = (A12 * b1 - A11 * b2) * Dd;
            nxp += dx;
            nyp += dy;
            outx = nxp + half;
            outy = nyp + half;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= prm.eps_sq)
                break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                outx -= dx * 0.5f;
                outy -= dy * 0.5f;
                break;
            }
            pdx = dx;
            pdy = dy;
        }

        // ---- 3. level-0 residual (err output of calcOpticalFlowPyrLK) ----
        if (st && level == 0) {
            const float qx = outx - half, qy = outy - half;
            const int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
            if (iqx < -WIN || iqx >= lw || iqy < -WIN || iqy >= lh) {
                st = 0;
                continue;
            }
            if (!have_tile || iqx < ox || iqx > ox + 2 * JR || iqy < oy || iqy > oy + 2 * JR) {
                ox = iqx - JR;
                oy = iqy - JR;
                wave_lds_sync();
                stage_tile<C>(TJ, TS, J, lw, lh, ox, oy, lane);
                wave_lds_sync();
                have_tile = true;
            }
            bilinear_weights(qx - (float)iqx, qy - (float)iqy, w00, w01, w10, w11);
            int s1, s2;
            lane_residual<C, true>(TJ, iqx - ox + wx, iqy - oy + wy, w00, w01, w10, w11, Iv, Ix, Iy, s1,
                                   s2);
            if (!active)
                s1 = 0;
            const long long sabs = wave_sum_i64(s1);
            errv = (float)sabs / (float)(32 * WIN * C * WIN);
        }
    }

    if (lane == 0) {
        next_pts[2 * p] = outx;
        next_pts[2 * p + 1] = outy;
        status[p] = (uint8_t)st;
        if (err)
            err[p] = st ? errv : 0.f;
        if (min_eig_out)
            min_eig_out[p] = mineig0;
    }
}

__global__ void grid_keypoints_kernel(int rows, int cols, int step, int nx, int total,
                                      float *__restrict__ out_xy)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total)
        return;
    int gy = i / nx, gx = i - gy * nx;
    out_xy[2 * i] = (float)((gx + 1) * step);
    out_xy[2 * i + 1] = (float)((gy + 1) * step);
}

}  // namespace

int svo_launch_lk(svo_ctx *ctx, const PyrDev &prev, const PyrDev &next, const float *prev_pts, int n,
                  float *next_pts, uint8_t *status, float *err, float *min_eig)
{
    if (n == 0)
        return SVO_OK;
    LkParams prm;
    prm.max_level = prev.levels - 1;
    prm.max_count = 30;
    prm.eps_sq = 0.01 * 0.01;
    prm.min_eig_thr = (float)1e-4;
    dim3 grid((n + WAVES - 1) / WAVES), block(64 * WAVES);
    ScopedKernelTime t(ctx, SVO_K_LK);
    switch (prev.c) {
    case 1:
        hipLaunchKernelGGL(lk_track_kernel<1>, grid, block, WAVES * Lds<1>::WAVE_BYTES, ctx->stream, prev,
                           next, prev_pts, n, next_pts, status, err, min_eig, prm);
        break;
    case 3:
        hipLaunchKernelGGL(lk_track_kernel<3>, grid, block, WAVES * Lds<3>::WAVE_BYTES, ctx->stream, prev,
                           next, prev_pts, n, next_pts, status, err, min_eig, prm);
        break;
    default:
        svo_set_error("lk: unsupported channel count %d (1 or 3)", prev.c);
        return SVO_ERR_ARG;
    }
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

// number of lattice points of the reference's loop `for (v = s; v < dim - s; v += s)`
static int grid_axis_count(int dim, int step)
{
    int k = 0;
    for (int v = step; v < dim - step; v += step)
        k++;
    return k;
}

int svo_launch_grid(svo_ctx *ctx, int rows, int cols, int step, float *out_xy, int cap)
{
    int nx = grid_axis_count(cols, step), ny = grid_axis_count(rows, step);
    int total = nx * ny;
    if (total > cap)
        total = cap;
    if (total <= 0)
        return SVO_OK;
    hipLaunchKernelGGL(grid_keypoints_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, rows,
                       cols, step, nx, total, out_xy);
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

extern "C" {

int svo_grid_keypoints(svo_ctx *ctx, int rows, int cols, int step, float *out_xy, int cap, int mem,
                       int *count)
{
    SVO_CHECK_ARG(ctx && step > 0 && rows > 0 && cols > 0 && cap >= 0);
    int total = grid_axis_count(cols, step) * grid_axis_count(rows, step);
    if (count)
        *count = total;
    if (!out_xy)
        return SVO_OK;
    if (total > cap) {
        svo_set_error("grid: %d keypoints exceed capacity %d", total, cap);
        return SVO_ERR_CAPACITY;
    }
    if (total == 0)
        return SVO_OK;
    float *d = out_xy;
    if (mem == SVO_MEM_HOST) {
        int rc = ctx->s_a.ensure((size_t)total * 8);
        if (rc)
            return rc;
        d = ctx->s_a.as<float>();
    }
    int rc = svo_launch_grid(ctx, rows, cols, step, d, total);
    if (rc)
        return rc;
    if (mem == SVO_MEM_HOST) {
        SVO_HIP(hipMemcpyAsync(out_xy, d, (size_t)total * 8, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SVO_OK;
}

int svo_lk_track(svo_ctx *ctx, const svo_pyramid *prev, const svo_pyramid *next, const float *prev_pts,
                 int n, float *next_pts, uint8_t *status, float *err, float *min_eig, int mem)
{
    SVO_CHECK_ARG(ctx && prev && next && n >= 0);
    SVO_CHECK_ARG(prev->w == next->w && prev->h == next->h && prev->c == next->c &&
                  prev->levels == next->levels);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    if (n == 0)
        return SVO_OK;
    SVO_CHECK_ARG(prev_pts && next_pts && status);
    if (mem == SVO_MEM_DEVICE)
        return svo_launch_lk(ctx, prev->dev, next->dev, prev_pts, n, next_pts, status, err, min_eig);

    int rc;
    if ((rc = ctx->s_a.ensure((size_t)n * 8)) || (rc = ctx->s_b.ensure((size_t)n * 8)) ||
        (rc = ctx->s_c.ensure((size_t)n)) || (rc = ctx->s_d.ensure((size_t)n * 4)) ||
        (rc = ctx->s_e.ensure((size_t)n * 4)))
        return rc;
    SVO_HIP(hipMemcpyAsync(ctx->s_a.p, prev_pts, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = svo_launch_lk(ctx, prev->dev, next->dev, ctx->s_a.as<float>(), n, ctx->s_b.as<float>(),
                       ctx->s_c.as<uint8_t>(), ctx->s_d.as<float>(), ctx->s_e.as<float>());
    if (rc)
        return rc;
    SVO_HIP(hipMemcpyAsync(next_pts, ctx->s_b.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipMemcpyAsync(status, ctx->s_c.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (err)
        SVO_HIP(hipMemcpyAsync(err, ctx->s_d.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (min_eig)
        SVO_HIP(hipMemcpyAsync(min_eig, ctx->s_e.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

}  // extern "C"
