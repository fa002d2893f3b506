// pyramid.hip -- 5-tap Gaussian pyramid (the pyrDown cv::calcOpticalFlowPyrLK applies to
// both images on every call; reference call sites src/tracking.cpp:18,52).
//
// One workgroup produces a 32x8-pixel output tile.  Its (2*32+3) x (2*8+3) source tile is
// staged in LDS with coalesced row loads (consecutive lanes -> consecutive bytes of one
// image row), the separable [1 4 6 4 1] filter runs LDS -> LDS (horizontal, uint16) and
// LDS -> HBM (vertical, (sum+128)>>8).  BORDER_REFLECT_101 is applied while staging.
// HBM traffic per level: source read ~once (+ halo), destination written once.
#include "svo_internal.h"

namespace {

__device__ __forceinline__ int reflect101(int p, int len)
{
    // cv::borderInterpolate(p, len, BORDER_REFLECT_101); loop form is safe for far halos
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

constexpr int TW = 32, TH = 8;
constexpr int SW = 2 * TW + 3, SH = 2 * TH + 3;

template <int C>
__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t *__restrict__ src, int w,
                                                       int h, uint8_t *__restrict__ dst, int dw,
                                                       int dh)
{
    constexpr int SROW = ((SW * C + 3) / 4) * 4;
    __shared__ uint8_t s_src[SH * SROW];
    __shared__ uint16_t s_h[SH][TW * C];
    const int tid = threadIdx.x;
    const int ox = blockIdx.x * TW, oy = blockIdx.y * TH;

    for (int i = tid; i < SH * SW * C; i += 256) {
        int r = i / (SW * C), cc = i - r * (SW * C);
        int px = cc / C, ch = cc - px * C;
        int sx = reflect101(2 * ox - 2 + px, w);
        int sy = reflect101(2 * oy - 2 + r, h);
        s_src[r * SROW + cc] = src[((size_t)sy * w + sx) * C + ch];
    }
    __syncthreads();
    for (int i = tid; i < SH * TW * C; i += 256) {
        int r = i / (TW * C), cc = i - r * (TW * C);
        int x = cc / C, ch = cc - x * C;
        const uint8_t *s = s_src + r * SROW + (2 * x) * C + ch;
        s_h[r][cc] = (uint16_t)(s[0] + 4 * s[C] + 6 * s[2 * C] + 4 * s[3 * C] + s[4 * C]);
    }
    __syncthreads();
    for (int i = tid; i < TH * TW * C; i += 256) {
        int y = i / (TW * C), cc = i - y * (TW * C);
        int X = ox + cc / C, Y = oy + y;
        if (X < dw && Y < dh) {
            int v = s_h[2 * y][cc] + 4 * s_h[2 * y + 1][cc] + 6 * s_h[2 * y + 2][cc] +
                    4 * s_h[2 * y + 3][cc] + s_h[2 * y + 4][cc];
            dst[((size_t)Y * dw + X) * C + (cc % C)] = (uint8_t)((v + 128) >> 8);
        }
    }
}

}  // namespace

int svo_launch_pyr_down(svo_ctx *ctx, const uint8_t *src, int w, int h, int c, uint8_t *dst)
{
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    dim3 grid((dw + TW - 1) / TW, (dh + TH - 1) / TH);
    switch (c) {
    case 1:
        hipLaunchKernelGGL(pyr_down_kernel<1>, grid, dim3(256), 0, ctx->stream, src, w, h, dst, dw, dh);
        break;
    case 3:
        hipLaunchKernelGGL(pyr_down_kernel<3>, grid, dim3(256), 0, ctx->stream, src, w, h, dst, dw, dh);
        break;
    case 4:
        hipLaunchKernelGGL(pyr_down_kernel<4>, grid, dim3(256), 0, ctx->stream, src, w, h, dst, dw, dh);
        break;
    default:
        svo_set_error("pyr_down: unsupported channel count %d (1, 3 or 4)", c);
        return SVO_ERR_ARG;
    }
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

extern "C" {

int svo_pyramid_create(svo_ctx *ctx, int width, int height, int channels, int levels,
                       svo_pyramid **out)
{
    SVO_CHECK_ARG(ctx && out);
    SVO_CHECK_ARG(width >= 2 * SVO_LK_WIN + 2 && height >= 2 * SVO_LK_WIN + 2);
    SVO_CHECK_ARG(channels == 1 || channels == 3 || channels == 4);
    SVO_CHECK_ARG(levels >= 1 && levels <= SVO_MAX_LEVELS);
    SVO_HIP(hipSetDevice(ctx->device));
    svo_pyramid *p = new svo_pyramid();
    p->w = width;
    p->h = height;
    p->c = channels;
    p->levels = levels;
    size_t off = 0;
    int w = width, h = height;
    for (int l = 0; l < levels; l++) {
        p->off[l] = off;
        p->dev.w[l] = w;
        p->dev.h[l] = h;
        off += ((size_t)w * h * channels + 255) & ~(size_t)255;
        w = (w + 1) / 2;
        h = (h + 1) / 2;
    }
    p->bytes = off;
    hipError_t e = hipMalloc((void **)&p->base, p->bytes);
    if (e != hipSuccess) {
        delete p;
        svo_set_error("hipMalloc pyramid -> %s", hipGetErrorString(e));
        return SVO_ERR_HIP;
    }
    for (int l = 0; l < SVO_MAX_LEVELS; l++)
        p->dev.lvl[l] = l < levels ? p->base + p->off[l] : nullptr;
    p->dev.levels = levels;
    p->dev.c = channels;
    *out = p;
    return SVO_OK;
}

int svo_pyramid_destroy(svo_ctx *ctx, svo_pyramid *pyr)
{
    if (!pyr)
        return SVO_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (pyr->base)
        (void)hipFree(pyr->base);
    delete pyr;
    return SVO_OK;
}

int svo_pyramid_build(svo_ctx *ctx, svo_pyramid *pyr, const uint8_t *image, int mem)
{
    SVO_CHECK_ARG(ctx && pyr && image);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    const size_t l0 = (size_t)pyr->w * pyr->h * pyr->c;
    SVO_HIP(hipMemcpyAsync(pyr->base, image, l0,
                           mem == SVO_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                           ctx->stream));
    {
        ScopedKernelTime t(ctx, SVO_K_PYRAMID);
        for (int l = 1; l < pyr->levels; l++) {
            int rc = svo_launch_pyr_down(ctx, pyr->base + pyr->off[l - 1], pyr->dev.w[l - 1],
                                         pyr->dev.h[l - 1], pyr->c, pyr->base + pyr->off[l]);
            if (rc != SVO_OK)
                return rc;
        }
    }
    if (mem == SVO_MEM_HOST)
        SVO_HIP(hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

int svo_pyramid_get_level(svo_ctx *ctx, const svo_pyramid *pyr, int level, uint8_t *out, int mem,
                          int *w, int *h)
{
    SVO_CHECK_ARG(ctx && pyr && level >= 0 && level < pyr->levels);
    if (w)
        *w = pyr->dev.w[level];
    if (h)
        *h = pyr->dev.h[level];
    if (out) {
        size_t sz = (size_t)pyr->dev.w[level] * pyr->dev.h[level] * pyr->c;
        SVO_HIP(hipMemcpyAsync(out, pyr->base + pyr->off[level], sz,
                               mem == SVO_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                               ctx->stream));
        if (mem == SVO_MEM_HOST)
            SVO_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SVO_OK;
}

}  // extern "C"
