// pyramid.hip -- 5-tap Gaussian pyramid (the pyrDown cv::calcOpticalFlowPyrLK applies to
// both images on every call; reference call sites src/tracking.cpp:18,52) and the Scharr
// derivative levels of the images that serve as a tracking pass's FIRST image.
//
// HBM layout: every level is stored with a reflect-101 border of SVO_PYR_PAD pixels and a
// 16-byte-aligned row pitch (what cv::buildOpticalFlowPyramid does with pyrBorder =
// BORDER_REFLECT_101), so the LK kernel stages its tiles with aligned 16-byte loads.
//
// FOUR launches build up to 2 x SVO_LK_MAX_JOBS pyramids (the left AND right images of a lock-step
// group of chunks), whatever their number -- the short kernels of this path are paced by launches,
// not by bytes (beside the other contexts' tracking launches every launch waits tens of
// microseconds for its first wave slots):
//   1. base:    padded level 0 (copy + border) and, in the same launch, level 1 straight from the raw
//               image (its own role of workgroups: no dependence on the padded level 0);
//   2. down:    level 2 from level 1;            3. down: level 3 from level 2
//               (one workgroup per 32x8 output tile; the (2*32+3) x (2*8+3) source tile is staged in LDS
//               with row-contiguous loads, the separable [1 4 6 4 1] filter runs LDS -> LDS (down the byte
//               columns, packed uint16) and LDS -> HBM (along the rows, (sum+128)>>8); edge tiles read the
//               INTERIOR of their source level through reflect-101 indices, so no level waits for a border);
//   4. finish:  the reflect-101 borders of levels 1.. and, in the same launch, the Scharr derivative
//               levels of the pyramids that carry them (edge pixels through reflect-101 indices again,
//               so the two roles are independent).
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
// Workgroups of ONE wave: beside a tracking launch, whose single-wave workgroups take every wave
// slot the moment it frees, a multi-wave workgroup waits until a CU has several slots free at once
// (measured: the pyramid stage 2.4x longer with 256-thread workgroups; the chip's throughput is the
// same either way -- the tracking launch gives up exactly the slots the pyramids win).
constexpr int PB = 64;
constexpr int SW = 2 * TW + 3, SH = 2 * TH + 3;
// ROWS_PER_BLOCK rows per workgroup in the byte movers: they are bound by workgroup dispatch, not by
// bandwidth (a wave per 256 bytes was 83 k waves per image)
constexpr int ROWS_PER_BLOCK = 8;
constexpr int PYR_JOBS = 2 * SVO_LK_MAX_JOBS;

// everything the four launches need, by value (2.4 KB of the 4 KB kernel-argument space)
// NJ: PYR_JOBS for the lock-step groups; 2 for a chunk on its own (its left and right image): a sixteenth of the kernel
// arguments per launch
template <int NJ> struct PyrBuildN {
    const uint8_t *img[NJ];                 // raw images, row stride w[0] * C
    uint8_t *lvl[NJ][SVO_MAX_LEVELS];       // pixel (0, 0) of the padded levels
    int *dlvl[NJ];                          // derivative levels (svo_pyramid::dbase) or null
    const int *gate[NJ];                    // optional: the pyramid is left untouched when *gate == 0 (chain runner)
    int doff[SVO_MAX_LEVELS];               // element (0, 0) of each derivative level, in ints
    int pitch[SVO_MAX_LEVELS], dpitch[SVO_MAX_LEVELS], w[SVO_MAX_LEVELS], h[SVO_MAX_LEVELS];
    int levels;
};
using PyrBuild = PyrBuildN<PYR_JOBS>;
constexpr int PYR_FEW = 2;
inline PyrBuildN<PYR_FEW> pyr_few(const PyrBuild &b)
{
    PyrBuildN<PYR_FEW> f;
    for (int a = 0; a < PYR_FEW; a++) {
        f.img[a] = b.img[a];
        for (int l = 0; l < SVO_MAX_LEVELS; l++)
            f.lvl[a][l] = b.lvl[a][l];
        f.dlvl[a] = b.dlvl[a];
        f.gate[a] = b.gate[a];
    }
    for (int l = 0; l < SVO_MAX_LEVELS; l++) {
        f.doff[l] = b.doff[l];
        f.pitch[l] = b.pitch[l];
        f.dpitch[l] = b.dpitch[l];
        f.w[l] = b.w[l];
        f.h[l] = b.h[l];
    }
    f.levels = b.levels;
    return f;
}
static_assert(sizeof(PyrBuild) <= 3072, "kernel arguments are limited to 4 KB");

// One 32x8 output tile of level l from level l-1 (RAW = false: the padded level; RAW = true: the raw
// image, rows of w * C bytes without padding or alignment).  The (2*32+3) x (2*8+3) source tile is staged in
// LDS with every row starting at byte 0 (misaligned source rows are shifted while they are staged), then
//   vertical   [1 4 6 4 1] down the byte COLUMNS, for the 8 output rows only: a thread takes one dword
//              column, five aligned dword reads, two uint16 pairs per dword and packed 16-bit arithmetic
//              (sums <= 16 * 255): 18 vector instructions per four values, no per-byte index arithmetic;
//   horizontal [1 4 6 4 1] along the rows of those sums (taps C apart), (sum + 128) >> 8, four output
//              bytes per thread and store.
// The filter is separable and the arithmetic integer, so the order of the two passes does not change a bit
// of the result; rows first (as until round 2) filtered 19 rows of which the vertical pass used 8 outputs'
// worth, and was 6 % of the vector instructions of a bench run.
template <int C, bool RAW>
__device__ __forceinline__ void down_tile(const uint8_t *__restrict__ src, int spitch, int w, int h,
                                          uint8_t *__restrict__ dst, int dpitch, int dw, int dh, int tx, int ty)
{
    constexpr int ND = (SW * C + 3) / 4;      // dwords of a staged row
    constexpr int SROW = (ND + 1) * 4;        // bytes per staged row (odd dword count: rows spread over the banks)
    constexpr int VROW = ND * 4 + 4;          // uint16 sums per row of the vertical pass's output (padded)
    __shared__ __attribute__((aligned(16))) uint8_t s_src[SH * SROW];
    __shared__ __attribute__((aligned(16))) uint16_t s_v[TH][VROW];
    const int tid = threadIdx.x;
    const int ox = tx * TW, oy = ty * TH;
    const int x0 = 2 * ox - 2, y0 = 2 * oy - 2;
    // a staged dword takes two source dwords (the shift): the furthest reaches 4 * ND + 3 bytes past the tile's
    // first byte -- inside the padded level always; inside the raw image unless the tile ends at the very last
    // bytes of the buffer
    const bool interior = x0 >= 0 && x0 + SW <= w && y0 >= 0 && y0 + SH <= h &&
                          (!RAW || (size_t)(y0 + SH - 1) * spitch + (size_t)x0 * C + (ND + 1) * 4 + 4 <= (size_t)h * spitch);
    if (interior) {
        for (int d = tid; d < ND; d += PB) {  // a thread per dword column (C = 4: 67 columns, two turns)
            const uint8_t *col = src + (ptrdiff_t)y0 * spitch + x0 * C;
#pragma unroll 1
            for (int r = 0; r < SH; r++) {
                const uint8_t *row = col + (ptrdiff_t)r * spitch;
                const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(row) & 3);
                const uint32_t *p = reinterpret_cast<const uint32_t *>(row - sh) + d;
                reinterpret_cast<uint32_t *>(s_src + r * SROW)[d] = __builtin_amdgcn_alignbyte(p[1], p[0], sh);
            }
        }
    } else {
        // an edge tile (at the small levels most tiles are): the same thread-per-dword-column shape, the four
        // source byte offsets of the column through reflect-101 once for all rows, the row index per row.  (A
        // byte per thread and turn with a division and a reflection each was 60 dependent turns: 20 us for a
        // workgroup with 2 us of filter work, and the launch lasts as long as its slowest workgroup.)
        for (int d = tid; d < ND; d += PB) {
            int sx[4];
#pragma unroll
            for (int bb = 0; bb < 4; bb++) {
                const int cc = min(4 * d + bb, SW * C - 1);  // the row's spare bytes repeat its last one
                const int px = cc / C, ch = cc - px * C;
                sx[bb] = reflect101(x0 + px, w) * C + ch;
            }
#pragma unroll 4
            for (int r = 0; r < SH; r++) {
                const uint8_t *row = src + (size_t)reflect101(y0 + r, h) * spitch;
                reinterpret_cast<uint32_t *>(s_src + r * SROW)[d] =
                    (uint32_t)row[sx[0]] | ((uint32_t)row[sx[1]] << 8) | ((uint32_t)row[sx[2]] << 16) | ((uint32_t)row[sx[3]] << 24);
            }
        }
    }
    __syncthreads();
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    for (int d = tid; d < ND; d += PB) {
        const uint32_t *colp = reinterpret_cast<const uint32_t *>(s_src) + d;
#pragma unroll
        for (int y = 0; y < TH; y++) {
            uint32_t r[5];
#pragma unroll
            for (int k = 0; k < 5; k++)
                r[k] = colp[(2 * y + k) * (SROW / 4)];
            uint32_t out[2];
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                // bytes (2 hh, 2 hh + 1) of each row's dword as a uint16 pair
                const unsigned sel = hh ? 0x0c030c02u : 0x0c010c00u;
                ushort2v v[5];
#pragma unroll
                for (int k = 0; k < 5; k++)
                    v[k] = __builtin_bit_cast(ushort2v, __builtin_amdgcn_perm(0u, r[k], sel));
                const ushort2v four = {4, 4}, six = {6, 6};
                const ushort2v acc = (v[0] + v[4]) + (v[1] + v[3]) * four + v[2] * six;
                out[hh] = __builtin_bit_cast(uint32_t, acc);
            }
            *reinterpret_cast<uint2 *>(&s_v[y][4 * d]) = make_uint2(out[0], out[1]);
        }
    }
    __syncthreads();
    // horizontal pass; a thread produces four consecutive bytes of an output row = one aligned dword
    // (ox * C and dpitch are multiples of 4), bytes beyond the level's width are not stored
    for (int i = tid; i < TH * (TW * C / 4); i += PB) {
        const int y = i / (TW * C / 4), q = i - y * (TW * C / 4);
        const int Y = oy + y;
        if (Y >= dh)
            continue;
        uint32_t out = 0;
        int nvalid = 0;
#pragma unroll
        for (int bb = 0; bb < 4; bb++) {
            const int cc = 4 * q + bb;
            const int x = cc / C, ch = cc - x * C;
            const uint16_t *sp = &s_v[y][2 * x * C + ch];  // source pixel 2x - 2 of the row, channel ch
            const int v = sp[0] + 4 * sp[C] + 6 * sp[2 * C] + 4 * sp[3 * C] + sp[4 * C];
            out |= (uint32_t)((v + 128) >> 8) << (8 * bb);
            nvalid += (ox + x) < dw ? 1 : 0;
        }
        uint8_t *drow = dst + (size_t)Y * dpitch + ox * C + 4 * q;
        if (nvalid == 4)
            *reinterpret_cast<uint32_t *>(drow) = out;
        else
            for (int bb = 0; bb < nvalid; bb++)
                drow[bb] = (uint8_t)(out >> (8 * bb));
    }
}

// one row of the padded level 0 from the raw image, one thread per output dword
template <int C>
__device__ __forceinline__ void pad_copy_row(const uint8_t *__restrict__ src, uint8_t *__restrict__ padded, int w, int h,
                                             int pitch, int row, int d)
{
    const int Y = reflect101(row - SVO_PYR_PAD, h);
    const uint8_t *srow = src + (size_t)Y * w * C;
    uint32_t out = 0;
    const int k = d * 4 - SVO_PYR_PAD * C;  // byte index inside the source row of the dword's first byte
    if (k >= 0 && k + 4 <= w * C && (Y < h - 1 || k + 8 <= w * C) && (Y > 0 || k >= 4)) {
        // interior: four consecutive source bytes, fetched as two aligned dwords (the source rows have
        // no particular alignment); neither dword reaches outside the image buffer
        const uintptr_t a = reinterpret_cast<uintptr_t>(srow + k);
        const uint32_t *p = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
        const uint32_t lo = p[0], hi = (a & 3) ? p[1] : 0u;
        out = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(a & 3));
    } else {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int cb = d * 4 + b;
            int px = cb / C, ch = cb - px * C;
            if (px < w + 2 * SVO_PYR_PAD) {
                int X = reflect101(px - SVO_PYR_PAD, w);
                out |= (uint32_t)srow[X * C + ch] << (8 * b);
            }
        }
    }
    reinterpret_cast<uint32_t *>(padded + (size_t)row * pitch)[d] = out;
}

// launch 1: workgroups [0, n_tiles) build level 1 from the raw image, the rest the padded level 0
template <int C, int NJ> __global__ __launch_bounds__(PB) void pyr_base_kernel(PyrBuildN<NJ> b, int tiles_x, int n_tiles, int pad_x)
{
    svo_chain_priority();
    const int job = blockIdx.y;
    if (b.gate[job] && *b.gate[job] == 0)
        return;
    if ((int)blockIdx.x < n_tiles) {
        if (b.levels > 1)
            down_tile<C, true>(b.img[job], b.w[0] * C, b.w[0], b.h[0], b.lvl[job][1], b.pitch[1], b.w[1], b.h[1],
                               blockIdx.x % tiles_x, blockIdx.x / tiles_x);
        return;
    }
    const int i = blockIdx.x - n_tiles;
    const int d = (i % pad_x) * PB + threadIdx.x, rb = i / pad_x;
    if (d >= (b.pitch[0] >> 2))
        return;
    uint8_t *padded = b.lvl[job][0] - (size_t)SVO_PYR_PAD * b.pitch[0] - SVO_PYR_PAD * C;
    for (int row = rb * ROWS_PER_BLOCK, rend = min(row + ROWS_PER_BLOCK, b.h[0] + 2 * SVO_PYR_PAD); row < rend; row++)
        pad_copy_row<C>(b.img[job], padded, b.w[0], b.h[0], b.pitch[0], row, d);
}

// launches 2 and 3: level l from level l - 1
template <int C, int NJ> __global__ __launch_bounds__(PB) void pyr_down_kernel(PyrBuildN<NJ> b, int l)
{
    svo_chain_priority();
    if (b.gate[blockIdx.z] && *b.gate[blockIdx.z] == 0)
        return;
    down_tile<C, false>(b.lvl[blockIdx.z][l - 1], b.pitch[l - 1], b.w[l - 1], b.h[l - 1], b.lvl[blockIdx.z][l], b.pitch[l],
                        b.w[l], b.h[l], blockIdx.x, blockIdx.y);
}

// one row of a level's reflect-101 border from its interior, one thread per dword
template <int C>
__device__ __forceinline__ void fill_border_row(uint8_t *__restrict__ level, int w, int h, int pitch, int row, int d)
{
    const bool interior_row = row >= SVO_PYR_PAD && row < SVO_PYR_PAD + h;
    const int cb0 = d * 4;
    if (interior_row && cb0 >= SVO_PYR_PAD * C && cb0 + 3 < (SVO_PYR_PAD + w) * C)
        return;  // dword entirely inside the image: already written by pyr_down
    uint8_t *prow = level + (size_t)row * pitch;
    const int Y = reflect101(row - SVO_PYR_PAD, h);
    const uint8_t *srow = level + (size_t)(Y + SVO_PYR_PAD) * pitch + SVO_PYR_PAD * C;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        int cb = cb0 + b;
        int px = cb / C, ch = cb - px * C;
        bool inside = interior_row && px >= SVO_PYR_PAD && px < SVO_PYR_PAD + w;
        if (inside)
            continue;
        uint8_t v = 0;
        if (px < w + 2 * SVO_PYR_PAD)
            v = srow[reflect101(px - SVO_PYR_PAD, w) * C + ch];
        prow[cb] = v;
    }
}

// Scharr derivatives of four consecutive channel-bytes of row y = one aligned 16-byte store of packed
// (4 dx | 4 dy << 16):
//   S(x) = 3 p(x, y-1) + 10 p(x, y) + 3 p(x, y+1),   V(x) = p(x, y+1) - p(x, y-1)
//   dx(x) = S(x+1) - S(x-1),                           dy(x) = 3 V(x-1) + 10 V(x) + 3 V(x+1)
// -- bit for bit what lk.hip derived per keypoint and level before (each pixel sat in ~5 keypoint
// tiles per level; that Scharr tile was 17 % of the tracker's VALU instructions).  Image edges: rows and
// pixels outside the image are their reflect-101 images (the level's border may not be written yet).
// `edge_off`: for a quad at the left or right end of the row (scharr_edge), the row offsets of the window's 12
// channel-bytes through reflect-101 -- computed ONCE per thread for all its rows (per row and byte they made the
// one thread of a workgroup that owns the row's end take 8 x 400 instructions: as long as the rest of the launch).
template <int C> __device__ __forceinline__ bool scharr_edge(int w, int q) { return q == 0 || 4 * q + 4 + C > w * C; }
template <int C> __device__ __forceinline__ void scharr_edge_offsets(int w, int q, int (&off)[12])
{
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const int cb = 4 * q - 4 + i;                                // channel-byte index in the row
        const int px = cb >= 0 ? cb / C : -((-cb + C - 1) / C);      // floor division
        off[i] = reflect101(px, w) * C + (cb - px * C);
    }
}

template <int C>
__device__ __forceinline__ void scharr_row(const uint8_t *__restrict__ lvl, int *__restrict__ dlvl, int pitch, int dpitch,
                                           int w, int h, int y, int q, const int (&edge_off)[12])
{
    const uint8_t *rows[3] = {lvl + (ptrdiff_t)reflect101(y - 1, h) * pitch, lvl + (ptrdiff_t)y * pitch,
                              lvl + (ptrdiff_t)reflect101(y + 1, h) * pitch};  // 4-byte aligned (pad * C and pitch are)
    uint32_t win[3][3];  // bytes 4q-4 .. 4q+7 of rows y-1, y, y+1
    if (!scharr_edge<C>(w, q)) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(rows[r]) + (q - 1);
            win[r][0] = p[0];
            win[r][1] = p[1];
            win[r][2] = p[2];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int d = 0; d < 3; d++) {
                uint32_t v = 0;
#pragma unroll
                for (int bb = 0; bb < 4; bb++)
                    v |= (uint32_t)rows[r][edge_off[4 * d + bb]] << (8 * bb);
                win[r][d] = v;
            }
    }
    // Packed 16-bit arithmetic over the window's 12 byte columns (pos 0..11 = channel-bytes 4q-4 .. 4q+7), two
    // columns per register: S = 3 (a + c) + 10 b <= 4080 and V = c - a fit int16, so do the outputs times 4.
    typedef short short2v __attribute__((ext_vector_type(2)));
    short2v S[6], V[6];  // [k] = columns (2k, 2k+1)
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const unsigned sel = (k & 1) ? 0x0c030c02u : 0x0c010c00u;  // bytes (2, 3) or (0, 1) of the dword, zero-extended
        const short2v a = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, win[0][k >> 1], sel));
        const short2v bq = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, win[1][k >> 1], sel));
        const short2v c = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, win[2][k >> 1], sel));
        const short2v three = {3, 3}, ten = {10, 10};
        S[k] = (a + c) * three + bq * ten;
        V[k] = c - a;
    }
    // the pair of columns (pos, pos + 1): a register as it is for an even pos, halves of two neighbours otherwise
    auto pair_at = [](const short2v (&P)[6], int pos) -> short2v {
        if ((pos & 1) == 0)
            return P[pos >> 1];
        return __builtin_bit_cast(short2v, __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, P[(pos >> 1) + 1]),
                                                                 __builtin_bit_cast(unsigned, P[pos >> 1]), 0x05040302u));
    };
    int out[4];
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        // stored times 4 (|4 d| <= 16320 fits int16): the tracker's descale by 2^14 of the interpolated
        // derivative then is "the high half of the sum", see lk.hip
        const short2v four = {4, 4}, twelve = {12, 12}, forty = {40, 40};
        const short2v dx = (pair_at(S, 4 + C + j) - pair_at(S, 4 - C + j)) * four;
        const short2v dy = (pair_at(V, 4 - C + j) + pair_at(V, 4 + C + j)) * twelve + pair_at(V, 4 + j) * forty;
        const unsigned ux = __builtin_bit_cast(unsigned, dx), uy = __builtin_bit_cast(unsigned, dy);
        out[j] = (int)__builtin_amdgcn_perm(uy, ux, 0x05040100u);      // (4 dx_j & 0xffff) | (4 dy_j << 16)
        out[j + 1] = (int)__builtin_amdgcn_perm(uy, ux, 0x07060302u);  // the same of column j + 1
    }
    int *drow = reinterpret_cast<int *>(reinterpret_cast<uint8_t *>(dlvl) + (ptrdiff_t)y * dpitch);
    const int rem = w * C - 4 * q;  // elements of this row from 4q on
    if (rem >= 4) {
        typedef int int4v __attribute__((ext_vector_type(4)));
        *reinterpret_cast<int4v *>(drow + 4 * q) = int4v{out[0], out[1], out[2], out[3]};
    } else {
        for (int j = 0; j < rem; j++)
            drow[4 * q + j] = out[j];
    }
}

// launch 4: blockIdx.y walks [border row blocks of levels 1..][Scharr row blocks of levels 0..]
struct FinishPlan {
    int border_y0[SVO_MAX_LEVELS + 1];  // first blockIdx.y of level l's border role (l >= 1); [levels] = end
    int scharr_y0[SVO_MAX_LEVELS + 1];  // first blockIdx.y of level l's Scharr role; [levels] = end
};
template <int C, int NJ> __global__ __launch_bounds__(PB) void pyr_finish_kernel(PyrBuildN<NJ> b, FinishPlan plan)
{
    svo_chain_priority();
    const int job = blockIdx.z, by = blockIdx.y;
    if (b.gate[job] && *b.gate[job] == 0)
        return;
    if (by < plan.border_y0[b.levels]) {
        int l = 1;
        for (int i = 2; i < b.levels; i++)
            l = by >= plan.border_y0[i] ? i : l;
        const int d = blockIdx.x * PB + threadIdx.x;
        if (d >= (b.pitch[l] >> 2))
            return;
        uint8_t *padded = b.lvl[job][l] - (size_t)SVO_PYR_PAD * b.pitch[l] - SVO_PYR_PAD * C;
        for (int row = (by - plan.border_y0[l]) * ROWS_PER_BLOCK, rend = min(row + ROWS_PER_BLOCK, b.h[l] + 2 * SVO_PYR_PAD);
             row < rend; row++)
            fill_border_row<C>(padded, b.w[l], b.h[l], b.pitch[l], row, d);
        return;
    }
    if (!b.dlvl[job])
        return;  // this pyramid carries no derivative levels (a right image)
    int l = 0;
    for (int i = 1; i < b.levels; i++)
        l = by >= plan.scharr_y0[i] ? i : l;
    const int q = blockIdx.x * PB + threadIdx.x;  // quad of channel-bytes 4q .. 4q+3 of the row
    if (4 * q >= b.w[l] * C)
        return;
    int edge_off[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (scharr_edge<C>(b.w[l], q))
        scharr_edge_offsets<C>(b.w[l], q, edge_off);
    for (int y = (by - plan.scharr_y0[l]) * ROWS_PER_BLOCK, yend = min(y + ROWS_PER_BLOCK, b.h[l]); y < yend; y++)
        scharr_row<C>(b.lvl[job][l], b.dlvl[job] + b.doff[l], b.pitch[l], b.dpitch[l], b.w[l], b.h[l], y, q, edge_off);
}

}  // namespace

static void fill_build(PyrBuild &b, int k, svo_pyramid *const *pyrs, const uint8_t *const *d_images, bool deriv_only,
                       const int *const *gates = nullptr)
{
    for (int a = 0; a < PYR_JOBS; a++)
        b.gate[a] = gates ? gates[a < k ? a : 0] : nullptr;
    const svo_pyramid *p0 = pyrs[0];
    for (int l = 0; l < SVO_MAX_LEVELS; l++) {
        b.pitch[l] = p0->dev.pitch[l];
        b.w[l] = p0->dev.w[l];
        b.h[l] = p0->dev.h[l];
        b.dpitch[l] = 0;
        b.doff[l] = 0;
    }
    b.levels = p0->levels;
    for (int a = 0; a < PYR_JOBS; a++) {
        svo_pyramid *p = pyrs[a < k ? a : 0];
        b.img[a] = d_images ? d_images[a < k ? a : 0] : nullptr;
        for (int l = 0; l < SVO_MAX_LEVELS; l++)
            b.lvl[a][l] = l < p->levels ? const_cast<uint8_t *>(p->dev.lvl[l]) : nullptr;
        const bool d = p->dbase && (deriv_only || p->want_deriv) && (p->c == 1 || p->c == 3);
        b.dlvl[a] = d ? p->dbase : nullptr;
        if (d)
            for (int l = 0; l < SVO_MAX_LEVELS; l++) {
                b.dpitch[l] = p->dpitch[l];
                b.doff[l] = (int)p->doff[l];
            }
    }
}

template <int C> static int launch_finish(svo_ctx *ctx, int k, const PyrBuild &b, bool borders)
{
    FinishPlan plan;
    int y = 0, xmax = 1;
    for (int l = 0; l <= SVO_MAX_LEVELS; l++)
        plan.border_y0[l] = plan.scharr_y0[l] = 0;
    for (int l = 1; l < b.levels; l++) {
        plan.border_y0[l] = y;
        if (borders) {
            y += (b.h[l] + 2 * SVO_PYR_PAD + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
            xmax = max(xmax, ((b.pitch[l] >> 2) + PB - 1) / PB);
        }
    }
    for (int l = b.levels; l <= SVO_MAX_LEVELS; l++)
        plan.border_y0[l] = y;
    plan.border_y0[b.levels] = y;
    bool any_deriv = false;
    for (int a = 0; a < k; a++)
        any_deriv = any_deriv || b.dlvl[a];
    for (int l = 0; l < b.levels; l++) {
        plan.scharr_y0[l] = y;
        if (any_deriv) {
            y += (b.h[l] + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
            xmax = max(xmax, ((b.w[l] * C + 3) / 4 + PB - 1) / PB);
        }
    }
    plan.scharr_y0[b.levels] = y;
    if (y == 0)
        return SVO_OK;
    if (k <= PYR_FEW)
        hipLaunchKernelGGL((pyr_finish_kernel<C, PYR_FEW>), dim3(xmax, y, k), dim3(PB), 0, ctx->stream, pyr_few(b), plan);
    else
        hipLaunchKernelGGL((pyr_finish_kernel<C, PYR_JOBS>), dim3(xmax, y, k), dim3(PB), 0, ctx->stream, b, plan);
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

// derivative levels of pyramids whose levels (and borders) are already built
int svo_build_derivatives(svo_ctx *ctx, int k, svo_pyramid *const *pyrs)
{
    if (k <= 0)
        return SVO_OK;
    if (k > PYR_JOBS) {
        svo_set_error("pyramid: at most %d per launch", PYR_JOBS);
        return SVO_ERR_ARG;
    }
    for (int a = 0; a < k; a++)
        if (!pyrs[a]->dbase || (pyrs[a]->c != 1 && pyrs[a]->c != 3)) {
            svo_set_error("pyramid: created without derivative levels");
            return SVO_ERR_STATE;
        }
    PyrBuild b;
    fill_build(b, k, pyrs, nullptr, true);
    const int rc = pyrs[0]->c == 1 ? launch_finish<1>(ctx, k, b, false) : launch_finish<3>(ctx, k, b, false);
    if (rc == SVO_OK)
        for (int a = 0; a < k; a++)
            pyrs[a]->has_deriv = true;
    return rc;
}

template <int C>
static int build_levels(svo_ctx *ctx, int k, svo_pyramid *const *pyrs, const uint8_t *const *d_images, const int *const *gates)
{
    PyrBuild b;
    fill_build(b, k, pyrs, d_images, false, gates);
    {
        const int tiles_x = b.levels > 1 ? (b.w[1] + TW - 1) / TW : 0, tiles_y = b.levels > 1 ? (b.h[1] + TH - 1) / TH : 0;
        const int n_tiles = tiles_x * tiles_y;
        const int pad_x = ((b.pitch[0] >> 2) + PB - 1) / PB,
                  pad_y = (b.h[0] + 2 * SVO_PYR_PAD + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
        if (k <= PYR_FEW)
            hipLaunchKernelGGL((pyr_base_kernel<C, PYR_FEW>), dim3(n_tiles + pad_x * pad_y, k), dim3(PB), 0, ctx->stream,
                               pyr_few(b), tiles_x > 0 ? tiles_x : 1, n_tiles, pad_x);
        else
            hipLaunchKernelGGL((pyr_base_kernel<C, PYR_JOBS>), dim3(n_tiles + pad_x * pad_y, k), dim3(PB), 0, ctx->stream, b,
                               tiles_x > 0 ? tiles_x : 1, n_tiles, pad_x);
    }
    for (int l = 2; l < b.levels; l++) {
        if (k <= PYR_FEW)
            hipLaunchKernelGGL((pyr_down_kernel<C, PYR_FEW>), dim3((b.w[l] + TW - 1) / TW, (b.h[l] + TH - 1) / TH, k),
                               dim3(PB), 0, ctx->stream, pyr_few(b), l);
        else
            hipLaunchKernelGGL((pyr_down_kernel<C, PYR_JOBS>), dim3((b.w[l] + TW - 1) / TW, (b.h[l] + TH - 1) / TH, k),
                               dim3(PB), 0, ctx->stream, b, l);
    }
    SVO_HIP(hipGetLastError());
    int rc = SVO_OK;
    if constexpr (C == 1 || C == 3)
        rc = launch_finish<C>(ctx, k, b, true);
    else {
        for (int a = 0; a < PYR_JOBS; a++)
            b.dlvl[a] = nullptr;
        rc = launch_finish<C>(ctx, k, b, true);
    }
    for (int a = 0; a < k; a++)
        pyrs[a]->has_deriv = rc == SVO_OK && b.dlvl[a] != nullptr;
    return rc;
}

// k pyramids of the same geometry from k device images, one set of launches
int svo_build_pyramids_from_device(svo_ctx *ctx, int k, svo_pyramid *const *pyrs, const uint8_t *const *d_images,
                                   const int *const *gates)
{
    if (k <= 0)
        return SVO_OK;
    if (k > PYR_JOBS) {
        svo_set_error("pyramid: at most %d per launch", PYR_JOBS);
        return SVO_ERR_ARG;
    }
    for (int a = 1; a < k; a++)
        if (pyrs[a]->w != pyrs[0]->w || pyrs[a]->h != pyrs[0]->h || pyrs[a]->c != pyrs[0]->c ||
            pyrs[a]->levels != pyrs[0]->levels) {
            svo_set_error("pyramid: the pyramids of one launch must share their geometry");
            return SVO_ERR_ARG;
        }
    ScopedKernelTime t(ctx, SVO_K_PYRAMID);
    switch (pyrs[0]->c) {
    case 1:
        return build_levels<1>(ctx, k, pyrs, d_images, gates);
    case 3:
        return build_levels<3>(ctx, k, pyrs, d_images, gates);
    case 4:
        return build_levels<4>(ctx, k, pyrs, d_images, gates);
    }
    svo_set_error("pyramid: unsupported channel count %d (1, 3 or 4)", pyrs[0]->c);
    return SVO_ERR_ARG;
}

int svo_build_pyramid_from_device(svo_ctx *ctx, svo_pyramid *pyr, const uint8_t *d_image)
{
    return svo_build_pyramids_from_device(ctx, 1, &pyr, &d_image, nullptr);
}

extern "C" {

int svo_pyramid_create(svo_ctx *ctx, int width, int height, int channels, int levels,
                       svo_pyramid **out)
{
    return svo_pyramid_create_ex(ctx, width, height, channels, levels, true, out);
}

}  // extern "C"

int svo_pyramid_create_ex(svo_ctx *ctx, int width, int height, int channels, int levels, bool want_deriv,
                          svo_pyramid **out)
{
    SVO_CHECK_ARG(ctx && out);
    SVO_CHECK_ARG(width >= 2 * SVO_LK_WIN + 2 && height >= 2 * SVO_LK_WIN + 2);
    SVO_CHECK_ARG(channels == 1 || channels == 3 || channels == 4);
    SVO_CHECK_ARG(levels >= 1 && levels <= SVO_MAX_LEVELS);
    SVO_CHECK_ARG((width >> (levels - 1)) >= 2 && (height >> (levels - 1)) >= 2);
    SVO_HIP(hipSetDevice(ctx->device));
    svo_pyramid *p = new svo_pyramid();
    p->w = width;
    p->h = height;
    p->c = channels;
    p->levels = levels;
    size_t off = 0;
    int w = width, h = height;
    for (int l = 0; l < SVO_MAX_LEVELS; l++) {
        p->off[l] = off;
        p->dev.w[l] = w;
        p->dev.h[l] = h;
        p->dev.pitch[l] = (((w + 2 * SVO_PYR_PAD) * channels + 15) / 16) * 16;
        if (l < levels)
            off += ((size_t)p->dev.pitch[l] * (h + 2 * SVO_PYR_PAD) + 255) & ~(size_t)255;
        w = (w + 1) / 2;
        h = (h + 1) / 2;
    }
    p->bytes = off + 256;  // slack: 16-byte tile loads may run a few bytes past the last row
    hipError_t e = hipMalloc((void **)&p->base, p->bytes);
    if (e != hipSuccess) {
        delete p;
        svo_set_error("hipMalloc pyramid -> %s", hipGetErrorString(e));
        return SVO_ERR_HIP;
    }
    e = hipMemsetAsync(p->base, 0, p->bytes, ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(p->base);
        delete p;
        svo_set_error("hipMemset pyramid -> %s", hipGetErrorString(e));
        return SVO_ERR_HIP;
    }
    p->dev.levels = levels;
    p->dev.c = channels;
    for (int l = 0; l < SVO_MAX_LEVELS; l++)
        p->dev.lvl[l] = l < levels ? p->origin(l) : nullptr;
    p->want_deriv = want_deriv && (channels == 1 || channels == 3);
    if (p->want_deriv) {
        size_t ints = 0;
        for (int l = 0; l < levels; l++) {
            p->dpitch[l] = (((p->dev.w[l] + 2 * SVO_DERIV_PAD) * channels * 4 + 15) / 16) * 16;
            const size_t lvl_ints = (size_t)(p->dpitch[l] / 4) * (p->dev.h[l] + 2 * SVO_DERIV_PAD);
            p->doff[l] = ints + (size_t)SVO_DERIV_PAD * (p->dpitch[l] / 4) + (size_t)SVO_DERIV_PAD * channels;
            ints += (lvl_ints + 63) & ~(size_t)63;  // levels start 256-byte aligned
        }
        const size_t dbytes = ints * 4 + 1024;  // slack: 16-byte tile loads run a few vectors past a row
        e = hipMalloc((void **)&p->dbase, dbytes);
        if (e == hipSuccess)
            e = hipMemsetAsync(p->dbase, 0, dbytes, ctx->stream);  // the zero border is never written again
        if (e != hipSuccess) {
            (void)hipFree(p->dbase);
            (void)hipFree(p->base);
            delete p;
            svo_set_error("hipMalloc derivative levels -> %s", hipGetErrorString(e));
            return SVO_ERR_HIP;
        }
    }
    *out = p;
    return SVO_OK;
}

extern "C" {

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
    if (pyr->dbase)
        (void)hipFree(pyr->dbase);
    delete pyr;
    return SVO_OK;
}

int svo_pyramid_build(svo_ctx *ctx, svo_pyramid *pyr, const uint8_t *image, int mem)
{
    SVO_CHECK_ARG(ctx && pyr && image);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    const size_t l0 = (size_t)pyr->w * pyr->h * pyr->c;
    const uint8_t *d_image = image;
    if (mem == SVO_MEM_HOST) {
        int rc = ctx->s_img.ensure(l0);
        if (rc)
            return rc;
        SVO_HIP(hipMemcpyAsync(ctx->s_img.p, image, l0, hipMemcpyHostToDevice, ctx->stream));
        d_image = ctx->s_img.as<uint8_t>();
    }
    int rc = svo_build_pyramid_from_device(ctx, pyr, d_image);
    if (rc)
        return rc;
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
        const size_t rowb = (size_t)pyr->dev.w[level] * pyr->c;
        SVO_HIP(hipMemcpy2DAsync(out, rowb, pyr->origin(level), pyr->dev.pitch[level], rowb,
                                 pyr->dev.h[level],
                                 mem == SVO_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                                 ctx->stream));
        if (mem == SVO_MEM_HOST)
            SVO_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SVO_OK;
}

}  // extern "C"
