"""An INDEPENDENT numpy restatement of cv::calcOpticalFlowPyrLK with default arguments, written from SURVEY.md
appendix A.1 alone (not from oracle/lk.c, not from lk.hip): pyramid, Scharr derivative levels, 14-bit fixed-point
bilinear patches, FLOAT32 accumulation of the normal matrix and of the mismatch vector, both stop rules, status.

OpenCV accumulates A11, A12, A22, b1, b2 in float32, and the ORDER of that accumulation depends on the build: the
scalar loop adds element after element; the SSE2 / AVX2 / NEON loops keep 4 or 8 partial sums per row chunk and add
them at the end.  `order` selects one of those orders, so that two "builds" of upstream can be compared with each
other and with the oracle, which sums the integer products EXACTLY and rounds once (deviation 1).

Test infrastructure (tests/test_lk_independent.py); slow by design (python loop over points), use a few hundred points.
"""
import numpy as np

W_BITS = 14
WIN = 21
HALF = np.float32(10.0)
FLT_SCALE = np.float32(1.0 / (1 << 20))
F32 = np.float32


def _reflect101(idx, n):
    idx = np.where(idx < 0, -idx, idx)
    return np.where(idx >= n, 2 * n - 2 - idx, idx)


def pyr_down(img):
    """5x5 [1 4 6 4 1] x [1 4 6 4 1] / 256 with BORDER_REFLECT_101, rounding (sum + 128) >> 8, size ((w+1)/2, (h+1)/2)."""
    h, w, c = img.shape
    k = np.array([1, 4, 6, 4, 1], np.int64)
    oh, ow = (h + 1) // 2, (w + 1) // 2
    a = img.astype(np.int64)
    cols = _reflect101(2 * np.arange(ow)[:, None] + np.arange(-2, 3)[None, :], w)      # (ow, 5)
    rows = _reflect101(2 * np.arange(oh)[:, None] + np.arange(-2, 3)[None, :], h)      # (oh, 5)
    horiz = (a[:, cols, :] * k[None, None, :, None]).sum(2)                           # (h, ow, c)
    out = (horiz[rows, :, :] * k[None, :, None, None]).sum(1)                         # (oh, ow, c)
    return ((out + 128) >> 8).astype(np.uint8)


def scharr(img):
    """dx = 3 p[y-1][x+1] - 3 p[y-1][x-1] + 10 p[y][x+1] - 10 p[y][x-1] + 3 p[y+1][x+1] - 3 p[y+1][x-1]; dy transposed;
    reflect-101 at the image border.  Returns int32 (h, w, c) arrays."""
    h, w, c = img.shape
    a = img.astype(np.int32)
    ym, yp = _reflect101(np.arange(h) - 1, h), _reflect101(np.arange(h) + 1, h)
    xm, xp = _reflect101(np.arange(w) - 1, w), _reflect101(np.arange(w) + 1, w)
    sm_y = 3 * a[ym] + 10 * a + 3 * a[yp]              # smoothed along y
    dx = sm_y[:, xp] - sm_y[:, xm]
    sm_x = 3 * a[:, xm] + 10 * a + 3 * a[:, xp]        # smoothed along x
    dy = sm_x[yp] - sm_x[ym]
    return dx, dy


def _pad_image(img, pad):
    h, w, c = img.shape
    ys = _reflect101(np.arange(-pad, h + pad), h)
    xs = _reflect101(np.arange(-pad, w + pad), w)
    return img[ys][:, xs].astype(np.int32)


def _pad_zero(a, pad):
    return np.pad(a, ((pad, pad), (pad, pad), (0, 0)))


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _weights(a, b):
    """iw00 .. iw11 from the fractional parts (float32 arithmetic, cvRound = round half to even)."""
    one = F32(1.0)
    s = F32(1 << W_BITS)
    iw00 = int(np.rint((one - a) * (one - b) * s))
    iw01 = int(np.rint(a * (one - b) * s))
    iw10 = int(np.rint((one - a) * b * s))
    return iw00, iw01, iw10, (1 << W_BITS) - iw00 - iw01 - iw10


def _patch(P, pad, ix, iy, wts, shift):
    """bilinear sample of the WIN x WIN window whose top-left integer corner is (ix, iy), descaled by `shift` bits"""
    iw00, iw01, iw10, iw11 = wts
    y0, x0 = iy + pad, ix + pad
    p00 = P[y0:y0 + WIN, x0:x0 + WIN]
    p01 = P[y0:y0 + WIN, x0 + 1:x0 + WIN + 1]
    p10 = P[y0 + 1:y0 + WIN + 1, x0:x0 + WIN]
    p11 = P[y0 + 1:y0 + WIN + 1, x0 + 1:x0 + WIN + 1]
    return _descale(p00 * iw00 + p01 * iw01 + p10 * iw10 + p11 * iw11, shift)


def _fsum(prod, order):
    """float32 sum of an integer array (rows x row elements, channels interleaved along the row as in memory) in the
    accumulation order of one upstream build:
      'scalar'  one accumulator, element after element, row after row
      'simd4'   per row: chunks of 4 elements into 4 partial sums, a scalar tail; partials + tail added at the end
      'simd8'   the same with 8 lanes"""
    rows = prod.reshape(prod.shape[0], -1).astype(np.float32)
    # np.add.accumulate adds sequentially (no pairwise blocking), in the array's dtype: exactly a running float32 sum
    if order == "scalar":
        return np.add.accumulate(rows.ravel(), dtype=np.float32)[-1]
    lanes = 4 if order == "simd4" else 8
    n = rows.shape[1]
    body = n - n % lanes
    part = np.add.accumulate(rows[:, :body].reshape(-1, lanes), axis=0, dtype=np.float32)[-1]
    tail = np.add.accumulate(rows[:, body:].ravel(), dtype=np.float32)[-1] if body < n else F32(0)
    acc = np.add.accumulate(part, dtype=np.float32)[-1]
    return F32(acc + tail)


def calc_optical_flow_pyr_lk(prev, nxt, pts, order="scalar", max_level=3, max_count=30, eps=0.01, min_eig_thr=1e-4):
    """-> (next_pts float32 (n, 2), status uint8, err float32, iterations per point [levels summed])"""
    prev_pyr, next_pyr = [prev], [nxt]
    for _ in range(max_level):
        prev_pyr.append(pyr_down(prev_pyr[-1]))
        next_pyr.append(pyr_down(next_pyr[-1]))
    pad = WIN + 2
    Ip = [_pad_image(p, pad) for p in prev_pyr]
    Jp = [_pad_image(p, pad) for p in next_pyr]
    D = [tuple(_pad_zero(d, pad) for d in scharr(p)) for p in prev_pyr]
    c = prev.shape[2]
    pts = np.asarray(pts, np.float32).reshape(-1, 2)
    n = len(pts)
    out = np.zeros((n, 2), np.float32)
    status = np.ones(n, np.uint8)
    err = np.zeros(n, np.float32)
    iters = np.zeros(n, np.int32)
    eps_sq = float(eps) * float(eps)
    for i in range(n):
        nx = ny = F32(0)
        for level in range(max_level, -1, -1):
            h, w = prev_pyr[level].shape[:2]
            sc = F32(1.0 / (1 << level))
            px, py = F32(pts[i, 0] * sc), F32(pts[i, 1] * sc)
            if level == max_level:
                nx, ny = px, py
            else:
                nx, ny = F32(nx * F32(2)), F32(ny * F32(2))
            out[i] = (nx, ny)
            px, py = F32(px - HALF), F32(py - HALF)
            ipx, ipy = int(np.floor(px)), int(np.floor(py))
            if ipx < -WIN or ipx >= w or ipy < -WIN or ipy >= h:
                if level == 0:
                    status[i] = 0
                    err[i] = 0
                continue
            wts = _weights(F32(px - F32(ipx)), F32(py - F32(ipy)))
            I = _patch(Ip[level], pad, ipx, ipy, wts, W_BITS - 5)
            Ix = _patch(D[level][0], pad, ipx, ipy, wts, W_BITS)
            Iy = _patch(D[level][1], pad, ipx, ipy, wts, W_BITS)
            A11 = F32(_fsum(Ix * Ix, order) * FLT_SCALE)
            A12 = F32(_fsum(Ix * Iy, order) * FLT_SCALE)
            A22 = F32(_fsum(Iy * Iy, order) * FLT_SCALE)
            Dd = F32(F32(A11 * A22) - F32(A12 * A12))
            dif = F32(A11 - A22)
            min_eig = F32(F32(F32(A22 + A11) - np.sqrt(F32(F32(dif * dif) + F32(F32(4) * F32(A12 * A12))), dtype=np.float32))
                          / F32(2 * WIN * WIN))
            if min_eig < F32(min_eig_thr) or Dd < np.finfo(np.float32).eps:
                if level == 0:
                    status[i] = 0
                continue
            Dd = F32(F32(1) / Dd)
            nx, ny = F32(nx - HALF), F32(ny - HALF)
            pdx = pdy = F32(0)
            for j in range(max_count):
                inx, iny = int(np.floor(nx)), int(np.floor(ny))
                if inx < -WIN or inx >= w or iny < -WIN or iny >= h:
                    if level == 0:
                        status[i] = 0
                    break
                wj = _weights(F32(nx - F32(inx)), F32(ny - F32(iny)))
                diff = _patch(Jp[level], pad, inx, iny, wj, W_BITS - 5) - I
                b1 = F32(_fsum(diff * Ix, order) * FLT_SCALE)
                b2 = F32(_fsum(diff * Iy, order) * FLT_SCALE)
                dx = F32(F32(F32(A12 * b2) - F32(A22 * b1)) * Dd)
                dy = F32(F32(F32(A12 * b1) - F32(A11 * b2)) * Dd)
                nx, ny = F32(nx + dx), F32(ny + dy)
                out[i] = (F32(nx + HALF), F32(ny + HALF))
                iters[i] += 1
                if float(dx) * float(dx) + float(dy) * float(dy) <= eps_sq:
                    break
                if j > 0 and abs(float(F32(dx + pdx))) < 0.01 and abs(float(F32(dy + pdy))) < 0.01:
                    out[i] = (F32(out[i, 0] - F32(dx * F32(0.5))), F32(out[i, 1] - F32(dy * F32(0.5))))
                    break
                pdx, pdy = dx, dy
            nx, ny = out[i]
            if level == 0 and status[i]:
                qx, qy = F32(out[i, 0] - HALF), F32(out[i, 1] - HALF)
                iqx, iqy = int(np.floor(qx)), int(np.floor(qy))
                if iqx < -WIN or iqx >= w or iqy < -WIN or iqy >= h:
                    status[i] = 0
                    continue
                wq = _weights(F32(qx - F32(iqx)), F32(qy - F32(iqy)))
                d = np.abs(_patch(Jp[level], pad, iqx, iqy, wq, W_BITS - 5) - I)
                err[i] = F32(F32(d.sum()) / F32(32 * WIN * WIN * c))
        # the guess carried between levels is nextPts of the level just finished
    return out, status, err, iters
