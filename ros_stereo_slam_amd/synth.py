"""Synthetic stereo sequences with ground-truth poses.

No dataset ships with the reference (``.MISSING_LARGE_BLOBS``) and none can be fetched, so
the benchmark and the parity tests ray-cast an analytic world: a ground plane below the
camera, two side walls (optionally two end walls, which turns the corridor into a yard a
trajectory can loop in) and a constant sky.  Surface albedo is seeded value noise of the
world coordinates, band-limited against the pixel footprint so that far texture does not
alias.  Camera intrinsics default to the reference's KITTI values
(``include/visualSLAM.h:68,82-87``): fx = fy = 718.856, cx = 607.1928, cy = 185.2157,
baseline 0.54 m, 1241 x 376.

Conventions: camera frame x right, y down, z forward (KITTI); a pose is camera-in-world,
``X_w = R @ X_c + t``.  The right camera sits at ``+baseline`` along the camera x axis,
matching ``P2 = K [I | (-b, 0, 0)]`` of ``src/triangulation.cpp:142-149``.
"""
from __future__ import annotations

import dataclasses

import numpy as np

KITTI_K = (718.856, 718.856, 607.1928, 185.2157)
KITTI_BASELINE = 0.54
KITTI_SIZE = (1241, 376)
DEFAULT_SEED = 20261003
# colour mode of Scene / textured_pair: seed offset, amplitude share and sky tint of a channel's own texture
CHANNEL_SEED = 15485863
COLOUR_MIX = 0.45
SKY_TINT = (38, -9, -52)          # B, G, R


def _hash2(ix: np.ndarray, iy: np.ndarray, seed: int) -> np.ndarray:
    """Integer lattice hash -> float in [0, 1).  Pure uint32 arithmetic (deterministic)."""
    with np.errstate(over="ignore"):
        h = ix.astype(np.uint32) * np.uint32(0x9E3779B1) ^ iy.astype(np.uint32) * np.uint32(0x85EBCA77)
        h ^= np.uint32(seed & 0xFFFFFFFF)
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x2C1B3C6D)
        h ^= h >> np.uint32(12)
        h *= np.uint32(0x297A2D39)
        h ^= h >> np.uint32(15)
    return (h >> np.uint32(8)).astype(np.float64) * (1.0 / (1 << 24))


def _value_noise(u: np.ndarray, v: np.ndarray, seed: int) -> np.ndarray:
    """Smooth value noise with unit lattice spacing, range [0, 1)."""
    fu, fv = np.floor(u), np.floor(v)
    iu, iv = fu.astype(np.int64), fv.astype(np.int64)
    a, b = u - fu, v - fv
    a = a * a * (3.0 - 2.0 * a)
    b = b * b * (3.0 - 2.0 * b)
    n00 = _hash2(iu, iv, seed)
    n10 = _hash2(iu + 1, iv, seed)
    n01 = _hash2(iu, iv + 1, seed)
    n11 = _hash2(iu + 1, iv + 1, seed)
    return (n00 * (1 - a) + n10 * a) * (1 - b) + (n01 * (1 - a) + n11 * a) * b


@dataclasses.dataclass
class Scene:
    """Ground plane y = +ground_y, walls x = +-wall_x, optional end walls z = z_min / z_max."""

    seed: int = DEFAULT_SEED
    ground_y: float = 1.65
    wall_x: float = 6.0
    z_min: float | None = None
    z_max: float | None = None
    wavelengths: tuple = (1.2, 0.45, 0.15)
    sky: int = 128
    # colour=False: R = G = B (every channel is the grey albedo).  colour=True: a hue-varying albedo -- channel c
    # (memory order B, G, R as cv::imread delivers KITTI's image_2/3, src/keyFrameManagement.cpp:52-54) mixes the
    # shared texture with a texture of its own (COLOUR_MIX of the amplitude) and the sky gets a tint, so that no two
    # channels of any pixel neighbourhood agree and a channel-stride or channel-order slip shows in the results.
    colour: bool = False

    def albedo(self, u, v, footprint, surface_id, channel=None):
        """Value-noise albedo in [0,255]; octaves fade out when the pixel footprint nears them.
        ``channel`` (colour mode): the channel's own texture is mixed in."""
        acc = np.zeros_like(u)
        wsum = np.zeros_like(u)
        amp = 1.0
        for k, wl in enumerate(self.wavelengths):
            fade = np.clip(1.5 - 2.0 * footprint / wl, 0.0, 1.0)
            n = _value_noise(u / wl, v / wl, self.seed + 7919 * surface_id + 104729 * k)
            if channel is not None:
                own = _value_noise(u / wl, v / wl, self.seed + 7919 * surface_id + 104729 * k + CHANNEL_SEED * (channel + 1))
                n = (1.0 - COLOUR_MIX) * n + COLOUR_MIX * own
            acc += amp * fade * (n - 0.5)
            wsum += amp
            amp *= 0.7
        return np.clip(128.0 + 230.0 * acc / wsum * 1.6, 0, 255)

    def render(self, R, t, K=KITTI_K, size=KITTI_SIZE, channels=3):
        """Ray-cast one view.  Returns (H, W, C) uint8 and the depth map (z in camera frame)."""
        fx, fy, cx, cy = K
        w, h = size
        R = np.asarray(R, np.float64).reshape(3, 3)
        t = np.asarray(t, np.float64).reshape(3)
        uu, vv = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        dc = np.stack([(uu - cx) / fx, (vv - cy) / fy, np.ones_like(uu)], -1)
        dw = dc @ R.T
        best_t = np.full((h, w), np.inf)
        tex_u = np.zeros((h, w))
        tex_v = np.zeros((h, w))
        surf = np.zeros((h, w), np.int64)

        def hit(tt, valid, u, v, sid):
            nonlocal best_t, tex_u, tex_v, surf
            m = valid & (tt > 1e-6) & (tt < best_t)
            best_t = np.where(m, tt, best_t)
            tex_u = np.where(m, u, tex_u)
            tex_v = np.where(m, v, tex_v)
            surf = np.where(m, sid, surf)

        with np.errstate(divide="ignore", invalid="ignore"):
            tt = (self.ground_y - t[1]) / dw[..., 1]
            hit(tt, dw[..., 1] > 1e-9, t[0] + tt * dw[..., 0], t[2] + tt * dw[..., 2], 1)
            for sid, xw in ((2, -self.wall_x), (3, self.wall_x)):
                tt = (xw - t[0]) / dw[..., 0]
                yw = t[1] + tt * dw[..., 1]
                hit(tt, (np.abs(dw[..., 0]) > 1e-9) & (yw < self.ground_y) & (yw > self.ground_y - 8.0),
                    yw, t[2] + tt * dw[..., 2], sid)
            for sid, zw in ((4, self.z_min), (5, self.z_max)):
                if zw is None:
                    continue
                tt = (zw - t[2]) / dw[..., 2]
                yw = t[1] + tt * dw[..., 1]
                hit(tt, (np.abs(dw[..., 2]) > 1e-9) & (yw < self.ground_y) & (yw > self.ground_y - 8.0),
                    t[0] + tt * dw[..., 0], yw, sid)
        found = np.isfinite(best_t)
        depth = np.where(found, best_t, 0.0)  # dc has z = 1, so ray parameter == camera depth
        footprint = depth / fx * 1.5

        def shade(channel):
            grey = np.zeros((h, w))
            for sid in (1, 2, 3, 4, 5):
                m = surf == sid
                if m.any():
                    grey = np.where(m, self.albedo(tex_u, tex_v, footprint, sid, channel), grey)
            sky = float(self.sky) if channel is None else float(np.clip(self.sky + SKY_TINT[channel % 3], 0, 255))
            return np.rint(np.where(found, grey, sky)).astype(np.uint8)

        if self.colour and channels > 1:
            img = np.stack([shade(c) for c in range(channels)], axis=2)
        else:
            img = np.repeat(shade(None)[..., None], channels, axis=2)
        return np.ascontiguousarray(img), depth

    def stereo(self, R, t, K=KITTI_K, size=KITTI_SIZE, channels=3, baseline=KITTI_BASELINE):
        R = np.asarray(R, np.float64).reshape(3, 3)
        t = np.asarray(t, np.float64).reshape(3)
        left, depth = self.render(R, t, K, size, channels)
        right, _ = self.render(R, t + R @ np.array([baseline, 0.0, 0.0]), K, size, channels)
        return left, right, depth


# ---- the same ray-caster on torch tensors (GPU): bench.py and the full-size tests render thousands of
# frames, which the numpy version (about 1 s per stereo pair) cannot supply in time.  Same arithmetic in
# float64 / wrapped 32-bit integer hashing; images agree with the numpy renderer up to last-ulp effects
# of the ray-direction product (tests/test_synth_torch.py).
def _t_hash2(ix, iy, seed: int):
    import torch

    M = 0xFFFFFFFF
    h = ((ix & M) * 0x9E3779B1 & M) ^ ((iy & M) * 0x85EBCA77 & M)
    h = h ^ (seed & M)
    h = h ^ (h >> 15)
    h = (h * 0x2C1B3C6D) & M
    h = h ^ (h >> 12)
    h = (h * 0x297A2D39) & M
    h = h ^ (h >> 15)
    return (h >> 8).to(torch.float64) * (1.0 / (1 << 24))


def _t_value_noise(u, v, seed: int):
    import torch

    fu, fv = torch.floor(u), torch.floor(v)
    iu, iv = fu.to(torch.int64), fv.to(torch.int64)
    a, b = u - fu, v - fv
    a = a * a * (3.0 - 2.0 * a)
    b = b * b * (3.0 - 2.0 * b)
    n00 = _t_hash2(iu, iv, seed)
    n10 = _t_hash2(iu + 1, iv, seed)
    n01 = _t_hash2(iu, iv + 1, seed)
    n11 = _t_hash2(iu + 1, iv + 1, seed)
    return (n00 * (1 - a) + n10 * a) * (1 - b) + (n01 * (1 - a) + n11 * a) * b


def render_torch(scene: "Scene", R, t, K=KITTI_K, size=KITTI_SIZE, channels=3, device="cuda"):
    """Scene.render for a batch of poses on a torch device.  R: (B, 3, 3), t: (B, 3) array-likes.
    Returns a (B, H, W, C) uint8 tensor on ``device``."""
    import torch

    fx, fy, cx, cy = K
    w, h = size
    R = torch.as_tensor(np.asarray(R, np.float64).reshape(-1, 3, 3), device=device)
    t = torch.as_tensor(np.asarray(t, np.float64).reshape(-1, 3), device=device)
    B = R.shape[0]
    vv, uu = torch.meshgrid(torch.arange(h, dtype=torch.float64, device=device),
                            torch.arange(w, dtype=torch.float64, device=device), indexing="ij")
    dc = [((uu - cx) / fx)[None], ((vv - cy) / fy)[None], torch.ones_like(uu)[None]]
    dw = [dc[0] * R[:, j, 0, None, None] + dc[1] * R[:, j, 1, None, None] + dc[2] * R[:, j, 2, None, None]
          for j in range(3)]
    tx, ty, tz = (t[:, k, None, None] for k in range(3))
    inf = float("inf")
    best_t = torch.full((B, h, w), inf, dtype=torch.float64, device=device)
    tex_u = torch.zeros((B, h, w), dtype=torch.float64, device=device)
    tex_v = torch.zeros_like(tex_u)
    surf = torch.zeros((B, h, w), dtype=torch.int64, device=device)

    def hit(tt, valid, u, v, sid):
        nonlocal best_t, tex_u, tex_v, surf
        m = valid & (tt > 1e-6) & (tt < best_t)
        best_t = torch.where(m, tt, best_t)
        tex_u = torch.where(m, u, tex_u)
        tex_v = torch.where(m, v, tex_v)
        surf = torch.where(m, torch.full_like(surf, sid), surf)

    tt = (scene.ground_y - ty) / dw[1]
    hit(tt, dw[1] > 1e-9, tx + tt * dw[0], tz + tt * dw[2], 1)
    for sid, xw in ((2, -scene.wall_x), (3, scene.wall_x)):
        tt = (xw - tx) / dw[0]
        yw = ty + tt * dw[1]
        hit(tt, (dw[0].abs() > 1e-9) & (yw < scene.ground_y) & (yw > scene.ground_y - 8.0), yw, tz + tt * dw[2], sid)
    for sid, zw in ((4, scene.z_min), (5, scene.z_max)):
        if zw is None:
            continue
        tt = (zw - tz) / dw[2]
        yw = ty + tt * dw[1]
        hit(tt, (dw[2].abs() > 1e-9) & (yw < scene.ground_y) & (yw > scene.ground_y - 8.0), tx + tt * dw[0], yw, sid)
    found = torch.isfinite(best_t)
    depth = torch.where(found, best_t, torch.zeros_like(best_t))
    footprint = depth / fx * 1.5

    def shade(channel):
        grey = torch.zeros_like(depth)
        for sid in (1, 2, 3, 4, 5):
            m = surf == sid
            acc = torch.zeros_like(depth)
            wsum = 0.0
            amp = 1.0
            for k, wl in enumerate(scene.wavelengths):
                fade = torch.clamp(1.5 - 2.0 * footprint / wl, 0.0, 1.0)
                n = _t_value_noise(tex_u / wl, tex_v / wl, scene.seed + 7919 * sid + 104729 * k)
                if channel is not None:
                    own = _t_value_noise(tex_u / wl, tex_v / wl,
                                         scene.seed + 7919 * sid + 104729 * k + CHANNEL_SEED * (channel + 1))
                    n = (1.0 - COLOUR_MIX) * n + COLOUR_MIX * own
                acc = acc + amp * fade * (n - 0.5)
                wsum += amp
                amp *= 0.7
            alb = torch.clamp(128.0 + 230.0 * acc / wsum * 1.6, 0, 255)
            grey = torch.where(m, alb, grey)
        sky = float(scene.sky) if channel is None else float(min(255, max(0, scene.sky + SKY_TINT[channel % 3])))
        grey = torch.where(found, grey, torch.full_like(grey, sky))
        return torch.round(grey).to(torch.uint8)  # round half to even, as np.rint

    if scene.colour and channels > 1:
        return torch.stack([shade(c) for c in range(channels)], dim=3).contiguous()
    return shade(None)[..., None].expand(B, h, w, channels).contiguous()


def stereo_torch(scene: "Scene", poses, K=KITTI_K, size=KITTI_SIZE, channels=3, baseline=KITTI_BASELINE,
                 device="cuda", batch: int = 8):
    """[(R, t)] -> (lefts, rights): two lists of (H, W, C) uint8 tensors on ``device``."""
    lefts, rights = [], []
    for s in range(0, len(poses), batch):
        Rs = np.stack([np.asarray(R, np.float64).reshape(3, 3) for R, _ in poses[s:s + batch]])
        ts = np.stack([np.asarray(t, np.float64).reshape(3) for _, t in poses[s:s + batch]])
        tr = ts + Rs @ np.array([baseline, 0.0, 0.0])
        both = render_torch(scene, np.concatenate([Rs, Rs]), np.concatenate([ts, tr]), K, size, channels, device)
        n = len(Rs)
        lefts.extend(both[i] for i in range(n))
        rights.extend(both[n + i] for i in range(n))
    return lefts, rights


def rot_y(a: float) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def corridor_trajectory(n: int, step: float = 0.9, yaw_amp: float = 0.02, sway: float = 0.6,
                        period: float = 60.0):
    """Forward motion along +z with slow sinusoidal yaw and lateral sway.  Returns [(R, t)]."""
    poses = []
    for i in range(n):
        ph = 2 * np.pi * i / period
        yaw = yaw_amp * period / (2 * np.pi) * 0.25 * np.sin(ph)
        x = sway * np.sin(ph * 0.5)
        poses.append((rot_y(yaw), np.array([x, 0.0, step * i])))
    return poses


def loop_trajectory(n: int, half_x: float = 14.0, half_z: float = 30.0, radius: float = 8.0,
                    step: float = 0.9, closed: bool = False):
    """Rounded-rectangle loop (camera looks along its direction of travel).  Returns [(R, t)].

    Use with ``Scene(wall_x=half_x+8, z_min=-half_z-8, z_max=half_z+8)``.  ``closed``: the step is
    adjusted so that a lap is a whole number of frames, i.e. every later lap revisits exactly the
    poses of the first (a loop-closure edge with identity measurement is then exact).
    """
    # build the closed path as straight segments + quarter circles, parametrised by arc length
    sx, sz = half_x - radius, half_z - radius
    segs = []  # (kind, length, data)
    segs.append(("line", 2 * sz, (np.array([half_x, -sz]), np.array([0.0, 1.0]))))
    segs.append(("arc", 0.5 * np.pi * radius, (np.array([sx, sz]), 0.0)))
    segs.append(("line", 2 * sx, (np.array([sx, half_z]), np.array([-1.0, 0.0]))))
    segs.append(("arc", 0.5 * np.pi * radius, (np.array([-sx, sz]), 0.5 * np.pi)))
    segs.append(("line", 2 * sz, (np.array([-half_x, sz]), np.array([0.0, -1.0]))))
    segs.append(("arc", 0.5 * np.pi * radius, (np.array([-sx, -sz]), np.pi)))
    segs.append(("line", 2 * sx, (np.array([-sx, -half_z]), np.array([1.0, 0.0]))))
    segs.append(("arc", 0.5 * np.pi * radius, (np.array([sx, -sz]), 1.5 * np.pi)))
    total = sum(s[1] for s in segs)
    lap = max(1, int(round(total / step)))
    if closed:
        step = total / lap
    poses = []
    for i in range(n):
        s = (step * (i % lap) if closed else step * i) % total
        for kind, length, data in segs:
            if s <= length:
                if kind == "line":
                    p0, d = data
                    pos = p0 + d * s
                    heading = d
                else:
                    c, a0 = data
                    a = a0 + s / radius
                    pos = c + radius * np.array([np.cos(a), np.sin(a)])
                    heading = np.array([-np.sin(a), np.cos(a)])
                break
            s -= length
        yaw = np.arctan2(heading[0], heading[1])  # rotation about y taking +z to heading
        poses.append((rot_y(yaw), np.array([pos[0], 0.0, pos[1]])))
    return poses


def textured_pair(w: int, h: int, c: int, shift=(0.0, 0.0), seed: int = 1, wavelength: float = 9.0,
                  colour: bool = False):
    """Two images of one band-limited texture, the second shifted by ``shift`` pixels
    (image2(x, y) = image1(x - dx, y - dy)); the analytic flow is ``shift`` everywhere.
    ``colour``: every channel mixes the shared texture with one of its own (see ``Scene.colour``)."""
    uu, vv = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))

    def tex(u, v, channel=None):
        acc = 0.0
        amp, wl = 1.0, wavelength * 4
        for k in range(3):
            n = _value_noise(u / wl, v / wl, seed + 31 * k)
            if channel is not None:
                n = (1.0 - COLOUR_MIX) * n + COLOUR_MIX * _value_noise(u / wl, v / wl,
                                                                        seed + 31 * k + CHANNEL_SEED * (channel + 1))
            acc = acc + amp * (n - 0.5)
            amp *= 0.6
            wl *= 0.5
        return np.clip(128 + 200 * acc / 1.96, 0, 255)

    if colour and c > 1:
        a = np.stack([np.rint(tex(uu, vv, ch)).astype(np.uint8) for ch in range(c)], 2)
        b = np.stack([np.rint(tex(uu - shift[0], vv - shift[1], ch)).astype(np.uint8) for ch in range(c)], 2)
        return np.ascontiguousarray(a), np.ascontiguousarray(b)
    a = np.rint(tex(uu, vv)).astype(np.uint8)
    b = np.rint(tex(uu - shift[0], vv - shift[1])).astype(np.uint8)
    a = np.ascontiguousarray(np.repeat(a[..., None], c, 2))
    b = np.ascontiguousarray(np.repeat(b[..., None], c, 2))
    return a, b


def loop_closures(poses, max_dist: float = 2.0, max_angle_deg: float = 10.0, min_gap: int = 100,
                  pick: str = "earliest"):
    """Stand-in for the reference's DBoW2 loop detector (vocabularies are not in the checkout):
    for every frame the earliest (``pick="earliest"``) or the closest (``"nearest"``) frame more than
    ``min_gap`` frames back whose pose lies within ``max_dist`` metres and ``max_angle_deg`` degrees,
    else -1."""
    n = len(poses)
    T = np.array([t for _, t in poses], np.float64).reshape(n, 3)
    Rm = np.array([R for R, _ in poses], np.float64).reshape(n, 9)
    cosmax = np.cos(np.radians(max_angle_deg))
    out = []
    for i in range(n):
        m = i - min_gap
        if m <= 0:
            out.append(-1)
            continue
        d = np.linalg.norm(T[:m] - T[i], axis=1)
        cosang = (Rm[:m] @ Rm[i] - 1) / 2  # trace(Ri^T Rj) = <Ri, Rj>
        ok = np.nonzero((d <= max_dist) & (cosang >= cosmax))[0]
        if ok.size == 0:
            out.append(-1)
        elif pick == "nearest":
            out.append(int(ok[np.argmin(d[ok])]))
        else:
            out.append(int(ok[0]))
    return out


# the benchmark's stream: 0.9 m per frame, yaw <= 0.02 rad per frame (radius 45 m),
# closing a rounded-rectangle loop of 492 frames; use with BENCH_SCENE
BENCH_LOOP = dict(half_x=50.0, half_z=80.0, radius=45.0, step=0.9, closed=True)


def bench_scene(seed: int = DEFAULT_SEED, colour: bool = False) -> Scene:
    return Scene(seed=seed, wall_x=BENCH_LOOP["half_x"] + 8, z_min=-BENCH_LOOP["half_z"] - 8,
                 z_max=BENCH_LOOP["half_z"] + 8, colour=colour)
