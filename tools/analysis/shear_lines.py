#!/usr/bin/env python3
"""CPU study (no GPU): distinct 128-byte lines one wave's gathers touch per sample step, for tile shapes / layouts / lane orders on the rotated camera.
Question: would 32 x 2 tiles SHEARED along the screen image of the volume's x rows, with skewed lock step, make the linear layout usable off the memory axis?
usage: python3 tools/analysis/shear_lines.py"""
import os, sys
import numpy as np
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python"))
import oracle_lib as O
import volviz_amd as vv

N, W, H, STEP = 1024, 1920, 1080, 1.0 / 512


def rays_of(cam, pix):
    rs = vv.analytic_rays(cam)
    out = []
    for (x, y) in pix:
        f, b = O.ray_endpoints(rs, cam, W, H, int(x), int(y))
        out.append((f.astype(np.float64), b.astype(np.float64)))
    return out


def lines_linear(p):            # p: [n, 3] voxel coordinates (continuous); 2x2x2 footprint; line = 32 voxels in x
    ix = np.floor(p[:, 0] - 0.5).astype(int); iy = np.floor(p[:, 1] - 0.5).astype(int); iz = np.floor(p[:, 2] - 0.5).astype(int)
    s = set()
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                for a, b, c in zip((ix + dx) >> 5, iy + dy, iz + dz):
                    s.add((a, b, c))
    return len(s)


def lines_bricked(p):           # 4x4x4 bricks, 2 lines per brick (layers z&3 < 2 / >= 2); halo voxel in x: x pair always inside
    ix = np.floor(p[:, 0] - 0.5).astype(int); iy = np.floor(p[:, 1] - 0.5).astype(int); iz = np.floor(p[:, 2] - 0.5).astype(int)
    s = set()
    for dz in (0, 1):
        for dy in (0, 1):
            for a, b, c in zip(ix >> 2, (iy + dy) >> 2, iz + dz):
                s.add((a, b, c >> 2, (c & 3) >> 1))
    return len(s)


def study(cam, name):
    eye = np.array(cam.origin, np.float64)
    rs = vv.analytic_rays(cam)
    # screen slope of the image of a volume x row near the cube centre: project c +- dx
    def to_screen(pt):          # brute force: nearest pixel whose ray passes closest (coarse grid then refine) -- use the camera basis instead
        return None
    # slope from two rays: find d(screen)/d(x) numerically with the pixel -> ray map (front points on a plane): use pixels around the centre
    cx, cy = W // 2, H // 2
    (f0, b0), (f1, b1), (f2, b2) = rays_of(cam, [(cx, cy), (cx + 64, cy), (cx, cy + 64)])
    d0 = b0 - f0; d0 /= np.linalg.norm(d0)
    # points at the cube centre depth along each ray
    t = np.dot(np.array([0.5, 0.5, 0.5]) - eye, d0)
    def at_depth(f, b):
        d = b - f; d /= np.linalg.norm(d); return eye + d * (t / np.dot(d, d0))
    p0, p1, p2 = at_depth(f0, b0), at_depth(f1, b1), at_depth(f2, b2)
    ex, ey = (p1 - p0) / 64.0, (p2 - p0) / 64.0          # world motion per pixel in screen x / y at the centre depth
    # want screen direction (1, m) whose world motion has no component outside span(x_hat, view dir)
    xh = np.array([1.0, 0, 0])
    nrm = np.cross(xh, d0); nrm /= np.linalg.norm(nrm)   # normal of the plane through the eye containing x_hat direction (locally)
    m = -np.dot(ex, nrm) / np.dot(ey, nrm)
    print(f"== {name}: eye {np.round(eye, 3)}, view dir {np.round(d0, 3)}, |x_hat . view| = {abs(d0[0]):.3f}, slope m of x rows on screen = {m:.3f}")
    axis = 2 if abs(d0[2]) >= abs(d0[1]) else 1
    rng = np.random.default_rng(1)
    res = {k: [] for k in ("8x8 bricked", "8x8 linear", "32x2 linear", "32x2 linear skew", "32x2 sheared skew (global m)", "32x2 sheared skew (local m)")}
    for _ in range(40):
        X0 = int(rng.integers(760, 1130)) // 32 * 32; Y0 = int(rng.integers(380, 690)) // 8 * 8
        # local slope at this tile
        (g0, h0), (g1, h1), (g2, h2) = rays_of(cam, [(X0, Y0), (X0 + 32, Y0), (X0, Y0 + 32)])
        dd = h0 - g0; dd /= np.linalg.norm(dd)
        tt = np.dot(np.array([0.5, 0.5, 0.5]) - eye, dd)
        def atd(f, b):
            d = b - f; d /= np.linalg.norm(d); return eye + d * (tt / np.dot(d, dd))
        q0, q1, q2 = atd(g0, h0), atd(g1, h1), atd(g2, h2)
        exl, eyl = (q1 - q0) / 32.0, (q2 - q0) / 32.0
        nl = np.cross(xh, dd); nl /= np.linalg.norm(nl)
        ml = -np.dot(exl, nl) / np.dot(eyl, nl)
        if not np.isfinite(ml) or abs(ml) > 1.5: continue
        tiles = {
            "8x8": [(X0 + i, Y0 + j) for j in range(8) for i in range(8)],
            "32x2": [(X0 + i, Y0 + j) for j in range(2) for i in range(32)],
            "sh_g": [(X0 + i, Y0 + j + int(np.floor(i * m))) for j in range(2) for i in range(32)],
            "sh_l": [(X0 + i, Y0 + j + int(np.floor(i * ml))) for j in range(2) for i in range(32)],
        }
        rr = {k: rays_of(cam, v) for k, v in tiles.items()}
        if any(np.linalg.norm(b - f) < 1e-6 for v in rr.values() for (f, b) in v): continue
        for depth in np.linspace(-0.35, 0.35, 5):
            R = tt + depth
            def pos(rays, skew):
                P = []
                ref = None
                for (f, b) in rays:
                    d = b - f; d /= np.linalg.norm(d)
                    p = eye + d * R
                    if skew:
                        if ref is None: ref = p[axis]
                        o = np.round((ref - p[axis]) / (d[axis] * STEP))
                        p = eye + d * (R + o * STEP)
                    P.append(p * N)
                return np.array(P)
            res["8x8 bricked"].append(lines_bricked(pos(rr["8x8"], False)))
            res["8x8 linear"].append(lines_linear(pos(rr["8x8"], False)))
            res["32x2 linear"].append(lines_linear(pos(rr["32x2"], False)))
            res["32x2 linear skew"].append(lines_linear(pos(rr["32x2"], True)))
            res["32x2 sheared skew (global m)"].append(lines_linear(pos(rr["sh_g"], True)))
            res["32x2 sheared skew (local m)"].append(lines_linear(pos(rr["sh_l"], True)))
    for k, v in res.items():
        print(f"   {k:32s} lines per wave-step: mean {np.mean(v):6.1f}  p90 {np.percentile(v, 90):6.1f}")


study(vv.Camera(), "view a (along the memory axis)")
study(vv.Camera.orbit(4.0, np.pi / 3, np.pi / 5), "view b (theta 60, phi 36)")
study(vv.Camera.orbit(4.0, np.pi / 4, np.pi / 4), "orbit 45,45")
study(vv.Camera.orbit(4.0, np.pi / 2, -np.pi / 4), "orbit 90,-45")
