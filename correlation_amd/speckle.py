"""Synthetic speckle image pairs (the workload generator of SURVEY.md section 8d).

N = W*H/40 Gaussian blobs, sigma 2.5 px, amplitude U(0.5,1), centres uniform; intensity
255*min(1, sum) rounded to u8.  The deformed image renders the same blobs with their
centres moved by a ground-truth affine map about the image centre.
"""
import numpy as np


def _render(h, w, xs, ys, amps, sigma):
    img = np.zeros(h * w, np.float32)
    rad = int(np.ceil(4 * sigma))
    x0 = np.floor(xs).astype(np.int64)
    y0 = np.floor(ys).astype(np.int64)
    inv = np.float32(-0.5 / (sigma * sigma))
    for dy in range(-rad, rad + 2):
        yy = y0 + dy
        ey = (yy.astype(np.float32) - ys) ** 2
        oky = (yy >= 0) & (yy < h)
        for dx in range(-rad, rad + 2):
            xx = x0 + dx
            ok = oky & (xx >= 0) & (xx < w)
            if not ok.any():
                continue
            ex = (xx[ok].astype(np.float32) - xs[ok]) ** 2
            val = amps[ok] * np.exp((ex + ey[ok]) * inv)
            np.add.at(img, yy[ok] * w + xx[ok], val)
    img = np.minimum(img, 1.0) * 255.0
    return np.rint(img).astype(np.uint8).reshape(h, w)


def _render_torch(h, w, xs, ys, amps, sigma, device):
    """Same image as _render, accumulated in float64 on the GPU (large images)."""
    import torch
    img = torch.zeros(h * w, dtype=torch.float64, device=device)
    xs_t = torch.from_numpy(xs).to(device)
    ys_t = torch.from_numpy(ys).to(device)
    am_t = torch.from_numpy(amps).to(device)
    rad = int(np.ceil(4 * sigma))
    x0 = torch.floor(xs_t).long()
    y0 = torch.floor(ys_t).long()
    inv = np.float32(-0.5 / (sigma * sigma))
    for dy in range(-rad, rad + 2):
        yy = y0 + dy
        ey = (yy.float() - ys_t) ** 2
        oky = (yy >= 0) & (yy < h)
        for dx in range(-rad, rad + 2):
            xx = x0 + dx
            ok = oky & (xx >= 0) & (xx < w)
            ex = (xx.float() - xs_t) ** 2
            val = am_t * torch.exp((ex + ey) * inv)
            img.index_add_(0, (yy * w + xx)[ok], val[ok].double())
    img = torch.clamp(img.float(), max=1.0) * 255.0
    return torch.round(img).to(torch.uint8).reshape(h, w).cpu().numpy()


def blobs(h, w, seed=7, density=40):
    rng = np.random.default_rng(seed)
    n = (h * w) // density
    xs = rng.uniform(0, w, n).astype(np.float32)
    ys = rng.uniform(0, h, n).astype(np.float32)
    amps = rng.uniform(0.5, 1.0, n).astype(np.float32)
    return xs, ys, amps


def deform(xs, ys, h, w, p):
    """Ground-truth map of blob centres: p = (u, v, ux, uy, vx, vy) about the image centre."""
    u, v, ux, uy, vx, vy = [np.float32(t) for t in p]
    cx, cy = np.float32(w / 2.0), np.float32(h / 2.0)
    dx, dy = xs - cx, ys - cy
    return xs + u + ux * dx + uy * dy, ys + v + vx * dx + vy * dy


def speckle_pair(h, w, p=(1.3, -0.7, 0.002, 0.0, 0.0, -0.001), seed=7, sigma=2.5, device=None):
    """Returns (undeformed, deformed) u8 images of shape (h, w).  device="cuda" renders with
    torch on the GPU (float64 accumulation; for the large configs)."""
    xs, ys, amps = blobs(h, w, seed)
    xd, yd = deform(xs, ys, h, w, p)
    if device is not None:
        return (_render_torch(h, w, xs, ys, amps, sigma, device),
                _render_torch(h, w, xd, yd, amps, sigma, device))
    und = _render(h, w, xs, ys, amps, sigma)
    dfm = _render(h, w, xd, yd, amps, sigma)
    return und, dfm


def speckle_sequence(h, w, n_frames, velocity=(0.8, -0.4), dilation=1e-4, seed=7, sigma=2.5, device=None):
    """Frames 0..n_frames-1 with constant-velocity translation and dilation (config C4).
    device="cuda" renders with torch on the GPU (the 64-frame 2048^2 sequence)."""
    xs, ys, amps = blobs(h, w, seed)
    frames = []
    for f in range(n_frames):
        p = (velocity[0] * f, velocity[1] * f, dilation * f, 0.0, 0.0, dilation * f)
        xd, yd = deform(xs, ys, h, w, p)
        frames.append(_render_torch(h, w, xd, yd, amps, sigma, device) if device is not None
                      else _render(h, w, xd, yd, amps, sigma))
    return frames
