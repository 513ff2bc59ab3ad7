"""Deterministic synthetic rectified stereo pairs (integer arithmetic only).

The reference ships calibration files but no images, and its camera /
MJPEG front end (/root/reference/stream/, /root/reference/decoder/) is out of scope, so the
hot path is fed by this generator: frame i of a stream uses seed STREAM_SEED + i.  The same
function is implemented on the device (csrc/synth.hip, rtdm_synth_pair) and the two must agree
bit for bit (tests/test_synth.py), so every rank of a multi-GPU run can make its own shard of
the stream without any communication.

Left image  : 5-octave integer value noise (cells 16, 8, 4, 2, 1), rich texture.
Disparity   : defined on the RIGHT image grid: a sloped background plus 3..6 rectangles of
              constant disparity, all inside [2, D-3].
Right image : Left sampled at x + d(x, y), plus hash noise in [-2, 2].
Stamps      : one flat patch (texture rejection), one 8-px-period stripe patch (uniqueness
              rejection), ~20 5x5 outlier patches with a different disparity (speckle filter).
"""
import numpy as np

STREAM_SEED = 0x5EED0000
_M = np.uint64(0xFFFFFFFFFFFFFFFF)
_K = [np.uint64(k) for k in (0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB,
                             0xD6E8FEB86659FD93, 0xA0761D6478BD642F)]


def _u64(x):
    return np.asarray(x).astype(np.uint64)


def mix(seed, a, b, c):
    """hash(seed, a, b, c) -> uint64; splitmix64 finaliser over a linear key (wrapping)."""
    with np.errstate(over="ignore"):
        z = (_u64(seed) * _K[3] + _u64(a) * _K[0] + _u64(b) * _K[1] + _u64(c) * _K[2] + _K[4])
        z = (z ^ (z >> np.uint64(30))) * _K[1]
        z = (z ^ (z >> np.uint64(27))) * _K[2]
        z = z ^ (z >> np.uint64(31))
    return z


def _octave(seed, o, x, y, lg):
    """Bilinear value noise with cell 2**lg at integer coords x, y (int64 arrays) -> 0..255."""
    s = 1 << lg
    X, Y = x >> lg, y >> lg
    fx, fy = x & (s - 1), y & (s - 1)
    def lat(i, j):
        return (mix(seed, 100 + o, X + i, Y + j) & np.uint64(255)).astype(np.int64)
    a, b, c, d = lat(0, 0), lat(1, 0), lat(0, 1), lat(1, 1)
    v = a * (s - fx) * (s - fy) + b * fx * (s - fy) + c * (s - fx) * fy + d * fx * fy
    return v >> (2 * lg)


def left_value(seed, x, y):
    """Left image intensity at (x, y); x, y broadcastable int64 arrays (any integer coords)."""
    x = np.asarray(x, np.int64) + 4096  # keep lattice coordinates non-negative
    y = np.asarray(y, np.int64) + 4096
    v = (2 * _octave(seed, 0, x, y, 4) + 2 * _octave(seed, 1, x, y, 3) + 2 * _octave(seed, 2, x, y, 2)
         + _octave(seed, 3, x, y, 1) + _octave(seed, 4, x, y, 0))
    return v >> 3


def _rects(seed, W, H, D):
    """3..6 foreground rectangles (x0, y0, x1, y1, d) on the right-image grid."""
    n = 3 + int(mix(seed, 1, 0, 0) % np.uint64(4))
    out = []
    lo, hi = 2, max(3, D - 3)
    for k in range(n):
        h = [int(mix(seed, 2, k, t) % np.uint64(1 << 20)) for t in range(5)]
        rw = max(8, W // 8 + h[0] % max(1, W // 4))
        rh = max(8, H // 8 + h[1] % max(1, H // 4))
        x0 = h[2] % max(1, W - rw)
        y0 = h[3] % max(1, H - rh)
        d = lo + h[4] % (hi - lo + 1)
        out.append((x0, y0, x0 + rw, y0 + rh, d))
    return out


def disparity_right(seed, W, H, D):
    """Ground-truth disparity on the right image grid, int64 HxW, values in [2, D-3]."""
    y, x = np.mgrid[0:H, 0:W].astype(np.int64)
    lo, hi = 2, max(3, D - 3)
    base = lo + int(mix(seed, 3, 0, 0) % np.uint64(max(1, (hi - lo) // 2)))
    gx = int(mix(seed, 3, 1, 0) % np.uint64(17))          # slope, 1/1024 px per px
    gy = int(mix(seed, 3, 2, 0) % np.uint64(33))
    d = base + ((gx * x + gy * y) >> 10)
    d = np.clip(d, lo, hi)
    for (x0, y0, x1, y1, dv) in _rects(seed, W, H, D):
        d[y0:y1, x0:x1] = dv
    return d


def make_pair(seed, W, H, D):
    """Returns (left, right) uint8 HxW for one frame; D only bounds the disparity field."""
    seed = int(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.int64)
    left = left_value(seed, x, y)
    d = disparity_right(seed, W, H, D)
    right = left_value(seed, x + d, y)
    noise = (mix(seed, 4, x, y) % np.uint64(5)).astype(np.int64) - 2
    right = np.clip(right + noise, 0, 255)

    # outlier patches: 5x5 blocks of the right image re-sampled with another disparity
    lo, hi = 2, max(3, D - 3)
    for k in range(20):
        h = [int(mix(seed, 5, k, t) % np.uint64(1 << 20)) for t in range(3)]
        if W <= 16 or H <= 16:
            break
        px, py = 4 + h[0] % (W - 12), 4 + h[1] % (H - 12)
        dalt = lo + h[2] % (hi - lo + 1)
        yy, xx = np.mgrid[py:py + 5, px:px + 5].astype(np.int64)
        right[py:py + 5, px:px + 5] = left_value(seed, xx + dalt, yy)

    # flat patch (same constant in both images)
    ps = max(4, min(64, W // 4, H // 4))
    fx, fy = int(mix(seed, 6, 0, 0) % np.uint64(max(1, W - ps))), int(mix(seed, 6, 1, 0) % np.uint64(max(1, H - ps)))
    left[fy:fy + ps, fx:fx + ps] = 100
    right[fy:fy + ps, fx:fx + ps] = 100

    # stripe patch, 8-px period, identical in both images -> ambiguous matches
    sx, sy = int(mix(seed, 7, 0, 0) % np.uint64(max(1, W - ps))), int(mix(seed, 7, 1, 0) % np.uint64(max(1, H - ps)))
    stripes = np.where(((x[sy:sy + ps, sx:sx + ps] >> 2) & 1) == 1, 200, 60)
    left[sy:sy + ps, sx:sx + ps] = stripes
    right[sy:sy + ps, sx:sx + ps] = stripes
    return left.astype(np.uint8), right.astype(np.uint8)


def make_stream(first_frame, n, W, H, D, seed=STREAM_SEED):
    """Frames [first_frame, first_frame+n) of the stream -> (L, R) uint8 arrays [n, H, W]."""
    L = np.empty((n, H, W), np.uint8)
    R = np.empty((n, H, W), np.uint8)
    for i in range(n):
        L[i], R[i] = make_pair(seed + first_frame + i, W, H, D)
    return L, R
