"""Second, independent CPU implementation, written straight from the mathematical definition
 in numpy: O(W*H*D*w*w) cost volume, argmin, the three rejection tests,
row-local left-right check, label-propagation speckle filter, hand-listed ellipse rows.

It shares no code with oracle/*.c and exists only to cross-check that restatement on small
images ("a second, independent brute-force implementation ... must agree
bit-for-bit").  Parity vs the real OpenCV-backed SWMatcherKonolige remains unpinned.
"""
import numpy as np


def prefilter_xsobel(img, cap):
    img = img.astype(np.int64)
    H, W = img.shape
    out = np.full((H, W), cap, np.int64)
    npair = (H // 2) * 2 if H >= 2 else 0
    for y in range(npair):
        ya = y - 1 if y > 0 else 1
        yb = y + 1 if y < H - 1 else H - 2
        g = (img[ya, 2:] - img[ya, :-2]) + 2 * (img[y, 2:] - img[y, :-2]) + (img[yb, 2:] - img[yb, :-2])
        out[y, 1:W - 1] = np.clip(g, -cap, cap) + cap
    return out.astype(np.uint8)


def valid_rect(W, H, D, minD, w, roi1=None, roi2=None):
    r1 = roi1 if roi1 and roi1[2] * roi1[3] > 0 else (0, 0, W, H)
    r2 = roi2 if roi2 and roi2[2] * roi2[3] > 0 else (0, 0, W, H)
    r = w // 2
    maxD = minD + D - 1
    xmin = max(r1[0], r2[0] + maxD) + r
    xmax = min(r1[0] + r1[2], r2[0] + r2[2] - minD) - r
    ymin = max(r1[1], r2[1]) + r
    ymax = min(r1[1] + r1[3], r2[1] + r2[3]) - r
    xmin, xmax = max(xmin, 0), min(xmax, W)
    ymin, ymax = max(ymin, r), min(ymax, H - r)
    if xmax - xmin <= 0 or ymax - ymin <= 0:
        return None
    return xmin, ymin, xmax - xmin, ymax - ymin


def cost_volume(Lp, Rp, D, minD, w, rows):
    """C[d, yi, x] for output columns x in [0, width1) and image rows `rows` (a range)."""
    Lp = Lp.astype(np.int64); Rp = Rp.astype(np.int64)
    H, W = Lp.shape
    r = w // 2
    lofs = max(D - 1 + minD, 0); rofs = -min(D - 1 + minD, 0)
    width1 = W - rofs - D + 1
    ys = np.arange(rows.start, rows.stop)
    x = np.arange(width1)
    C = np.zeros((D, len(ys), width1), np.int64)
    T = np.zeros((len(ys), width1), np.int64)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            lcol = np.clip(lofs + x + dx, 0, W - 1)
            rcol = np.clip(rofs + x + dx, 0, W - D)
            lv = Lp[(ys + dy)[:, None], lcol[None, :]]
            for d in range(D):
                C[d] += np.abs(lv - Rp[(ys + dy)[:, None], (rcol + d)[None, :]])
    return C, lofs, width1


def texture_volume(Lp, cap, D, minD, w, rows):
    Lp = Lp.astype(np.int64)
    H, W = Lp.shape
    r = w // 2
    lofs = max(D - 1 + minD, 0); rofs = -min(D - 1 + minD, 0)
    width1 = W - rofs - D + 1
    ys = np.arange(rows.start, rows.stop); x = np.arange(width1)
    T = np.zeros((len(ys), width1), np.int64)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            lcol = np.clip(lofs + x + dx, 0, W - 1)
            T += np.abs(Lp[(ys + dy)[:, None], lcol[None, :]] - cap)
    return T


def select(C, T, D, minD, tex, uniq):
    """-> (disp16, cost) arrays of shape C.shape[1:]"""
    FIL = (minD - 1) * 16
    mind = np.argmin(C, axis=0)             # first minimum
    minsad = np.take_along_axis(C, mind[None], 0)[0]
    out = np.full(mind.shape, FIL, np.int64)
    ok = T >= tex
    if uniq > 0:
        thresh = minsad + (minsad * uniq) // 100
        dd = np.arange(D)[:, None, None]
        outside = (dd < mind[None] - 1) | (dd > mind[None] + 1)
        ok &= ~np.any(outside & (C <= thresh[None]), axis=0)
    ip = np.where(mind + 1 < D, mind + 1, D - 2)
    im = np.where(mind > 0, mind - 1, 1)
    p = np.take_along_axis(C, ip[None], 0)[0]
    n = np.take_along_axis(C, im[None], 0)[0]
    den = p + n - 2 * minsad + np.abs(p - n)
    num = (p - n) * 256
    q = np.where(den != 0, np.sign(num) * (np.abs(num) // np.where(den != 0, den, 1)), 0)
    val = ((D - mind - 1 + minD) * 256 + q + 15) >> 4
    out = np.where(ok, val, FIL)
    return out, minsad


def validate(disp, cost, minD, D, maxdiff):
    disp = disp.copy()
    H, W = disp.shape
    INV = (minD - 1) * 16
    minX1, maxX1 = max(minD + D, 0), W + min(minD, 0)
    for y in range(H):
        d2 = [INV] * W; c2 = [None] * W
        for x in range(minX1, maxX1):
            d = int(disp[y, x])
            if d == INV:
                continue
            x2 = x - ((d + 8) >> 4)
            if 0 <= x2 < W and (c2[x2] is None or c2[x2] > cost[y, x]):
                c2[x2] = int(cost[y, x]); d2[x2] = d
        row = disp[y].copy()
        for x in range(minX1, maxX1):
            d = int(row[x])
            if d == INV:
                continue
            x0, x1 = x - (d >> 4), x - ((d + 15) >> 4)
            bad0 = 0 <= x0 < W and d2[x0] > INV and abs(d2[x0] - d) > maxdiff * 16
            bad1 = 0 <= x1 < W and d2[x1] > INV and abs(d2[x1] - d) > maxdiff * 16
            if bad0 and bad1:
                disp[y, x] = INV
    return disp


def speckle(disp, newval, maxsize, maxdiff):
    """Min-label propagation to a fixed point, then size histogram."""
    d = disp.astype(np.int64)
    H, W = d.shape
    valid = d != newval
    lab = np.where(valid, np.arange(H * W).reshape(H, W), -1)
    eh = valid[:, 1:] & valid[:, :-1] & (np.abs(d[:, 1:] - d[:, :-1]) <= maxdiff)
    ev = valid[1:, :] & valid[:-1, :] & (np.abs(d[1:, :] - d[:-1, :]) <= maxdiff)
    while True:
        old = lab.copy()
        m = np.minimum(lab[:, 1:], lab[:, :-1])
        lab[:, 1:] = np.where(eh, m, lab[:, 1:]); lab[:, :-1] = np.where(eh, np.minimum(lab[:, :-1], m), lab[:, :-1])
        m = np.minimum(lab[1:, :], lab[:-1, :])
        lab[1:, :] = np.where(ev, m, lab[1:, :]); lab[:-1, :] = np.where(ev, np.minimum(lab[:-1, :], m), lab[:-1, :])
        if (lab == old).all():
            break
    sizes = np.bincount(lab[valid].ravel(), minlength=H * W)
    out = disp.copy()
    small = valid & (sizes[np.where(valid, lab, 0)] <= maxsize)
    out[small] = newval
    return out


def stereo_bm(L, R, preFilterCap=31, blockSize=13, minDisparity=0, numDisparities=64, textureThreshold=10,
              uniquenessRatio=10, speckleWindowSize=100, speckleRange=32, disp12MaxDiff=1,
              roi1=None, roi2=None):
    H, W = L.shape
    D, minD, w = numDisparities, minDisparity, blockSize
    FIL = (minD - 1) * 16
    out = np.full((H, W), FIL, np.int64)
    lofs = max(D - 1 + minD, 0); rofs = -min(D - 1 + minD, 0); width1 = W - rofs - D + 1
    rect = valid_rect(W, H, D, minD, w, roi1, roi2)
    if lofs >= W or rofs >= W or width1 < 1 or rect is None:
        return out.astype(np.int16)
    Lp, Rp = prefilter_xsobel(L, preFilterCap), prefilter_xsobel(R, preFilterCap)
    rows = range(rect[1], rect[1] + rect[3])
    C, lofs, width1 = cost_volume(Lp, Rp, D, minD, w, rows)
    T = texture_volume(Lp, preFilterCap, D, minD, w, rows)
    d16, cost = select(C, T, D, minD, textureThreshold, uniquenessRatio)
    band = np.full((len(rows), W), FIL, np.int64); cband = np.zeros((len(rows), W), np.int64)
    ncol = min(width1, W - lofs)     # minD > 0: columns past the row end are dropped
    band[:, lofs:lofs + ncol] = d16[:, :ncol]; cband[:, lofs:lofs + ncol] = cost[:, :ncol]
    if disp12MaxDiff >= 0:
        band = validate(band, cband, minD, D, disp12MaxDiff)
    band[:, :rect[0]] = FIL; band[:, rect[0] + rect[2]:] = FIL
    out[rows.start:rows.stop] = band
    if speckleRange >= 0 and speckleWindowSize > 0:
        out = speckle(out, FIL, speckleWindowSize, speckleRange)
    return out.astype(np.int16)


# Appendix B: rows of getStructuringElement(MORPH_ELLIPSE, Size(10,10)), hand-listed.
ELLIPSE_10 = [(5, 5), (2, 8), (1, 9), (0, 9), (0, 9), (0, 9), (0, 9), (0, 9), (1, 9), (2, 8)]


def morph(img, dilate):
    img = img.astype(np.int64)
    H, W = img.shape
    pad = np.full((H + 10, W + 10), 0 if dilate else 255, np.int64)
    pad[5:5 + H, 5:5 + W] = img
    acc = np.full((H, W), 0 if dilate else 255, np.int64)
    for i, (j1, j2) in enumerate(ELLIPSE_10):
        for j in range(j1, j2 + 1):
            s = pad[i:i + H, j:j + W]      # src(y+i-5, x+j-5)
            acc = np.maximum(acc, s) if dilate else np.minimum(acc, s)
    return acc.astype(np.uint8)


def morph_open_close(img):
    return morph(morph(morph(morph(img, False), True), True), False)


# ---- cv::StereoSGBM restated a second time (MODE_SGBM = 5 paths, MODE_HH = 8), straight from the rules R1-R12 listed
# ---- in oracle/sgm_oracle.c; nothing is shared with that file ------------------------------------------------------
def _sgm_grad(img):
    img = img.astype(np.int64)
    H, W = img.shape
    up = np.vstack([img[:1], img[:-1]]); dn = np.vstack([img[1:], img[-1:]])
    g = np.full((H, W), 15, np.int64)
    v = (img[:, 2:] - img[:, :-2]) * 2 + (up[:, 2:] - up[:, :-2]) + (dn[:, 2:] - dn[:, :-2])
    g[:, 1:W - 1] = np.clip(v, -15, 15) + 15
    return g


def _bt_bounds(a):
    """per-pixel (value, min, max) of the half-sample interval, integer halves, C truncation == floor here"""
    a = a.astype(np.int64)
    left = np.concatenate([a[:, :1], (a[:, 1:] + a[:, :-1]) // 2], axis=1)
    right = np.concatenate([(a[:, :-1] + a[:, 1:]) // 2, a[:, -1:]], axis=1)
    return a, np.minimum(np.minimum(left, right), a), np.maximum(np.maximum(left, right), a)


def sgm_volumes(L, R, D, minD=0, blockSize=5, P1=600, P2=2400, paths=8):
    H, W = L.shape
    x0, x1 = max(minD + D, 0), W + min(minD, 0)
    W1 = x1 - x0
    pix = np.zeros((H, W1, D), np.int64)
    Lb, Rb = L.astype(np.int64).copy(), R.astype(np.int64).copy()
    Lb[:, 0] = Lb[:, -1] = 15; Rb[:, 0] = Rb[:, -1] = 15        # R1: the raw rows' border columns are overwritten too
    for a, b, sh in ((_sgm_grad(L), _sgm_grad(R), 0), (Lb, Rb, 2)):
        u, u0, u1 = _bt_bounds(a)
        v, v0, v1 = _bt_bounds(b)
        xs = np.arange(x0, x1)
        for d in range(D):
            xr = xs - (d + minD)
            c0 = np.maximum(0, np.maximum(u[:, xs] - v1[:, xr], v0[:, xr] - u[:, xs]))
            c1 = np.maximum(0, np.maximum(v[:, xr] - u1[:, xs], u0[:, xs] - v[:, xr]))
            pix[:, :, d] += np.minimum(c0, c1) >> sh
    r = blockSize // 2
    C = np.zeros_like(pix)
    yy = np.arange(H); xx = np.arange(W1)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            C += pix[np.clip(yy + dy, 0, H - 1)[:, None], np.clip(xx + dx, 0, W1 - 1)[None, :]]
    S = np.zeros_like(C)
    for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, 1), (1, -1), (-1, -1)):
        if paths == 5 and dy < 0:                  # the five directions of the library's default mode: none runs upwards
            continue
        Lr = np.zeros_like(C)
        ys = range(H) if dy >= 0 else range(H - 1, -1, -1)
        xs_ = range(W1) if dx >= 0 else range(W1 - 1, -1, -1)
        for y in ys:
            for x in xs_:
                px, py = x - dx, y - dy
                if 0 <= px < W1 and 0 <= py < H:
                    prev = Lr[py, px]
                    m = prev.min()
                    best = np.minimum(prev, m + P2)
                    best[1:] = np.minimum(best[1:], prev[:-1] + P1)
                    best[:-1] = np.minimum(best[:-1], prev[1:] + P1)
                    Lr[y, x] = C[y, x] + best - m
                else:
                    Lr[y, x] = C[y, x]
        S += Lr
    S = np.minimum(S, 32767)                           # R5: saturating 16-bit sum of non-negative terms
    return pix, C, S


def median3x3(img):
    """R10: 3x3 median with clamped coordinates."""
    p = np.pad(img.astype(np.int64), 1, mode="edge")
    H, W = img.shape
    st = np.stack([p[dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)])
    return np.sort(st, axis=0)[4]


def sgm(L, R, numDisparities=32, minDisparity=0, blockSize=5, P1=600, P2=2400, uniquenessRatio=10,
        speckleWindowSize=100, speckleRange=32, disp12MaxDiff=1, paths=8):
    H, W = L.shape
    D, minD = numDisparities, minDisparity
    INV = (minD - 1) * 16
    out = np.full((H, W), INV, np.int64)
    x0, x1 = max(minD + D, 0), W + min(minD, 0)
    if x1 - x0 <= 0:
        return out.astype(np.int16)
    P1 = P1 if P1 > 0 else 2                                           # R12
    P2 = max(P2 if P2 > 0 else 5, P1 + 1)
    uniq = uniquenessRatio if uniquenessRatio >= 0 else 10             # R6
    maxdiff = disp12MaxDiff if disp12MaxDiff > 0 else 1                # R9: the check is never off
    _, _, S = sgm_volumes(L, R, D, minD, blockSize, P1, P2, paths)
    for y in range(H):
        d2 = [INV] * W; c2 = [32767] * W                               # R9: initialised with the scaled invalid value
        for x in range(x1 - 1, x0 - 1, -1):                            # R7: right to left
            s = S[y, x - x0]
            bd = int(np.argmin(s)); mins = int(s[bd])
            if mins >= 32767:
                continue
            far = np.abs(np.arange(D) - bd) > 1
            if np.any(far & (s * (100 - uniq) < mins * 100)):
                continue
            x2 = x - bd - minD
            if 0 <= x2 < W and c2[x2] > mins:
                c2[x2] = mins; d2[x2] = bd + minD
            if 0 < bd < D - 1:
                den = max(int(s[bd - 1]) + int(s[bd + 1]) - 2 * mins, 1)
                num = (int(s[bd - 1]) - int(s[bd + 1])) * 16 + den
                q = abs(num) // (den * 2) * (1 if num >= 0 else -1)      # C truncation
                d16 = bd * 16 + q
            else:
                d16 = bd * 16
            out[y, x] = d16 + minD * 16
        row = out[y].copy()
        for x in range(x0, x1):
            d1 = int(row[x])
            if d1 == INV:
                continue
            da, db = d1 >> 4, (d1 + 15) >> 4
            xa, xb = x - da, x - db
            if (0 <= xa < W and d2[xa] >= minD and abs(d2[xa] - da) > maxdiff and
                    0 <= xb < W and d2[xb] >= minD and abs(d2[xb] - db) > maxdiff):
                out[y, x] = INV
    out = median3x3(out)
    if speckleWindowSize > 0:                                          # R11
        out = speckle(out.astype(np.int16), INV, speckleWindowSize, 16 * speckleRange).astype(np.int64)
    return out.astype(np.int16)


# ---- depth statistics after the matcher (estimator.cpp:75-77, 206-263), numpy version --------------------
def depth_stats(disp16, Q, mask, regions, unit=25.0):
    d = np.rint(disp16.astype(np.float64) / 16.0).astype(np.int64)       # numpy rint = ties to even
    H, W = d.shape
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    Q = np.asarray(Q, np.float64).reshape(4, 4)
    Zh = Q[2, 0] * x + Q[2, 1] * y + Q[2, 2] * d + Q[2, 3]
    Wh = Q[3, 0] * x + Q[3, 1] * y + Q[3, 2] * d + Q[3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (Zh / Wh).astype(np.float32)
    z[d == d.min()] = np.float32(10000.0)
    zz = z.astype(np.float64)
    ok = ~(np.abs(zz - 10000.0) < np.finfo(np.float32).eps) & ~(np.abs(zz) > 10000.0) & (mask != 0)
    means, counts = [], []
    for (rx, ry, rw, rh) in regions:
        sel = ok[ry:ry + rh, rx:rx + rw]
        c = int(sel.sum())
        counts.append(c)
        means.append(float(zz[ry:ry + rh, rx:rx + rw][sel].sum() / c * unit / 10.0) if c else 0.0)
    return np.array(means), np.array(counts)


# ---- rectification in front of the matcher (independent of oracle/rectify_oracle.c) ---------------------------
def rgb2gray(rgb):
    c = rgb.astype(np.int64)
    return ((c[..., 0] * 4899 + c[..., 1] * 9617 + c[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def bilinear_table():
    """The 32x32 table of 2x2 weights the way the library builds it: float products scaled by 2**15, rounded to
    short, with the fix-up step that forces every 2x2 block to sum to 32768 (never triggered for 5-bit fractions)."""
    tab = np.zeros((32, 32, 4), np.int32)
    for fy in range(32):
        for fx in range(32):
            ax, ay = np.float32(fx) / np.float32(32), np.float32(fy) / np.float32(32)
            w = np.array([(1 - ay) * (1 - ax), (1 - ay) * ax, ay * (1 - ax), ay * ax], np.float32)
            iw = np.rint(w * np.float32(32768)).astype(np.int32)
            d = int(iw.sum()) - 32768
            if d < 0:
                iw[np.argmax(iw)] -= d
            elif d > 0:
                iw[np.argmin(iw)] -= d
            tab[fy, fx] = iw
    return tab


def remap_bilinear(src, map1, map2):
    """Vectorised, via the table above and zero padding instead of per-sample range checks."""
    tab = bilinear_table()
    src3 = src if src.ndim == 3 else src[..., None]
    sH, sW, cn = src3.shape
    pad = np.zeros((sH + 2, sW + 2, cn), np.int64)
    pad[1:-1, 1:-1] = src3
    sx = map1[..., 0].astype(np.int64); sy = map1[..., 1].astype(np.int64)
    fx = (map2 & 31).astype(np.int64); fy = ((map2 >> 5) & 31).astype(np.int64)
    w = tab[fy, fx].astype(np.int64)                                  # dH x dW x 4
    acc = np.zeros(sx.shape + (cn,), np.int64)
    for k in range(4):
        xx = np.clip(sx + (k & 1), -1, sW) + 1
        yy = np.clip(sy + (k >> 1), -1, sH) + 1
        acc += w[..., k, None] * pad[yy, xx]
    out = ((acc + (1 << 14)) >> 15).astype(np.uint8)
    return out if src.ndim == 3 else out[..., 0]


def init_undistort_rectify_map(M, D, R, P, W, H):
    """Closed-form rays (no running sums along the row), numpy.linalg.inv: an ulp-level different route to the
    same maps; entries may differ from the oracle's where u*32 lands within an ulp of a rounding boundary."""
    M = np.asarray(M, np.float64).reshape(3, 3); R = np.asarray(R, np.float64).reshape(3, 3)
    P = np.asarray(P, np.float64).reshape(3, 4)
    d = np.zeros(14); dd = np.asarray(D, np.float64).reshape(-1); d[:len(dd)] = dd
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = d[:12]
    ir = np.linalg.inv(P[:, :3] @ R)
    j, i = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    X = ir[0, 0] * j + ir[0, 1] * i + ir[0, 2]
    Y = ir[1, 0] * j + ir[1, 1] * i + ir[1, 2]
    Wd = ir[2, 0] * j + ir[2, 1] * i + ir[2, 2]
    x, y = X / Wd, Y / Wd
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    xd = x * kr + p1 * 2 * x * y + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * 2 * x * y + s3 * r2 + s4 * r2 * r2
    u = M[0, 0] * xd + M[0, 2]; v = M[1, 1] * yd + M[1, 2]
    iu = np.rint(u * 32).astype(np.int64); iv = np.rint(v * 32).astype(np.int64)
    map1 = np.stack([iu >> 5, iv >> 5], -1).astype(np.int16)
    map2 = ((iv & 31) * 32 + (iu & 31)).astype(np.uint16)
    return map1, map2


# ---- object detection -> ROI (independent of oracle/objects_oracle.c) -----------------------------------------
def rgb2hsv(rgb):
    """Per-pixel Python ints, straight from the published fixed-point formulas (hue in [0,180))."""
    sdiv = [0] + [int(round((255 << 12) / (1.0 * i))) for i in range(1, 256)]
    hdiv = [0] + [int(round((180 << 12) / (6.0 * i))) for i in range(1, 256)]
    H, W, _ = rgb.shape
    out = np.zeros_like(rgb)
    for y in range(H):
        for x in range(W):
            r, g, b = (int(v) for v in rgb[y, x])
            v, vmin = max(r, g, b), min(r, g, b)
            diff = v - vmin
            s = (diff * sdiv[v] + 2048) >> 12
            if v == r:
                h = g - b
            elif v == g:
                h = b - r + 2 * diff
            else:
                h = r - g + 4 * diff
            h = (h * hdiv[diff] + 2048) >> 12
            if h < 0:
                h += 180
            out[y, x] = (min(max(h, 0), 255), s, v)
    return out


def external_boxes(mask, min_area=100, zero_border=True):
    """scipy.ndimage labelling; a component is external iff one of its pixels is 4-adjacent to the background that
    reaches the frame (or lies on the frame itself)."""
    from scipy import ndimage as ndi
    m = mask != 0
    if zero_border:
        m = m.copy(); m[0, :] = m[-1, :] = False; m[:, 0] = m[:, -1] = False
    H, W = m.shape
    pad = np.zeros((H + 2, W + 2), bool); pad[1:-1, 1:-1] = m
    bg, _ = ndi.label(~pad)                                   # 4-connected by default
    outer = bg == bg[0, 0]
    near_outer = ndi.binary_dilation(outer, structure=ndi.generate_binary_structure(2, 1))[1:-1, 1:-1]
    fg, n = ndi.label(m, structure=np.ones((3, 3), bool))      # 8-connected
    found = []
    for i, sl in enumerate(ndi.find_objects(fg), 1):
        comp = fg[sl] == i
        if not (comp & near_outer[sl]).any():
            continue
        ys, xs = sl
        first = np.flatnonzero(fg.ravel() == i)[0]
        box = (xs.start, ys.start, xs.stop - xs.start, ys.stop - ys.start)
        if box[2] * box[3] >= min_area:
            found.append((first, box))
    return [b for _, b in sorted(found, reverse=True)]
