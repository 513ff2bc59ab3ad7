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
