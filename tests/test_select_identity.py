"""CPU check of the arithmetic identities the selection kernels rest on (rt-depth-map_amd/csrc/rtdm_select.h), against the
plain rule of the oracle (oracle/bm_oracle.c:175-191, cv::StereoBM's uniqueness test): a pixel is rejected iff some index
outside [a-1, a+1] has sad <= T = minsad + minsad * ratio / 100.

  * the sum identity (select_disparity, select_disparity_lds): sum_e max(T+1 - sad[e], 0) over ALL e equals the same sum over
    {a-1, a, a+1} iff no other index has sad <= T -- with saturating 16-bit halves (even / odd indices);
  * the split test of GroupSelect: (A) the number of groups of eight whose minimum is <= T equals 1 (+1 if the group the
    neighbour a-1 / a+1 falls into has a minimum <= T), (B) the sum identity restricted to those one or two groups;
  * the one-correction truncating division of the sub-pixel step.
"""
import numpy as np
import pytest


def reject_plain(sad, ratio):
    m = int(sad.min()); a = int(np.argmin(sad))           # first minimum
    T = m + m * ratio // 100
    idx = np.arange(len(sad))
    return bool(np.any((sad <= T) & ((idx < a - 1) | (idx > a + 1)))), a, m, T


def sat16(v):
    return min(int(v), 65535)


def reject_sum_identity(sad, ratio):
    _, a, m, T = reject_plain(sad, ratio)
    T = min(T, 32766); T1 = T + 1
    terms = np.maximum(T1 - sad.astype(np.int64), 0)
    z = sat16(terms[0::2].sum()) + sat16(terms[1::2].sum())           # two saturating halves, as the packed u16 chains
    want = sum(int(terms[e]) for e in (a - 1, a, a + 1) if 0 <= e < len(sad))
    return z != want


def reject_group_select(sad, ratio):
    D = len(sad)
    _, a, m, T = reject_plain(sad, ratio)
    T = min(T, 32766); T1 = T + 1
    gmin = sad.reshape(D // 8, 8).min(axis=1)
    gs, e = a // 8, a % 8
    nb = gs - 1 if (e == 0 and a > 0) else gs + 1 if (e == 7 and a + 1 < D) else gs
    cnt = int((gmin < T1).sum())
    expect = 1 + int(nb != gs and gmin[nb] < T1)
    vals = np.concatenate([sad[8 * gs:8 * gs + 8], sad[8 * nb:8 * nb + 8] if nb != gs else np.zeros(0, sad.dtype)])
    pos = np.concatenate([np.arange(8 * gs, 8 * gs + 8), np.arange(8 * nb, 8 * nb + 8) if nb != gs else np.zeros(0, int)])
    terms = np.maximum(T1 - vals.astype(np.int64), 0)
    z = sat16(terms[pos % 2 == 0].sum()) + sat16(terms[pos % 2 == 1].sum())
    want = sum(int(max(T1 - int(sad[k]), 0)) for k in (a - 1, a, a + 1) if 0 <= k < D)
    return (z != want) or (cnt != expect)


def vectors(rng, D, n):
    for i in range(n):
        kind = i % 6
        if kind == 0: s = rng.integers(0, 32767, D)
        elif kind == 1: s = rng.integers(900, 1100, D)                       # many near-ties
        elif kind == 2: s = np.full(D, int(rng.integers(0, 5000)))           # plateau
        elif kind == 3:                                                      # a clear minimum at a group edge
            s = rng.integers(3000, 9000, D); s[int(rng.choice([0, 7, 8, 15, D - 8, D - 1, D // 2, D // 2 - 1]))] = int(rng.integers(0, 3000))
        elif kind == 4:                                                      # minimum with a close neighbour across a group edge
            s = rng.integers(4000, 9000, D); p = int(rng.choice(np.arange(7, D - 1, 8))); s[p] = 1000; s[p + 1] = int(rng.integers(1000, 1200))
        else:                                                                # saturating sums: everything far below the threshold
            s = rng.integers(0, 40, D); s[int(rng.integers(0, D))] = 0
        yield np.minimum(s, 32766).astype(np.int32)


@pytest.mark.parametrize("D", [16, 32, 48, 64, 96, 128])
def test_uniqueness_identities_equal_the_plain_rule(D):
    rng = np.random.default_rng(1000 + D)
    for ratio in (1, 10, 25, 100, 400):
        for sad in vectors(rng, D, 1500):
            want = reject_plain(sad, ratio)[0]
            assert reject_sum_identity(sad, ratio) == want, (D, ratio, sad.tolist())
            assert reject_group_select(sad, ratio) == want, (D, ratio, sad.tolist())


def test_one_correction_division():
    # sel_div_trunc: q = trunc(num / den) with num = 256 (p - n), den = p + n - 2 m + |p - n| <= 65532, |q| <= 128, from the
    # float estimate an * rcp(den) -- v_rcp_f32 is good to one ulp, so the reciprocal is also tried one ulp either way --
    # and ONE correction each way
    rng = np.random.default_rng(7)
    one = np.float32(1.0)
    for _ in range(200000):
        m = int(rng.integers(0, 32000)); p = m + int(rng.integers(0, 766)); n = m + int(rng.integers(0, 766))
        den = p + n - 2 * m + abs(p - n)
        if den == 0: continue
        an = abs(256 * (p - n))
        assert an // den <= 128
        r0 = one / np.float32(den)
        for r in (r0, np.nextafter(r0, np.float32(0)), np.nextafter(r0, np.float32(2))):
            q = int(np.float32(an) * r)
            rem = an - q * den
            if rem < 0: q -= 1
            elif rem >= den: q += 1
            assert q == an // den, (p, n, m, float(r))
