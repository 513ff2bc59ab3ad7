"""The oracle against the committed vectors (tests/golden/, made by make_golden.py from this
repository's own oracle: the reference has no fixtures for this path -- PARITY UNPINNED)."""
import numpy as np
import pytest

import golden_util as gu


@pytest.mark.parametrize("name", gu.bm_cases())
def test_oracle_reproduces_bm_golden(oracle, name):
    L, R, disp, kw = gu.load_bm(name)
    assert np.array_equal(oracle.bm_compute(L, R, **kw), disp)


@pytest.mark.parametrize("name", gu.morph_cases())
def test_oracle_reproduces_morph_golden(oracle, name):
    z = gu.load_morph(name)
    assert np.array_equal(oracle.morph_open_close(z["mask"]), z["mask_out"])
    assert np.array_equal(oracle.morph_open_close(z["gray"]), z["gray_out"])


def test_golden_inputs_are_the_synthetic_stream(synth):
    L, R, _, kw = gu.load_bm("bm_64x48_d16_w5")
    l2, r2 = synth.make_pair(synth.STREAM_SEED + 1001, 64, 48, kw["numDisparities"])
    assert np.array_equal(L, l2) and np.array_equal(R, r2)
