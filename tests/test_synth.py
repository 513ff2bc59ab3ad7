import numpy as np


def test_stream_is_deterministic_and_frame_indexed(synth):
    a = synth.make_pair(synth.STREAM_SEED + 3, 80, 60, 16)
    b = synth.make_pair(synth.STREAM_SEED + 3, 80, 60, 16)
    c = synth.make_pair(synth.STREAM_SEED + 4, 80, 60, 16)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert not np.array_equal(a[0], c[0])
    L, R = synth.make_stream(3, 2, 80, 60, 16)
    assert np.array_equal(L[0], a[0]) and np.array_equal(R[1], c[1])


def test_disparity_field_in_range(synth):
    d = synth.disparity_right(synth.STREAM_SEED, 320, 240, 32)
    assert d.min() >= 2 and d.max() <= 29


def test_textured_and_u8(synth):
    L, R = synth.make_pair(synth.STREAM_SEED, 320, 240, 32)
    assert L.dtype == np.uint8 and L.std() > 15 and R.std() > 15
