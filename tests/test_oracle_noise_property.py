"""The reference pins `RawData.add_noise` only through a hypothesis test over (data_length, n_channels,
noise_data_length) (/root/reference/tests/test_core.py:6-39: crash-freedom on that domain, including a channel whose
range runs past the end of its row).  Restated here for the CPU oracle's add_noise, and strengthened: the result must
equal the reference's expression noise[(ix_rand + ix - ch_left) mod N, ch] (rawdata.py:419-437) sample by sample."""
import numpy as np
from hypothesis import example, given, settings, strategies

from oracle import oracle as O


def _expected(data, mask, left, right, noise, ix_rand):
    out = data.copy()
    n_len, n_ch = noise.shape
    for ch in range(data.shape[0]):
        if ch >= n_ch or not mask[ch]:
            continue
        for ix in range(left[ch], right[ch] + 1):
            if ix >= data.shape[1]:
                continue
            out[ch, ix] += noise[(ix_rand + ix - left[ch]) % n_len, ch]
    return out


@settings(max_examples=100, deadline=None)
@given(strategies.integers(min_value=1, max_value=1_000), strategies.integers(min_value=1, max_value=4),
       strategies.integers(min_value=5, max_value=1_000), strategies.integers(min_value=0, max_value=10 ** 6))
@example(data_length=101, n_channels=4, noise_data_length=1000, pick=3)
def test_add_noise_matches_reference_expression(data_length, n_channels, noise_data_length, pick):
    rng = np.random.default_rng(data_length * 7919 + n_channels * 31 + noise_data_length)
    data = rng.integers(-100, 0, size=(n_channels, data_length)).astype(np.int64)
    # the reference's test masks only the last channel, with left = n_channels - 1, right = noise_data_length - n_channels
    mask = np.zeros(n_channels, dtype=np.uint8)
    left = np.full(n_channels, 9223372036399775857, dtype=np.int64)
    right = np.full(n_channels, -454999850, dtype=np.int64)
    mask[-1], left[-1], right[-1] = 1, n_channels - 1, noise_data_length - n_channels
    noise = rng.integers(-10, 10, size=(noise_data_length, n_channels)).astype(np.int16)
    high = O.noise_high(mask, left, right, noise_data_length)
    # rawdata.py:407-417
    span = noise_data_length - right[-1] + left[-1] - 1
    assert high == (noise_data_length - 1 if span < 0 else span)
    ix_rand = 0 if high <= 0 else pick % high
    want = _expected(data, mask, left, right, noise, ix_rand)
    O.add_noise(data, mask, left, right, noise, ix_rand)
    assert np.array_equal(data, want)


def test_add_noise_skips_channels_without_noise_columns_and_unmasked_rows():
    rng = np.random.default_rng(1)
    data = np.zeros((5, 40), dtype=np.int64)
    noise = rng.integers(-9, 9, size=(17, 3)).astype(np.int16)             # shorter than the rows: wraps; 3 of 5 channels
    mask = np.array([1, 0, 1, 1, 1], dtype=np.uint8)
    left = np.array([0, 0, 5, 2, 0], dtype=np.int64); right = np.array([39, 39, 30, 39, 39], dtype=np.int64)
    want = _expected(data, mask, left, right, noise, 11)
    O.add_noise(data, mask, left, right, noise, 11)
    assert np.array_equal(data, want)
    assert not data[1].any() and not data[3].any() and not data[4].any() and data[0].any() and data[2, 5:31].any()


def test_float_noise_stores_the_truncated_sum_like_the_reference():
    """tests/golden/noise_float.npz: the reference's add_noise (numba) run on int64 rows with a float noise array --
    `data[ch, ix] += noise[ix, ch]` stores trunc(data + noise), not data + trunc(noise)"""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'noise_float.npz'))
    data, noise, out = d['data'], d['noise'], d['out']
    n_ch, n = data.shape
    mine = data.copy()
    O.add_noise(mine, np.ones(n_ch, dtype=np.uint8), np.zeros(n_ch, dtype=np.int64), np.full(n_ch, n - 1, dtype=np.int64), noise, 0)
    assert np.array_equal(mine, out)
    assert not np.array_equal(out, data + np.trunc(noise.T).astype(np.int64))        # the difference is visible in the fixture
