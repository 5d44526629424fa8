"""The summation orders the device kernels promise, restated in NumPy (no device): the exchange tree of
`wave_sum_many` (csrc/mvn_multi_kernel.h) gives, for every one of its NV values, BITWISE the sum `wave_sum` (csrc/device_math.h)
gives for that value alone -- the balanced pairwise tree over the 64 lanes, (0 + 1), (2 + 3), pairs of pairs, ... -- which is what
lets a chain inside a merged launch reproduce the chain alone bit for bit (tests/test_gpu_chain_group.py checks the device)."""

import numpy as np
import pytest

WAVE = 64


def wave_sum(v):
    """device_math.h `wave_sum`: row_shr 1, 2, 4, 8 inside rows of 16 lanes, then row_bcast 15 and 31 -- lane 63 ends up with
    ((rows 3 + 2) + (rows 1 + 0)), every row total being the balanced tree of its 16 lanes.  `a + b` is computed as upper + lower."""
    x = np.array(v, dtype=np.float64)
    for sh in (1, 2, 4, 8):
        moved = np.zeros(WAVE)
        for lane in range(WAVE):
            if lane % 16 >= sh:
                moved[lane] = x[lane - sh]
        x = x + moved
    rows = [x[15], x[31], x[47], x[63]]
    r1 = rows[1] + rows[0]
    r3 = rows[3] + rows[2]
    return r3 + r1


def wave_sum_many(vals):
    """mvn_multi_kernel.h `wave_sum_many`: vals[lane][i]; at step s lanes l and l ^ 2^s keep one half of their values each and add
    what the partner hands over; the remaining steps all-reduce the single value left.  Returns {value index: total}."""
    nv = vals.shape[1]
    nvp = 1
    while nvp < nv:
        nvp *= 2
    v = np.zeros((WAVE, nvp))
    v[:, :nv] = vals
    idx = np.zeros(WAVE, dtype=int)
    n, s = nvp, 0
    while n > 1:
        half = n // 2
        new = np.zeros((WAVE, half))
        for lane in range(WAVE):
            up = (lane >> s) & 1
            partner = lane ^ (1 << s)
            pup = (partner >> s) & 1
            for i in range(half):
                keep = v[lane, i + half] if up else v[lane, i]
                recv = v[partner, i] if pup else v[partner, i + half]      # what the partner SENDS: the half it does not keep
                new[lane, i] = keep + recv
            idx[lane] += half if up else 0
        v, n, s = new, half, s + 1
    x = v[:, 0].copy()
    while (1 << s) < WAVE:
        x = x + x[np.arange(WAVE) ^ (1 << s)]
        s += 1
    return {int(idx[lane]): x[lane] for lane in range(nvp)}, idx, x


@pytest.mark.parametrize("nv", [1, 2, 4, 8, 16, 24, 32])
def test_exchange_tree_reproduces_wave_sum_bit_for_bit(nv):
    rng = np.random.default_rng(nv)
    for scale in (1.0, 1e8, 1e-8):
        vals = rng.normal(size=(WAVE, nv)) * scale * np.exp(rng.normal(size=(WAVE, 1)) * 3.0)     # badly scaled: order matters
        totals, idx, x = wave_sum_many(vals)
        assert sorted(k for k in totals if k < nv) == list(range(nv))
        for i in range(nv):
            assert totals[i] == wave_sum(vals[:, i]), (nv, i)          # bitwise
        # every lane that ends with value i holds the same total
        for lane in range(WAVE):
            if idx[lane] < nv:
                assert x[lane] == totals[int(idx[lane])]


def test_wave_sum_is_a_sum():
    """(the restated `wave_sum` adds every lane exactly once)"""
    rng = np.random.default_rng(0)
    v = rng.normal(size=WAVE) * np.exp(rng.normal(size=WAVE) * 6.0)
    seq = 0.0
    for t in v:
        seq += t
    assert abs(wave_sum(v) - seq) <= 1e-9 * np.abs(v).sum()      # the same sum to rounding ...
    assert wave_sum(np.ones(WAVE)) == 64.0 and wave_sum(np.arange(WAVE, dtype="float64")) == 2016.0
