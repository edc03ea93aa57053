"""tests/dropout_ref.py (the numpy restatement the GPU tests check the kernels' dropout masks against) cross-checked on the CPU against
tests/dropout_ref.c (the same specification in plain C with native uint32_t arithmetic), plus the properties a mask must have."""
import subprocess

import numpy as np
import pytest

from tests import dropout_ref as R


@pytest.fixture(scope='module')
def cref(tmp_path_factory):
    import os
    exe = tmp_path_factory.mktemp('dref') / 'dropout_ref'
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dropout_ref.c')
    subprocess.run(['gcc', '-O1', '-o', str(exe), src], check=True)

    def run(*args):
        out = subprocess.run([str(exe)] + [str(a) for a in args], check=True, capture_output=True, text=True).stdout.strip()
        return np.frombuffer(out.encode(), dtype=np.uint8) == ord('1')
    return run


@pytest.mark.parametrize('seed,p,first,count', [(4242, 0.1, 0, 70001), (0x5EED0000BEEF, 0.5, 1, 4097), ((1 << 48) - 3, 0.25, (1 << 33) - 7, 64),
                                                 (R.eff_seed(77, 5), 0.1, 12345, 9999)])
def test_flat_mask_numpy_equals_c(cref, seed, p, first, count):
    idx = np.arange(first, first + count, dtype=np.uint64)
    assert np.array_equal(R.keep_flat(seed, idx, p), cref('flat', seed, p, first, count))


@pytest.mark.parametrize('seed,p,Bn,H,Sq,Sk', [(4242, 0.1, 2, 3, 17, 33), (R.eff_seed(9, 3), 0.3, 1, 8, 5, 300), (1 << 40, 0.1, 3, 2, 4, 16)])
def test_attention_mask_numpy_equals_c(cref, seed, p, Bn, H, Sq, Sk):
    assert np.array_equal(R.keep_attention(seed, Bn, H, Sq, Sk, p).reshape(-1), cref('attn', seed, p, Bn, H, Sq, Sk))


def test_masks_have_the_rate_and_no_neighbour_correlation():
    """what the round-6 miscompile pattern would break: a word's flags replicated into its neighbours keep the RATE and correlate
    adjacent elements.  The specification itself must not: lag-1 .. lag-8 correlations of the keep bits within 4 sigma of zero."""
    n = 1 << 20
    for keep in (R.keep_flat(4242, np.arange(n, dtype=np.uint64), 0.1).astype(np.float64),
                 R.keep_attention(4242, 4, 8, 128, 256, 0.1).reshape(-1).astype(np.float64)):
        assert abs(keep.mean() - 0.9) < 2e-3
        c = keep - keep.mean()
        for lag in range(1, 9):
            r = float((c[:-lag] * c[lag:]).mean() / c.var())
            assert abs(r) < 4.0 / np.sqrt(n), (lag, r)
    assert R.drop_thresh(0.1) == int(float(np.float32(0.1)) * 2 ** 32) and R.drop_thresh(0.0) == 0 and R.drop_thresh(1.0) == 0xFFFFFFFF
    assert R.eff_seed(5) == 5 and R.eff_seed(5, 0) == 5 and R.eff_seed(5, 1) == 5 ^ 0x9E3779B97F4A7C15
