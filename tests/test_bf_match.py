"""Tracking::BfMatch (src/Tracking.cc:1747-1766): brute-force Hamming 2-NN
with the 0.6 ratio test.  PARITY UNPINNED against OpenCV (absent here): the oracle restates
cv::BFMatcher::knnMatch's published behaviour; the HIP path must equal the oracle exactly."""
import importlib

import numpy as np
import pytest

pkg = importlib.import_module("lc-crf-slam_amd")


def descriptors(n_query, n_train, seed, planted=0.4, flips=(0, 40)):
    """Random 256-bit descriptors; a share of the queries gets a noisy copy in the train set (and
    some of those a second, worse copy), so that the ratio test sees accepts, rejects and ties."""
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 256, (n_query, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (n_train, 32), dtype=np.uint8)
    if n_train and n_query:
        for i in rng.choice(n_query, int(planted * n_query), replace=False):
            j = int(rng.integers(0, n_train))
            bits = rng.choice(256, int(rng.integers(flips[0], flips[1] + 1)), replace=False)
            d = q[i].copy()
            for b in bits:
                d[b >> 3] ^= np.uint8(1 << (b & 7))
            t[j] = d
            if rng.random() < 0.3:                         # an exact duplicate elsewhere: a tie on the best distance
                t[int(rng.integers(0, n_train))] = d
    return q, t


def test_oracle_bf_match_known_answers(po):
    z = np.zeros((1, 32), np.uint8)
    one = z.copy(); one[0, 0] = 0x01                       # distance 1 from z
    far = np.full((1, 32), 0xff, np.uint8)                 # distance 256 from z
    # best 0, second 256: 0 < 256 * 0.6 -> match with the FIRST of the train rows
    out, n = po.oracle_bf_match(z, np.concatenate([far, z, far]))
    assert out.tolist() == [1] and n == 1
    # two exact copies: 0 < 0 * 0.6 is false -> no match (and the tie would keep the lower index)
    out, n = po.oracle_bf_match(z, np.concatenate([z, z, far]))
    assert out.tolist() == [-1] and n == 0
    # ratio boundary: d0 = 3, d1 = 5: 5 * 0.6 rounds to exactly 3.0 in double, and 3 < 3.0 is false
    three = z.copy(); three[0, 0] = 0x07
    five = z.copy(); five[0, 0] = 0x1f
    two = z.copy(); two[0, 0] = 0x03
    out, _ = po.oracle_bf_match(z, np.concatenate([five, three]))
    assert out.tolist() == [-1]
    out, _ = po.oracle_bf_match(z, np.concatenate([five, two]))
    assert out.tolist() == [1]
    out, _ = po.oracle_bf_match(z, np.concatenate([five, two]), ratio=0.4)
    assert out.tolist() == [-1]
    # fewer than two train descriptors: knnMatch returns < 2 neighbours, the reference skips the query
    assert po.oracle_bf_match(z, one)[0].tolist() == [-1]
    assert po.oracle_bf_match(z, np.zeros((0, 32), np.uint8))[0].tolist() == [-1]
    assert po.oracle_bf_match(np.zeros((0, 32), np.uint8), z)[1] == 0


def test_bf_match_argument_checks():
    lib = pkg.lib()
    assert lib.lccrf_bf_match(0, -1, None, 0, None, 0.6, None, None) == -1
    assert lib.lccrf_bf_match(0, 4, None, 4, None, 0.6, None, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("n_query,n_train,seed", [(2000, 2000, 1), (1, 2, 2), (0, 5, 3), (7, 1, 4), (5, 0, 5),
                                                  (1500, 1023, 6), (333, 1025, 7), (64, 3000, 8), (65, 4096, 9)])
def test_hip_bf_match_equals_oracle(po, n_query, n_train, seed):
    q, t = descriptors(n_query, n_train, seed)
    for ratio in (0.6, 0.95):
        o, no = po.oracle_bf_match(q, t, ratio)
        h, nh = pkg.bf_match(q, t, ratio)
        assert np.array_equal(o, h) and no == nh
    if n_query >= 1500:
        assert 0.2 * n_query < no < 0.6 * n_query           # the planted pairs are found, random pairs are not


@pytest.mark.gpu
def test_hip_bf_match_ties_keep_the_lower_train_index(po):
    q, t = descriptors(400, 900, seed=11, planted=0.0)
    t[500] = t[17]                                         # exact duplicates in the train set
    t[300] = q[5]; t[800] = q[5]                           # and two perfect copies of a query: d0 = d1 = 0 -> rejected
    q[9] = t[17]                                           # query 9: d0 = d1 = 0 as well
    far = q[3].copy(); far[:12] ^= 0xff                    # query 3: a lone near copy at two places, 96 bits away ...
    t[40] = far; t[41] = far                               # ... a tie at the best distance, accepted only at ratio ~1
    o, no = po.oracle_bf_match(q, t, 1.01)
    h, nh = pkg.bf_match(q, t, 1.01)
    assert np.array_equal(o, h) and no == nh
    assert h[3] == 40                                      # a tie at the best distance: the lower train index wins
    assert h[5] == -1 and h[9] == -1                       # d0 = d1 = 0: 0 < 0 * ratio never holds
