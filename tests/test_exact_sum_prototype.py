"""scripts/exact_sum_prototype.py (not part of the product: the arithmetic behind the costed / shelved plan for the ordered
row sums, notes/r3_experiments.md): the binade-segmented evaluation must equal the plain left-to-right fp32 sum bit for bit."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("exact_sum_prototype", os.path.join(ROOT, "scripts", "exact_sum_prototype.py"))
esp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(esp)


def test_segmented_sum_equals_the_plain_loop():
    rng = np.random.default_rng(2026)
    for i in range(240):
        n = int(rng.integers(1, 500))
        t = esp.slam_like_row(rng, n) if i % 3 == 0 else esp.adversarial_row(rng, n)
        assert esp.sequential(t).view(np.uint32) == esp.segmented(t).view(np.uint32), (i, n)


def test_every_addition_a_tie_and_the_dependent_chain_is_short():
    rng = np.random.default_rng(7)
    stats = []
    for _ in range(20):
        n = 400
        t = (rng.integers(0, 4096, n) * 2.0 ** -23 + 2.0 ** -24).astype(np.float32)
        t[0] = np.float32(1.0)
        assert esp.sequential(t).view(np.uint32) == esp.segmented(t, stats).view(np.uint32)
    assert np.mean([s[3] for s in stats]) > 20              # parity-dependent blocks really occur ...
    assert np.mean([s[1] for s in stats]) < 80               # ... and the sequential pass stays short (400 terms)


def test_ineligible_rows_are_refused():
    assert esp.segmented(np.array([1.0, -0.5], np.float32)) is None
    assert esp.segmented(np.array([np.nan], np.float32)) is None
    assert esp.segmented(np.zeros(0, np.float32)) == 0.0
