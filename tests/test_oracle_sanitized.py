"""The CPU restatement under AddressSanitizer + UBSan (the reference has no sanitizer
coverage; GPU ASAN is not available on this pool, so the sanitised target is the oracle).

oracle/Makefile's `liblccrf_oracle_asan.so` is built here and driven, in a child process with the
sanitizer runtime preloaded, over golden fixtures of every shape class (N%4 phantoms, d = 1..6, L up to
21, the C5 miniature); any report makes the child exit non-zero."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as po
po.ORACLE_SO = os.path.join(ROOT, "oracle", "liblccrf_oracle_asan.so")
import crf_cases as cc
n = 0
for name, pick in (("slam", ("N5", "N6", "N7", "N1001", "N2000")), ("generic", None), ("bilateral", ("c5",))):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    for case in [str(c) for c in z["cases"]]:
        if pick is not None and case not in pick:
            continue
        c = cc.setup(po.OracleCRF, cc.case_problem(z, case))
        cc.check_against_expected(c, cc.case_expected(z, case))
        c.close()
        n += 1
assert n >= 8, n
# the side rows of the restatement: unary builder and BfMatch on small seeded inputs
import importlib
wl = importlib.import_module("lc-crf-slam_amd.workloads")
sc = wl.map_point_scene(300, 8, seed=5)
po.oracle_unary_build(sc["Xw"], sc["obs_ptr"], sc["obs_kf"], sc["obs_kp"], sc["kf_pose"], sc["kf_intr"], sc["kf_bounds"])
rng = np.random.default_rng(3)
po.oracle_bf_match(rng.integers(0, 256, (70, 32), dtype=np.uint8), rng.integers(0, 256, (90, 32), dtype=np.uint8))
print("sanitized oracle ok:", n, "fixtures")
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not installed")
def test_oracle_under_asan_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liblccrf_oracle_asan.so"], check=True,
                   stdout=subprocess.DEVNULL)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True, check=True).stdout.strip()
    if not os.path.isabs(libasan):
        pytest.skip("libasan runtime not found")
    env = dict(os.environ, LD_PRELOAD=libasan + (":" + libubsan if os.path.isabs(libubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="halt_on_error=1:exitcode=24")
    r = subprocess.run([sys.executable, "-c", "ROOT=%r\n" % ROOT + CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitized oracle ok" in r.stdout
