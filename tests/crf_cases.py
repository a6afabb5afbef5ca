"""Helpers shared by the parity tests: fixture decoding and a backend-agnostic runner.

A "backend" is any class with the reference's operator surface on numpy arrays
(oracle.pyoracle.OracleCRF / RefCRF, or the HIP mirror DenseCRFHIP):
  cls(N, L); set_unary | set_unary_from_label; add_pairwise(features, w);
  start_inference(); step_inference(relax); build_map(); probability(); map(); kernel(k)
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INSTR_LIB = os.path.join(ROOT, "lc-crf-slam_amd", "liblccrf_hip_instr.so")


def switch_env(env=None, **more):
    """Environment for a child process that sets A/B / cross-check switches: they exist in the INSTRUMENTED library only
    (csrc/engine.h: ab_env), so a non-empty set of switches also selects that library through LCCRF_LIB."""
    sw = dict(env or {}, **more)
    out = dict(os.environ, **sw)
    if sw:
        out["LCCRF_LIB"] = INSTR_LIB
    return out


def bits(a):
    """Bit pattern view, so float comparisons are bit-exact (and NaN-safe)."""
    a = np.ascontiguousarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def case_problem(z, prefix):
    """Decode one fixture case (tests/golden/make_golden.py: pack())."""
    p = prefix + "_"
    pb = dict(N=int(z[p + "N"]), L=int(z[p + "L"]))
    K = int(z[p + "K"])
    if p + "unary" in z.files:
        pb["unary"] = z[p + "unary"]
    else:
        pb["label"] = z[p + "label"]
        pb["conf"] = np.float32(z[p + "conf"])
    pb["kernels"] = [(z[p + "feat%d" % k], np.float32(z[p + "w%d" % k])) for k in range(K)]
    return pb


def case_expected(z, prefix):
    p = prefix + "_"
    exp = dict(V=[], norm=[], Q={}, map={})
    K = int(z[p + "K"])
    for k in range(K):
        exp["V"].append(int(z[p + "V%d" % k]))
        exp["norm"].append(z[p + "norm%d" % k])
    for name in z.files:
        if name.startswith(p + "Q"):
            exp["Q"][int(name[len(p) + 1:])] = z[name]
        if name.startswith(p + "map"):
            exp["map"][int(name[len(p) + 3:])] = z[name]
    exp["relax"] = float(z[p + "relax"]) if p + "relax" in z.files else 1.0
    return exp


def setup(cls, pb, **kw):
    c = cls(pb["N"], pb["L"], **kw)
    if "unary" in pb:
        c.set_unary(pb["unary"])
    else:
        c.set_unary_from_label(pb["label"], pb["conf"])
    for f, w in pb["kernels"]:
        c.add_pairwise(f, w)
    return c


def check_against_expected(c, exp, q_tol=None):
    """Step a prepared CRF and compare with the reference's recorded outputs.

    q_tol=None  -> Q must be bit-identical;  q_tol=x -> max|dQ| <= x.
    Labels and lattice sizes must always be identical.
    """
    for k, V in enumerate(exp["V"]):
        kv = c.kernel(k)
        assert kv["V"] == V, "kernel %d: V=%d, reference M_=%d" % (k, kv["V"], V)
        if q_tol is None:
            assert same_bits(kv["norm"], exp["norm"][k]), "kernel %d: norm differs" % k
        else:
            np.testing.assert_allclose(kv["norm"], exp["norm"][k], rtol=1e-5, atol=0)
    iters = sorted(exp["Q"])
    c.start_inference()
    for t in range(max(iters) + 1):
        if t:
            c.step_inference(exp["relax"])
        if t in exp["Q"]:
            q = c.probability()
            if q_tol is None:
                assert same_bits(q, exp["Q"][t]), "Q differs after %d iterations (max %g)" % (
                    t, np.abs(q - exp["Q"][t]).max() if q.size else 0.0)
            else:
                assert q.shape == exp["Q"][t].shape
                if q.size:
                    assert np.abs(q - exp["Q"][t]).max() <= q_tol
            c.build_map()
            assert np.array_equal(c.map(), exp["map"][t]), "labels differ after %d iterations" % t
