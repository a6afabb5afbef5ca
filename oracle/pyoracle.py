"""ctypes bindings for the CHECKER libraries -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

  OracleCRF  -> oracle/liblccrf_oracle.so   (own C restatement, travels to the GPU box)
  RefCRF     -> oracle/_ref/liblccrf_ref.so (the reference's headers compiled in place;
                                             prebuilt file travels, sources do not)

Both expose the reference's operator surface (densecrf_base.h:54-91) on numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liblccrf_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "liblccrf_ref.so")

_f32p = C.POINTER(C.c_float)
_i16p = C.POINTER(C.c_int16)
_i32p = C.POINTER(C.c_int)


def build(quiet=True):
    """(Re)build the checker libraries with oracle/Makefile."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def have_ref():
    return os.path.exists(REF_SO)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t):
    return a.ctypes.data_as(t)


class _Lattice(C.Structure):
    _fields_ = [("N", C.c_int), ("Npad", C.c_int), ("d", C.c_int), ("V", C.c_int),
                ("offset", _i32p), ("bary", _f32p), ("nbr", _i32p), ("keys", _i16p)]


class _Pairwise(C.Structure):
    _fields_ = [("lat", _Lattice), ("w", C.c_float), ("norm", _f32p)]


class _Crf(C.Structure):
    _fields_ = [("N", C.c_int), ("L", C.c_int), ("K", C.c_int),
                ("unary", _f32p), ("current", _f32p), ("next", _f32p), ("tmp", _f32p),
                ("map", _i16p), ("pw", C.POINTER(_Pairwise) * 8)]


class CrfParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "w1", "w2", "u_alpha", "stdev_alpha", "u_beta", "stdev_beta", "u_gamma",
        "stdev_gamma", "point3d_stdev", "point2d_stdev", "u_depth", "pth", "confidence")]


_olib = None


def oracle_lib():
    global _olib
    if _olib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        lib = C.CDLL(ORACLE_SO)
        lib.orc_crf_create.restype = C.POINTER(_Crf)
        lib.orc_crf_create.argtypes = [C.c_int, C.c_int]
        lib.orc_crf_destroy.argtypes = [C.POINTER(_Crf)]
        lib.orc_crf_set_unary.argtypes = [C.POINTER(_Crf), _f32p]
        lib.orc_crf_set_unary_from_label.argtypes = [C.POINTER(_Crf), _i16p, _f32p]
        lib.orc_crf_add_pairwise.argtypes = [C.POINTER(_Crf), _f32p, C.c_int, C.c_float]
        lib.orc_crf_start_inference.argtypes = [C.POINTER(_Crf)]
        lib.orc_crf_step_inference.argtypes = [C.POINTER(_Crf), C.c_float]
        lib.orc_crf_build_map.argtypes = [C.POINTER(_Crf)]
        lib.orc_pairwise_apply.argtypes = [C.POINTER(_Crf), C.c_int, _f32p, _f32p]
        lib.orc_crf_inference.argtypes = [C.POINTER(_Crf), C.c_int, C.c_int, C.c_float]
        lib.orc_exp_and_normalize.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_float, C.c_float]
        lib.orc_fast_exp.restype = C.c_float
        lib.orc_fast_exp.argtypes = [C.c_float]
        lib.orc_lattice_init.argtypes = [C.POINTER(_Lattice), _f32p, C.c_int, C.c_int]
        lib.orc_lattice_compute.argtypes = [C.POINTER(_Lattice), _f32p, _f32p, C.c_int]
        lib.orc_lattice_free.argtypes = [C.POINTER(_Lattice)]
        lib.orc_image_features.argtypes = [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                           C.c_float, _f32p]
        lib.orc_appearance_features.argtypes = [C.c_int, _f32p, _f32p, C.c_float, C.c_float, _f32p]
        lib.orc_smooth_features.argtypes = [C.c_int, _f32p, C.c_float, _f32p]
        lib.orc_default_params.argtypes = [C.POINTER(CrfParams)]
        lib.orc_rough_classify.argtypes = [C.c_int, _f32p, _f32p, _f32p, C.POINTER(C.c_double),
                                           C.POINTER(CrfParams), _i16p]
        lib.orc_map_point_err_observ.argtypes = [
            C.c_int, _f32p, _f32p, _f32p, _f32p, C.POINTER(C.c_double),
            C.POINTER(C.c_int), _f32p, _f32p]
        lib.orc_unary_build.argtypes = [C.c_int, _f32p, _i32p, _i32p, C.POINTER(C.c_double), _f32p, _f32p, _f32p,
                                        C.POINTER(C.c_double), C.POINTER(CrfParams), _f32p, _f32p, _f32p, _i16p]
        lib.orc_bf_match.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_double, _i32p, _i32p]
        lib.orc_pose_optimization.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_void_p, _f32p, C.c_float, _f32p, _f32p,
                                              C.c_void_p, C.POINTER(C.c_int)]
        _olib = lib
    return _olib


_rlib = None


def ref_lib():
    global _rlib
    if _rlib is None:
        if not have_ref():
            raise FileNotFoundError(REF_SO + " (build it where /root/reference exists: make -C oracle)")
        lib = C.CDLL(REF_SO)
        lib.ref_crf_create.restype = C.c_void_p
        lib.ref_crf_create.argtypes = [C.c_int, C.c_int]
        lib.ref_crf_destroy.argtypes = [C.c_void_p]
        lib.ref_crf_set_unary.argtypes = [C.c_void_p, _f32p]
        lib.ref_crf_set_unary_from_label.argtypes = [C.c_void_p, _i16p, _f32p]
        lib.ref_crf_add_pairwise.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_float]
        lib.ref_crf_start_inference.argtypes = [C.c_void_p]
        lib.ref_crf_step_inference.argtypes = [C.c_void_p, C.c_float]
        lib.ref_crf_inference.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
        lib.ref_crf_build_map.argtypes = [C.c_void_p]
        lib.ref_kernel_apply.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        lib.ref_crf_probability.restype = _f32p
        lib.ref_crf_probability.argtypes = [C.c_void_p]
        lib.ref_crf_map.restype = _i16p
        lib.ref_crf_map.argtypes = [C.c_void_p]
        lib.ref_kernel_V.argtypes = [C.c_void_p, C.c_int]
        for n, t in (("norm", _f32p), ("offset", _i32p), ("bary", _f32p), ("nbr", _i32p)):
            f = getattr(lib, "ref_kernel_" + n)
            f.restype = t
            f.argtypes = [C.c_void_p, C.c_int]
        lib.ref_lattice_filter.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int]
        lib.ref_fast_exp.restype = C.c_float
        lib.ref_fast_exp.argtypes = [C.c_float]
        lib.ref_example_image.argtypes = [C.c_int, C.c_int, C.c_void_p, _i16p, C.c_float,
                                          C.c_int, _i16p]
        _rlib = lib
    return _rlib


class _CrfBase:
    """Shared numpy-facing surface; subclasses bind one library."""

    def inference(self, n_iter, with_map=False, relax=1.0):
        self.start_inference()
        for _ in range(n_iter):
            self.step_inference(relax)
        if with_map:
            self.build_map()

    def run_trace(self, n_iter, relax=1.0):
        """Q after startInference and after every step: [n_iter+1, N, L]."""
        out = np.empty((n_iter + 1, self.N, self.L), np.float32)
        self.start_inference()
        out[0] = self.probability()
        for t in range(n_iter):
            self.step_inference(relax)
            out[t + 1] = self.probability()
        return out


class OracleCRF(_CrfBase):
    def __init__(self, N, L):
        self.lib = oracle_lib()
        self.N, self.L = int(N), int(L)
        self.h = self.lib.orc_crf_create(self.N, self.L)
        if not self.h:
            raise MemoryError("orc_crf_create")

    def close(self):
        if self.h:
            self.lib.orc_crf_destroy(self.h)
            self.h = None

    __del__ = close

    def set_unary(self, unary):
        u = _f32(unary).reshape(-1)
        assert u.size == self.N * self.L
        self.lib.orc_crf_set_unary(self.h, _ptr(u, _f32p))

    def set_unary_from_label(self, label, conf):
        lab = np.ascontiguousarray(label, np.int16)
        cf = _f32(np.broadcast_to(np.asarray(conf, np.float32), (self.L,)))
        self.lib.orc_crf_set_unary_from_label(self.h, _ptr(lab, _i16p), _ptr(cf, _f32p))

    def add_pairwise(self, features, w):
        f = _f32(features)
        f = f.reshape(self.N, f.shape[-1] if f.ndim == 2 else -1)
        rc = self.lib.orc_crf_add_pairwise(self.h, _ptr(f, _f32p), f.shape[1], float(w))
        if rc:
            raise RuntimeError("orc_crf_add_pairwise rc=%d" % rc)

    def start_inference(self):
        self.lib.orc_crf_start_inference(self.h)

    def step_inference(self, relax=1.0):
        self.lib.orc_crf_step_inference(self.h, float(relax))

    def build_map(self):
        self.lib.orc_crf_build_map(self.h)

    def inference_native(self, n_iter, with_map=True, relax=1.0):
        self.lib.orc_crf_inference(self.h, int(n_iter), int(with_map), float(relax))

    def apply(self, k, out, x):
        """PairwisePotential::apply of kernel k: returns out + w * norm * compute(x)."""
        o = _f32(out).reshape(self.N, self.L).copy()
        xin = _f32(x).reshape(self.N, self.L)
        self.lib.orc_pairwise_apply(self.h, int(k), _ptr(o, _f32p), _ptr(xin, _f32p))
        return o

    def probability(self):
        return np.ctypeslib.as_array(self.h.contents.current, (self.N * self.L,)).reshape(
            self.N, self.L).copy()

    def unary(self):
        return np.ctypeslib.as_array(self.h.contents.unary, (self.N * self.L,)).reshape(
            self.N, self.L).copy()

    def map(self):
        if self.N == 0:
            return np.zeros(0, np.int16)
        return np.ctypeslib.as_array(self.h.contents.map, (self.N,)).copy()

    def kernel(self, k):
        p = self.h.contents.pw[k].contents
        lat = p.lat
        D1, V, N = lat.d + 1, lat.V, lat.N
        n = max(N, 0)
        return dict(
            d=lat.d, V=V,
            norm=np.ctypeslib.as_array(p.norm, (max(n, 1),))[:n].copy(),
            offset=np.ctypeslib.as_array(lat.offset, (max(n * D1, 1),))[:n * D1].reshape(n, D1).copy(),
            bary=np.ctypeslib.as_array(lat.bary, (max(n * D1, 1),))[:n * D1].reshape(n, D1).copy(),
            nbr=np.ctypeslib.as_array(lat.nbr, (max(D1 * V * 2, 1),))[:D1 * V * 2].reshape(D1, V, 2).copy(),
            keys=np.ctypeslib.as_array(lat.keys, (max(V * lat.d, 1),))[:V * lat.d].reshape(V, lat.d).copy(),
        )


class RefCRF(_CrfBase):
    def __init__(self, N, L):
        self.lib = ref_lib()
        self.N, self.L = int(N), int(L)
        self.h = self.lib.ref_crf_create(self.N, self.L)
        if not self.h:
            raise ValueError("reference shim instantiates L in {2,3,4,21} only")
        self._d = []

    def close(self):
        if self.h:
            self.lib.ref_crf_destroy(self.h)
            self.h = None

    __del__ = close

    def set_unary(self, unary):
        u = _f32(unary).reshape(-1)
        assert u.size == self.N * self.L
        self.lib.ref_crf_set_unary(self.h, _ptr(u, _f32p))

    def set_unary_from_label(self, label, conf):
        lab = np.ascontiguousarray(label, np.int16)
        cf = _f32(np.broadcast_to(np.asarray(conf, np.float32), (self.L,))).copy()
        self.lib.ref_crf_set_unary_from_label(self.h, _ptr(lab, _i16p), _ptr(cf, _f32p))

    def add_pairwise(self, features, w):
        f = _f32(features)
        f = f.reshape(self.N, f.shape[-1] if f.ndim == 2 else -1)
        rc = self.lib.ref_crf_add_pairwise(self.h, _ptr(f, _f32p), f.shape[1], float(w))
        if rc:
            raise ValueError("reference shim instantiates d in 1..6 only")
        self._d.append(f.shape[1])

    def apply(self, k, out, x):
        o = _f32(out).reshape(self.N, self.L).copy()
        xin = _f32(x).reshape(self.N, self.L)
        self.lib.ref_kernel_apply(self.h, int(k), _ptr(o, _f32p), _ptr(xin, _f32p))
        return o

    def start_inference(self):
        self.lib.ref_crf_start_inference(self.h)

    def step_inference(self, relax=1.0):
        self.lib.ref_crf_step_inference(self.h, float(relax))

    def build_map(self):
        self.lib.ref_crf_build_map(self.h)

    def inference_native(self, n_iter, with_map=True, relax=1.0):
        self.lib.ref_crf_inference(self.h, int(n_iter), int(with_map), float(relax))

    def probability(self):
        if self.N == 0:
            return np.zeros((0, self.L), np.float32)
        p = self.lib.ref_crf_probability(self.h)
        return np.ctypeslib.as_array(p, (self.N * self.L,)).reshape(self.N, self.L).copy()

    def map(self):
        if self.N == 0:
            return np.zeros(0, np.int16)
        return np.ctypeslib.as_array(self.lib.ref_crf_map(self.h), (self.N,)).copy()

    def kernel(self, k):
        d = self._d[k]
        D1, N = d + 1, self.N
        V = self.lib.ref_kernel_V(self.h, k)
        n = max(N, 0)

        def arr(p, cnt, shape):
            if cnt == 0:
                return np.zeros(shape, np.ctypeslib.as_array(p, (1,)).dtype)
            return np.ctypeslib.as_array(p, (cnt,)).reshape(shape).copy()

        return dict(
            d=d, V=V,
            norm=arr(self.lib.ref_kernel_norm(self.h, k), n, (n,)),
            offset=arr(self.lib.ref_kernel_offset(self.h, k), n * D1, (n, D1)),
            bary=arr(self.lib.ref_kernel_bary(self.h, k), n * D1, (n, D1)),
            nbr=arr(self.lib.ref_kernel_nbr(self.h, k), D1 * V * 2, (D1, V, 2)),
        )


def oracle_lattice_filter(features, x):
    """y = compute(x) on the lattice of `features` (oracle).  Returns (y, V)."""
    lib = oracle_lib()
    f = _f32(features)
    N, d = f.shape
    xin = _f32(x).reshape(N, -1)
    out = np.empty_like(xin)
    lat = _Lattice()
    rc = lib.orc_lattice_init(C.byref(lat), _ptr(f, _f32p), d, N)
    if rc:
        raise RuntimeError("orc_lattice_init rc=%d" % rc)
    lib.orc_lattice_compute(C.byref(lat), _ptr(out, _f32p), _ptr(xin, _f32p), xin.shape[1])
    V = lat.V
    lib.orc_lattice_free(C.byref(lat))
    return out, V


def ref_lattice_filter(features, x):
    lib = ref_lib()
    f = _f32(features)
    N, d = f.shape
    xin = _f32(x).reshape(N, -1)
    out = np.empty_like(xin)
    V = lib.ref_lattice_filter(_ptr(f, _f32p), d, N, _ptr(xin, _f32p), _ptr(out, _f32p), xin.shape[1])
    return out, V


def oracle_image_features(W, H, posdev, rgb=None, featuredev=0.0):
    lib = oracle_lib()
    Cn = 0 if rgb is None else rgb.shape[-1]
    out = np.empty((W * H, 2 + Cn), np.float32)
    p = None
    if rgb is not None:
        rgb = np.ascontiguousarray(rgb, np.uint8)
        p = rgb.ctypes.data_as(C.c_void_p)
    lib.orc_image_features(W, H, float(posdev), p, Cn, float(featuredev), _ptr(out, _f32p))
    return out


def oracle_rough_classify(obs, err, depth, params=None, match_prob=None):
    lib = oracle_lib()
    if params is None:
        params = CrfParams()
        lib.orc_default_params(C.byref(params))
    obs, err, depth = _f32(obs), _f32(err), _f32(depth)
    out = np.empty(obs.size, np.int16)
    mp = None
    if match_prob is not None:
        match_prob = np.ascontiguousarray(match_prob, np.float64)
        mp = match_prob.ctypes.data_as(C.POINTER(C.c_double))
    lib.orc_rough_classify(obs.size, _ptr(obs, _f32p), _ptr(err, _f32p), _ptr(depth, _f32p), mp,
                           C.byref(params), _ptr(out, _i16p))
    return out


def default_params():
    p = CrfParams()
    oracle_lib().orc_default_params(C.byref(p))
    return p


def oracle_unary_build(Xw, obs_ptr, obs_kf, obs_kp, kf_pose, kf_intr, kf_bounds, match_prob=None, params=None):
    """Whole-frame ComputeMapPointErrAndObserv + RroughClassify (oracle)."""
    lib = oracle_lib()
    if params is None:
        params = default_params()
    Xw = _f32(Xw).reshape(-1, 3)
    n = Xw.shape[0]
    ptr = np.ascontiguousarray(obs_ptr, np.int32)
    kf = np.ascontiguousarray(obs_kf, np.int32)
    kp = np.ascontiguousarray(obs_kp, np.float64).reshape(-1, 2)
    pose, intr, bnd = _f32(kf_pose).reshape(-1, 12), _f32(kf_intr).reshape(-1, 4), _f32(kf_bounds).reshape(-1, 4)
    mp = None
    if match_prob is not None:
        match_prob = np.ascontiguousarray(match_prob, np.float64)
        mp = match_prob.ctypes.data_as(C.POINTER(C.c_double))
    obs, err, dep = (np.empty(n, np.float32) for _ in range(3))
    lab = np.empty(n, np.int16)
    lib.orc_unary_build(n, _ptr(Xw, _f32p), _ptr(ptr, _i32p), _ptr(kf, _i32p),
                        kp.ctypes.data_as(C.POINTER(C.c_double)), _ptr(pose, _f32p), _ptr(intr, _f32p),
                        _ptr(bnd, _f32p), mp, C.byref(params), _ptr(obs, _f32p), _ptr(err, _f32p),
                        _ptr(dep, _f32p), _ptr(lab, _i16p))
    return obs, err, dep, lab


def oracle_bf_match(desc_query, desc_train, ratio=0.6):
    """Tracking::BfMatch (oracle): train index per query or -1, and the number of matches."""
    lib = oracle_lib()
    q = np.ascontiguousarray(desc_query, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(desc_train, np.uint8).reshape(-1, 32)
    out = np.empty(q.shape[0], np.int32)
    nm = np.zeros(1, np.int32)
    lib.orc_bf_match(q.shape[0], q.ctypes.data, t.shape[0], t.ctypes.data, float(ratio), _ptr(out, _i32p), _ptr(nm, _i32p))
    return out, int(nm[0])


def oracle_pose_optimization(Xw, kp, u_right, inv_sigma2, valid, K4, bf, Tcw):
    """Optimizer::PoseOptimization restated (parity unpinned, see lccrf_oracle.c): returns (Tcw_out [4,4] f32, outlier
    u8[n], n_inliers, n_initial)."""
    lib = oracle_lib()
    Xw, kp = _f32(Xw).reshape(-1, 3), _f32(kp).reshape(-1, 2)
    n = Xw.shape[0]
    ur, is2 = _f32(u_right), _f32(inv_sigma2)
    va = np.ascontiguousarray(valid, np.uint8)
    K, T = _f32(K4), _f32(Tcw).reshape(16)
    out = np.empty(16, np.float32)
    outl = np.zeros(n, np.uint8)
    ninit = C.c_int(0)
    r = lib.orc_pose_optimization(n, _ptr(Xw, _f32p), _ptr(kp, _f32p), _ptr(ur, _f32p), _ptr(is2, _f32p), va.ctypes.data,
                                  _ptr(K, _f32p), float(bf), _ptr(T, _f32p), _ptr(out, _f32p), outl.ctypes.data, C.byref(ninit))
    return out.reshape(4, 4), outl, int(r), ninit.value
