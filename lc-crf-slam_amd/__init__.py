"""lc-crf-slam_amd -- MI355X-native dense-CRF mean-field path of LC-CRF-SLAM.

Python here is plumbing only: a ctypes binding of the C-ABI in include/lccrf.h
(liblccrf_hip.so, hand-written gfx950 HIP kernels) plus numpy-facing mirrors of the
reference's operator interface (densecrf_base.h: DenseCRF / PairwisePotential) so that
the parity tests read like the reference's call site (src/Tracking.cc:1919-1930).

There is no CPU fallback: if the library is missing, or no GPU is usable, calls raise.

Import with  importlib.import_module("lc-crf-slam_amd")  (the directory name is fixed
by the project layout and is not a valid identifier).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LCCRF_LIB selects another build of the same library (e.g. liblccrf_hip_instr.so from `make INSTRUMENT=1`,
# used by scripts/gpu_stamps.sh); it must exist -- there is no fallback of any kind.
LIB_PATH = os.environ.get("LCCRF_LIB") or os.path.join(_HERE, "liblccrf_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lccrf.h")

MAX_KERNELS = 8
OPT_SINGLE_WORKGROUP = 1        # lccrf_option (include/lccrf.h)
OPT_VERTEX_ORDER = 2
OK = 0
_STATUS = {0: "OK", -1: "E_INVALID", -2: "E_NO_DEVICE", -3: "E_HIP", -4: "E_NOMEM", -5: "E_STATE",
           -6: "E_CAPACITY"}

_f32p = C.POINTER(C.c_float)
_i16p = C.POINTER(C.c_int16)
_i32p = C.POINTER(C.c_int32)


class LccrfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lccrf %s (%d): %s" % (_STATUS.get(code, "?"), code, msg))
        self.code = code


class CrfParams(C.Structure):
    """lccrf_crf_params: the CRF block of TUM3.yaml (Tracking.cc:151-171)."""
    _fields_ = [(n, C.c_float) for n in (
        "w1", "w2", "u_alpha", "stdev_alpha", "u_beta", "stdev_beta", "u_gamma", "stdev_gamma",
        "point3d_stdev", "point2d_stdev", "u_depth", "pth", "confidence")]


class BatchDesc(C.Structure):
    _fields_ = [("max_frames", C.c_int), ("max_points", C.c_int), ("n_labels", C.c_int),
                ("n_kernels", C.c_int), ("feat_dims", C.c_int * MAX_KERNELS),
                ("weights", C.c_float * MAX_KERNELS)]


def build_library(quiet=True):
    """Compile liblccrf_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", _HERE, "-j4", "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)
    return LIB_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            LIB_PATH + " is missing: build it with `make -C lc-crf-slam_amd` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.lccrf_last_error.restype = C.c_char_p
    L.lccrf_device_count.argtypes = [C.POINTER(C.c_int)]
    L.lccrf_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int]
    L.lccrf_destroy.argtypes = [vp]
    L.lccrf_destroy.restype = None
    L.lccrf_trim_cache.argtypes = []
    L.lccrf_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.lccrf_batch_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.lccrf_set_default_option.argtypes = [C.c_int, C.c_int]
    L.lccrf_set_unary.argtypes = [vp, _f32p]
    L.lccrf_set_unary_from_label.argtypes = [vp, _i16p, _f32p]
    L.lccrf_add_pairwise.argtypes = [vp, _f32p, C.c_int, C.c_float]
    L.lccrf_add_appearance_kernel.argtypes = [vp, C.c_float, _f32p, _f32p, C.c_float, C.c_float]
    L.lccrf_add_smooth_kernel.argtypes = [vp, C.c_float, _f32p, C.c_float]
    L.lccrf_start_inference.argtypes = [vp]
    L.lccrf_step_inference.argtypes = [vp, C.c_float]
    L.lccrf_build_map.argtypes = [vp]
    L.lccrf_inference.argtypes = [vp, C.c_int, C.c_int, C.c_float]
    L.lccrf_get_map.argtypes = [vp, _i16p]
    L.lccrf_get_probability.argtypes = [vp, _f32p]
    L.lccrf_get_unary.argtypes = [vp, _f32p]
    L.lccrf_get_lattice_size.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.lccrf_pairwise_apply.argtypes = [vp, C.c_int, _f32p, _f32p]
    L.lccrf_exp_and_normalize.argtypes = [vp, _f32p, _f32p, C.c_float, C.c_float]
    L.lccrf_step_init.argtypes = [vp, _f32p]
    L.lccrf_map_of.argtypes = [vp, _f32p, _i16p]
    L.lccrf_lattice_filter.argtypes = [C.c_int, _f32p, C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.POINTER(C.c_int)]
    L.lccrf_get_norm.argtypes = [vp, C.c_int, _f32p]
    L.lccrf_get_lattice.argtypes = [vp, C.c_int, _i32p, _f32p, _i32p]
    L.lccrf_batch_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(BatchDesc)]
    L.lccrf_batch_destroy.argtypes = [vp]
    L.lccrf_batch_destroy.restype = None
    L.lccrf_batch_set_inputs_host.argtypes = [vp, C.c_int, _i32p, _f32p, _i16p, _f32p, C.POINTER(_f32p)]
    L.lccrf_batch_bind_inputs_device.argtypes = [vp, C.c_int, vp, vp, vp, _f32p, C.POINTER(vp)]
    L.lccrf_batch_set_inputs_host_async.argtypes = [vp, C.c_int, vp, vp, vp, _f32p, C.POINTER(vp), C.c_int]
    L.lccrf_batch_wait_inputs.argtypes = [vp]
    L.lccrf_batch_download_async.argtypes = [vp, C.c_int]
    L.lccrf_batch_wait_download.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(vp), C.POINTER(vp)]
    L.lccrf_batch_build.argtypes = [vp, vp]
    L.lccrf_batch_inference.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp]
    L.lccrf_batch_run.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp]
    L.lccrf_batch_synchronize.argtypes = [vp]
    L.lccrf_batch_get_map_host.argtypes = [vp, _i16p]
    L.lccrf_batch_get_probability_host.argtypes = [vp, _f32p]
    L.lccrf_batch_get_lattice_sizes_host.argtypes = [vp, C.c_int, _i32p]
    L.lccrf_batch_get_norm_host.argtypes = [vp, C.c_int, _f32p]
    L.lccrf_batch_device_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.lccrf_batch_device_label_bits.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int)]
    L.lccrf_batch_set_engine.argtypes = [vp, C.c_int]
    L.lccrf_batch_get_engine.argtypes = [vp, C.POINTER(C.c_int)]
    L.lccrf_batch_get_fallback_frames.argtypes = [vp, C.POINTER(C.c_int)]
    L.lccrf_batch_get_fused_shape.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lccrf_batch_get_locality_mode.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lccrf_batch_pose_set_crf_counts.argtypes = [vp, vp]
    L.lccrf_batch_last_timing.argtypes = [vp, _f32p, _f32p]
    L.lccrf_batch_last_prepare.argtypes = [vp, _f32p, C.POINTER(C.c_int)]
    L.lccrf_batch_get_stream.argtypes = [vp, C.POINTER(vp)]
    L.lccrf_batch_time_blur_pass.argtypes = [vp, C.c_int, C.c_int, _f32p, C.POINTER(C.c_int64)]
    L.lccrf_bf_match.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_double, _i32p, _i32p]
    L.lccrf_pose_optimization.argtypes = [C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_void_p, _i16p, _f32p, C.c_float,
                                          _f32p, _f32p, C.c_void_p, _i32p]
    L.lccrf_batch_pose_optimization.argtypes = [vp, vp, vp, vp, vp, vp, _f32p, C.c_float, vp, vp, vp, vp, vp, vp]
    L.lccrf_default_params.argtypes = [C.POINTER(CrfParams)]
    L.lccrf_default_params.restype = None
    L.lccrf_unary_build.argtypes = [C.c_int, C.c_int, _f32p, _i32p, _i32p, C.POINTER(C.c_double), C.c_int, _f32p,
                                    _f32p, _f32p, C.POINTER(C.c_double), C.POINTER(CrfParams), _f32p, _f32p, _f32p,
                                    _i16p]
    _lib = L
    return L


def _check(rc):
    if rc != OK:
        raise LccrfError(rc, lib().lccrf_last_error().decode("utf-8", "replace"))


def set_default_option(option, value):
    _check(lib().lccrf_set_default_option(int(option), int(value)))


def device_count():
    n = C.c_int(0)
    _check(lib().lccrf_device_count(C.byref(n)))
    return n.value


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(t)


class DenseCRFHIP:
    """Mirror of DenseCRF3D<M> + PottsPotential3D<M,F> (densecrf3d.h, pairwise3d.h) on the
    HIP path.  Same method meaning and call order as the reference; host arrays in/out."""

    def __init__(self, N, L, device=0):
        self.N, self.L = int(N), int(L)
        self._d = []
        self.h = C.c_void_p()
        _check(lib().lccrf_create(C.byref(self.h), int(device), self.N, self.L))

    def close(self):
        if getattr(self, "h", None):
            lib().lccrf_destroy(self.h)
            self.h = None

    __del__ = close

    def set_option(self, option, value):
        _check(lib().lccrf_set_option(self.h, int(option), int(value)))

    # -- unary -------------------------------------------------------------------------
    def set_unary(self, unary):
        u = _f32(unary).reshape(-1)
        if u.size != self.N * self.L:
            raise ValueError("unary must have N*L entries")
        _check(lib().lccrf_set_unary(self.h, _p(u, _f32p)))

    def set_unary_from_label(self, label, conf):
        lab = np.ascontiguousarray(label, np.int16)
        if lab.size != self.N:
            raise ValueError("label must have N entries")
        cf = _f32(np.broadcast_to(np.asarray(conf, np.float32), (self.L,)))
        _check(lib().lccrf_set_unary_from_label(self.h, _p(lab, _i16p), _p(cf, _f32p)))

    # -- pairwise ----------------------------------------------------------------------
    def add_pairwise(self, features, w):
        f = _f32(features)
        f = f.reshape(self.N, f.shape[-1] if f.ndim == 2 else -1)
        _check(lib().lccrf_add_pairwise(self.h, _p(f, _f32p), f.shape[1], float(w)))
        self._d.append(f.shape[1])

    def add_appearance_kernel(self, w, vobserv, verror, sd_observ, sd_error):
        a, b = _f32(vobserv), _f32(verror)
        _check(lib().lccrf_add_appearance_kernel(self.h, float(w), _p(a, _f32p), _p(b, _f32p),
                                                 float(sd_observ), float(sd_error)))
        self._d.append(2)

    def add_smooth_kernel(self, w, xy, sd2d):
        a = _f32(xy)
        _check(lib().lccrf_add_smooth_kernel(self.h, float(w), _p(a, _f32p), float(sd2d)))
        self._d.append(2)

    # -- inference ---------------------------------------------------------------------
    def start_inference(self):
        _check(lib().lccrf_start_inference(self.h))

    def step_inference(self, relax=1.0):
        _check(lib().lccrf_step_inference(self.h, float(relax)))

    def build_map(self):
        _check(lib().lccrf_build_map(self.h))

    def inference(self, n_iter, with_map=False, relax=1.0):
        _check(lib().lccrf_inference(self.h, int(n_iter), int(bool(with_map)), float(relax)))

    inference_native = inference

    def run_trace(self, n_iter, relax=1.0):
        out = np.empty((n_iter + 1, self.N, self.L), np.float32)
        self.start_inference()
        out[0] = self.probability()
        for t in range(n_iter):
            self.step_inference(relax)
            out[t + 1] = self.probability()
        return out

    # -- the reference's plug-in points (densecrf_base.h:18,34-36) on host arrays ------------
    def apply(self, k, out, x):
        """PairwisePotential::apply of kernel k: returns out + w * norm * compute(x)."""
        o = _f32(out).reshape(self.N, self.L).copy()
        xin = _f32(x).reshape(self.N, self.L)
        _check(lib().lccrf_pairwise_apply(self.h, int(k), _p(o, _f32p), _p(xin, _f32p)))
        return o

    def exp_and_normalize(self, x, scale=1.0, relax=1.0, old=None):
        xin = _f32(x).reshape(self.N, self.L)
        o = np.zeros_like(xin) if old is None else _f32(old).reshape(self.N, self.L).copy()
        _check(lib().lccrf_exp_and_normalize(self.h, _p(o, _f32p), _p(xin, _f32p), float(scale), float(relax)))
        return o

    def step_init(self):
        o = np.empty((self.N, self.L), np.float32)
        _check(lib().lccrf_step_init(self.h, _p(o, _f32p)))
        return o

    def map_of(self, prob):
        p = _f32(prob).reshape(self.N, self.L)
        m = np.empty(self.N, np.int16)
        _check(lib().lccrf_map_of(self.h, _p(p, _f32p), _p(m, _i16p)))
        return m

    # -- results -----------------------------------------------------------------------
    def map(self):
        out = np.empty(self.N, np.int16)
        _check(lib().lccrf_get_map(self.h, _p(out, _i16p)))
        return out

    def probability(self):
        out = np.empty((self.N, self.L), np.float32)
        _check(lib().lccrf_get_probability(self.h, _p(out, _f32p)))
        return out

    def unary(self):
        out = np.empty((self.N, self.L), np.float32)
        _check(lib().lccrf_get_unary(self.h, _p(out, _f32p)))
        return out

    def kernel(self, k):
        d = self._d[k]
        V = C.c_int(0)
        _check(lib().lccrf_get_lattice_size(self.h, k, C.byref(V)))
        V = V.value
        norm = np.empty(self.N, np.float32)
        off = np.empty((self.N, d + 1), np.int32)
        bary = np.empty((self.N, d + 1), np.float32)
        nbr = np.empty((d + 1, V, 2), np.int32)
        _check(lib().lccrf_get_norm(self.h, k, _p(norm, _f32p)))
        _check(lib().lccrf_get_lattice(self.h, k, _p(off, _i32p), _p(bary, _f32p), _p(nbr, _i32p)))
        return dict(d=d, V=V, norm=norm, offset=off, bary=bary, nbr=nbr)


class BatchCRF:
    """Frames-in-flight: F independent CRFs with a common stride (include/lccrf.h section 2)."""

    def __init__(self, max_frames, max_points, n_labels, feat_dims, weights, device=0):
        self.F, self.maxN, self.L = int(max_frames), int(max_points), int(n_labels)
        self.dims = [int(d) for d in feat_dims]
        d = BatchDesc()
        d.max_frames, d.max_points, d.n_labels, d.n_kernels = self.F, self.maxN, self.L, len(self.dims)
        for i, (fd, w) in enumerate(zip(self.dims, weights)):
            d.feat_dims[i] = fd
            d.weights[i] = float(w)
        self.h = C.c_void_p()
        self.n_frames = 0
        _check(lib().lccrf_batch_create(C.byref(self.h), int(device), C.byref(d)))
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            lib().lccrf_batch_destroy(self.h)
            self.h = None

    __del__ = close

    def set_option(self, option, value):
        _check(lib().lccrf_batch_set_option(self.h, int(option), int(value)))

    def set_inputs_host(self, n_points, features, unary=None, label=None, conf=None):
        npts = np.ascontiguousarray(n_points, np.int32)
        F = npts.size
        feats = [_f32(f).reshape(F, self.maxN, d) for f, d in zip(features, self.dims)]
        arr = (_f32p * len(feats))(*[_p(f, _f32p) for f in feats])
        u = l = cf = None
        if unary is not None:
            u = _f32(unary).reshape(F, self.maxN, self.L)
        if label is not None:
            l = np.ascontiguousarray(label, np.int16).reshape(F, self.maxN)
            cf = _f32(np.broadcast_to(np.asarray(conf, np.float32), (self.L,)))
        _check(lib().lccrf_batch_set_inputs_host(
            self.h, F, _p(npts, _i32p), _p(u, _f32p) if u is not None else None,
            _p(l, _i16p) if l is not None else None, _p(cf, _f32p) if cf is not None else None, arr))
        self.n_frames = F

    HOST_PINNED = 1
    DOWNLOAD_LABEL_BITS, DOWNLOAD_MAP, DOWNLOAD_PROBABILITY = 1, 2, 4
    OPT_COPY_THREADS = 3
    OPT_EVENT_TIMING = 4

    def set_inputs_host_async(self, n_points, features, unary=None, label=None, conf=None, pinned=False):
        """lccrf_batch_set_inputs_host_async: the arrays are staged and uploaded without a host wait.  The arrays must already be
        C-contiguous of the right dtype (nothing is converted here: a hidden copy would be timed as part of the call); with
        pinned=True they must live in pinned memory and stay untouched until wait_inputs() or a result of this batch."""
        npts = np.ascontiguousarray(n_points, np.int32)
        F = npts.size

        def addr(a, dtype, count):
            assert a.dtype == dtype and a.flags["C_CONTIGUOUS"] and a.size == count, (a.dtype, a.shape, count)
            return C.c_void_p(a.ctypes.data)
        arr = (C.c_void_p * len(features))(*[addr(f, np.float32, F * self.maxN * d) for f, d in zip(features, self.dims)])
        u = l = cf = None
        if unary is not None:
            u = addr(unary, np.float32, F * self.maxN * self.L)
        if label is not None:
            l = addr(label, np.int16, F * self.maxN)
            cf = _f32(np.broadcast_to(np.asarray(conf, np.float32), (self.L,)))
        _check(lib().lccrf_batch_set_inputs_host_async(
            self.h, F, C.c_void_p(npts.ctypes.data), u, l, _p(cf, _f32p) if cf is not None else None, arr,
            self.HOST_PINNED if pinned else 0))
        self.n_frames = F

    def wait_inputs(self):
        _check(lib().lccrf_batch_wait_inputs(self.h))

    def download_async(self, what=1):
        _check(lib().lccrf_batch_download_async(self.h, int(what)))

    def wait_download(self, copy=True):
        """-> dict with the arrays that were asked for ('bits' uint64 [F][words], 'map' int16 [F][maxN], 'prob' float32 [F][maxN][L]);
        copy=False hands out views of the batch's pinned memory (valid until the next download_async)."""
        bits, mp, pr, words = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
        _check(lib().lccrf_batch_wait_download(self.h, C.byref(bits), C.byref(words), C.byref(mp), C.byref(pr)))
        out, F = {}, self.n_frames

        def view(ptr, ctype, shape, dtype):
            n = int(np.prod(shape))
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (max(n, 1),))[:n].reshape(shape).view(dtype)
            return a.copy() if copy else a
        if bits.value:
            out["bits"] = view(bits, C.c_uint64, (F, words.value), np.uint64)
        if mp.value:
            out["map"] = view(mp, C.c_int16, (F, self.maxN), np.int16)
        if pr.value:
            out["prob"] = view(pr, C.c_float, (F, self.maxN, self.L), np.float32)
        return out

    def bind_inputs_device(self, n_frames, d_n_points, d_features, d_unary=None, d_label=None, conf=None):
        """Pointers are raw device addresses (e.g. torch_tensor.data_ptr())."""
        arr = (C.c_void_p * len(d_features))(*[C.c_void_p(int(p)) for p in d_features])
        cf = None
        if d_label is not None:
            cf = _f32(np.broadcast_to(np.asarray(conf, np.float32), (self.L,)))
        _check(lib().lccrf_batch_bind_inputs_device(
            self.h, int(n_frames), C.c_void_p(int(d_n_points)),
            C.c_void_p(int(d_unary)) if d_unary is not None else None,
            C.c_void_p(int(d_label)) if d_label is not None else None,
            _p(cf, _f32p) if cf is not None else None, arr))
        self.n_frames = int(n_frames)

    def build(self, stream=None):
        _check(lib().lccrf_batch_build(self.h, C.c_void_p(stream) if stream else None))

    def inference(self, n_iter, with_map=True, relax=1.0, stream=None):
        _check(lib().lccrf_batch_inference(self.h, int(n_iter), int(bool(with_map)), float(relax),
                                           C.c_void_p(stream) if stream else None))

    def run(self, n_iter, with_map=True, relax=1.0, stream=None):
        """Lattice build + normalisation + inference of every frame in one launch (lccrf_batch_run)."""
        _check(lib().lccrf_batch_run(self.h, int(n_iter), int(bool(with_map)), float(relax),
                                     C.c_void_p(stream) if stream else None))

    def synchronize(self):
        _check(lib().lccrf_batch_synchronize(self.h))

    def set_engine(self, engine):
        _check(lib().lccrf_batch_set_engine(self.h, int(engine)))

    def engine(self):
        e = C.c_int(0)
        _check(lib().lccrf_batch_get_engine(self.h, C.byref(e)))
        return e.value

    def fused_shape(self):
        """(lanes per frame, frames per CU) of the last fused-engine launch; (0, 0) if there was none."""
        a, b = C.c_int(0), C.c_int(0)
        _check(lib().lccrf_batch_get_fused_shape(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def locality_mode(self):
        """(internal point order in use, sorted build in use) for the lattices now in HBM."""
        a, b = C.c_int(0), C.c_int(0)
        _check(lib().lccrf_batch_get_locality_mode(self.h, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)

    def fallback_frames(self):
        """Frames of the last run() that did not fit the one-launch kernel and were re-run on the two-kernel path."""
        n = C.c_int(0)
        _check(lib().lccrf_batch_get_fallback_frames(self.h, C.byref(n)))
        return n.value

    def pose_set_crf_counts(self, d_n_total):
        """lccrf_batch_pose_set_crf_counts: device int32[F] of CRF points + extra non-CRF edges per frame (None: off)."""
        _check(lib().lccrf_batch_pose_set_crf_counts(self.h, C.c_void_p(int(d_n_total)) if d_n_total else None))

    def map(self):
        out = np.empty((self.n_frames, self.maxN), np.int16)
        _check(lib().lccrf_batch_get_map_host(self.h, _p(out, _i16p)))
        return out

    def probability(self):
        out = np.empty((self.n_frames, self.maxN, self.L), np.float32)
        _check(lib().lccrf_batch_get_probability_host(self.h, _p(out, _f32p)))
        return out

    def lattice_sizes(self, k):
        out = np.empty(self.n_frames, np.int32)
        _check(lib().lccrf_batch_get_lattice_sizes_host(self.h, k, _p(out, _i32p)))
        return out

    def norm(self, k):
        out = np.empty((self.n_frames, self.maxN), np.float32)
        _check(lib().lccrf_batch_get_norm_host(self.h, k, _p(out, _f32p)))
        return out

    def device_buffers(self):
        m, q = C.c_void_p(), C.c_void_p()
        _check(lib().lccrf_batch_device_buffers(self.h, C.byref(m), C.byref(q)))
        return m.value, q.value

    def device_label_bits(self):
        """(device address, words per frame) of the bit-packed MAP labels (binary CRFs only)."""
        p, w = C.c_void_p(), C.c_int(0)
        _check(lib().lccrf_batch_device_label_bits(self.h, C.byref(p), C.byref(w)))
        return p.value, w.value

    def time_blur_pass(self, kernel=0, reps=50):
        """(ms per launch, vertices per launch) of one blur pass of `kernel` over all frames (streaming engine)."""
        ms, nv = C.c_float(0), C.c_int64(0)
        _check(lib().lccrf_batch_time_blur_pass(self.h, int(kernel), int(reps), C.byref(ms), C.byref(nv)))
        return ms.value, nv.value

    def pose_optimization(self, d_Xw, d_kp, d_u_right, d_inv_sigma2, K4, bf, d_Tcw_in, d_Tcw_out, d_outlier, d_n_inliers,
                          d_n_initial, d_valid=None, stream=None):
        """lccrf_batch_pose_optimization: raw device addresses; consumes the labels of the last inference on the device."""
        K = _f32(K4)
        v = lambda p: C.c_void_p(int(p)) if p is not None else None
        _check(lib().lccrf_batch_pose_optimization(self.h, v(d_Xw), v(d_kp), v(d_u_right), v(d_inv_sigma2), v(d_valid), _p(K, _f32p),
                                                   float(bf), v(d_Tcw_in), v(d_Tcw_out), v(d_outlier), v(d_n_inliers),
                                                   v(d_n_initial), C.c_void_p(stream) if stream else None))

    def last_timing(self):
        a, b = C.c_float(0), C.c_float(0)
        _check(lib().lccrf_batch_last_timing(self.h, C.byref(a), C.byref(b)))
        return dict(inference_ms=a.value, build_ms=b.value)

    def own_stream(self):
        """The batch's own hipStream_t as an integer (what stream=None means): wrap it with torch.cuda.ExternalStream to order torch work behind it."""
        p = C.c_void_p()
        _check(lib().lccrf_batch_get_stream(self.h, C.byref(p)))
        return p.value

    def last_prepare(self):
        """(prepare_ms, runs) of the two-frames-per-CU kernel's prepared launch records (include/lccrf.h: lccrf_batch_last_prepare)."""
        ms, n = C.c_float(0), C.c_int(0)
        _check(lib().lccrf_batch_last_prepare(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def lattice_filter(features, x, device=0):
    """PermutohedralLatticeCPU::init + compute on the GPU (lccrf_lattice_filter): returns (y, V)."""
    f = _f32(features)
    N, d = f.shape
    xa = _f32(x)
    xin = xa.reshape(N, xa.shape[-1] if xa.ndim == 2 else max(xa.size // max(N, 1), 1))
    out = np.empty_like(xin)
    V = C.c_int(0)
    _check(lib().lccrf_lattice_filter(int(device), _p(f, _f32p), N, d, _p(xin, _f32p), xin.shape[1], _p(out, _f32p), C.byref(V)))
    return out, V.value


def pose_optimization(Xw, kp, u_right, inv_sigma2, K4, bf, Tcw, valid=None, label=None, device=0):
    """Optimizer::PoseOptimization on the GPU (lccrf_pose_optimization): (Tcw_out [4,4], outlier u8[n], n_inliers)."""
    Xw, kp = _f32(Xw).reshape(-1, 3), _f32(kp).reshape(-1, 2)
    n = Xw.shape[0]
    ur, is2, K, T = _f32(u_right), _f32(inv_sigma2), _f32(K4), _f32(Tcw).reshape(16)
    va = None if valid is None else np.ascontiguousarray(valid, np.uint8)
    la = None if label is None else np.ascontiguousarray(label, np.int16)
    out = np.empty(16, np.float32)
    outl = np.zeros(n, np.uint8)
    ninl = np.zeros(1, np.int32)
    _check(lib().lccrf_pose_optimization(int(device), n, _p(Xw, _f32p), _p(kp, _f32p), _p(ur, _f32p), _p(is2, _f32p),
                                         va.ctypes.data if va is not None else None, _p(la, _i16p) if la is not None else None,
                                         _p(K, _f32p), float(bf), _p(T, _f32p), _p(out, _f32p), outl.ctypes.data, _p(ninl, _i32p)))
    return out.reshape(4, 4), outl, int(ninl[0])


def default_params():
    p = CrfParams()
    lib().lccrf_default_params(C.byref(p))
    return p


def unary_build(Xw, obs_ptr, obs_kf, obs_kp, kf_pose, kf_intr, kf_bounds, match_prob=None, params=None, device=0):
    """ComputeMapPointErrAndObserv + RroughClassify for a whole frame on the GPU
    (src/Tracking.cc:1803-1839, 1961-2013; include/lccrf.h section 3)."""
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
    _check(lib().lccrf_unary_build(int(device), n, _p(Xw, _f32p), _p(ptr, _i32p), _p(kf, _i32p),
                                   kp.ctypes.data_as(C.POINTER(C.c_double)), pose.shape[0], _p(pose, _f32p),
                                   _p(intr, _f32p), _p(bnd, _f32p), mp, C.byref(params), _p(obs, _f32p),
                                   _p(err, _f32p), _p(dep, _f32p), _p(lab, _i16p)))
    return obs, err, dep, lab


def bf_match(desc_query, desc_train, ratio=0.6, device=0):
    """Tracking::BfMatch on the GPU (src/Tracking.cc:1747-1766; include/lccrf.h section 4):
    train index per query or -1, and the number of matches."""
    q = np.ascontiguousarray(desc_query, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(desc_train, np.uint8).reshape(-1, 32)
    out = np.empty(q.shape[0], np.int32)
    nm = np.zeros(1, np.int32)
    _check(lib().lccrf_bf_match(int(device), q.shape[0], q.ctypes.data, t.shape[0], t.ctypes.data, float(ratio),
                                _p(out, _i32p), _p(nm, _i32p)))
    return out, int(nm[0])
