"""Reader / writer of capture-replay records (include/lccrf_record.h).

Pure numpy.  A frame is a dict with the arrays of the call site at
/root/reference/src/Tracking.cc:1871-1930: vobservs, verrors, vdepths (float32 [n]), coord2d
(float32 [n,2]), init_label (int16 [n]), params (dict of the 13 CRF settings), frame_id,
n_iterations, and optionally match_prob (float64 [n]), ref_label (int16 [n]), ref_prob (float32 [n,2]).

Version 2 (read and written here; version-1 files are still read): a frame may carry `sections`, a dict with any of
  "unary"    inputs / outputs of ComputeMapPointErrAndObserv + RroughClassify (Tracking.cc:1803-1839, 1961-2013)
  "bfmatch"  Tracking::BfMatch (Tracking.cc:1747-1766)
  "pose"     Optimizer::PoseOptimization (Optimizer.cc:239-450)
each a dict of the arrays include/lccrf_record.h lists for that section; sections with an unknown tag are kept as
(tag, flags, payload bytes) under `unknown_sections` and written back untouched.
"""
import os
import struct

import numpy as np

MAGIC = b"LCCRFREC"
VERSION = 2
HAS_MATCH_PROB, HAS_REF_LABEL, HAS_REF_PROB, HAS_SECTIONS = 1, 2, 4, 8
ORIGIN_REFERENCE, ORIGIN_SYNTHETIC = 0, 1
SEC_UNARY, SEC_BFMATCH, SEC_POSE = 0x59524e55, 0x544d4642, 0x45534f50      # "UNRY", "BFMT", "POSE"
_SEC_HDR = struct.Struct("<IIQ")                # 16 bytes
PARAM_NAMES = ("w1", "w2", "u_alpha", "stdev_alpha", "u_beta", "stdev_beta", "u_gamma", "stdev_gamma",
               "point3d_stdev", "point2d_stdev", "u_depth", "pth", "confidence")
_FILE_HDR = struct.Struct("<8sIII3I")           # 32 bytes: magic, version, header bytes, frame header bytes, origin, 2 reserved
_FRAME_HDR = struct.Struct("<4I13fI2f")         # 80 bytes: ..., n_sections, 2 reserved


class RecordError(ValueError):
    pass


def write_records(path, frames, origin=ORIGIN_REFERENCE):
    """Write `frames` (an iterable of frame dicts) to `path`; returns the number written.  `origin` says where the
    recorded OUTPUTS come from (ORIGIN_SYNTHETIC: this repository's restatements -- a format sample, not evidence)."""
    n = 0
    with open(path, "wb") as fh:
        fh.write(_FILE_HDR.pack(MAGIC, VERSION, _FILE_HDR.size, _FRAME_HDR.size, int(origin), 0, 0))
        for fr in frames:
            fh.write(encode_frame(fr))
            n += 1
    return n


def encode_frame(fr):
    npts = int(len(fr["init_label"]))
    flags = 0
    parts = []

    def arr(name, dtype, shape):
        a = np.ascontiguousarray(fr[name], dtype=dtype)
        if a.shape != shape:
            raise RecordError("%s has shape %s, expected %s" % (name, a.shape, shape))
        return a.tobytes()

    parts.append(arr("vobservs", "<f4", (npts,)))
    parts.append(arr("verrors", "<f4", (npts,)))
    parts.append(arr("vdepths", "<f4", (npts,)))
    parts.append(arr("coord2d", "<f4", (npts, 2)))
    parts.append(arr("init_label", "<i2", (npts,)))
    if fr.get("match_prob") is not None:
        flags |= HAS_MATCH_PROB
        parts.append(arr("match_prob", "<f8", (npts,)))
    if fr.get("ref_label") is not None:
        flags |= HAS_REF_LABEL
        parts.append(arr("ref_label", "<i2", (npts,)))
    if fr.get("ref_prob") is not None:
        flags |= HAS_REF_PROB
        parts.append(arr("ref_prob", "<f4", (npts, 2)))
    secs = []
    for name, enc in (("unary", _encode_unary), ("bfmatch", _encode_bfmatch), ("pose", _encode_pose)):
        sec = (fr.get("sections") or {}).get(name)
        if sec is not None:
            secs.append(enc(sec))
    for tag, sflags, payload in fr.get("unknown_sections") or []:
        secs.append((int(tag), int(sflags), bytes(payload)))
    if secs:
        flags |= HAS_SECTIONS
    p = fr["params"]
    hdr = _FRAME_HDR.pack(npts, flags, int(fr.get("frame_id", 0)), int(fr.get("n_iterations", 5)),
                          *[np.float32(p[k]) for k in PARAM_NAMES], len(secs), 0.0, 0.0)
    body = b"".join(parts)
    out = hdr + body + b"\0" * (-(len(hdr) + len(body)) % 8)
    for tag, sflags, payload in secs:
        out += _SEC_HDR.pack(tag, sflags, len(payload)) + payload + b"\0" * (-len(payload) % 8)
    return out


def _arr(sec, name, dtype, shape):
    a = np.ascontiguousarray(sec[name], dtype=dtype)
    if a.shape != tuple(shape):
        raise RecordError("section array %s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a.tobytes()


def _pad4(b):
    return b + b"\0" * (-len(b) % 4)


def _encode_unary(u):
    nc, nk = len(u["fid"]), len(u["kf_pose"])
    no = int(np.asarray(u["obs_ptr"])[-1]) if nc else 0
    flags = 1 if u.get("match_prob") is not None else 0
    b = struct.pack("<4I", nc, no, nk, 0)
    b += _arr(u, "Xw", "<f4", (nc, 3)) + _arr(u, "fid", "<i4", (nc,)) + _arr(u, "obs_ptr", "<i4", (nc + 1,))
    b += _arr(u, "obs_kf", "<i4", (no,)) + _arr(u, "obs_kp", "<f8", (no, 2))
    b += _arr(u, "kf_pose", "<f4", (nk, 12)) + _arr(u, "kf_intr", "<f4", (nk, 4)) + _arr(u, "kf_bounds", "<f4", (nk, 4))
    if flags:
        b += _arr(u, "match_prob", "<f8", (nc,))
    b += _arr(u, "observs", "<f4", (nc,)) + _arr(u, "error", "<f4", (nc,)) + _arr(u, "depth", "<f4", (nc,))
    b += _arr(u, "rough_label", "<i2", (nc,))
    return SEC_UNARY, flags, b


def _encode_bfmatch(m):
    nq, nt = len(m["desc_query"]), len(m["desc_train"])
    b = struct.pack("<IId", nq, nt, float(m.get("ratio", 0.6)))
    b += _pad4(_arr(m, "desc_query", "u1", (nq, 32)) + _arr(m, "desc_train", "u1", (nt, 32)))
    b += _arr(m, "asso", "<i4", (nq,))
    return SEC_BFMATCH, 0, b


def _encode_pose(q):
    n = len(q["valid"])
    K = np.asarray(q["K4"], np.float32)
    flags = 1 if q.get("crf_index") is not None else 0
    b = struct.pack("<Ii5fI", n, int(q["n_inliers"]), float(K[0]), float(K[1]), float(K[2]), float(K[3]), float(np.float32(q["bf"])), 0)
    b += _arr(q, "Xw", "<f4", (n, 3)) + _arr(q, "kp", "<f4", (n, 2)) + _arr(q, "u_right", "<f4", (n,)) + _arr(q, "inv_sigma2", "<f4", (n,))
    b += _pad4(_arr(q, "valid", "u1", (n,)) + _arr(q, "outlier", "u1", (n,)))
    b += _arr(q, "Tcw_in", "<f4", (4, 4)) + _arr(q, "Tcw_out", "<f4", (4, 4))
    if flags:
        b += _arr(q, "crf_index", "<i4", (n,))
    return SEC_POSE, flags, b


class _Cursor:
    def __init__(self, b):
        self.b, self.o = b, 0

    def take(self, dtype, shape):
        cnt = 1
        for x in shape:                                 # (Python integers: header counts cannot wrap)
            cnt *= int(x)
        nbytes = cnt * np.dtype(dtype).itemsize
        if self.o + nbytes > len(self.b):
            raise RecordError("section payload too short")
        a = np.frombuffer(self.b, dtype=dtype, count=cnt, offset=self.o).reshape(shape).copy()
        self.o += nbytes
        return a

    def align4(self):
        self.o += -self.o % 4


def _decode_unary(flags, b):
    if len(b) < 16:
        raise RecordError("unary section too short")
    nc, no, nk, _ = struct.unpack_from("<4I", b)
    c = _Cursor(b)
    c.o = 16
    u = dict(Xw=c.take("<f4", (nc, 3)), fid=c.take("<i4", (nc,)), obs_ptr=c.take("<i4", (nc + 1,)), obs_kf=c.take("<i4", (no,)),
             obs_kp=c.take("<f8", (no, 2)), kf_pose=c.take("<f4", (nk, 12)), kf_intr=c.take("<f4", (nk, 4)),
             kf_bounds=c.take("<f4", (nk, 4)))
    u["match_prob"] = c.take("<f8", (nc,)) if flags & 1 else None
    u.update(observs=c.take("<f4", (nc,)), error=c.take("<f4", (nc,)), depth=c.take("<f4", (nc,)), rough_label=c.take("<i2", (nc,)))
    if nc and (int(u["obs_ptr"][-1]) != no or np.any(np.diff(u["obs_ptr"]) < 0) or (no and (u["obs_kf"].min() < 0 or u["obs_kf"].max() >= nk))):
        raise RecordError("unary section: inconsistent observation CSR")
    return u


def _decode_bfmatch(flags, b):
    if len(b) < 16:
        raise RecordError("bfmatch section too short")
    nq, nt, ratio = struct.unpack_from("<IId", b)
    c = _Cursor(b)
    c.o = 16
    m = dict(ratio=ratio, desc_query=c.take("u1", (nq, 32)), desc_train=c.take("u1", (nt, 32)))
    c.align4()
    m["asso"] = c.take("<i4", (nq,))
    return m


def _decode_pose(flags, b):
    if len(b) < 32:
        raise RecordError("pose section too short")
    n, ninl, fx, fy, cx, cy, bf, _ = struct.unpack_from("<Ii5fI", b)
    c = _Cursor(b)
    c.o = 32
    q = dict(n_inliers=ninl, K4=np.array([fx, fy, cx, cy], np.float32), bf=np.float32(bf), Xw=c.take("<f4", (n, 3)),
             kp=c.take("<f4", (n, 2)), u_right=c.take("<f4", (n,)), inv_sigma2=c.take("<f4", (n,)), valid=c.take("u1", (n,)),
             outlier=c.take("u1", (n,)))
    c.align4()
    q.update(Tcw_in=c.take("<f4", (4, 4)), Tcw_out=c.take("<f4", (4, 4)))
    q["crf_index"] = c.take("<i4", (n,)) if flags & 1 else None
    return q


_DECODERS = {SEC_UNARY: ("unary", _decode_unary), SEC_BFMATCH: ("bfmatch", _decode_bfmatch), SEC_POSE: ("pose", _decode_pose)}


def file_origin(path):
    """ORIGIN_* of a record file (0 for version-1 files)."""
    with open(path, "rb") as fh:
        raw = fh.read(_FILE_HDR.size)
    if len(raw) != _FILE_HDR.size or raw[:8] != MAGIC:
        raise RecordError("not a record file")
    return _FILE_HDR.unpack(raw)[4]


def read_records(path):
    """Yield the frames of `path`.  Raises RecordError on a malformed or truncated file."""
    with open(path, "rb") as fh:
        fsize = os.fstat(fh.fileno()).st_size           # every size field of the file is checked against what is left of it BEFORE
        raw = fh.read(_FILE_HDR.size)                   #   anything is read or allocated (a corrupt or hostile file must not drive memory)
        if len(raw) != _FILE_HDR.size:
            raise RecordError("truncated file header")
        magic, version, hb, fhb = _FILE_HDR.unpack(raw)[:4]
        if magic != MAGIC:
            raise RecordError("bad magic %r" % magic)
        if version not in (1, VERSION):
            raise RecordError("unsupported version %d" % version)
        if hb < _FILE_HDR.size or fhb < _FRAME_HDR.size:
            raise RecordError("header sizes smaller than version 1")
        if hb > fsize or fhb > fsize:
            raise RecordError("header sizes larger than the file")
        fh.seek(hb)
        while True:
            raw = fh.read(fhb)
            if not raw:
                return
            if len(raw) != fhb:
                raise RecordError("truncated frame header")
            v = _FRAME_HDR.unpack(raw[:_FRAME_HDR.size])
            npts, flags, frame_id, n_it = v[:4]
            n_sections = v[17] if (version >= 2 and flags & HAS_SECTIONS) else 0
            fr = dict(frame_id=frame_id, n_iterations=n_it,
                      params={k: np.float32(x) for k, x in zip(PARAM_NAMES, v[4:17])})
            size = fhb

            def take(dtype, shape):
                nonlocal size
                cnt = 1
                for x in shape:
                    cnt *= int(x)
                nbytes = cnt * np.dtype(dtype).itemsize
                if nbytes > fsize - fh.tell():
                    raise RecordError("truncated frame %d" % frame_id)
                b = fh.read(nbytes)
                if len(b) != nbytes:
                    raise RecordError("truncated frame %d" % frame_id)
                size += nbytes
                return np.frombuffer(b, dtype=dtype).reshape(shape).copy()

            fr["vobservs"] = take("<f4", (npts,))
            fr["verrors"] = take("<f4", (npts,))
            fr["vdepths"] = take("<f4", (npts,))
            fr["coord2d"] = take("<f4", (npts, 2))
            fr["init_label"] = take("<i2", (npts,))
            fr["match_prob"] = take("<f8", (npts,)) if flags & HAS_MATCH_PROB else None
            fr["ref_label"] = take("<i2", (npts,)) if flags & HAS_REF_LABEL else None
            fr["ref_prob"] = take("<f4", (npts, 2)) if flags & HAS_REF_PROB else None
            pad = -size % 8
            if pad and len(fh.read(pad)) != pad:
                raise RecordError("truncated padding after frame %d" % frame_id)
            fr["sections"], fr["unknown_sections"] = {}, []
            for _ in range(n_sections):
                raw = fh.read(_SEC_HDR.size)
                if len(raw) != _SEC_HDR.size:
                    raise RecordError("truncated section header in frame %d" % frame_id)
                tag, sflags, nbytes = _SEC_HDR.unpack(raw)
                if nbytes > fsize - fh.tell():
                    raise RecordError("section of frame %d claims %d bytes, %d left in the file" % (frame_id, nbytes, fsize - fh.tell()))
                payload = fh.read(nbytes)
                spad = -nbytes % 8
                if len(payload) != nbytes or len(fh.read(spad)) != spad:
                    raise RecordError("truncated section in frame %d" % frame_id)
                if tag in _DECODERS:
                    name, dec = _DECODERS[tag]
                    fr["sections"][name] = dec(sflags, payload)
                else:                                   # a newer writer's section: carried along, never interpreted
                    fr["unknown_sections"].append((tag, sflags, payload))
            yield fr


def synthetic_frame(wl, n, seed, params=None, frame_id=0):
    """A record-shaped frame from lc-crf-slam_amd.workloads (no reference results attached)."""
    p = dict(wl.TUM3 if params is None else params)
    f = wl.slam_frame(n, seed)
    return dict(frame_id=frame_id, n_iterations=5, params=p, vobservs=f["obs"], verrors=f["err"],
                vdepths=f["depth"], coord2d=f["uv"], init_label=f["init_label"], match_prob=None,
                ref_label=None, ref_prob=None)
