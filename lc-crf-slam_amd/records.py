"""Reader / writer of capture-replay records (include/lccrf_record.h).

Pure numpy.  A frame is a dict with the arrays of the call site at
/root/reference/src/Tracking.cc:1871-1930: vobservs, verrors, vdepths (float32 [n]), coord2d
(float32 [n,2]), init_label (int16 [n]), params (dict of the 13 CRF settings), frame_id,
n_iterations, and optionally match_prob (float64 [n]), ref_label (int16 [n]), ref_prob (float32 [n,2]).
"""
import struct

import numpy as np

MAGIC = b"LCCRFREC"
VERSION = 1
HAS_MATCH_PROB, HAS_REF_LABEL, HAS_REF_PROB = 1, 2, 4
PARAM_NAMES = ("w1", "w2", "u_alpha", "stdev_alpha", "u_beta", "stdev_beta", "u_gamma", "stdev_gamma",
               "point3d_stdev", "point2d_stdev", "u_depth", "pth", "confidence")
_FILE_HDR = struct.Struct("<8sIII3I")           # 32 bytes
_FRAME_HDR = struct.Struct("<4I13f3f")          # 80 bytes


class RecordError(ValueError):
    pass


def write_records(path, frames):
    """Write `frames` (an iterable of frame dicts) to `path`; returns the number written."""
    n = 0
    with open(path, "wb") as fh:
        fh.write(_FILE_HDR.pack(MAGIC, VERSION, _FILE_HDR.size, _FRAME_HDR.size, 0, 0, 0))
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
    p = fr["params"]
    hdr = _FRAME_HDR.pack(npts, flags, int(fr.get("frame_id", 0)), int(fr.get("n_iterations", 5)),
                          *[np.float32(p[k]) for k in PARAM_NAMES], 0.0, 0.0, 0.0)
    body = b"".join(parts)
    return hdr + body + b"\0" * (-(len(hdr) + len(body)) % 8)


def read_records(path):
    """Yield the frames of `path`.  Raises RecordError on a malformed or truncated file."""
    with open(path, "rb") as fh:
        raw = fh.read(_FILE_HDR.size)
        if len(raw) != _FILE_HDR.size:
            raise RecordError("truncated file header")
        magic, version, hb, fhb = _FILE_HDR.unpack(raw)[:4]
        if magic != MAGIC:
            raise RecordError("bad magic %r" % magic)
        if version != VERSION:
            raise RecordError("unsupported version %d" % version)
        if hb < _FILE_HDR.size or fhb < _FRAME_HDR.size:
            raise RecordError("header sizes smaller than version 1")
        fh.seek(hb)
        while True:
            raw = fh.read(fhb)
            if not raw:
                return
            if len(raw) != fhb:
                raise RecordError("truncated frame header")
            v = _FRAME_HDR.unpack(raw[:_FRAME_HDR.size])
            npts, flags, frame_id, n_it = v[:4]
            fr = dict(frame_id=frame_id, n_iterations=n_it,
                      params={k: np.float32(x) for k, x in zip(PARAM_NAMES, v[4:17])})
            size = fhb

            def take(dtype, shape):
                nonlocal size
                cnt = int(np.prod(shape))
                nbytes = cnt * np.dtype(dtype).itemsize
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
            yield fr


def synthetic_frame(wl, n, seed, params=None, frame_id=0):
    """A record-shaped frame from lc-crf-slam_amd.workloads (no reference results attached)."""
    p = dict(wl.TUM3 if params is None else params)
    f = wl.slam_frame(n, seed)
    return dict(frame_id=frame_id, n_iterations=5, params=p, vobservs=f["obs"], verrors=f["err"],
                vdepths=f["depth"], coord2d=f["uv"], init_label=f["init_label"], match_prob=None,
                ref_label=None, ref_prob=None)
