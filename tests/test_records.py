"""Capture/replay records (include/lccrf_record.h): the format, the committed
sample file (reference results from oracle/_ref) against the oracle, and the replay through the HIP path."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import crf_cases as cc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rec = importlib.import_module("lc-crf-slam_amd.records")
SAMPLE = os.path.join(ROOT, "tests", "golden", "sample_frames.lccrfrec")


def test_c_header_matches_the_python_layout(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lccrf_record.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %u\\n", sizeof(lccrf_rec_file_header), '
                   'sizeof(lccrf_rec_frame_header), offsetof(lccrf_rec_frame_header, w1), '
                   'offsetof(lccrf_rec_frame_header, confidence), LCCRF_REC_VERSION);return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).split()
    assert [int(x) for x in out] == [rec._FILE_HDR.size, rec._FRAME_HDR.size, 16, 16 + 12 * 4, rec.VERSION]
    assert rec._FILE_HDR.size == 32 and rec._FRAME_HDR.size == 80


def test_round_trip_is_bit_exact(tmp_path, wl):
    frames = [rec.synthetic_frame(wl, n, 7 + n, frame_id=n) for n in (0, 1, 3, 250)]
    frames[1]["match_prob"] = np.array([0.25])
    frames[2]["ref_label"] = np.array([1, 0, 1], np.int16)
    frames[3]["ref_prob"] = np.random.default_rng(0).random((250, 2)).astype(np.float32)
    frames[3]["params"]["w1"] = 11.5
    path = tmp_path / "f.lccrfrec"
    assert rec.write_records(path, frames) == 4
    assert os.path.getsize(path) % 8 == 0
    back = list(rec.read_records(path))
    assert len(back) == 4
    for a, b in zip(frames, back):
        assert a["frame_id"] == b["frame_id"] and a["n_iterations"] == b["n_iterations"]
        for k in rec.PARAM_NAMES:
            assert np.float32(a["params"][k]) == b["params"][k]
        for k in ("vobservs", "verrors", "vdepths", "coord2d", "init_label", "match_prob", "ref_label", "ref_prob"):
            if a[k] is None:
                assert b[k] is None
            else:
                assert np.asarray(a[k]).tobytes() == b[k].tobytes(), k


def test_malformed_files_are_rejected(tmp_path, wl):
    path = tmp_path / "f.lccrfrec"
    rec.write_records(path, [rec.synthetic_frame(wl, 40, 1)])
    blob = path.read_bytes()
    (tmp_path / "trunc").write_bytes(blob[:-9])
    with pytest.raises(rec.RecordError):
        list(rec.read_records(tmp_path / "trunc"))
    (tmp_path / "magic").write_bytes(b"X" + blob[1:])
    with pytest.raises(rec.RecordError):
        list(rec.read_records(tmp_path / "magic"))
    (tmp_path / "ver").write_bytes(blob[:8] + (99).to_bytes(4, "little") + blob[12:])
    with pytest.raises(rec.RecordError):
        list(rec.read_records(tmp_path / "ver"))
    with pytest.raises(rec.RecordError):
        rec.encode_frame(dict(rec.synthetic_frame(wl, 5, 1), vobservs=np.zeros(4, np.float32)))


def test_sample_records_agree_with_the_oracle(po):
    """The committed file's reference results (made by oracle/_ref) against the C restatement."""
    frames = list(rec.read_records(SAMPLE))
    assert len(frames) == 8 and sum(f["match_prob"] is not None for f in frames) == 4
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import replay
    for fr in frames:
        n, p = len(fr["init_label"]), fr["params"]
        app, smooth = replay.features(fr)
        o = po.OracleCRF(n, 2)
        o.set_unary_from_label(fr["init_label"], p["confidence"])
        o.add_pairwise(app, p["w1"])
        o.add_pairwise(smooth, p["w2"])
        o.inference_native(fr["n_iterations"], True)
        assert np.array_equal(o.map(), fr["ref_label"]), fr["frame_id"]
        assert cc.same_bits(o.probability(), fr["ref_prob"]), fr["frame_id"]
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("batch,engine", [(256, 0), (3, 0), (1, 1)])
def test_replay_reproduces_the_reference_results(batch, engine):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import replay
    out = replay.replay(SAMPLE, batch=batch, engine=engine)
    assert out["frames"] == 8 and out["checked_frames"] == 8 and out["points"] == 10735
    assert out["label_mismatches"] == 0 and out["prob_mismatches"] == 0 and out["max_abs_dQ"] == 0.0
