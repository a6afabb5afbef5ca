"""The line the driver parses: bench.py prints its sub-records first and ONE compact line last.
BENCH_r05.json was unparsed because the single line had grown to 20 KB; these tests pin the size and the keys."""
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (stdlib imports only at module level)

FULL = os.path.join(ROOT, "profiles", "r5_fused_c2", "bench_default_with_extras.json")   # a complete default-run record (20 KB, round 5)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _full():
    with open(FULL) as fh:
        return json.load(fh)


def test_compact_line_is_last_small_and_complete(tmp_path):
    full = _full()
    buf = io.StringIO()
    text = bench.emit(full, str(tmp_path / "full.json"), buf)
    lines = buf.getvalue().splitlines()
    assert lines[-1] == text and buf.getvalue().endswith(text + "\n")
    assert len(text) < 4096
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-5)
    assert line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert line["config"]["workload"].startswith("C2: 2000 keypoints")
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms")) <= set(line["roofline"])
    assert line["roofline"]["frac"] == pytest.approx(line["roofline"]["achieved"] / line["roofline"]["peak"], rel=1e-4)
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert line["label_match_vs_cpu_reference"] == 1.0 and line["tiles_identical"] is True
    for sub in ("c1", "c3", "c4", "n500", "c5", "c5_single_frame", "single_frame_latency_us", "single_frame_latency_us_n500"):
        assert sub in line, sub
    # the complete record is on disk and every sub-record is a line of its own before the compact one
    assert json.load(open(tmp_path / "full.json"))["c4"]["value"] == full["c4"]["value"]
    recs = [json.loads(l).get("record") for l in lines[:-1]]
    assert {"c1", "c3", "c4", "n500", "c5", "cpu_baseline", "end_to_end", "roofline_detail"} <= set(recs)


def test_compact_line_has_no_note_strings():
    line = bench.compact_line(_full())

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(s) for s in strings(line)) <= 80


def test_compact_line_stays_small_with_long_numbers_and_multi_gpu():
    """Worst case: every float at full repr length, 8 ranks, every optional key present."""
    full = _full()

    def inflate(o):
        if isinstance(o, dict):
            return {k: inflate(v) for k, v in o.items()}
        if isinstance(o, list):
            return [inflate(v) for v in o]
        if isinstance(o, float):
            return o * 1.0000001234567 + 1e-9
        return o
    full = inflate(full)
    full["n_gpus"] = 8
    full["multi_gpu"] = {"ms_per_step_by_rank": [1.8234567891234] * 8, "ranks_in_collective": 8, "backend": "nccl",
                         "label_gather": {"serial_ms_per_step": 1.9, "overlapped_ms_per_step": 1.83, "exposed_ms_per_step": 0.07123456789,
                                          "bytes_per_rank": 4194304}}
    full["label_gather_ok"] = True
    full["device"] = "AMD Instinct MI355X pci 0000:f5:00 uuid 0123456789abc"
    full["end_to_end"]["host_to_host"] = dict(full["end_to_end"]["host_to_host"], best_frames_per_s=1.3312345678e6, best_frac_of_link=0.8412345678)
    full["c5"]["single_frame"]["frac_wall"] = 0.40123456789
    text = bench.emit(full, None, io.StringIO())
    assert len(text) < 4096
    assert json.loads(text)["multi_gpu"]["ranks_in_collective"] == 8


def test_oversized_line_is_refused(monkeypatch):
    monkeypatch.setattr(bench, "COMPACT_LIMIT", 512)
    with pytest.raises(SystemExit):
        bench.emit(_full(), None, io.StringIO())
