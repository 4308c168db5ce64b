"""The driver parses the LAST stdout line of bench.py out of an ~8 KB tail.  Round 4's line had grown to 25 KB and the
record came back `parsed: null`.  These tests keep the last line small and complete, on the real 25 KB record."""
import importlib.util
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _full_record():
    return json.load(open(os.path.join(ROOT, "profiles", "r04_final_bench_lines.json")))["default_command_python_bench_py"]


def test_last_line_is_small_and_carries_the_contract():
    b = _bench()
    full = _full_record()
    assert len(json.dumps(full)) > 20000          # the input really is the oversized record
    out = b.compact_line(full, "gpurun_out/bench_secondary.json")
    text = json.dumps(out)
    assert len(text) < b.LAST_LINE_BUDGET < 6000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["value"] == full["value"] and out["ms_per_step"] == full["ms_per_step"]
    assert out["config"]["workload"].startswith("blob_to_kzg_commitment batch of 1024 blobs")
    r = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert r["traffic"] == full["roofline"]["pmc_cross_check"]["traffic_bytes_per_launch"]   # counters, not the model
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert "predicted_scaling" not in out and "secondary" not in out      # the model and the sweeps stay out
    # both forms of the headline travel in the one line: resident (the contract's `value`) and PCIe-inclusive
    assert "value_host_pointer" in out and "resident" in out["value_semantics"] and "H2D+D2H" in out["value_semantics"]
    assert out["baseline_configs"]["configs[2]"]["wide_tables_ms_per_call"] == full["baseline_configs"]["configs[2]"]["wide_tables_ms_per_call"]


def test_emit_prints_the_compact_line_last(tmp_path, monkeypatch):
    b = _bench()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    buf, err = io.StringIO(), io.StringIO()
    monkeypatch.setattr(sys, "stdout", buf)
    monkeypatch.setattr(sys, "stderr", err)
    b.emit(_full_record())
    monkeypatch.undo()
    lines = buf.getvalue().strip().split("\n")
    assert len(lines) == 1                                   # ONE JSON line on stdout
    assert "bench_full_record" in json.loads(err.getvalue())  # the full record: stderr + side file
    last = json.loads(lines[-1])
    assert len(lines[-1]) < b.LAST_LINE_BUDGET and last["metric"] and last["secondary_file"] == "gpurun_out/bench_secondary.json"
    side = json.load(open(tmp_path / "gpurun_out" / "bench_secondary.json"))
    assert "secondary" in side and "predicted_scaling" in side


def test_oversized_extras_are_dropped_not_the_head():
    b = _bench()
    full = _full_record()
    full["per_rank"] = [{"rank": r, "pad": "x" * 100} for r in range(8)]
    full["predicted"] = {"value": 1, "pad": "y" * 1100}
    out = b.compact_line(full, None)
    assert len(json.dumps(out)) <= b.LAST_LINE_BUDGET
    assert out["roofline"]["frac"] and out["cpu_baseline"]["value"]


def test_round5_record_also_fits():
    """the full record of a round-5 run (one-rank shares, percentiles in every caller row: 29 KB)"""
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_last_bench_full_record.json")))
    out = b.compact_line(full, "gpurun_out/bench_secondary.json")
    assert len(json.dumps(out)) < b.LAST_LINE_BUDGET
    assert out["roofline"]["traffic"] and out["roofline"]["traffic_source"].startswith("profiles/r05_final_pmc")
    assert out["cpu_baseline"]["all_cores"]["driver"].startswith("pthreads")
    assert out["baseline_configs"]["configs[4]_recover_256_ms"] == full["baseline_configs"]["configs[4]"]
