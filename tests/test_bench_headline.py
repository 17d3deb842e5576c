"""bench.py's final stdout line: the driver parses it, so it must stay small and well formed whatever the legs hold
(round 5's line carried every leg, grew past the driver's tail buffer and was recorded as `parsed: null`)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from benchlib import headline as H


def canned():
    """the shape of a full bench.py result: round 5's own line (profiles/r05/bench_n1.json.log), a 20.9 KB object"""
    with open(os.path.join(ROOT, "profiles", "r05", "bench_n1.json.log")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_round5_line_was_too_long_and_the_headline_is_not():
    full = canned()
    assert len(json.dumps(full)) > 20000
    line = H.headline_line(full, "gpurun_out/bench_legs.json")
    assert len(line.encode()) < H.MAX_BYTES and "\n" not in line
    h = json.loads(line)
    for k in H.REQUIRED:
        assert k in h, k
    assert h["value"] == float("%.9g" % full["value"]) and h["n_gpus"] == 1 and h["steps"] == full["steps"] and h["warmup"] == full["warmup"]
    assert h["metric"] == "aligned reads/s (100 bp, band=15)" and h["unit"] == "reads/s" and h["higher_is_better"] is True and h["vs_baseline"] is None
    assert "workload" in h["config"] and "model" not in h["config"]
    # the dtype names the result type first and says where 16-bit lanes are used
    assert h["dtype"].startswith("int32") and "int16" in h["dtype"]
    r = h["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "hbm_GBs", "hbm_frac", "counters", "a32"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["a32"]["identical_results"] is True
    rr = h["rank_roofline"]
    assert rr["target"] == 0.60 and rr["met"] is False and abs(rr["frac"] - 0.255) < 0.01 and rr["traffic"] > 0
    c = h["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert h["parity"] == {"checked": 100000, "bit_exact": True}
    assert h["legs_file"] == "gpurun_out/bench_legs.json"


def test_headline_survives_missing_and_oversized_legs():
    full = canned()
    # a leg-less N > 1 line (no rank / cpu legs on the multi-rank path)
    slim = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "data", "config", "roofline", "parity")}
    slim["n_gpus"] = 8
    slim["config"] = dict(slim["config"], gather_path="cxx_rccl", rccl_ranks_seen=8, per_rank_ms_per_step=[2.8] * 8)
    h = json.loads(H.headline_line(slim))
    assert h["n_gpus"] == 8 and h["cpu_baseline"] is None and "rank_roofline" not in h
    assert h["config"]["rccl_ranks_seen"] == 8 and len(h["config"]["per_rank_ms_per_step"]) == 8
    # absurdly long strings in the legs do not reach the line
    full["cpu_baseline"]["sample"] = "x" * 50000
    full["config"]["workload"] = "w" * 50000
    full["e2e_leg"]["config4_full_size"]["note"] = "n" * 50000
    line = H.headline_line(full)
    assert len(line.encode()) < H.MAX_BYTES
    json.loads(line)


def test_leg_lines_are_json_and_carry_every_leg():
    full = canned()
    lines = H.leg_lines(full)
    names = [json.loads(l)["leg"] for l in lines]
    for k in ("roofline", "rank_roofline", "seed_leg", "e2e_leg", "full_dp_leg", "compat_stream_leg", "cpu_baseline"):
        assert k in names, k


def test_bench_emit_prints_the_headline_last(tmp_path, capsys):
    import importlib
    try:
        bench = importlib.import_module("bench")
    except Exception as e:      # noqa: BLE001  (bench.py imports the library; on a box without the built .so this test has nothing to check)
        import pytest
        pytest.skip("bench.py not importable here: %r" % (e,))
    legs = tmp_path / "legs.json"
    bench.emit(canned(), str(legs))
    out = capsys.readouterr().out.strip().splitlines()
    last = json.loads(out[-1])
    assert len(out[-1].encode()) < H.MAX_BYTES and last["metric"].startswith("aligned reads/s") and "roofline" in last and "cpu_baseline" in last
    assert len(out) > 3 and all(json.loads(l).get("leg") for l in out[:-1])
    assert json.load(open(legs))["e2e_leg"]["reads"] == 10_000_000
