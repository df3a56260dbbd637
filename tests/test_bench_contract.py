"""CPU suite: the committed bench line (profiles/) carries what the driver's contract asks for, and its numbers are
consistent with each other.  (bench.py itself needs the GPU; this guards the shape of what it prints.)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    return json.load(open(os.path.join(ROOT, "profiles", "r02b_bench.json")))


def test_bench_line_has_the_contract_fields():
    b = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["higher_is_better"] is True and b["vs_baseline"] is None
    assert b["dtype"] == "f32" and b["data"] == "synthetic" and "workload" in b["config"] and "model" not in b["config"]
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    c = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] == 1


def test_bench_line_is_self_consistent():
    b = _line()
    r = b["roofline"]
    # value = records per second of the whole job
    assert abs(b["value"] - b["config"]["records_per_volume"] / (b["ms_per_step"] * 1e-3)) <= 1e-3 * b["value"]
    # achieved = algorithmic bytes per launch / average launch time; frac = achieved / peak
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # compulsory bytes: 52 B/voxel over the five launches of a 512^3 volume (initial blur 8, L1 8, L2..L4 with their DoG 12 each)
    assert abs(r["alg_bytes_per_launch"] - 52.0 / 5.0 * 512 ** 3) < 1.0
    # PMC traffic (per launch) is above the algorithmic bytes and within 2x of them
    assert r["alg_bytes_per_launch"] < r["traffic"] < 2.0 * r["alg_bytes_per_launch"]
    # the per-instantiation launches add up to the dominant kernel's launches
    assert sum(p["launches"] for p in r["per_instantiation"]) == r["launches"] == 5 * b["steps"]
