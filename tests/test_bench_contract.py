"""CPU suite: the NEWEST committed bench line (profiles/rNN_bench.json) carries what the driver's contract asks for, and its
numbers are consistent with each other.  (bench.py itself needs the GPU; this guards the shape of what it prints.)  Round 4's
review: the guard must follow the newest line and derive the byte sum from `per_instantiation`, not from a constant."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest():
    best = None
    for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json")):
        n = int(re.match(r"r(\d\d)_bench\.json", os.path.basename(p)).group(1))
        if best is None or n > best[0]:
            best = (n, p)
    assert best is not None and best[0] >= 4, "no committed bench line of round 4 or later under profiles/"
    return best


def _line():
    return json.load(open(_newest()[1]))


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
    nvox = 512 ** 3
    # value = records per second of the whole job
    assert abs(b["value"] - b["config"]["records_per_volume"] / (b["ms_per_step"] * 1e-3)) <= 1e-3 * b["value"]
    # achieved = algorithmic bytes per launch / average launch time; frac = achieved / peak
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # the per-instantiation launches add up to the dominant kernel's launches, and their compulsory bytes to the line's
    # bytes per launch: every fused launch is credited what it reads once and what it keeps -- 8 B/voxel (level only),
    # 12 (level + DoG), 12.5 (level + DoG + the next octave's level 0) -- and nothing else
    per = r["per_instantiation"]
    assert sum(p["launches"] for p in per) == r["launches"] and r["launches"] % b["steps"] == 0
    for p in per:
        assert p["alg_bytes_per_voxel"] in (8.0, 12.0, 12.5), p
    total = sum(p["launches"] * p["alg_bytes_per_voxel"] * nvox for p in per)
    assert abs(total / r["launches"] - r["alg_bytes_per_launch"]) <= 1e-6 * r["alg_bytes_per_launch"]
    assert abs(r["alg_bytes_per_voxel"] - r["alg_bytes_per_launch"] / nvox) < 0.01
    # and the average launch time is the launch-weighted mean of the instantiations'
    mean_ms = sum(p["launches"] * p["avg_launch_ms"] for p in per) / r["launches"]
    assert abs(mean_ms - r["avg_launch_ms"]) <= 0.01 * r["avg_launch_ms"]
    # PMC traffic (per launch) is above the algorithmic bytes and within 2x of them
    if r["traffic"] is not None:
        assert r["alg_bytes_per_launch"] < r["traffic"] < 2.0 * r["alg_bytes_per_launch"]
    # nothing in the line may read as "the 0.70 target is met" while frac is below it
    s8 = b["pyramid"].get("survey_8d_accounting", {})
    if _newest()[0] >= 5:
        assert "met" not in s8
        c = r["ceiling"]
        assert c and "frac" in c, c
        assert abs(r["ceiling_frac"] - c["frac"]) < 1e-9 and abs(r["frac_of_ceiling"] - r["frac"] / c["frac"]) < 2e-3
        assert 0.5 < c["frac"] < 0.9 and r["frac"] <= c["frac"] * 1.02   # a kernel does not beat the march of its own tiles by more than noise
        assert "within_three_pass_time_budget" in s8
