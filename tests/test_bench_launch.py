"""bench.py --gpus N starts its own ranks (round-3 review, missing 1): without a launcher around it the script must not
read WORLD_SIZE = 1 and measure one GPU under an N-GPU label.  CPU only: `--launch-check` brings the ranks up over gloo
and reports the world size the process group holds, with no GPU call anywhere."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")
    return {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_")}


def _result_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_n_without_a_launcher_starts_n_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", "ok"], capture_output=True, text=True,
                       env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _result_line(r.stdout)
    assert d["n_gpus"] == 2 and d["launched_by"] == "self"
    assert "without a launcher" in r.stderr


def test_self_launch_relays_the_exit_code_of_a_failing_rank():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", "fail"], capture_output=True, text=True,
                       env=_env(), timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]   # no result line from a failed job


def test_under_an_external_launcher_no_second_launch():
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--launch-check", "ok"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _result_line(r.stdout)
    assert d["n_gpus"] == 2 and d["launched_by"] == "external launcher"
    assert "without a launcher" not in r.stderr


def test_a_label_that_is_not_the_world_size_is_refused():
    """--gpus 2 inside a one-rank environment (WORLD_SIZE=1 set by some wrapper) must fail, not print n_gpus 2 or 1."""
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-sample", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "process group has 1 rank" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_watchdog_names_the_phase_and_ends_the_process():
    code = ("import sys, time; sys.path.insert(0, %r); import bench; w = bench.Watchdog(5); w.phase('first', 60); "
            "w.phase('stuck exchange', 1); time.sleep(30)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3
    assert "rank 5: phase 'stuck exchange' exceeded its limit" in r.stderr and "first" in r.stderr


def test_volume_presets_and_the_one_gpu_refusal():
    """--config c4 / c5 are BASELINE's multi-GPU configurations (their own shape and descriptor); --dims and --size name other
    volumes; a preset on one GPU is refused before anything touches the device."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    ns = lambda **kw: argparse.Namespace(**dict(dict(config=None, dims=None, size=512, desc=None), **kw))
    assert bench.resolve_volume(ns()) == (512, 512, 512, 0, "512^3")
    assert bench.resolve_volume(ns(config="c4")) == (1024, 1024, 512, 0, "1024 x 1024 x 512")
    assert bench.resolve_volume(ns(config="c5")) == (2048, 2048, 1024, 3, "2048 x 2048 x 1024")
    assert bench.resolve_volume(ns(config="c5", desc=0))[3] == 0          # an explicit --desc wins
    assert bench.resolve_volume(ns(dims="96,80,160", desc=2)) == (96, 80, 160, 2, "96 x 80 x 160")
    assert bench.resolve_volume(ns(size=256)) == (256, 256, 256, 0, "256^3")
    r = subprocess.run([sys.executable, BENCH, "--config", "c4"], capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode != 0 and "multi-GPU configuration" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
