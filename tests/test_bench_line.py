"""The bench line the driver parses (bench.py: compact_line): short enough to survive the driver's 8 KB tail of
stdout, and carrying every key the contract names -- round 4's line was 22 KB and `parsed` came back null.
CPU only: canned measurements (the round-4 detail record under profiles/, and a synthetic worst case)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU = ("value", "unit", "cores", "kind", "sample")


def _canned():
    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as fh:
        return json.load(fh)


def _check(line):
    assert "\n" not in line
    assert len(line) < bench.COMPACT_LIMIT, len(line)
    c = json.loads(line)
    for k in REQUIRED:
        assert k in c, k
    for k in ROOFLINE:
        assert k in c["roofline"], k
    for k in CPU:
        assert k in c["cpu_baseline"], k
    assert c["config"]["workload"] and "model" not in c["config"]
    assert abs(c["roofline"]["frac"] - c["roofline"]["achieved"] / c["roofline"]["peak"]) < 1e-3
    return c


def test_round4_record_fits_and_keeps_the_contract():
    out = _canned()
    assert len(json.dumps(out)) > 16000            # the record that broke the driver's parser
    c = _check(bench.compact_line(out, "bench_detail.json"))
    assert c["detail"] == "bench_detail.json"
    assert set(c["other_configs"]) == {"C2", "C5"}
    for o in c["other_configs"].values():
        assert {"ms_per_step", "frac", "bound", "traffic"} <= set(o)
    assert c["steady_state"]["ms_per_step"] > 0
    assert c["cpu_baseline"]["all_cores"]["cores"] >= 1
    assert "what" not in json.dumps(c)             # no prose in the line


def test_a_bloated_record_still_fits():
    out = _canned()
    out["per_gpu"] = [{"rank": g, "shard": g, "device": g, "channels": 16384, "ms_per_step": 0.5 + g * 1e-7,
                       "pci": "0000:%02x:00.0" % g, "host_thread_pinned_to_cpus": list(range(64))} for g in range(64)]
    out["kernel_ms"] = {"kernel_%d" % i: 0.1 * i for i in range(200)}
    out["config"]["workload"] = "w" * 3000
    out["cpu_baseline"]["sample"] = "s" * 3000
    _check(bench.compact_line(out, "x.json"))


def test_a_node_line_without_pmc_fits():
    out = _canned()
    out["roofline"] = {"bound": "hbm", "kernel": "chain (all devices)", "achieved": 22000.0, "peak": 64000.0,
                       "unit": "GB/s", "frac": 22000.0 / 64000.0, "traffic": None,
                       "algorithmic_bytes_per_launch": 8 * 1.57e9}
    out["n_gpus"] = 8
    out.pop("other_configs", None)
    c = _check(bench.compact_line(out))
    assert c["roofline"]["traffic"] is None and c["n_gpus"] == 8


def test_a_run_that_cannot_start_ends_in_one_line_quickly():
    """bench.py's preflight (device count, free memory per shard): on a host without a HIP device -- and for a device index
    that does not exist -- the run ends within seconds in ONE parseable line whose `error` says why, status 3, no hang."""
    import subprocess
    import time
    import torch
    cmds = [["--gpus", "8", "--steps", "1", "--warmup", "1"]]
    if torch.cuda.is_available():
        cmds = [["--gpus", "2", "--devices", "0,%d" % (torch.cuda.device_count() + 3), "--steps", "1", "--warmup", "1"]]
    for extra in cmds:
        t = time.perf_counter()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, timeout=120)
        took = time.perf_counter() - t
        assert p.returncode == 3, p.stderr.decode()[-500:]
        line = p.stdout.decode().strip().splitlines()[-1]
        c = json.loads(line)
        assert c["value"] is None and c["n_gpus"] == int(extra[1]) and c["error"]
        assert ("visible" in c["error"]) or ("device" in c["error"])
        assert took < 60, took


def test_preflight_names_the_device_and_both_figures(monkeypatch):
    """bench.preflight() over a stand-in for torch.cuda: a missing index, a device without room, a healthy node."""
    import types
    import torch
    fake = types.SimpleNamespace(is_available=lambda: True, device_count=lambda: 4,
                                 mem_get_info=lambda d: ((2 << 30) if d == 2 else (200 << 30), 288 << 30))
    monkeypatch.setattr(torch, "cuda", fake)
    cfg = bench.CONFIGS["C3"]
    assert bench.preflight([0, 1, 3], cfg) is None
    e = bench.preflight(list(range(8)), cfg)
    assert "[4, 5, 6, 7]" in e and "4 device(s) visible" in e
    e = bench.preflight([0, 2], cfg)
    assert e.startswith("device 2:") and "GB free" in e and "16384 channels" in e
    e = bench.preflight([1] * 200, cfg)                 # 200 shards on one device do not fit 200 GB
    assert e.startswith("device 1: 200 shard(s)")


def test_ranks_of_torch_distributed_run_that_cannot_start_say_so_together():
    """Under `python -m torch.distributed.run` (how the driver starts N > 1): every rank checks its own device, the ranks
    exchange what they found through a TCP store BEFORE any process group exists, and rank 0 prints ONE line whose `error`
    names every failing rank -- within seconds, status 3, nobody left in a barrier.  (Here: no rank has a device.)"""
    import socket
    import subprocess
    import time
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the ranks would start")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    t = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "1"], capture_output=True, timeout=180)
    took = time.perf_counter() - t
    assert p.returncode != 0 and took < 90, took
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    c = json.loads(lines[0])
    assert c["value"] is None and c["n_gpus"] == 2
    assert "rank 0 (device 0)" in c["error"] and "rank 1 (device 1)" in c["error"]
    assert [g["rank"] for g in c["per_gpu"]] == [0, 1]
