"""GPU: bench.py's output contract on a small configuration (2 frames of 128x128): one JSON line with the headline
fields, a `roofline` object computed from this run's own event timings, the whole-step fractions and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract(dev):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--img-size", "128"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "step_roofline", "kernel_breakdown"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["unit"] == "frames/s" and d["value"] > 0 and abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s")
    assert 0 < ro["frac"] <= 1.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9
    # recomputable from the line alone: achieved = algorithmic work per launch / average launch duration
    scale = 1e9 if ro["unit"] == "GB/s" else 1e12
    assert abs(ro["algorithmic_per_launch"] / (ro["avg_launch_ms"] * 1e-3) / scale - ro["achieved"]) < 1e-6 * ro["achieved"]
    # the dominant kernel is the one with the largest event-timed total of THIS run (entry points that launch the same
    # kernel — the split-operand Winograd GEMMs — are added up, as rocprofv3's per-kernel statistics do)
    tot = {}
    for k in d["kernel_breakdown"]:
        tot[k["kernel"]] = tot.get(k["kernel"], 0.0) + k["ms_per_step"]
    assert max(tot, key=tot.get) == ro["entry_point"]
    assert "matmul" in d["config"] and d["fp32_mfma_only"]["value"] > 0
    ts = ro["traffic_source"]
    assert ts is None or (ts["stale"] == (ro["traffic"] is None) or ro["traffic"] is None)
    sr = d["step_roofline"]
    # ideal_ms = sum over launches of max(bytes / HBM peak, FLOPs / peak of that launch's matrix instruction): a lower
    # bound of the step, recomputable from the per-entry-point rows
    assert 0 < sr["ideal_ms"] < d["ms_per_step"] and abs(sr["achieved"] - sr["ideal_ms"] / d["ms_per_step"]) < 1e-9
    assert sr["ideal_ms"] >= sum(k["ideal_ms_per_step"] for k in d["kernel_breakdown"]) * 0.999
    assert 0 < sr["hbm_fraction"] < 1 and "compute_path" in d
    assert "hbm_bytes_source" not in sr or (sr["hbm_bytes_source"]["stale"] == (sr["hbm_bytes_per_step_measured"] is None)
                                             or sr["hbm_bytes_per_step_measured"] is None)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "warm" in cb["sample"]


def test_bench_two_ranks_one_card(dev):
    """`python bench.py --gpus 2` exactly as the driver calls it — no torch.distributed.run, WORLD_SIZE unset: the parent
    starts the two ranks itself (before touching the GPU), both run the full step and exchange gradients, rank 0 prints ONE
    JSON line with the whole-job rate.  Two ranks share this box's single card, so the exchange runs over gloo
    (WFAE_DIST_BACKEND; RCCL needs a device per rank) — the launch path, the rank bookkeeping and the timing protocol are
    the ones an 8-GPU node runs."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WFAE_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "2", "--img-size", "128", "--no-fp32-leg"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 4
    assert d["dp"]["ranks_seen"] == 2 and d["dp"]["backend"] == "gloo" and d["dp"]["grad_exchange_ms_per_step"] > 0
    assert abs(d["value"] - 2 * 2 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only
