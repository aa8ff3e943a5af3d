"""bench.py's multi-rank branch, driven end to end on CPU (gloo, world_size 2) with the GPU classes mocked
(tests/helpers/bench_mock_driver.py): the 8-GPU node is not ours to launch, so the plumbing is proven here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(world, port, extra_env=None):
    env = dict(os.environ, **(extra_env or {}))
    env.pop("CLEARCAM_BENCH_FORCE_SHARDED", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "helpers", "bench_mock_driver.py"),
           "--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "4", "--res", "64"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]                      # ONE JSON line, from rank 0 only
    # the driver parses the LAST line of an 8 KB stdout tail (BENCH_r05.json.parsed was null: the r05 line was 25 KB)
    assert r.stdout.rstrip().splitlines()[-1] == lines[0]
    assert len(lines[0]) < 4096, len(lines[0])
    line = json.loads(lines[0])
    detail = json.load(open(os.path.join(ROOT, "bench_detail.json")))       # the full record lives beside bench.py
    assert detail["value"] == line["value"] and "kernel_timing_note" in detail["roofline"]
    return line


def test_two_ranks_produce_one_weak_scaling_line():
    line = run_bench(2, 29641)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["unit"] == "frames/s" and line["higher_is_better"] is True and line["vs_baseline"] is None
    # value = all ranks' frames / max-over-ranks time: rank 1 sleeps 2 ms per step, so the job cannot look faster than that
    assert 0 < line["value"] <= 2 * 4 / 0.002 * 1.01
    assert abs(line["value"] - 2 * 4 * 3 / (line["ms_per_step"] * 3e-3)) / line["value"] < 1e-3
    sm = line["streams_multi_gpu"]
    assert sm["cams8_per_gpu"] == {"cameras_total": 16, "ranks_ok": 2, "frames_per_sec_total": 1600.0, "min_fps_per_camera": 100.0, "h2d_GBps_total": 2.0}
    assert sm["cams64_per_gpu"]["cameras_total"] == 128 and sm["cams64_per_gpu"]["ranks_ok"] == 2
    sh = line["search_sharded"]
    assert sh["rows_total"] == 4000 and sh["rows_per_gpu"] == 2000 and sh["ranks_ok"] == 2 and sh["result_sorted_and_in_range"] is True
    assert "roofline" in line and "cpu_baseline" not in line     # the CPU leg runs at N=1 only
    # what the collective backend saw: an all-reduce of ones over the ranks, and every rank's own step time (rank 1 is the slow one)
    assert line["ranks_seen"] == 2
    pr = line["ms_per_step_by_rank"]
    assert len(pr) == 2 and all(v > 0 for v in pr) and pr[1] >= 2.0 and pr[1] <= line["ms_per_step"] * 1.5 + 1.0


def test_single_rank_line_is_short_and_carries_roofline_and_cpu_baseline():
    line = run_bench(1, 29645, {"CLEARCAM_CPU_SWEEP": "8"})
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["value"] > 0
    for key in ("metric", "unit", "steps", "warmup", "ms_per_step", "dtype", "storage_mode", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"]) and line["cpu_baseline"]["kind"] == "port"
    assert "workload" in line["config"] and "model" not in line["config"]


def test_a_failing_side_metric_on_one_rank_does_not_hang_or_kill_the_line():
    line = run_bench(2, 29643, {"MOCK_FAIL_STREAMS_ON_RANK": "1"})
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["streams_multi_gpu"]["cams8_per_gpu"]["ranks_ok"] == 1    # the other rank still reported; nobody waited forever
