"""-m gpu: bench.py honours its contract -- one JSON line with the required keys at N = 1, and
the N > 1 launch path (torch.distributed.run, barrier, max over ranks, whole-job aggregate)
works; the latter with the gloo backend because the test box has a single GPU (on a real
node the driver uses the default nccl = RCCL backend)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def _line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_single_gpu_line():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "2",
                                   "--cpu-blocks", "1"], cwd=ROOT)
    j = _line(out)
    for k in REQUIRED + ["cpu_baseline"]:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 40 and j["warmup"] == 2 and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert abs(j["value"] - 4.0e6 * 40 / (j["ms_per_step"] * 40 / 1e3) / 1e6) / j["value"] < 1e-3
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    B = j["config"]["blocks_per_launch"]
    assert B == 4 and r["frames_per_launch"] == 4.0e6 * B and 4 <= r["launches_timed"] <= 40 // B
    # the dominant kernel IS the step (the previous block's post stage rides in the same launch); its
    # event-timed mean covers groups of 8 launches with their gaps
    assert r["kernel_ms"] <= j["ms_per_step"] * B * 1.05
    one = j["secondary"]["c2_one_block_per_launch"]
    assert one["blocks_per_launch"] == 1 and one["value"] > 0 and one["kernel_ms"] <= one["ms_per_step"] * 1.05
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == min(256, len(os.sched_getaffinity(0))) and c["value"] > 0
    assert c["one_core"]["cores"] == 1 and c["one_core"]["value"] > 0
    assert c["value"] >= c["one_core"]["value"]                    # more threads do not lose
    c3 = j["secondary"]["c3"]                                        # BASELINE config 3 beside the headline
    assert c3["value"] > 0 and c3["roofline"]["algorithmic_bytes_per_frame"] == 12 * 65536
    assert abs(c3["roofline"]["frac"] - c3["roofline"]["achieved"] / 8000.0) < 1e-4
    c1 = j["secondary"]["c1"]
    assert c1["value"] > 2.048 and abs(c1["times_real_time"] - c1["value"] / 2.048) < 0.1


def test_two_rank_launch_path():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                   "--master-addr", "127.0.0.1", "--master-port", "29711", os.path.join(ROOT, "bench.py"),
                                   "--gpus", "2", "--steps", "4", "--warmup", "1", "--backend", "gloo"], cwd=ROOT, env=env)
    j = _line(out)
    assert j["n_gpus"] == 2 and "cpu_baseline" not in j
    # whole-job aggregate: two tuners' samples over the slowest rank's time
    assert abs(j["value"] - 2 * 4.0e6 * 4 / (j["ms_per_step"] * 4 / 1e3) / 1e6) / j["value"] < 1e-3


def test_c5_workload_line_and_two_rank_ring():
    """bench.py --workload c5 (BASELINE config 5): the JSON line at N = 1, and the two-rank path --
    chunks dealt round-robin, the halo through torch.distributed send/recv -- over gloo with both
    ranks on the one GPU of the test box (the driver's node uses nccl = RCCL)."""
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--steps", "3",
                                   "--warmup", "1"], cwd=ROOT)
    j = _line(out)
    for k in REQUIRED:
        assert k in j, k
    assert j["config"]["halo_frames"] == 260_000 and j["config"]["chunk_frames"] % 20_000 == 0
    assert abs(j["value"] - j["config"]["chunk_frames"] * 3 / (j["ms_per_step"] * 3 / 1e3) / 1e6) / j["value"] < 1e-3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                   "--master-addr", "127.0.0.1", "--master-port", "29713", os.path.join(ROOT, "bench.py"),
                                   "--workload", "c5", "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo"],
                                  cwd=ROOT, env=env)
    j = _line(out)
    assert j["n_gpus"] == 2
    assert abs(j["value"] - 2 * j["config"]["chunk_frames"] * 3 / (j["ms_per_step"] * 3 / 1e3) / 1e6) / j["value"] < 1e-3
