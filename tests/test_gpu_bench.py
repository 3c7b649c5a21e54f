"""-m gpu: bench.py honours its contract -- one JSON line with the required keys at N = 1, and
the N > 1 launch path (torch.distributed.run, barrier, max over ranks, whole-job aggregate)
works; the latter with the gloo backend because the test box has a single GPU (on a real
node the driver uses the default nccl = RCCL backend)."""
import json
import os
import sys

import pytest

import _proc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def _line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_single_gpu_line():
    out = _proc.output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "2",
                                   "--cpu-blocks", "1"], cwd=ROOT, timeout=420)
    j = _line(out)
    for k in REQUIRED + ["cpu_baseline"]:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 40 and j["warmup"] == 2 and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert abs(j["value"] - 4.0e6 * 40 / (j["ms_per_step"] * 40 / 1e3) / 1e6) / j["value"] < 1e-3
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # r05: the headline is the STREAMING launch -- the 40 timed steps are 40 rings of the doorbell of one persistent launch,
    # opened by the first and closed by the flush inside the timed region: a block's audio is complete without another
    # block behind it (dspblock.cxx:169-212), nothing is held back
    assert j["config"]["streaming"] is True and j["config"]["blocks_per_launch"] == 1
    assert r["launches_timed"] == 1 and r["frames_per_launch"] == 4.0e6 * 40
    assert r["kernel_ms"] <= j["ms_per_step"] * 40 * 1.05           # the one launch IS the timed region's GPU work
    if r.get("power"):                                               # r06: what the package draws under the headline's kernel (rocm-smi)
        assert 300.0 < r["power"]["package_w"] <= r["power"]["package_w_max"] < 1600.0 and r["power"]["samples"] >= 3
    one = j["secondary"]["c2_one_block_per_launch"]                  # a kernel launch per block: r01-r04's like-for-like figure
    assert one["streaming"] is False and one["blocks_per_launch"] == 1 and one["value"] > 0
    assert one["kernel_ms"] <= one["ms_per_step"] * 1.05 and j["value_one_block_per_launch"] == one["value"]
    four = j["secondary"]["c2_four_blocks_per_launch"]               # r02-r04's headline, demoted: up to 120 ms of added latency
    assert four["blocks_per_launch"] == 4 and four["value"] > 0 and four["launches_timed"] >= 4
    c = j["cpu_baseline"]
    # the reference's own classes where oracle/_ref/libwr_ref_chain.so is there (it is wherever build() ran with
    # /root/reference present), the oracle's port beside it -- the two agree within the noise of a shared memory system
    have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libwr_ref_chain.so"))
    assert c["kind"] == ("reference" if have_ref else "port"), c
    assert c["cores"] == min(256, len(os.sched_getaffinity(0))) and c["value"] > 0
    if have_ref:
        assert c["port"]["kind"] == "port" and 0.3 < c["port"]["value"] / c["value"] < 3.0, c
        assert c["audio_frames"] > 0 and c["audio_abs_sum"] > 0
    assert c["one_core"]["cores"] == 1 and c["one_core"]["value"] > 0
    assert c["value"] >= c["one_core"]["value"]                    # more threads do not lose
    c3 = j["secondary"]["c3"]                                        # BASELINE config 3 beside the headline
    assert c3["value"] > 0 and c3["roofline"]["algorithmic_bytes_per_frame"] == 12 * 65536
    assert abs(c3["roofline"]["frac"] - c3["roofline"]["achieved"] / 8000.0) < 1e-4
    c1 = j["secondary"]["c1"]
    hf = j["secondary"]["host_fed"]                                  # the drop-in path, block in host memory (PCIe inside)
    assert len(hf["runs"]) == 10 and all("error" not in r and r["msps_tuner_input"] > 1000 for r in hf["runs"])
    assert max(r["msps_tuner_input"] for r in hf["runs"]) < j["value"]          # never the headline
    # r06: a source that produces its blocks in GPU memory (DeviceBlock) is streamed by the host classes themselves -- the rows
    # say so with the library's own count (wr_tuner_stream_info), the same source with WEBRADIO_STREAM=0 beside them
    dev = [r for r in hf["runs"] if r["source"].startswith("device")]
    assert len(dev) == 4
    streamed = [r for r in dev if "WEBRADIO_STREAM=0" not in r["staging"]]
    assert len(streamed) == 2 and all(r["stream_info"]["blocks"] >= r["blocks"] for r in streamed)
    assert all(r["stream_info"]["blocks"] == 0 for r in hf["runs"] if r not in streamed)
    assert c1["value"] > 2.048 and abs(c1["times_real_time"] - c1["value"] / 2.048) < 0.1
    assert c1["streaming"]["blocks"] == c1["steps"] and c1["streaming"]["launches"] >= 1 and c1["streaming"]["value"] > 0
    fe = j["secondary"]["frontend"]                                  # r06: C2 and C3 on the same blocks of one tuner
    for mode in ("per_block_launches", "per_block_launches_newest_frame", "streaming_closed_per_block", "streaming_newest_frame"):
        assert fe[mode]["us_per_block"] > 0 and 0 < fe[mode]["c2_frac_of_hbm_roof"] < 1
    assert fe["per_block_launches"]["streaming_launches"] == 0
    assert fe["streaming_closed_per_block"]["streaming_launches"] == fe["blocks"]       # every waterfall batch closes the launch
    assert fe["streaming_newest_frame"]["streaming_launches"] <= fe["blocks"] // 5 + 1  # a poll does, a block does not
    assert fe["lazy"]["pushes_kept"] > 0 and fe["lazy"]["transformed_on_demand"] > 0


def test_two_rank_launch_path():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = _proc.output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                   "--master-addr", "127.0.0.1", "--master-port", "29711", os.path.join(ROOT, "bench.py"),
                                   "--gpus", "2", "--steps", "4", "--warmup", "1", "--backend", "gloo"], cwd=ROOT, env=env)
    j = _line(out)
    assert j["n_gpus"] == 2 and "cpu_baseline" not in j
    # whole-job aggregate: two tuners' samples over the slowest rank's time
    assert abs(j["value"] - 2 * 4.0e6 * 4 / (j["ms_per_step"] * 4 / 1e3) / 1e6) / j["value"] < 1e-3
    _check_ranks(j, 2, "gloo")


def _check_ranks(j, world, backend):
    """r05: the line says for itself what every rank ran on: one record per rank, in rank order, with
    the device's PCI bus id, the rank's own rate, the communicator's size -- and whether the devices are distinct (two gloo
    ranks on the test box's one GPU are not, and say so; over RCCL that is an error, not a measurement)."""
    rk = j["ranks"]
    assert [r["rank"] for r in rk] == list(range(world))
    for r in rk:
        assert r["msps"] > 0 and r["seconds"] > 0 and r["pci_bus_id"] and r["device_name"]
        assert r["comm_ranks"] == world and r["backend"] == backend
        assert "ring_neighbour" in r and r["pid"] > 0
    assert len({r["pid"] for r in rk}) == world                   # one process per rank
    assert j["devices_distinct"] == (len({r["pci_bus_id"] for r in rk}) == world)
    # the slowest rank's time is the job's
    assert max(r["seconds"] for r in rk) <= j["ms_per_step"] * j["steps"] / 1e3 * 1.001 + 1e-6


@pytest.mark.parametrize("workload", ["c2", "c5"])
def test_eight_rank_line_on_one_gpu(workload):
    """BASELINE configs 4 and 5 as far as a one-GPU box can walk them: EIGHT ranks started by bench.py itself (file store,
    gloo: they share the GPU), each with its own 256-channel tuner (c2) or its chunks of the one stream and a halo from its
    ring neighbour (c5).  The line carries eight rank records."""
    extra = ["--workload", "c5"] if workload == "c5" else []
    out = _proc.output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", "--settle-ms", "20", "--resident-blocks", "4", "--spawn-timeout", "400",
                        "--rdzv-timeout", "200"] + extra, cwd=ROOT, env=_plain_env(), timeout=460)
    j = _line(out)
    assert j["n_gpus"] == 8 and j["steps"] == 3
    _check_ranks(j, 8, "gloo")
    if workload == "c5":
        assert all(r["ring_exchanges"] >= 3 for r in j["ranks"])


def test_c5_workload_line_and_two_rank_ring():
    """bench.py --workload c5 (BASELINE config 5): the JSON line at N = 1, and the two-rank path --
    chunks dealt round-robin, the halo through torch.distributed send/recv -- over gloo with both
    ranks on the one GPU of the test box (the driver's node uses nccl = RCCL)."""
    out = _proc.output([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--steps", "3",
                                   "--warmup", "1"], cwd=ROOT)
    j = _line(out)
    for k in REQUIRED:
        assert k in j, k
    assert j["config"]["halo_frames"] == 260_000 and j["config"]["chunk_frames"] % 20_000 == 0
    assert abs(j["value"] - j["config"]["chunk_frames"] * 3 / (j["ms_per_step"] * 3 / 1e3) / 1e6) / j["value"] < 1e-3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = _proc.output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                   "--master-addr", "127.0.0.1", "--master-port", "29713", os.path.join(ROOT, "bench.py"),
                                   "--workload", "c5", "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo"],
                                  cwd=ROOT, env=env)
    j = _line(out)
    assert j["n_gpus"] == 2
    assert abs(j["value"] - 2 * j["config"]["chunk_frames"] * 3 / (j["ms_per_step"] * 3 / 1e3) / 1e6) / j["value"] < 1e-3
    _check_ranks(j, 2, "gloo")
    assert all(r["ring_exchanges"] >= 3 and r["halo_wait_us_per_exchange"] is not None for r in j["ranks"])


def test_c5_ranks_take_the_same_number_of_settle_steps():
    """Every C5 step holds a halo exchange with the neighbour, so the ranks must agree on how many untimed settle steps they
    take.  r03 let each rank decide by its own clock: two ranks a moment apart at the mark deadlocked about one run in twenty
    (the hang that cost GPUTEST_r03 its time limit, profiles/r04_spawn_runs_before_fix.txt).  Here rank 1's settle clock is
    made to run 30 ms ahead of rank 0's: they disagree at every look, and the job still ends -- the decision is collective."""
    out = _proc.output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c5", "--steps", "3", "--warmup",
                        "1", "--backend", "gloo", "--settle-ms", "60", "--settle-skew-ms", "30", "--spawn-timeout", "120",
                        "--rdzv-timeout", "40"], cwd=ROOT, env=_plain_env(), timeout=170)
    j = _line(out)
    assert j["n_gpus"] == 2 and j["steps"] == 3


def test_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's call style) starts the two
    ranks itself and reports n_gpus = 2 -- it used to run one GPU and say n_gpus 1.  gloo: the test box
    has one GPU, which the two ranks share."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    for extra in ([], ["--workload", "c5"]):
        out = _proc.output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                                       "--warmup", "1", "--backend", "gloo", "--spawn-timeout", "120", "--rdzv-timeout", "60"] + extra,
                           cwd=ROOT, env=env, timeout=170)
        j = _line(out)
        assert j["n_gpus"] == 2 and j["steps"] == 3


def _plain_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}


@pytest.mark.parametrize("extra", [[], ["--workload", "c5", "--halo", "ring"]], ids=["c2", "c5-ring"])
def test_spawned_rank_over_rccl_at_world_size_one(extra):
    """The launch path a multi-GPU node takes, as far as one GPU can walk it: spawn_ranks starts the (one) rank,
    the rank joins through the file store with the nccl backend (= RCCL), barriers and the max-over-ranks reduction
    run on the device, C5's halo goes through wr_ring_*."""
    out = _proc.output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary", "--settle-ms", "0", "--spawn-timeout", "150",
                        "--rdzv-timeout", "60"] + extra, cwd=ROOT, env=_plain_env(), timeout=200)
    j = _line(out)
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["value"] > 0
    if extra:
        assert j["config"]["halo"] == "ring" and j["config"]["ring_exchanges"] == 3 + 1 + 1


def test_spawn_watchdog_kills_ranks_that_overrun():
    """A job that does not finish inside --spawn-timeout is killed -- both ranks -- and the launcher exits 124 with
    a message: a hang in the N > 1 path costs its own limit, not the caller's."""
    import time
    t0 = time.monotonic()
    p = _proc.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                   "--backend", "gloo", "--spawn-timeout", "0.2"], cwd=ROOT, env=_plain_env(), check=False, timeout=300)
    assert p.returncode == 124, (p.returncode, p.stderr[-2000:])
    assert b"did not finish within --spawn-timeout" in p.stderr and b"all ranks killed" in p.stderr
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert time.monotonic() - t0 < 280


def test_one_failing_rank_ends_the_job():
    """Rank 1 exits with 3 after the rendezvous: the launcher takes rank 0 down (it would otherwise sit in a
    barrier until the collective's timeout) and exits 3."""
    p = _proc.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                   "--backend", "gloo", "--fail-rank", "1", "--spawn-timeout", "150", "--rdzv-timeout", "100"],
                  cwd=ROOT, env=_plain_env(), check=False, timeout=200)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert b"rank 1 exited with 3" in p.stderr
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_more_gpus_than_the_box_has_is_an_error():
    """--gpus 8 on a box with fewer GPUs exits non-zero with a message instead of measuring what is there."""
    import torch
    want = torch.cuda.device_count() + 7
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    for extra in ([], ["--workload", "c5"]):
        p = _proc.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "2",
                            "--warmup", "1"] + extra, cwd=ROOT, env=env, check=False)
        assert p.returncode != 0
        assert b"refusing to measure fewer GPUs" in p.stderr and not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    # under a launcher that made fewer ranks than --gpus says: an error too
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = _proc.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], cwd=ROOT, env=env2,
                  check=False)
    assert p.returncode != 0 and b"WORLD_SIZE=1" in p.stderr


_RCCL_PROBE = r"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29733")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(1024, dtype=torch.float32, device="cuda")
dist.all_reduce(x)                                   # RCCL: ncclAllReduce on one rank
y = torch.empty_like(x)
ops = [dist.P2POp(dist.isend, x, 0), dist.P2POp(dist.irecv, y, 0)]     # the halo ring's send/recv pair, to itself
for w in dist.batch_isend_irecv(ops): w.wait()
torch.cuda.synchronize()
assert bool((y == torch.arange(1024, device="cuda")).all())
import ctypes
maps = open("/proc/self/maps").read()
assert "librccl" in maps, "RCCL is not mapped"
print("rccl ok", torch.cuda.nccl.version())
dist.destroy_process_group()
"""


def test_rccl_loads_and_reduces_at_world_size_one():
    """The box proves RCCL loads: the nccl backend (= RCCL on ROCm) initialises at world size 1, an
    all-reduce and a grouped send/recv pair (the C5 halo ring's primitive) run on the device."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = _proc.output([sys.executable, "-c", _RCCL_PROBE], cwd=ROOT, env=env, timeout=240)
    assert b"rccl ok" in out


def test_c5_halo_through_the_native_ring():
    """bench.py --workload c5 --halo ring at N = 1: the rank is its own ring neighbour, so every chunk's halo goes
    through wr_ring_* -- ncclSend/ncclRecv on the ring's stream, posted a round ahead -- the branch a multi-GPU
    node takes over nccl.  One exchange per step plus the one posted ahead."""
    out = _proc.output([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--halo", "ring",
                                   "--steps", "3", "--warmup", "1", "--settle-ms", "0"], cwd=ROOT)
    j = _line(out)
    assert j["n_gpus"] == 1 and j["config"]["halo"] == "ring" and "wr_ring" in j["config"]["workload"]
    assert j["config"]["ring_exchanges"] == 3 + 1 + 1
