"""CPU: bench.py's own launcher (spawn_ranks) refuses to run without a GPU -- the HIP path has no CPU
fallback -- and says so, for every way of asking for N > 1."""
import os
import sys

import pytest

import _proc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_gpus() > 0, reason="this box has a GPU: tests/test_gpu_bench.py covers the launcher there")
@pytest.mark.parametrize("extra", [["--backend", "gloo"], [], ["--workload", "c5"]], ids=["gloo", "nccl", "c5"])
def test_no_gpu_is_an_error_not_a_cpu_run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = _proc.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + extra,
                  cwd=ROOT, env=env, check=False, timeout=200)
    assert p.returncode != 0
    assert b"no GPU" in p.stderr or b"refusing to measure fewer GPUs" in p.stderr
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_ranks_that_share_a_gpu_over_rccl_are_an_error_not_a_measurement():
    """`bench.py --gpus N` over RCCL whose ranks turn out to sit on fewer than N distinct GPUs
    (by PCI bus id) exits with an error that names the devices -- it never prints a line; over gloo (the tests' worlds on one
    GPU) the line is printed and says `devices_distinct: false` for itself."""
    sys.path.insert(0, ROOT)
    import bench

    class FakeDist:
        def __init__(self, recs):
            self.recs = recs

        def all_gather_object(self, out, rec):
            out[:] = self.recs

    same = [{"rank": r, "host": "box", "device_index": 0, "pci_bus_id": "0000:05:00.0"} for r in range(2)]
    with pytest.raises(SystemExit) as e:
        bench.gather_ranks(FakeDist(same), 2, same[0], "nccl")
    assert "2 ranks over RCCL on 1 distinct GPU(s)" in str(e.value) and "0000:05:00.0" in str(e.value)
    recs, distinct = bench.gather_ranks(FakeDist(same), 2, same[0], "gloo")
    assert not distinct and len(recs) == 2
    two = [dict(same[0]), dict(same[1], pci_bus_id="0000:15:00.0", device_index=1)]
    recs, distinct = bench.gather_ranks(FakeDist(two), 2, two[0], "nccl")
    assert distinct
