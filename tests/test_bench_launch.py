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
