"""Child processes of the test suite: every one runs in its own session under a deadline.  When the
deadline passes the whole session is killed and the test FAILS with the child's output -- a hang
becomes a named failure instead of eating the suite's time limit (GPUTEST_r03: one un-timed
subprocess cost the whole parity run)."""
import os
import signal
import subprocess

DEFAULT_TIMEOUT = 180.0


class ChildTimeout(AssertionError):
    pass


def run(cmd, timeout=DEFAULT_TIMEOUT, check=True, **kw):
    """subprocess.run with stdout/stderr captured, its own session, a deadline, and the child's
    stderr in the failure message.  Returns the CompletedProcess."""
    kw.setdefault("stdout", subprocess.PIPE)
    kw.setdefault("stderr", subprocess.PIPE)
    kw.setdefault("stdin", subprocess.DEVNULL)
    p = subprocess.Popen(cmd, start_new_session=True, **kw)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            out, err = p.communicate(timeout=10)
        except Exception:
            out, err = b"", b""
        raise ChildTimeout("child did not finish in %.0f s: %r\n---- stdout\n%s\n---- stderr\n%s" % (
            timeout, cmd, (out or b"").decode(errors="replace")[-4000:], (err or b"").decode(errors="replace")[-8000:]))
    r = subprocess.CompletedProcess(cmd, p.returncode, out, err)
    if check and p.returncode != 0:
        raise AssertionError("child exited with %d: %r\n---- stdout\n%s\n---- stderr\n%s" % (
            p.returncode, cmd, (out or b"").decode(errors="replace")[-4000:], (err or b"").decode(errors="replace")[-8000:]))
    return r


def output(cmd, timeout=DEFAULT_TIMEOUT, **kw):
    """check_output with a deadline (see run)."""
    return run(cmd, timeout=timeout, **kw).stdout


def spawn_ranks(fn, world, args, timeout=DEFAULT_TIMEOUT):
    """torch.multiprocessing.spawn(fn, args, nprocs=world) under a deadline: ranks that have not
    finished by then are killed and the test fails.  fn(rank, *args)."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.spawn(fn, args=args, nprocs=world, join=False)
    deadline = time.monotonic() + timeout
    try:
        while not ctx.join(timeout=1.0):                  # raises when a rank failed
            if time.monotonic() > deadline:
                raise ChildTimeout("%d ranks of %s did not finish in %.0f s" % (world, getattr(fn, "__name__", fn), timeout))
    finally:
        for p in ctx.processes:
            if p.is_alive():
                p.kill()
        for p in ctx.processes:
            p.join(10)


def init_gloo(rank, world, rdzv_file, seconds=60.0):
    """gloo ranks of one test meet through a file store (no port to pick or collide on), and wait for
    each other no longer than `seconds`."""
    import datetime
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + rdzv_file, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=seconds))
    return dist
