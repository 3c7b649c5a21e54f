#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

  metric   complex Msamples/s of tuner input consumed by the node with 256 DDC+NFM
           receiver channels per tuner (one tuner per GPU, no collective: "weak").
  step     one 4 000 000-frame block (40 ms of a 100 Msps stream) of synthetic IQ,
           already resident in HBM, pushed through all 256 receiver chains of the
           rank's tuner (BASELINE config 2 / SURVEY C2).

  python bench.py --gpus N --steps K --warmup W
  N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...
  bench.py --gpus N ...: RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* come from the environment), or plain
  `python bench.py --gpus N`, which starts the N ranks itself (spawn_ranks below) -- one per GPU,
  RCCL ("nccl") by default.  N GPUs must be there: with fewer the job exits non-zero instead of
  quietly measuring one GPU (`--backend gloo` lets ranks share a GPU; that is for the tests).

Prints ONE JSON line on rank 0 (see the contract in the task description) carrying
`roofline` (dominant kernel, HIP-event timed inside the timed region) and, at N = 1,
`cpu_baseline` (the oracle -- a scalar port of the reference CPU path -- timed on ALL host
cores, one pipeline thread per core with its own subset of the receivers, and on one core,
over a bounded sample of the same workload) and `secondary.c3` (BASELINE config 3: the
SpectrumSink waterfall, 65536-point FFT at 50 % overlap, off the same resident stream),
`secondary.c1` (BASELINE config 1) and `secondary.host_fed` (the same C2 job through the C++ host
classes with the block in HOST memory: the PCIe-inclusive rates -- reported beside `value`, never as it).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_SAMPLE = 8.0 + 4.0 * 256 / (400 * 5)      # 8.512 B per input sample
VALU_PER_TAP = {"rotate": 7, "split": 11}                 # VALU instructions per channel-tap
HBM_PEAK_GBPS = 8000.0                                    # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3                           # MI355X_MICROARCH.md: peak FP32 (vector)
ALGO_FLOP_PER_SAMPLE = 445.0                              # flop per tuner input sample, 256 channels
C3_FFT, C3_HOP = 65536, 32768                             # BASELINE config 3
C3_BYTES_PER_FRAME = 12 * C3_FFT                          # 8 B in + 4 B dB out per bin


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["c2", "c5"], default="c2",
                    help="c2 (default, the headline): one 256-channel tuner per GPU off its own 100 Msps stream, no "
                         "collective.  c5: ONE 1 Gsps stream cut in time, chunks dealt round-robin to the ranks, the "
                         "halo of every chunk from the ring neighbour (RCCL send/recv)")
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--nco", choices=["rotate", "split", "exact"], default="rotate")
    ap.add_argument("--resident-blocks", type=int, default=12,
                    help="consecutive blocks of the stream kept in HBM and cycled through (12 x 32 MB is "
                         "more than the 256 MB Infinity Cache holds, so every step reads its block from HBM)")
    ap.add_argument("--no-stream", dest="stream", action="store_false",
                    help="headline without wr_tuner_set_streaming: a kernel launch per step (or per --blocks-per-launch steps).  "
                         "Default: the streaming launch -- the tuner's blocks go to ONE persistent launch through a doorbell, a "
                         "block's audio is complete ~15 us after its last channel-rate frame whether or not another block follows "
                         "(dspblock.cxx:169-212), same bits (tests/test_gpu_stream.py); the launch-per-block figures are "
                         "reported as secondary.c2_one_launch_per_block / c2_four_blocks_per_launch")
    ap.add_argument("--blocks-per-launch", type=int, default=1,
                    help="only with --no-stream: wr_tuner_set_blocks_per_launch, consecutive resident blocks the tuner holds and "
                         "launches as one (a step stays one 40 ms block; audio is delivered per launch, up to n - 1 blocks late)")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="after the W warm-up steps, keep stepping (untimed) for this long before the timed region: an "
                         "MI355X that was idle starts a kernel stream at a reduced clock and takes ~50 ms of continuous "
                         "load to reach its steady state (measured: the same launch 42-44 us in the first 6 ms, 34.4 us "
                         "from 45 ms on, profiles/r02_clock_ramp.txt).  0 turns it off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-blocks", type=int, default=0,
                    help="blocks of the all-cores CPU baseline (0: as many as take about 10-20 s)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 (SpectrumSink) measurement")
    ap.add_argument("--profile-stride", type=int, default=64,
                    help="one HIP event pair around every n consecutive launches of the dominant kernel, at most all of the "
                         "timed region's (an event record between two launches costs the stream ~10 us; n = 1: every "
                         "launch stamps its own start and stop, ~4 us of stream time per launch)")
    ap.add_argument("--halo", choices=["auto", "copy", "ring"], default="auto",
                    help="c5: where a chunk's halo comes from.  ring: the C ABI's wr_ring_* (RCCL ncclSend/ncclRecv on a side "
                         "stream, issued a round ahead) -- what N > 1 over nccl uses; at N = 1 the rank is its own neighbour.  "
                         "copy: a device copy (N = 1 only).  auto: ring for N > 1 over nccl, copy at N = 1; gloo ranks "
                         "(tests: two ranks on one GPU) exchange host tensors through torch.distributed")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks from this process even at --gpus 1 (tests: the launch path itself -- file-store "
                         "rendezvous, watchdog, RCCL at world size 1)")
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)    # tests: this rank exits 3 after the rendezvous
    ap.add_argument("--settle-skew-ms", type=float, default=0.0, help=argparse.SUPPRESS)   # tests: rank r's settle clock runs r x this ahead
    ap.add_argument("--spawn-timeout", type=float, default=1500.0,
                    help="N > 1 without a launcher: seconds the ranks this process starts may take before it kills them "
                         "and exits 124 with their stderr")
    ap.add_argument("--rdzv-timeout", type=float, default=300.0,
                    help="N > 1: seconds init_process_group (and every host-side collective) may wait for the other "
                         "ranks -- a fresh box pages torch in for a minute or two, not at the same pace in every rank")
    ap.add_argument("--watchdog", type=float, default=1200.0,
                    help="seconds after which a rank that is still running dumps every thread's stack to stderr and "
                         "exits (faulthandler): a hang becomes an error that says where.  0 turns it off")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo lets two ranks share one GPU in tests)")
    return ap.parse_args()


def _hip_runtime():
    import ctypes
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            pass
    return None


def rank_record(torch, dist, rank, local_rank, world, device_index, have, backend, units, seconds, extra=None):
    """What THIS rank ran on and what it did -- gathered on rank 0 into the JSON line's `ranks`, so that a multi-GPU line
    says for itself whether N distinct GPUs took part (radio.cxx:56-59: one front end, one tuner, per device): the
    device's PCI bus id and UUID, whether the ring neighbour's GPU is reachable peer to peer, the RCCL the process
    loaded and the size of its communicator, and the rank's own rate over its own clock."""
    import ctypes
    import socket
    rec = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "pid": os.getpid(),
           "device_index": device_index, "seconds": round(seconds, 6),
           "msps": round(units / seconds / 1e6, 2) if seconds > 0 else None}
    try:
        p = torch.cuda.get_device_properties(device_index)
        rec["device_name"] = p.name
        rec["device_uuid"] = str(getattr(p, "uuid", "")) or None
        rec["cus"] = p.multi_processor_count
    except Exception as e:                                           # (never the reason a measurement is lost)
        rec["device_error"] = str(e)[:100]
    hip = _hip_runtime()
    if hip is not None:
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, device_index) == 0:
            rec["pci_bus_id"] = buf.value.decode()
        if world > 1:
            nb = ((local_rank + 1) % world) % have                   # the device of rank + 1, the halo ring's neighbour
            can = ctypes.c_int(-1)
            if nb != device_index and hip.hipDeviceCanAccessPeer(ctypes.byref(can), device_index, nb) == 0:
                rec["ring_neighbour"] = {"device_index": nb, "peer_access": bool(can.value)}
            else:
                rec["ring_neighbour"] = {"device_index": nb, "peer_access": None, "same_device": nb == device_index}
    rec["backend"] = backend if dist is not None else None
    if dist is not None:
        rec["comm_ranks"] = dist.get_world_size()
        if backend == "nccl":
            try:
                rec["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rec["rccl_version"] = None
    if extra:
        rec.update(extra)
    return rec


def gather_ranks(dist, world, rec, backend):
    """-> (records of all ranks in rank order, are their devices distinct) on every rank; N = 1: just this one"""
    recs = [rec]
    if dist is not None and world > 1:
        recs = [None] * world
        dist.all_gather_object(recs, rec)
    ids = [r.get("pci_bus_id") or "%s/%s" % (r.get("host"), r.get("device_index")) for r in recs]
    distinct = len(set(ids)) == len(ids)
    if backend == "nccl" and world > 1 and not distinct:
        raise SystemExit("bench.py: %d ranks over RCCL on %d distinct GPU(s) (%s): not an %d-GPU measurement"
                         % (world, len(set(ids)), ", ".join(ids), world))
    return recs, distinct


def _gpu_count(timeout_s=240.0):
    """The number of GPUs torch sees, asked of a CHILD process: the launching process never opens the
    device (it only waits for its ranks), and a runtime that does not come up costs a timeout, not the job."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        return int(r.stdout.decode().strip().splitlines()[-1])
    except Exception as e:                                          # incl. TimeoutExpired
        raise SystemExit("bench.py: could not count the GPUs (%s)" % str(e)[:200])


def _kill_ranks(procs):
    import signal
    for p in procs:
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)                    # every rank leads its own session
            except (ProcessLookupError, PermissionError):
                pass
    for p in procs:
        try:
            p.wait(timeout=10)
        except Exception:
            pass


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (Radio::run pumps its front
    ends one after the other, radio.cxx:56-59; here every front end has its own process and GPU).

    r04: the ranks are this script's own children -- no torch.distributed.run agent in between -- and meet
    through a FILE store in a fresh temp directory (`init_method=file://...`), so no rendezvous port is
    picked, raced for or left in TIME_WAIT; gloo's and RCCL's own data sockets bind to kernel-chosen
    ports.  This process never imports torch or opens the GPU.  It is also the watchdog: a rank that exits
    non-zero takes the others down at once, and a job that overruns --spawn-timeout is killed (every
    rank's session) with the ranks' stderr tails on ours and exit code 124 -- a hang here becomes an error
    with a name, not a silent wait.  Rank 0's stdout is ours (the ONE JSON line); the other ranks' goes to
    stderr."""
    import shutil
    import subprocess
    import tempfile
    have = _gpu_count()
    if have < args.gpus and args.backend == "nccl":
        raise SystemExit("bench.py: --gpus %d but this box has %d GPU(s): refusing to measure fewer GPUs than asked for "
                         "(--backend gloo lets ranks share a GPU, for tests only)" % (args.gpus, have))
    if have < 1:
        raise SystemExit("bench.py: no GPU (the HIP path has no CPU fallback)")
    rdzv = tempfile.mkdtemp(prefix="wr_bench_rdzv_")
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                                                             "MASTER_PORT", "GROUP_RANK", "ROLE_RANK")}
    base["WR_BENCH_RDZV_FILE"] = os.path.join(rdzv, "store")
    base["WORLD_SIZE"] = base["LOCAL_WORLD_SIZE"] = str(args.gpus)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("OMP_NUM_THREADS", "1")                         # what torch.distributed.run gives its workers
    if args.backend == "nccl":
        base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this host driver
    procs, errs = [], []
    t0 = time.monotonic()
    try:
        for r in range(args.gpus):
            err = open(os.path.join(rdzv, "rank%d.err" % r), "w+b")
            errs.append(err)
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdin=subprocess.DEVNULL, stdout=None if r == 0 else sys.stderr, stderr=err,
                                          start_new_session=True))
        rc, why = 0, None
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc, why = (bad[0][1] if bad[0][1] > 0 else 1), "rank %d exited with %d" % bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() - t0 > args.spawn_timeout:
                rc, why = 124, "the %d ranks did not finish within --spawn-timeout %.0f s (still running: %s)" % (
                    args.gpus, args.spawn_timeout, [r for r, c in enumerate(codes) if c is None])
                break
            time.sleep(0.05)
        _kill_ranks(procs)
        for r, err in enumerate(errs):                              # the ranks' stderr, in rank order, on ours
            err.flush()
            err.seek(0)
            data = err.read()
            if rc != 0:
                data = data[-6000:]
            if data:
                sys.stderr.write("".join("[rank %d] %s\n" % (r, l) for l in data.decode(errors="replace").splitlines()))
        if why:
            sys.stderr.write("bench.py: %s -- all ranks killed\n" % why)
        sys.stderr.flush()
    finally:
        _kill_ranks(procs)
        for err in errs:
            err.close()
        shutil.rmtree(rdzv, ignore_errors=True)
    raise SystemExit(rc)


def reference_cpu_baseline(blocks, channels):
    """cpu_baseline with kind "reference" (r04): the reference's OWN DownConverter / LowPass / Demodulator classes --
    /root/reference's sources compiled into oracle/_ref/libwr_ref_chain.so where they lay, which travels to the GPU box
    prebuilt -- on the host cores, through oracle/ref_cpu_baseline.py in a process of its own (its FFTW calls, two 64-point
    transforms per receiver at start(), go to the image's hipFFTW and so to the system's HIP runtime; this process holds
    torch's).  None when that library is not there or the run fails: the caller then times the port."""
    import subprocess
    script = os.path.join(ROOT, "oracle", "ref_cpu_baseline.py")
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libwr_ref_chain.so")) or os.environ.get("WR_BENCH_CPU_PORT"):
        return None
    try:
        r = subprocess.run([sys.executable, script, str(int(blocks)), str(int(channels))], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, timeout=600)
        d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
        return d if "value" in d and d["value"] > 0 else None
    except Exception:
        return None


def cpu_baseline(cfg, ifs, blocks):
    """The oracle's Receiver chains (a faithful scalar port of the reference CPU path:
    full-rate mixer, block copy, 64-tap FIRs, atan2f) over the same synthetic stream:
    T pipeline threads, T = all host cores, each with its own subset of the receivers of
    the one tuner buffer (the reference pipeline itself is single-threaded, radio.cxx:56-59),
    and the one-thread figure (one block)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wr_oracle as oracle
    from webradio_amd import synth
    n = cfg["block_frames"]
    iq = synth.fm_stream(n, cfg["input_rate"], ifs[::4], seed=12345)
    args = (cfg["input_rate"], ifs, cfg["chan_passband"], cfg["chan_rate"], oracle.FM,
            cfg["audio_passband"], cfg["audio_rate"], iq)
    cores = max(1, min(len(os.sched_getaffinity(0)), len(ifs)))
    one = oracle.bench_receivers(*args, 1)
    if blocks <= 0:
        # calibrate on two blocks (the threads share the memory system: they do not scale like the
        # cores), then as many blocks as take about 12 s
        probe = oracle.bench_receivers_mt(*args, 2, cores)
        blocks = int(max(2, min(256, round(12.0 / max(probe / 2.0, 1e-3)))))
    secs = oracle.bench_receivers_mt(*args, blocks, cores)
    return {
        "value": round(n * blocks / secs / 1e6, 4),
        "unit": "complex Msamples/s (tuner input, all %d channels)" % len(ifs),
        "cores": cores,
        "kind": "port",
        "sample": "%d channels x %d block(s) of %d frames on %d threads (disjoint channel subsets), %.1f s" % (
            len(ifs), blocks, n, cores, secs),
        "one_core": {"value": round(n / one / 1e6, 4), "cores": 1,
                     "sample": "%d channels x 1 block of %d frames, %.1f s" % (len(ifs), n, one)},
    }


def committed_traffic():
    """HBM bytes of the dominant kernels from the PMC passes of tools/profile_round.sh (profiles/traffic.json): they
    cannot be collected from inside this process."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return {}


def c3_secondary(torch, dev, blocks, n, steps, settle_ms):
    """BASELINE config 3: SpectrumSink (io/spectrumsink.cxx:88-142) as a waterfall -- 65536-point
    Hamming-windowed FFT every 32768 frames, dB with fft-shift -- over the same resident stream:
    consecutive 4 M-frame blocks (more than the Infinity Cache holds), one row buffer per block.
    Timed with events on the stream the kernels run on."""
    from webradio_amd.device import Spectrum
    rows = (n - C3_FFT) // C3_HOP + 1
    spec = Spectrum(dev, C3_FFT, C3_HOP)
    nb = len(blocks)
    outs = [torch.empty(rows * C3_FFT, dtype=torch.float32, device="cuda") for _ in range(nb)]
    for b in range(min(2, nb)):
        spec.batch_db(blocks[b], rows, outs[b])
    torch.cuda.synchronize()
    t_settle = time.perf_counter()                     # clocks: see --settle-ms
    while settle_ms > 0 and (time.perf_counter() - t_settle) * 1e3 < settle_ms:
        for i in range(50):
            spec.batch_db(blocks[i % nb], rows, outs[i % nb])
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        spec.batch_db(blocks[i % nb], rows, outs[i % nb])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    db = outs[(steps - 1) % nb][:C3_FFT]
    assert bool(torch.isfinite(db).all())
    spec.destroy()
    achieved = rows * C3_BYTES_PER_FRAME / (ms / 1e3) / 1e9
    return {
        "workload": "C3: SpectrumSink waterfall, %d-point FFT, hop %d (50 %% overlap), %d frames per 4 000 000-frame "
                    "block, %d resident blocks cycled" % (C3_FFT, C3_HOP, rows, nb),
        "metric": "FFT frames/s",
        "value": round(rows / (ms / 1e3), 1),
        "msps_new_samples": round(rows * C3_HOP / (ms / 1e3) / 1e6, 1),
        "ms_per_block": round(ms, 5),
        "steps": steps,
        "roofline": {"bound": "hbm", "kernel": "k_fft64k (window + 65536-point FFT + dB, fft-shift)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 5),
                     "algorithmic_bytes_per_frame": C3_BYTES_PER_FRAME,
                     "algorithmic_bytes_per_launch": rows * C3_BYTES_PER_FRAME,
                     # PMC FETCH_SIZE + WRITE_SIZE of pass 1 + pass 2 per launch pair of `rows` frames, FETCH_SIZE
                     # doubled as MI355X_MICROARCH.md (HBM) prescribes for wide coalesced loads on gfx950
                     "traffic": (committed_traffic().get("c3") or {}).get("hbm_bytes_per_launch")
                     if (committed_traffic().get("c3") or {}).get("frames_per_launch") == rows else None,
                     "traffic_unit": "bytes per launch pair (k_fft64k_pass1 + pass2), profiles/traffic.json"},
    }


def frontend_secondary(torch, dev, tuner, blocks, n, steps, streaming_ok):
    """The whole FrontEnd (radio.cxx:120-133: a tuner feeds its receivers AND a SpectrumSink, DspBlock::run hands both every
    block): BASELINE configs 2 and 3 on the SAME resident blocks of one tuner -- 256 receivers and the 65536-point waterfall at
    50 % overlap -- instead of timed apart.  Three ways:
      per_block_launches        a tuner launch and the waterfall's two passes per block, one stream: what r01-r04's path does;
      streaming_closed_per_block  wr_tuner_set_streaming with a waterfall batch every block: every batch closes the launch
                                (the FFT passes need the CUs the launch holds) -- what a stream costs a front end that
                                wants EVERY frame's row;
      streaming_newest_frame    the reference's own semantics (spectrumsink.cxx:114-116,136-141: only the most recent frame
                                is observable; waterfallhandler.cxx:56-61 reads it at 5 Hz = every fifth 40 ms block): a
                                push per block beside the open launch (kept, not transformed), one getSpectrum per five
                                blocks (one closed launch per poll).
    Times are wall clock over `steps` blocks between two synchronisations; the roofline fractions take each config's
    algorithmic bytes over the COMBINED time."""
    from webradio_amd.device import Spectrum
    rows = (n - C3_FFT) // C3_HOP + 1
    nb = len(blocks)
    outs = [torch.empty(rows * C3_FFT, dtype=torch.float32, device="cuda") for _ in range(min(nb, 4))]
    spec = Spectrum(dev, C3_FFT, C3_HOP)

    def run(stream_on, mode, k, poll_every=5):
        tuner.flush()
        tuner.streaming(stream_on)
        tuner.blocks_per_launch(1)
        for phase in range(2):                              # the same loop untimed first (clocks, allocations), then timed
            tuner.flush()
            torch.cuda.synchronize()
            opened0 = tuner.stream_info()[1]
            t0 = time.perf_counter()
            for i in range(k):
                tuner.submit_device(blocks[i % nb], n)
                if mode == "waterfall":
                    spec.batch_db(blocks[i % nb], rows, outs[i % len(outs)])
                else:
                    spec.push_device(blocks[i % nb], n)
                    if poll_every and (i + 1) % poll_every == 0:
                        spec.get_db()
            tuner.flush()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        us = dt / k * 1e6
        c2 = n * ALGO_BYTES_PER_SAMPLE / (us * 1e-6) / 1e9
        c3 = (rows * C3_BYTES_PER_FRAME if mode == "waterfall" else C3_BYTES_PER_FRAME / 5.0) / (us * 1e-6) / 1e9
        return {"us_per_block": round(us, 2), "msps": round(n / us, 1), "streaming_launches": tuner.stream_info()[1] - opened0,
                "c2_frac_of_hbm_roof": round(c2 / HBM_PEAK_GBPS, 4), "c3_frac_of_hbm_roof": round(c3 / HBM_PEAK_GBPS, 4),
                "spectrum": "%d waterfall rows per block" % rows if mode == "waterfall" else
                            "newest frame, polled every 5th block" if poll_every else "newest frame kept, never polled"}

    k = max(10, min(steps, 60)) // 5 * 5
    out = {"workload": "C2 + C3 on the same blocks: 256 receivers and the SpectrumSink (65536 points, hop 32768) of one FrontEnd, "
                       "%d-frame blocks resident in HBM" % n,
           "blocks": k,
           "per_block_launches": run(False, "waterfall", k),
           "per_block_launches_newest_frame": run(False, "newest", k)}
    if streaming_ok:
        out["streaming_closed_per_block"] = run(True, "waterfall", k)
        out["streaming_newest_frame"] = run(True, "newest", k)
        quiet = run(True, "newest", k, 0)
        out["streaming_newest_frame"]["us_per_block_between_polls"] = quiet["us_per_block"]
        out["streaming_newest_frame"]["us_per_poll"] = round((out["streaming_newest_frame"]["us_per_block"] - quiet["us_per_block"]) * 5, 1)
        out["streaming_newest_frame"]["note"] = ("a poll = wr_spectrum_get_db: closes the launch (its drain), transforms the kept frame, "
                                                 "brings 256 KB of dB values to the host and returns -- synchronous, the GPU idles meanwhile; "
                                                 "at the UI's 5 Hz that is once per 200 ms of signal")
        out["lazy"] = dict(zip(("pushes_kept", "transformed_on_demand"), spec.lazy_info()))
    tuner.flush()
    tuner.streaming(streaming_ok)
    spec.destroy()
    return out


def power_secondary(torch, tuner, blocks, n, seconds=2.5):
    """What the package draws while the headline's kernel runs back to back (rocm-smi, sampled from a thread beside
    ~2.5 s of 400-block streaming launches; outside every timed region).  r06: the launch sits at the package's power
    limit -- the resource that binds it (profiles/r06_power.txt).  None where rocm-smi says nothing."""
    import re
    import subprocess
    import threading
    smi = "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    samples, stop = [], [False]

    def read():
        o = subprocess.run([smi, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        w = re.findall(r'Package Power \(W\)": "([0-9.]+)"', o)
        c = re.findall(r'"sclk clock speed:": "\(([0-9]+)Mhz\)"', o)
        return (time.perf_counter(), float(w[0]) if w else None, int(c[0]) if c else None)

    def sampler():
        while not stop[0]:
            try:
                samples.append(read())
            except Exception:
                pass
            time.sleep(0.1)

    try:
        tuner.flush()
        torch.cuda.synchronize()
        time.sleep(0.5)
        idle = read()
        th = threading.Thread(target=sampler)
        th.start()
        t0 = time.perf_counter()
        done = 0
        while time.perf_counter() - t0 < seconds:
            for i in range(400):
                tuner.submit_device(blocks[i % len(blocks)], n)
            tuner.flush()
            torch.cuda.synchronize()
            done += 400
        t1 = time.perf_counter()
        stop[0] = True
        th.join()
        w = [x[1] for x in samples if x[1] is not None and t0 + 0.6 < x[0] <= t1]
        c = [x[2] for x in samples if x[2] is not None and t0 + 0.6 < x[0] <= t1]
        if not w:
            return None
        return {"package_w": round(sum(w) / len(w), 1), "package_w_max": max(w), "samples": len(w),
                "sclk_mhz": round(sum(c) / len(c)) if c else None, "idle_w": idle[1], "idle_sclk_mhz": idle[2],
                "limit_w": 1400, "us_per_block": round((t1 - t0) / done * 1e6, 2),
                "how": "rocm-smi --showpower --showclocks beside %.1f s of 400-block streaming launches, samples of the first 0.6 s dropped" % seconds,
                "reading": "the launch draws what the package may (MI355X: 1400 W) and the shader clock gives way (2.4 GHz idle): "
                           "the kernel's time is its energy -- DESIGN.md 3.1, profiles/r06_power.txt"}
    except Exception:
        stop[0] = True
        return None


def host_fed_secondary():
    """The drop-in path with the block in HOST memory (PCIe inside the timing; never `value`): tests/cxx/host_bench --
    one FrontEnd, 256 Receivers wired as radio.cxx wires them, Radio::run() pumping 4 000 000-frame blocks through the C++
    host classes -- from a byte-format source and from a float32 source, the sinks' audio one block late
    (WEBRADIO_AUDIO_LATE=1) and on time.  Runs after everything timed above; None when the binary has not been built."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cxx", "host_bench")
    if not os.path.exists(exe):
        return None
    rows = []
    # (r06) "dev": a source that produces its blocks in GPU memory (DeviceBlock): the host classes STREAM it -- Radio::run() rings
    # the doorbell of k_tuner_stream, the kernel the headline times; the same source with WEBRADIO_STREAM=0 beside it
    for kind in ("dev", "dev-nostream", "u8", "f32"):
        src = kind.split("-")[0]
        for late, sparse in (("1", "1"), ("0", "1"), ("0", "0")):
            if src == "dev" and sparse == "0":
                continue
            env = dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_AUDIO_LATE=late, WEBRADIO_SPARSE=sparse,
                       WEBRADIO_STREAM="0" if kind == "dev-nostream" else "1")
            try:
                r = subprocess.run([exe, "256", "100", "4000000", src], env=env, capture_output=True, text=True, timeout=120)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
                d = json.loads(line)
            except Exception as e:                      # the secondary figures never fail the headline
                rows.append({"source": src, "audio_late": int(late), "error": str(e)[:200]})
                continue
            rows.append({"source": d["source"], "audio": d["audio"],
                         "stream_info": d.get("stream_info"),
                         "staging": "none: the block lies in GPU memory" + ("" if env["WEBRADIO_STREAM"] == "1" else " (WEBRADIO_STREAM=0: a launch per block)")
                                    if src == "dev" else
                                    "sparse: only the frames under the channel filters' taps (64 of every 400) and the block's "
                                    "tail cross PCIe, read by a kernel (wr_stage_windows_from_host)" if sparse == "1"
                                    else "the whole block crosses PCIe (WEBRADIO_SPARSE=0: r03's path)",
                         "ms_per_block": d["ms_per_block"], "msps_tuner_input": d["msps_tuner_input"], "blocks": d["blocks"]})
    return {"workload": "C2 through the C++ host classes (Radio::run, 256 Receivers): the block in GPU memory (streamed: k_tuner_stream) "
                        "and in host memory (PCIe inside the timing)",
            "unit": "complex Msamples/s of tuner input", "runs": rows}


def c1_secondary(torch, dev, steps, settle_ms):
    """BASELINE config 1 on the GPU: ONE receiver (DDC + FM) off a 2.048 Msps stream in the RTL-SDR byte
    format, 131 072-frame blocks resident in HBM (the reference's own CPU-runnable case; parity in
    tests/test_gpu_tuner.py::test_c1_single_receiver_u8_file).  One small launch sequence per 64 ms
    block: what it measures is launch latency, not the chip."""
    from webradio_amd import capi, synth
    from webradio_amd.device import Tuner
    c1 = synth.C1
    n, nb = c1["block_frames"], 16
    raw = torch.from_numpy(synth.rtl_u8_stream(n * nb, c1["input_rate"], c1["if_hz"])).cuda()
    blocks = [raw[2 * n * b: 2 * n * (b + 1)] for b in range(nb)]
    t = Tuner(dev, c1["input_rate"], 1, n, capi.WR_NCO_ROTATE)
    t.add_receiver(c1["if_hz"], c1["chan_passband"], c1["chan_rate"], capi.WR_FM, c1["audio_passband"], c1["audio_rate"])
    for i in range(8):
        t.submit_u8_device(blocks[i % nb], n)
    torch.cuda.synchronize()
    t_settle = time.perf_counter()
    while settle_ms > 0 and (time.perf_counter() - t_settle) * 1e3 < settle_ms / 3:
        for i in range(100):
            t.submit_u8_device(blocks[i % nb], n)
        torch.cuda.synchronize()
    def timed(k):
        t.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            t.submit_u8_device(blocks[i % nb], n)
        t.flush()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    dt = timed(steps)
    a = t.fetch(0, capi.WR_STAGE_AUDIO, n)
    assert a.size == n // 8 // 8 and bool((a == a).all()) and float(abs(a).max()) > 0.0
    # r06: the same blocks through ONE streaming launch (D2 = 8 is eligible): a doorbell per 64 ms block
    # instead of a launch -- parity: tests/test_gpu_stream.py::test_c1_capture_streams_against_the_reference_vectors
    streamed = None
    t.streaming(True)
    timed(50)
    opened0, blocks0 = t.stream_info()[1:]
    dts = timed(steps)
    opened, taken = t.stream_info()[1] - opened0, t.stream_info()[2] - blocks0
    t.streaming(False)
    if opened >= 1 and taken == steps:
        b = t.fetch(0, capi.WR_STAGE_AUDIO, n)
        assert b.size == a.size and bool((b == b).all()) and float(abs(b).max()) > 0.0
        streamed = {"value": round(n * steps / dts / 1e6, 2), "ms_per_block": round(dts / steps * 1e3, 5),
                    "times_real_time": round(n * steps / dts / c1["input_rate"], 1), "launches": opened, "blocks": taken}
    t.destroy()
    msps = n * steps / dt / 1e6
    return {
        "workload": "C1: one receiver (DDC + FM) off a 2.048 Msps RTL-SDR byte stream, %d-frame blocks resident in HBM, "
                    "D1=8 (256 kHz), D2=8 (32 kHz)" % n,
        "value": round(msps, 2), "unit": "complex Msamples/s of tuner input",
        "ms_per_block": round(dt / steps * 1e3, 5), "steps": steps,
        "times_real_time": round(msps / (c1["input_rate"] / 1e6), 1),
        "note": "launch-latency bound: ONE small launch per 64 ms block (r03: the fused demodulator + audio filter also for "
                "D2 = 8, riding in the next block's DDC launch; r02: three launches, 28.6-30.8 us)",
        "streaming": streamed,
        "streaming_note": "wr_tuner_set_streaming(1): the %d blocks ring the doorbell of one persistent launch (opened and closed "
                          "inside the timing)" % steps,
    }


def run_c5(args, torch, dist, rank, world, device_index, local_rank=0, have=1):
    """BASELINE config 5: one synthetic 1 Gsps stream, D1 = 4000, sharded IN TIME.
    Chunk c of T frames belongs to rank c mod world; a rank computes [halo | chunk] from the state
    of a stream that starts at the halo's first frame (wr_tuner_seek: closed-form NCO phase, empty
    histories) and drops the audio the empty history contaminates.  The halo -- the last
    H = 260 000 frames of the previous chunk -- comes from the ring neighbour: one send/recv pair per
    chunk (RCCL over one xGMI link; at world 1 a device copy).  A step = one chunk per rank."""
    from webradio_amd import capi, synth, timeshard
    from webradio_amd.device import Device, Tuner
    cfg = synth.C5
    fs, T = cfg["input_rate"], cfg["block_frames"]
    d1, d2 = fs // cfg["chan_rate"], cfg["chan_rate"] // cfg["audio_rate"]
    H = timeshard.halo_frames(d1, d2)
    assert H == 260_000 and T % (d1 * d2) == 0 and T >= H
    ifs = synth.c2_ifs(args.channels, cfg)
    nb = max(3, min(args.resident_blocks, 4))               # (H + T) * 8 B = 162 MB each; >= 3: a halo posted a round
                                                            # ahead lands in a buffer no chunk in flight reads
    # every buffer: [halo H | chunk T]; the chunks are consecutive pieces of one stream per rank
    bufs = [torch.empty(2 * (H + T), dtype=torch.float32, device="cuda") for _ in range(nb)]
    for b in range(nb):
        bufs[b][2 * H:] = synth.fm_stream_torch(T, fs, ifs[::4], "cuda", start_frame=(b * world + rank) * T,
                                                seed=777 + b * world + rank)
        bufs[b][:2 * H] = 0.0
    stream = torch.cuda.current_stream().cuda_stream
    dev = Device(device_index, stream)
    tuner = Tuner(dev, fs, args.channels, H + T, capi.WR_NCO_ROTATE)
    for f in ifs:
        tuner.add_receiver(f, cfg["chan_passband"], cfg["chan_rate"], capi.WR_FM, cfg["audio_passband"], cfg["audio_rate"])
    native = args.halo == "ring" or (args.halo == "auto" and world > 1 and args.backend == "nccl")
    if args.halo == "copy" and world > 1:
        raise SystemExit("bench.py: --halo copy needs --gpus 1")
    if native and world > 1 and args.backend != "nccl":
        raise SystemExit("bench.py: --halo ring between ranks needs one GPU per rank (RCCL): use the nccl backend")
    ring = None
    if native:
        ring = timeshard.RingHalo(dist, rank, world, dev=dev)        # wr_ring_* on RCCL
        tuner.mark_launches(True)                                    # the exchanges wait for the tuner's launches, not the stream
    elif world > 1:
        ring = timeshard.RingHalo(dist, rank, world)                 # torch.distributed (gloo: host tensors)
    posted = [None]
    halo_wait = [0.0, 0]                                             # seconds this rank's host spent waiting for halos, waits

    def tail_of(i):
        return bufs[i % nb][2 * T:]                                  # last H frames of round i's [halo | chunk]

    def halo_of(i):
        # the tail received in round i is the one of chunk i * world + rank - 1: this rank's halo of the same
        # round -- except on rank 0, whose predecessor chunk is rank world - 1's of the round before
        return bufs[(i + 1) % nb][:2 * H] if rank == 0 else bufs[i % nb][:2 * H]

    def step(i):
        buf = bufs[i % nb]
        c = i * world + rank                                # this rank's chunk of round i
        tail = buf[2 * T:]                                  # last H frames of [halo | chunk]
        if native:
            # the halo is input, not a result: round i + 1's pair goes out on the ring's own stream before this
            # round's chunk is submitted and travels while it computes; nothing here waits on the host
            if posted[0] is None:
                ring.post(tail_of(i), halo_of(i), tuner)
            tw = time.perf_counter()
            ring.wait()                                     # round i's pair (and rank 0's halo, which came a round earlier)
            halo_wait[0] += time.perf_counter() - tw
            halo_wait[1] += 1
            ring.post(tail_of(i + 1), halo_of(i + 1), tuner)
            posted[0] = i + 1
        elif world == 1:
            bufs[(i + 1) % nb][:2 * H].copy_(tail)          # next chunk's halo: a device copy
        else:
            # (gloo -- the two-ranks-on-one-GPU test -- moves host tensors only)
            tw = time.perf_counter()
            got = ring.exchange(tail if args.backend == "nccl" else tail.cpu())   # from rank - 1: the tail of chunk c - 1 ...
            halo_wait[0] += time.perf_counter() - tw
            halo_wait[1] += 1
            if rank == 0:
                bufs[(i + 1) % nb][:2 * H].copy_(got, non_blocking=False)    # ... which rank 0 needs one round later
            else:
                buf[:2 * H].copy_(got, non_blocking=False)
        if c == 0:
            tuner.seek(0)
            tuner.submit_device(buf[2 * H:], T)
        else:
            tuner.seek(c * T - H)
            tuner.submit_device(buf, H + T)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    done = args.warmup
    if args.settle_ms > 0:                              # clocks: see --settle-ms
        torch.cuda.synchronize()
        t_settle = time.perf_counter()
        while True:
            # Every step holds a halo exchange with the ring neighbour, so all ranks must take the SAME number of steps:
            # whether to go on is decided together (a MAX over the ranks' own clocks), never by a rank on its own.
            # (r03 let each rank look at its own clock: two ranks a millisecond apart at the 150 ms mark left one of them
            # in an exchange nobody answered -- about one run in twenty over gloo, until the collective's 30-minute
            # timeout: the hang that cost GPUTEST_r03 its time limit; profiles/r04_spawn_runs.txt caught it.)
            more = (time.perf_counter() - t_settle) * 1e3 + rank * args.settle_skew_ms < args.settle_ms
            if dist is not None:
                flag = torch.tensor([1 if more else 0], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                more = bool(flag.item())
            if not more:
                break
            for _ in range(50):
                step(done)
                done += 1
            torch.cuda.synchronize()
    tuner.profile(max(1, min(args.profile_stride, args.steps)))     # (one launch per step; a pair must close inside the timed region)
    barrier()
    t0 = time.perf_counter()
    for i in range(done, done + args.steps):
        step(i)
    tuner.flush()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0         # (the closing barrier's own latency is nobody's step: see timed_steps)
    barrier()
    launches, ddc_ms = tuner.profile_read()
    tuner.profile(False)
    ranks, distinct = gather_ranks(dist, world, rank_record(
        torch, dist, rank, local_rank, world, device_index, have, args.backend, float(T) * args.steps, elapsed,
        {"launches_timed": launches, "kernel_ms": round(ddc_ms, 5),
         "ring_exchanges": ring.native.exchanges() if native else halo_wait[1],
         "halo_wait_us_per_exchange": round(halo_wait[0] / halo_wait[1] * 1e6, 2) if halo_wait[1] else None}), args.backend)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    a = tuner.fetch(0, capi.WR_STAGE_AUDIO, H + T)
    assert a.size == (H + T) // (d1 * d2) and bool((a == a).all()) and float(abs(a).max()) > 0.0
    out = None
    if rank == 0:
        algo = 8.0 + 4.0 * args.channels / (d1 * d2)
        achieved = ((H + T) * algo / 1e9) / (ddc_ms / 1e3) if ddc_ms > 0 else 0.0
        out = {
            "metric": "complex Msamples/sec (node), 256-ch DDC+NFM demod, one 1 Gsps stream time-sharded",
            "value": round(float(T) * args.steps * world / elapsed / 1e6, 2),
            "unit": "complex Msamples/s of the one stream (whole job; halo recomputation not counted)",
            "n_gpus": world, "ranks": ranks, "devices_distinct": distinct, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "C5: ONE synthetic 1 Gsps complex-f32 stream, %d channels, D1=4000 (250 kHz, passband 64 MHz), "
                            "FM, D2=5; chunks of %d frames dealt round-robin to the ranks, halo of %d frames per chunk "
                            "from the ring neighbour (%s)" % (args.channels, T, H,
                                                              "wr_ring: RCCL ncclSend/ncclRecv on a side stream, issued a round ahead"
                                                              if native else "torch.distributed %s send/recv" % args.backend
                                                              if world > 1 else "device copy at world size 1"),
                "channels": args.channels, "chunk_frames": T, "halo_frames": H, "resident_chunks": nb,
                "halo": "ring" if native else ("copy" if world == 1 else "torch.distributed"),
                "ring_exchanges": ring.native.exchanges() if native else None,
                "nco": "rotate", "parallelism": "time sharding, ring halo exchange (RCCL), no other collective",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_tuner_ddc on [halo | chunk], ONE launch per step since r03: wr_tuner_seek launches nothing (the "
                          "DDC takes the phase in closed form and reads all-zero state sets) and the demod + audio filter of the "
                          "chunk before ride in it; kernel_ms is the mean per step over groups of consecutive steps "
                          "(WR_LAZY_SEEK=0: k_seek, k_tuner_ddc, k_tuner_post as three launches, r02's schedule)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None,
                "kernel_ms": round(ddc_ms, 5), "launches_timed": launches,
                "algorithmic_bytes_per_launch": (H + T) * algo,
                "note": "at D1 = 4000 a tap reaches 64 of every 4000 input frames and the kernel reads nothing else: "
                        "the algorithmic figure (every input byte once) can exceed what HBM could deliver",
            },
        }
    if ring is not None:
        if native:
            ring.wait()
            torch.cuda.synchronize()
        ring.close()
    tuner.destroy()
    dev.close()
    return out


def finish(out, dist, gpus):
    """The JSON line is the LAST line the job writes: RCCL prints its version banner through C stdio
    when NCCL_DEBUG=VERSION is set (it is on the GPU boxes); every rank pushes that out, the ranks
    meet, and only then does rank 0 print."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
    if out is not None:
        assert out["n_gpus"] == gpus, "the line says %d GPUs, --gpus asked for %d" % (out["n_gpus"], gpus)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if (args.gpus > 1 or args.spawn) and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                          # does not return
    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and args.backend == "nccl":
        # RCCL between processes needs dmabuf IPC on this host driver (the task's environment exports it; a launcher that
        # scrubbed the environment must not cost the job): before the HIP runtime comes up
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    from webradio_amd import capi, synth
    from webradio_amd.device import Device, Tuner

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: the line would not be an %d-GPU measurement"
                         % (args.gpus, world, args.gpus))
    have = torch.cuda.device_count()
    if have < 1:
        raise SystemExit("bench.py: no GPU (the HIP path has no CPU fallback)")
    if world > have and args.backend == "nccl":
        raise SystemExit("bench.py: %d ranks but %d GPU(s): one rank per GPU (--backend gloo lets ranks share a GPU, "
                         "for tests only)" % (world, have))
    device_index = local_rank % have
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1 or os.environ.get("WR_BENCH_RDZV_FILE"):
        import torch.distributed as dist
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"timeout": datetime.timedelta(seconds=args.rdzv_timeout)}
        if os.environ.get("WR_BENCH_RDZV_FILE"):                    # started by spawn_ranks: a file store, no port
            kw.update(init_method="file://" + os.environ["WR_BENCH_RDZV_FILE"], rank=rank, world_size=world)
        if args.backend == "nccl":
            kw["device_id"] = torch.device("cuda", device_index)
        dist.init_process_group(args.backend, **kw)
        if rank == args.fail_rank:
            raise SystemExit(3)

    if args.workload == "c5":
        finish(run_c5(args, torch, dist, rank, world, device_index, local_rank, have), dist, args.gpus)
        return

    cfg = synth.C2
    n = cfg["block_frames"]
    ifs = synth.c2_ifs(args.channels)
    # one independent tuner per GPU: its own stream of FM carriers, seed 12345 + tuner index
    # (ranks that SHARE a GPU -- the tests' gloo worlds -- take a launch per block: a streaming launch fills the device, and two
    # processes' launches on one device can each hold CUs the other waits for; one process per GPU is the deployment)
    streaming = bool(args.stream) and args.nco == "rotate" and world <= have
    B = 1 if streaming else max(1, args.blocks_per_launch)
    BMAX = max(B, 4)                            # (the secondary four-blocks-per-launch run needs the room)
    nb = max(1, args.resident_blocks)
    nb = (nb + BMAX - 1) // BMAX * BMAX         # whole launches before the resident stream wraps
    stream_iq = synth.fm_stream_torch(n * nb, cfg["input_rate"], ifs[::4], "cuda", seed=12345 + rank)
    blocks = [stream_iq[2 * n * b: 2 * n * (b + 1)] for b in range(nb)]
    stream = torch.cuda.current_stream().cuda_stream
    dev = Device(device_index, stream)
    nco = {"rotate": capi.WR_NCO_ROTATE, "split": capi.WR_NCO_SPLIT, "exact": capi.WR_NCO_EXACT}[args.nco]
    tuner = Tuner(dev, cfg["input_rate"], args.channels, n * BMAX, nco)
    for f in ifs:
        tuner.add_receiver(f, cfg["chan_passband"], cfg["chan_rate"], capi.WR_FM, cfg["audio_passband"],
                           cfg["audio_rate"])
    tuner.blocks_per_launch(B)
    tuner.streaming(streaming)

    def gpu_sync():
        # an open streaming launch ends when it is told to (wr_tuner_flush), not by itself: close it before waiting
        tuner.flush()
        torch.cuda.synchronize()

    def barrier():
        gpu_sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step = 0
    for _ in range(args.warmup):
        tuner.submit_device(blocks[step % nb], n)
        step += 1
    # untimed: the same steps until the clocks have settled (see --settle-ms)
    settle_steps = 0
    if args.settle_ms > 0:
        gpu_sync()
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for _ in range(100):
                tuner.submit_device(blocks[step % nb], n)
                step += 1
            settle_steps += 100
            gpu_sync()
    def launches_of(steps, B=B, streaming=streaming):
        if streaming:
            return (steps + capi.WR_STREAM_MAX_BLOCKS - 1) // capi.WR_STREAM_MAX_BLOCKS     # blocks one streaming launch takes at most
        # the launches `steps` consecutive steps from the start of the resident stream make: a launch
        # closes when it holds B blocks or the next block does not follow on in memory (the wrap)
        count, held = 0, 0
        for i in range(steps):
            held += 1
            if held == B or (i + 1) % nb == 0 or i + 1 == steps:
                count, held = count + 1, 0
        return count

    def timed_steps(steps, stride):
        # exactly `steps` steps between two barriers; what was held before goes out first, untimed
        tuner.flush()
        tuner.profile(stride)
        barrier()
        opened0 = tuner.stream_info()[1]
        t0 = time.perf_counter()
        for i in range(steps):
            tuner.submit_device(blocks[i % nb], n)
        # the demod + audio filter of a block ride along with the NEXT launch (wr_tuner_flush in
        # include/webradio_amd.h): have the last one's run too, inside the timed region
        tuner.flush()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0          # this rank's K steps, from the common start to its own last kernel's end ...
        barrier()                              # ... the ranks meet again, and the job's time is the MAX over them (below):
        got, ms = tuner.profile_read()         # a collective's own latency (RCCL: tens of us) is not part of anybody's steps
        tuner.profile(False)
        opened = tuner.stream_info()[1] - opened0
        #  the streaming launches the region opened are the library's count, not this script's guess
        assert opened in (0, launches_of(steps, 1, True)), \
            "the timed region opened %d streaming launches, launches_of() expected %d" % (opened, launches_of(steps, 1, True))
        return dt, got, ms

    n_launches = launches_of(args.steps)
    # (an event record between two launches costs the stream ~10 us: as few pairs as the stride allows -- at the driver's 20 steps
    # ONE pair around the five launches, recorded before the first and behind the last)
    stride = max(1, min(args.profile_stride, n_launches))
    tuner.flush()
    if tuner.stream_info()[2]:                           # (not with --warmup 0 --settle-ms 0: nothing to fetch yet)
        tuner.fetch(0, capi.WR_STAGE_AUDIO, n * B)       # a fetch looks at the launch the settling steps left closed: its count is in
    long0 = tuner.stream_long_blocks()
    elapsed, launches, ddc_ms = timed_steps(args.steps, stride)
    frames_per_launch = float(n) * args.steps / n_launches

    ranks, distinct = gather_ranks(dist, world, rank_record(torch, dist, rank, local_rank, world, device_index, have, args.backend,
                                                            float(n) * args.steps, elapsed,
                                                            {"launches_timed": launches, "kernel_ms": round(ddc_ms, 5)}),
                                   args.backend)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # sanity: the audio of a carrier channel is finite and non-trivial (nothing was skipped)
    a = tuner.fetch(0, capi.WR_STAGE_AUDIO, n * B)
    assert a.size and a.size % (n // 400 // 5) == 0 and bool((a == a).all()) and float(abs(a).max()) > 0.0
    long_blocks = tuner.stream_long_blocks() - long0     # (the fetch above looked at the closed launch: the count is in)

    # the same job the other ways the library can run it, outside the headline's timed region: a kernel launch per block
    # (r01-r04's like-for-like figure) and four held blocks per launch (r02-r04's headline: up to 120 ms of added audio latency)
    def other_mode(stream_on, bpl, steps):
        tuner.flush()
        tuner.streaming(stream_on)
        tuner.blocks_per_launch(bpl)
        for i in range(600 // bpl * bpl):        # ~25 ms of this launch shape before its timed steps
            tuner.submit_device(blocks[i % nb], n)
        nl_ = launches_of(steps, bpl, stream_on)
        dt1, got1, ms1 = timed_steps(steps, max(1, min(args.profile_stride, nl_)))
        per = float(n) * steps / max(1, nl_)       # frames per launch
        r = {"streaming": stream_on, "blocks_per_launch": bpl, "steps": steps, "ms_per_step": round(dt1 / steps * 1e3, 5),
             "value": round(float(n) * steps / dt1 / 1e6, 2), "unit": "complex Msamples/s",
             "kernel_ms": round(ms1, 5), "launches_timed": got1,
             "roofline_frac": round((per * ALGO_BYTES_PER_SAMPLE / 1e9) / (ms1 / 1e3) / HBM_PEAK_GBPS, 5) if ms1 > 0 else None}
        tuner.flush()
        tuner.streaming(streaming)
        tuner.blocks_per_launch(B)
        return r

    one = four = streamed = None
    if world == 1 and not args.no_secondary:
        k1 = min(args.steps, 96) // 4 * 4 or 4
        if streaming or B > 1:
            one = other_mode(False, 1, k1)
        if streaming or B != 4:
            four = other_mode(False, 4, k1)
        if not streaming and args.nco == "rotate":
            streamed = other_mode(True, 1, k1)

    # BASELINE config 3 off the same resident stream, outside the timed region of the headline
    c3 = c1 = fe = pw = None
    if world == 1 and not args.no_secondary:
        tuner.flush()
        torch.cuda.synchronize()
        c3 = c3_secondary(torch, dev, blocks, n, max(nb, min(args.steps, 60)), args.settle_ms)
        fe = frontend_secondary(torch, dev, tuner, blocks, n, args.steps, streaming)
        pw = power_secondary(torch, tuner, blocks, n) if streaming else None
        c1 = c1_secondary(torch, dev, 400, args.settle_ms)

    if rank == 0:
        total_samples = float(n) * args.steps * world
        value = total_samples / elapsed / 1e6
        nl = frames_per_launch                    # tuner input frames per launch
        achieved = (nl * ALGO_BYTES_PER_SAMPLE / 1e9) / (ddc_ms / 1e3) if ddc_ms > 0 else 0.0
        # lane groups x channel-rate frames x 64 taps x VALU instructions per tap
        tap_instr = ((args.channels + 63) // 64) * (nl / (cfg["input_rate"] // cfg["chan_rate"])) * 64 \
            * VALU_PER_TAP.get(args.nco, 0)
        # HBM bytes of the dominant kernel from the PMC passes of tools/profile_round.sh (they
        # cannot be collected from inside this process); null when the committed figure is
        # not for this configuration
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if args.channels == 256 and args.nco == tj.get("nco", "split"):
                if streaming and "streaming" in tj:
                    # one streaming launch's bytes grow with the blocks it takes: the committed figure is per block
                    traffic = int(round(tj["streaming"]["hbm_bytes_per_block"] * args.steps))
                elif not streaming and tj.get("frames_per_launch", n) == nl:
                    traffic = tj["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "complex Msamples/sec (node), 256-ch DDC+NFM demod",
            "value": round(value, 2),
            "unit": "complex Msamples/s of tuner input (whole job)",
            "n_gpus": world,
            "ranks": ranks,                       # every rank's device and its own rate over its own clock (rank_record)
            "devices_distinct": distinct,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle": {"ms": args.settle_ms, "steps": settle_steps,
                       "why": "untimed steps after the warm-up until the GPU clock has ramped up (an idle MI355X needs "
                              "~50 ms of continuous load); the timed region is the K steps after them"},
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "C2: %d-channel DDC+NFM off a synthetic 100 Msps complex-f32 stream, "
                            "4 000 000-frame blocks resident in HBM, D1=400 (250 kHz), FM, D2=5 (50 kHz)"
                            % args.channels,
                "channels": args.channels,
                "block_frames": n,
                "resident_blocks": nb,
                "streaming": streaming,
                "streaming_note": "wr_tuner_set_streaming(1): the K timed steps are K rings of the doorbell of ONE persistent "
                                  "launch opened by the first of them and closed by wr_tuner_flush inside the timed region "
                                  "(its ramp and tail are in `value`); every block's demod + audio filter run as soon as its "
                                  "last channel-rate frame is out -- no added latency, same bits as a launch per block "
                                  "(tests/test_gpu_stream.py)" if streaming else None,
                "post_stage_long_run_blocks": long_blocks if streaming else None,   # of the K timed blocks: demod + audio filter in runs of 5 tiles
                                                                                    # (the host was two blocks ahead); the rest -- the stream's end -- in runs of 2
                "blocks_per_launch": B,
                "blocks_per_launch_note": None if B == 1 else
                                          "wr_tuner_set_blocks_per_launch(%d): the tuner holds consecutive blocks and "
                                          "launches them as one (bit-identical audio, tests/test_gpu_ring.py); a block's "
                                          "audio is available when its launch has run, up to %d blocks (%d ms of signal) "
                                          "later than with one launch per block" % (B, B - 1, (B - 1) * 40),
                "nco": args.nco,
                "tuners_per_gpu": 1,
                "parallelism": "one tuner per GPU, no collective",
                "parity": "this mode (ROTATE) against the oracle in tests/: channel IQ <= 1e-6 on every channel (measured "
                          "<= 2.5e-7 at this size); FM audio <= 1e-5 on the channels that hold a carrier (64 of the 256 here) "
                          "and, on ALL 256 channels, within what the channel's own IQ difference allows: per demodulator "
                          "frame 1.25 * (|dz[k]|/|z[k]| + |dz[k-1]|/|z[k-1]|) / (2 pi) + 2.4e-7 cycles, through the audio "
                          "filter's |taps| (tests/fm_bound.py; test_c2_full_size_properties asserts it at this size)",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_tuner_stream (ONE persistent launch for the K steps: NCO mix + 64-tap decimating channel FIR of all channels "
                           "in 2/3 of the resident workgroups, every block's demod + audio filter in the other third)") if streaming else
                          "k_tuner_ddc (NCO mix + 64-tap decimating channel FIR, all channels; the previous block's demod + audio filter ride along in extra workgroups of the same launch)",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": traffic,
                "traffic_unit": "bytes per launch, PMC 2 x FETCH_SIZE + WRITE_SIZE (profiles/traffic.json: gfx950 tallies every 128-byte "
                                "read request at 64 bytes, whatever the load width -- profiles/r05_fetch_calibration.txt)",
                "kernel_ms": round(ddc_ms, 5),
                "launches_timed": launches,
                "timing": "HIP events on the launch stream inside the timed region, one pair around every %d consecutive "
                          "launches: the mean per launch includes the gaps between launches" % stride,
                "algorithmic_bytes_per_launch": nl * ALGO_BYTES_PER_SAMPLE,
                "frames_per_launch": nl,
                "note": "the path is VALU-bound at 256 channels, not HBM-bound (DESIGN.md 3.1): a channel-tap of the "
                        "index-exact NCO is 7 VALU instructions, and at the issue ceiling this mix reaches on gfx950 "
                        "(150 G wave-taps/s = 1.05 T wave-instructions/s with two recurrences per wave, "
                        "profiles/r03_ubench_tap.txt) the 2.56 M wave-taps of a block alone take 17.1 us: the "
                        "algorithmic 34.05 MB per block then are 1.99 TB/s = 24.9 % of the 8 TB/s roof -- the ceiling of "
                        "this formulation (29 % at the nominal 1.22 T/s); `frac` is to be read against that.  r06: the launch "
                        "runs at the package's 1400 W limit with the shader clock giving way (`power`, profiles/r06_power.txt): "
                        "what it takes per block is what it spends per block, and the channel IQ's way to memory and back is "
                        "4 us of its 28 (it was 30 before the ring of channel IQ shrank to what the Infinity Cache holds and the post stage took to longer runs)",
                "valu_ceiling_frac_of_hbm_roof": 0.249,
                # the binding resource, for context: VALU wave-instructions the DDC taps need (7 per
                # channel-tap in the ROTATE mode, 64 lanes per wave) against the rate the same
                # instruction mix reaches in isolation (profiles/r03_ubench_tap.txt: two recurrences per wave)
                "valu": {
                    "tap_wave_instr_per_launch": tap_instr,
                    "isolated_rate_wave_instr_per_s": 1.05e12,
                    "frac": round(tap_instr / 1.05e12 / (ddc_ms / 1e3), 4) if ddc_ms > 0 else None,
                    # SURVEY 8d's algorithmic flop count against the fp32 vector peak
                    "fp32_tflops": round(nl * ALGO_FLOP_PER_SAMPLE * args.channels / 256 / (ddc_ms / 1e3) / 1e12, 2) if ddc_ms > 0 else None,
                    "fp32_frac_of_vector_peak": round(nl * ALGO_FLOP_PER_SAMPLE * args.channels / 256 / (ddc_ms / 1e3) / 1e12
                                                      / FP32_VECTOR_PEAK_TFLOPS, 4) if ddc_ms > 0 else None,
                },
            },
        }
        if one is not None:
            # the same job with every 40 ms block a kernel launch of its own: the like-for-like successor of the r01 figure
            out["value_one_block_per_launch"] = one["value"]
        if world == 1 and (c3 is not None or one is not None or four is not None or streamed is not None):
            out["secondary"] = {}
            if one is not None:
                out["secondary"]["c2_one_block_per_launch"] = one
            if four is not None:
                out["secondary"]["c2_four_blocks_per_launch"] = four
            if streamed is not None:
                out["secondary"]["c2_streaming"] = streamed
            if c3 is not None:
                out["secondary"]["c3"] = c3
            if fe is not None:
                out["secondary"]["frontend"] = fe
            if c1 is not None:
                out["secondary"]["c1"] = c1
            if pw is not None:
                out["roofline"]["power"] = pw
            hf = host_fed_secondary()
            if hf is not None:
                out["secondary"]["host_fed"] = hf
        if world == 1 and not args.no_cpu_baseline:
            # the reference's own classes on the host cores where oracle/_ref is there (kind "reference"), with the
            # oracle's port beside it; the port alone (kind "port") otherwise
            port = cpu_baseline(cfg, ifs, args.cpu_blocks)
            ref = reference_cpu_baseline(args.cpu_blocks, args.channels)
            if ref:
                ref["port"] = port
                out["cpu_baseline"] = ref
            else:
                out["cpu_baseline"] = port
    else:
        out = None

    tuner.destroy()
    dev.close()
    finish(out, dist, args.gpus)


if __name__ == "__main__":
    main()
