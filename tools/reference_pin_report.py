#!/usr/bin/env python3
"""Measured distances between the REAL reference blocks (oracle/_ref/libwr_ref_chain.so: the reference's own
downconverter/lowpass/demodulator/spectrumsink over the image's hipFFTW), the oracle and the HIP path, for every case
of tests/refcases.py.  Runs on the GPU box; the output is committed as profiles/r0N_reference_pin.txt.
r06: a row per case for the STREAMING launch (k_tuner_stream, the kernel bench.py times), and the spectrum distances on
the bins within 60, 70 and 80 dB of the frame's peak (what tests/refcases.py DB_TOL / DB_MASK assert)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("WEBRADIO_QUIET", "1")

import wr_oracle as o  # noqa: E402
import refcases  # noqa: E402
from webradio_amd import capi  # noqa: E402
from webradio_amd.device import Device, Tuner, Spectrum  # noqa: E402


def main():
    # the reference runs in a process of its own: hipFFTW brings the system's HIP runtime, this one uses torch's
    import subprocess
    import tempfile
    live_path = os.path.join(tempfile.mkdtemp(), "live.npz")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "make_reference_chain_golden.py"), live_path],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    live = np.load(live_path)
    dev = Device(0)
    print("reference = /root/reference/src/{dsp/dspblock,dsp/downconverter,dsp/lowpass,dsp/demodulator,io/spectrumsink}.cxx over hipFFTW")
    print("\nLowPass taps (impulse response of the running reference block): max |oracle - ref|, max |wr_lowpass_design - ref|, max |tap|")
    import ctypes as C
    for pb, rate in refcases.LOWPASS:
        want = live["taps_%d_%d" % (pb, rate)]
        ours = np.empty(64, np.float32)
        capi.load().wr_lowpass_design(pb, rate, capi.ptr(ours), None)
        print("  passband %10d rate %11d maxbin %2d: %.2e  %.2e  %.4f" % (pb, rate, o.lowpass_maxbin(pb, rate),
              np.abs(o.lowpass_design(pb, rate) - want).max(), np.abs(ours - want).max(), np.abs(want).max()))
    print("\nDownConverter::process: bit-identical words, oracle / k_mix")
    import torch
    for name, c in sorted(refcases.MIXES.items()):
        iq = refcases.mix_input(c)
        want = live["mix_" + name]
        step = o.phase_step(c["if_hz"], c["fs"])
        got, _ = o.mix(o.sin_table(), 0, step, iq)
        x = torch.from_numpy(iq).cuda()
        y = torch.empty_like(x)
        ph = C.c_uint(0)
        dev.lib.wr_mix(dev.h, capi.ptr(x), capi.ptr(y), iq.size // 2, C.byref(ph), step)
        torch.cuda.synchronize()
        print("  %-10s %d of %d / %d of %d" % (name, int((got.view(np.uint32) == want.view(np.uint32)).sum()), want.size,
                                             int((y.cpu().numpy().view(np.uint32) == want.view(np.uint32)).sum()), want.size))
    print("\nReceiver chain: max |x - ref| for channel IQ / demodulator / audio  (oracle; HIP EXACT; HIP ROTATE), max |audio|")
    for name, c in sorted(refcases.CHAINS.items()):
        iq = refcases.chain_input(c)
        w = [live["chain_%s_%s" % (name, k)] for k in ("audio", "chan", "demod")]
        rx = o.Receiver(c["fs"], c["if_hz"], c["cpb"], c["crate"], c["mode"], c["apb"], c["arate"])
        n = c["block"]
        parts = [rx.run(iq[2 * n * b: 2 * n * (b + 1)]) for b in range(c["blocks"])]
        rows = [[np.concatenate([p[i] for p in parts]) for i in range(3)]]
        for nco in (capi.WR_NCO_EXACT, capi.WR_NCO_ROTATE):
            t = Tuner(dev, c["fs"], 1, n, nco)
            ch = t.add_receiver(c["if_hz"], c["cpb"], c["crate"], c["mode"], c["apb"], c["arate"])
            t.keep_stages(capi.WR_STAGE_DEMOD)
            a, z, d = [], [], []
            for b in range(c["blocks"]):
                t.submit_host(iq[2 * n * b: 2 * n * (b + 1)])
                z.append(t.fetch(ch, capi.WR_STAGE_CHAN_IQ, 2 * n))
                d.append(t.fetch(ch, capi.WR_STAGE_DEMOD, n))
                a.append(t.fetch(ch, capi.WR_STAGE_AUDIO, n))
            t.destroy()
            rows.append([np.concatenate(a), np.concatenate(z), np.concatenate(d)])
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_gpu_reference_pin as pin
        s_audio, s_chan = pin._hip_chain_stream(dev, c, iq)
        stream_txt = "stream (k_tuner_stream, blocks of %d): last block's IQ %.1e, audio %.1e" % (
            pin._stream_cut(c), np.abs(s_chan - w[1][-s_chan.size:]).max(), np.abs(s_audio - w[0]).max())
        txt = "; ".join("%.1e / %.1e / %.1e" % (np.abs(r[1] - w[1]).max(), np.abs(r[2] - w[2]).max(), np.abs(r[0] - w[0]).max())
                        for r in rows)
        print("  %-6s %s   max|audio| %.3f\n         %s" % (name, txt, np.abs(w[0]).max(), stream_txt))
    print("\nSpectrumSink dB: max |x - ref| on bins within 60 / 70 / 80 dB of the peak (oracle; HIP), peak bin ref / oracle / HIP")
    for name, c in sorted(refcases.SPECTRA.items()):
        iq = refcases.spectrum_input(c)
        want = live["spec_" + name]
        strong = want >= want.max() - 60.0
        oo, s = o.Spectrum(c["n"]), Spectrum(dev, c["n"])
        n = c["block"]
        for b in range(c["blocks"]):
            oo.process(iq[2 * n * b: 2 * n * (b + 1)])
            s.push_host(iq[2 * n * b: 2 * n * (b + 1)])
        got = s.get_db()
        s.destroy()
        txt = []
        for mask in (60.0, 70.0, 80.0):
            strong = want >= want.max() - mask
            txt.append("%g dB: %.2e ; %.2e (%d bins)" % (mask, np.abs(oo.get() - want)[strong].max(), np.abs(got - want)[strong].max(),
                                                          int(strong.sum())))
        print("  %-10s %s   %d / %d / %d" % (name, "   ".join(txt), int(np.argmax(want)), int(np.argmax(oo.get())), int(np.argmax(got))))
    # full-size cases (live only): BASELINE configs 2, 5 (parameters) and 3
    full_path = os.path.join(os.path.dirname(live_path), "full.npz")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "make_reference_chain_golden.py"), "--full", full_path],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=400)
    full = np.load(full_path)
    print("\nFull size, the whole 256-receiver tuner on the HIP path (ROTATE) against probed receivers of the reference's own chain: max |chan IQ - ref| / max |audio - ref|")
    for name, c in sorted(refcases.FULL.items()):
        iq = refcases.full_input(c)
        ifs = refcases.full_ifs(c)
        n = c["block"]
        t = Tuner(dev, c["fs"], c["channels"], n, capi.WR_NCO_ROTATE)
        chans = [t.add_receiver(f, c["cpb"], c["crate"], refcases.full_mode(c, i), c["apb"], c["arate"]) for i, f in enumerate(ifs)]
        got = {ch: ([], []) for ch in c["probe"]}
        for b in range(c["blocks"]):
            t.submit_host(iq[2 * n * b: 2 * n * (b + 1)])
            for ch in c["probe"]:
                got[ch][0].append(t.fetch(chans[ch], capi.WR_STAGE_CHAN_IQ, 2 * n))
                got[ch][1].append(t.fetch(chans[ch], capi.WR_STAGE_AUDIO, n))
        t.destroy()
        print("  %-10s %d frames x %d block(s): " % (name, n, c["blocks"]) + "; ".join(
            "rx %d: %.1e / %.1e" % (ch, np.abs(np.concatenate(got[ch][0]) - full["full_%s_%d_chan" % (name, ch)]).max(),
                                    np.abs(np.concatenate(got[ch][1]) - full["full_%s_%d_audio" % (name, ch)]).max()) for ch in c["probe"]))
        # the same through ONE streaming launch (k_tuner_stream<5, 2> at C2: bench.py's headline kernel and configuration)
        t = Tuner(dev, c["fs"], c["channels"], n, capi.WR_NCO_ROTATE)
        chans = [t.add_receiver(f, c["cpb"], c["crate"], refcases.full_mode(c, i), c["apb"], c["arate"]) for i, f in enumerate(ifs)]
        xd = torch.from_numpy(iq).cuda()
        torch.cuda.synchronize()
        t.audio_ring(c["blocks"])
        t.streaming(True)
        for b in range(c["blocks"]):
            t.submit_device(xd[2 * n * b: 2 * n * (b + 1)], n)
        info = t.stream_info()
        t.flush()
        sa = {ch: [] for ch in c["probe"]}
        for b in range(c["blocks"]):
            a, seq = t.ring_acquire()
            for ch in c["probe"]:
                sa[ch].append(a[t.slot(chans[ch])].copy())
            t.ring_release()
        sz = {ch: t.fetch(chans[ch], capi.WR_STAGE_CHAN_IQ, 2 * n) for ch in c["probe"]}
        t.destroy()
        del xd
        print("  %-10s   streaming launch (live %s, launches %d, blocks %d; last block's IQ / every block's audio): " % ("", info[0], info[1], info[2])
              + "; ".join("rx %d: %.1e / %.1e" % (ch, np.abs(sz[ch] - full["full_%s_%d_chan" % (name, ch)][-sz[ch].size:]).max(),
                                                  np.abs(np.concatenate(sa[ch]) - full["full_%s_%d_audio" % (name, ch)]).max()) for ch in c["probe"]))
    c3 = refcases.C3_FULL
    c = refcases.FULL[c3["case"]]
    iq = refcases.full_input(c)
    nrows = (c["block"] - c3["n"]) // c3["hop"] + 1
    x = torch.from_numpy(iq).cuda()
    out = torch.empty(nrows * c3["n"], dtype=torch.float32, device="cuda")
    s = Spectrum(dev, c3["n"], c3["hop"])
    s.batch_db(x, nrows, out)
    torch.cuda.synchronize()
    rows = out.cpu().numpy().reshape(nrows, c3["n"])
    s.destroy()
    for mask in (60.0, 70.0, 80.0):
        print("  C3 waterfall, %d rows of 65536 points, rows %s against the reference's SpectrumSink: max |dB - ref| on bins within %g dB of the peak: %s" % (
            nrows, c3["rows"], mask, ", ".join("%.1e (%d)" % (np.abs(rows[r] - full["c3_row_%d" % r])[full["c3_row_%d" % r] >= full["c3_row_%d" % r].max() - mask].max(),
                                                        int((full["c3_row_%d" % r] >= full["c3_row_%d" % r].max() - mask).sum())) for r in c3["rows"])))
    dev.close()


if __name__ == "__main__":
    main()
