#!/usr/bin/env python3
"""Generates tests/golden/reference_c1.npz: BASELINE config 1's recorded RTL-SDR capture through the REAL reference
Receiver chain -- io/rtlsdrtuner.cxx:106's (u8 - 128) / 128, then dsp/downconverter.cxx -> dsp/lowpass.cxx ->
dsp/demodulator.cxx (FM) -> dsp/lowpass.cxx wired as radio.cxx:68-83 -- run through oracle/_ref/libwr_ref_chain.so
(oracle/ref_chain.cxx; its FFTW3 API is the image's hipFFTW, so this runs ON THE GPU BOX):

    gpurun -- 'python tests/golden/make_c1_reference_golden.py gpurun_out/reference_c1.npz'

and the file it writes is copied to tests/golden/reference_c1.npz and committed.  Only data is stored: the capture's
bytes (synth.rtl_u8_stream(4 * 16384): the same 131 072 bytes oracle_selfcheck_c1.npz held), the block size, and the reference's channel IQ, demodulator
output and audio.  Until r05 the C1 tests compared the HIP path with the ORACLE's outputs for this capture
(oracle_selfcheck_c1.npz); they compare it with the reference's now (tests/test_gpu_tuner.py, tests/test_gpu_host.py),
and tests/test_oracle_reference_chain.py holds the oracle to the same vectors on the CPU.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import wr_oracle as o  # noqa: E402
from webradio_amd import synth  # noqa: E402


def main(path):
    if o.ref_chain() is None:
        print("oracle/_ref/libwr_ref_chain.so has not been built (make -C oracle ref_chain, where /root/reference is)")
        return 1
    n = 16384
    u8 = synth.rtl_u8_stream(4 * n)                               # the capture: seeded, the same bytes since r01
    if os.path.exists(os.path.join(HERE, "reference_c1.npz")):
        assert np.array_equal(np.load(os.path.join(HERE, "reference_c1.npz"))["u8"], u8), "the capture changed"
    c1 = synth.C1
    iq = o.u8_to_float(u8)                                        # io/rtlsdrtuner.cxx:106
    audio, chan, dem = o.ref_receiver(c1["input_rate"], c1["if_hz"], c1["chan_passband"], c1["chan_rate"], o.FM,
                                      c1["audio_passband"], c1["audio_rate"], iq, n)
    np.savez_compressed(path, u8=u8, block_frames=np.int64(n), chan_iq=chan, demod=dem, audio=audio)
    print("wrote", path, "chan", chan.shape, "demod", dem.shape, "audio", audio.shape, os.path.getsize(path), "bytes")
    # how far the oracle is from it (reported, not asserted here: tests/test_oracle_reference_chain.py asserts)
    rx = o.Receiver(c1["input_rate"], c1["if_hz"], c1["chan_passband"], c1["chan_rate"], o.FM, c1["audio_passband"],
                    c1["audio_rate"])
    oa, oc, od = [], [], []
    for b in range(iq.size // 2 // n):
        a, c, d = rx.run(iq[2 * n * b: 2 * n * (b + 1)])
        oa.append(a); oc.append(c); od.append(d)
    print("oracle vs reference: chan %.3g demod %.3g audio %.3g" % (np.abs(np.concatenate(oc) - chan).max(),
                                                                     np.abs(np.concatenate(od) - dem).max(),
                                                                     np.abs(np.concatenate(oa) - audio).max()))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "reference_c1.npz")))
