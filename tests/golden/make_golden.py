#!/usr/bin/env python3
"""Generates the committed golden fixtures.  Run in the build container (where
/root/reference exists): `python tests/golden/make_golden.py`.

  demod_reference.npz   inputs + outputs of the REAL reference Demodulator
                        (dsp/demodulator.cxx driven through DspSource::run by
                        oracle/ref_harness.cxx), all four modes, 8 blocks of 512 frames.
  dspblock_traces.json  scheduling traces of the REAL reference DspBlock runtime
                        (dsp/dspblock.cxx) for the scenarios of oracle/ref_harness.cxx.
(r01-r04 also wrote oracle_selfcheck_c1.npz here -- the ORACLE's outputs for BASELINE config 1's capture, a
self-check.  r05: the capture goes through the REAL reference chain instead, on the GPU box:
tests/golden/make_c1_reference_golden.py -> reference_c1.npz; the capture itself is synth.rtl_u8_stream(4 * 16384).)
Only data is written; no reference source text is stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import wr_oracle as o  # noqa: E402
from webradio_amd import synth  # noqa: E402


def main():
    R = o.ref()
    if R is None:
        print("no /root/reference: nothing generated")
        return 1
    rng = np.random.default_rng(20260928)
    iq = rng.standard_normal(2 * 4096).astype(np.float32)
    iq[:16] = 0.0
    iq[32:36] = [-0.0, 1.0, 0.0, -1.0]
    out = {"iq": iq}
    for m in ("AM", "FM", "USB", "LSB"):
        out["out_" + m] = o.ref_demod(m, iq, 512)
    np.savez_compressed(os.path.join(HERE, "demod_reference.npz"), **out)

    traces = [o.harness_trace(R, i) for i in range(R.wr_harness_scenarios())]
    json.dump(traces, open(os.path.join(HERE, "dspblock_traces.json"), "w"), indent=0)

    print("golden fixtures written to", HERE)
    return 0


if __name__ == "__main__":
    sys.exit(main())
