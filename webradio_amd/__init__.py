"""webradio_amd -- MI355X (gfx950) backend for webradio's per-tuner DSP hot path.

The product is the HIP library ``webradio_amd/lib/libwebradio_amd.so`` behind the C ABI
of ``include/webradio_amd.h`` and the C++ host classes in ``webradio_amd/host/`` that keep
the reference's DspBlock operator API.  The Python modules here are plumbing for tests
and bench.py:

  capi    ctypes declarations of the C ABI (fails loudly if the library is not built)
  device  Device / Tuner / Spectrum handles over the C ABI
  synth   synthetic tuner streams (test and bench inputs)
"""
__all__ = ["capi", "device", "synth"]
