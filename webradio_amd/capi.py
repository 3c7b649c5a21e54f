"""ctypes binding of the product's C ABI (include/webradio_amd.h).

This module is plumbing for tests and bench.py: it loads
``webradio_amd/lib/libwebradio_amd.so`` (built by ``__graft_entry__.build()`` /
``make -C webradio_amd/csrc``) and declares every entry point.  It fails loudly if the
library is missing -- there is no Python or CPU implementation behind it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwebradio_amd.so")

WR_OK, WR_ERR_ARG, WR_ERR_HIP, WR_ERR_STATE, WR_ERR_NOMEM, WR_ERR_NODEV, WR_ERR_RATE = range(7)
WR_AM, WR_FM, WR_USB, WR_LSB = range(4)
WR_STAGE_CHAN_IQ, WR_STAGE_DEMOD, WR_STAGE_AUDIO = 1, 2, 3
WR_NCO_SPLIT, WR_NCO_EXACT, WR_NCO_ROTATE = 0, 1, 2
WR_HOST, WR_DEVICE = 0, 1
WR_ABI_VERSION = 6            # include/webradio_amd.h
WR_STREAM_MAX_BLOCKS = 512    # blocks one streaming launch takes (include/webradio_amd.h)
WR_FIR_LENGTH = 64
WR_TABLE_SIZE = 65536

_fp = C.POINTER(C.c_float)
_vp = C.c_void_p
_u32 = C.c_uint
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/webradio_amd.h one to one
SIGNATURES = {
    "wr_abi_version": (C.c_int, []),
    "wr_last_error": (C.c_char_p, []),
    "wr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "wr_tune": (C.c_int, [C.c_int, C.c_long, C.POINTER(C.c_long)]),
    "wr_phase_step": (C.c_int, [C.c_int, _u32, C.POINTER(C.c_int)]),
    "wr_sin_table": (C.c_int, [_vp]),
    "wr_lowpass_design": (C.c_int, [_u32, _u32, _vp, C.POINTER(_u32)]),
    "wr_lowpass_design_n": (C.c_int, [_u32, _u32, _u32, _vp, C.POINTER(_u32)]),
    "wr_spectrum_window": (C.c_int, [_u32, _vp]),
    "wr_dev_open": (C.c_int, [C.POINTER(_vp), C.c_int, _vp]),
    "wr_dev_close": (C.c_int, [_vp]),
    "wr_dev_sync": (C.c_int, [_vp]),
    "wr_dev_stream": (_vp, [_vp]),
    "wr_dev_malloc": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "wr_dev_free": (C.c_int, [_vp, _vp]),
    "wr_dev_upload": (C.c_int, [_vp, _vp, _vp, _sz]),
    "wr_dev_download": (C.c_int, [_vp, _vp, _vp, _sz]),
    "wr_dev_host_register": (C.c_int, [_vp, _vp, _sz]),
    "wr_dev_host_unregister": (C.c_int, [_vp, _vp]),
    "wr_dev_upload_async": (C.c_int, [_vp, _vp, _vp, _sz]),
    "wr_dev_wait_uploads": (C.c_int, [_vp]),
    "wr_dev_wait_uploads_but": (C.c_int, [_vp, C.c_uint]),
    "wr_dev_upload_ahead": (C.c_int, [_vp, _vp, _vp, _sz]),
    "wr_mix": (C.c_int, [_vp, _vp, _vp, _sz, C.POINTER(_u32), C.c_int]),
    "wr_fir_decimate": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _vp, _vp, _vp]),
    "wr_fir_decimate_n": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _u32, _vp, _vp, _vp]),
    "wr_demod": (C.c_int, [_vp, C.c_int, _vp, _sz, _vp, _vp]),
    "wr_u8_to_f32": (C.c_int, [_vp, _vp, _vp, _sz]),
    "wr_stage_windows_from_host": (C.c_int, [_vp, _vp, C.c_int, _vp, _sz, _u32, _u32, _sz]),
    "wr_u8_to_f32_from_host": (C.c_int, [_vp, _vp, _vp, _sz]),
    "wr_tuner_create": (C.c_int, [C.POINTER(_vp), _vp, _u32, _u32, _sz, C.c_int]),
    "wr_tuner_destroy": (C.c_int, [_vp]),
    "wr_chan_add": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "wr_chan_remove": (C.c_int, [_vp, C.c_int]),
    "wr_chan_count": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "wr_chan_set_if": (C.c_int, [_vp, C.c_int, C.c_int]),
    "wr_chan_set_filter": (C.c_int, [_vp, C.c_int, C.c_int, _u32, _u32]),
    "wr_chan_set_taps": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _u32]),
    "wr_chan_set_filter_n": (C.c_int, [_vp, C.c_int, C.c_int, _u32, _u32, _u32]),
    "wr_chan_set_taps_n": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _u32, _u32]),
    "wr_chan_set_mode": (C.c_int, [_vp, C.c_int, C.c_int]),
    "wr_chan_set_af_gain": (C.c_int, [_vp, C.c_int, C.c_float]),
    "wr_chan_set_squelch": (C.c_int, [_vp, C.c_int, C.c_float, C.c_int]),
    "wr_tuner_keep_stages": (C.c_int, [_vp, _u32]),
    "wr_tuner_flush": (C.c_int, [_vp]),
    "wr_tuner_set_blocks_per_launch": (C.c_int, [_vp, _u32]),
    "wr_tuner_set_streaming": (C.c_int, [_vp, C.c_int]),
    "wr_tuner_stream_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "wr_tuner_stream_host_blocks": (C.c_int, [_vp, C.POINTER(C.c_ulonglong)]),
    "wr_tuner_stream_long_blocks": (C.c_int, [_vp, C.POINTER(C.c_ulonglong)]),
    "wr_tuner_seek": (C.c_int, [_vp, C.c_ulonglong]),
    "wr_tuner_audio_ring": (C.c_int, [_vp, _u32]),
    "wr_tuner_audio_ring_acquire": (C.c_int, [_vp, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t),
                                              C.POINTER(C.c_size_t), C.POINTER(_u32), C.POINTER(C.c_ulonglong)]),
    "wr_tuner_audio_ring_release": (C.c_int, [_vp]),
    "wr_tuner_audio_ring_ready": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "wr_tuner_audio_ring_stats": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(C.c_ulonglong)]),
    "wr_tuner_submit_count": (C.c_int, [_vp, C.POINTER(C.c_ulonglong)]),
    "wr_chan_get_state": (C.c_int, [_vp, C.c_int, C.POINTER(_u32), _vp]),
    "wr_chan_set_state": (C.c_int, [_vp, C.c_int, _u32, _vp]),
    "wr_tuner_submit": (C.c_int, [_vp, _vp, _sz, C.c_int]),
    "wr_tuner_last_staging": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "wr_block_kernel_calls": (C.c_ulonglong, []),
    "wr_tuner_submit_u8": (C.c_int, [_vp, _vp, _sz, C.c_int]),
    "wr_chan_fetch": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "wr_tuner_audio_dev": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_sz)]),
    "wr_chan_slot": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "wr_tuner_fetch_audio_all": (C.c_int, [_vp, _vp, _sz, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_u32)]),
    "wr_chan_reset_history": (C.c_int, [_vp, C.c_int]),
    "wr_ring_id_bytes": (C.c_int, []),
    "wr_ring_version": (C.c_int, [C.POINTER(C.c_int)]),
    "wr_ring_make_id": (C.c_int, [_vp, _sz]),
    "wr_ring_create": (C.c_int, [C.POINTER(_vp), _vp, _vp, _sz, C.c_int, C.c_int]),
    "wr_ring_exchange": (C.c_int, [_vp, _vp, _vp, _sz]),
    "wr_ring_exchange_after": (C.c_int, [_vp, _vp, _vp, _vp, _sz]),
    "wr_tuner_mark_launches": (C.c_int, [_vp, C.c_int]),
    "wr_ring_wait": (C.c_int, [_vp]),
    "wr_ring_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_ulonglong)]),
    "wr_ring_destroy": (C.c_int, [_vp]),
    "wr_tuner_set_audio_scale": (C.c_int, [_vp, C.c_float]),
    "wr_spectrum_get_waterfall_row": (C.c_int, [_vp, _u32, C.c_int, _vp, _vp]),
    "wr_tuner_profile": (C.c_int, [_vp, C.c_int]),
    "wr_tuner_profile_read": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(C.c_double)]),
    "wr_spectrum_create": (C.c_int, [C.POINTER(_vp), _vp, _u32, _u32]),
    "wr_spectrum_destroy": (C.c_int, [_vp]),
    "wr_spectrum_push": (C.c_int, [_vp, _vp, _sz, C.c_int]),
    "wr_spectrum_lazy_info": (C.c_int, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "wr_spectrum_get_db": (C.c_int, [_vp, _vp]),
    "wr_spectrum_get_bins": (C.c_int, [_vp, _vp]),
    "wr_spectrum_frames_done": (C.c_int, [_vp, C.POINTER(C.c_ulong)]),
    "wr_spectrum_batch_db": (C.c_int, [_vp, _vp, _sz, _vp]),
}


class WrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("webradio_amd error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load the C-ABI library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIB_PATH)
    try:
        # torch bundles its own libamdhip64.so.7; importing it first makes this library
        # bind to the same HIP runtime instead of loading a second copy from /opt/rocm
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != WR_OK:
        raise WrError(rc, load().wr_last_error().decode(errors="replace"))
    return rc


def ptr(x):
    """void* of a numpy array, a torch tensor, an int address or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)
