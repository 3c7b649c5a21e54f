"""Thin Python handles over the C ABI: Device, Tuner (one FrontEnd's tuner with its
Receivers), Spectrum (SpectrumSink).  Used by tests, bench.py and the examples; the
arithmetic all happens in the HIP library.  Names follow the reference (radio.h:42-102).
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import check, ptr


class Device:
    """wr_dev: a gfx950 device + the HIP stream all work is issued on."""

    def __init__(self, index=0, stream=None):
        self.lib = capi.load()
        h = C.c_void_p()
        check(self.lib.wr_dev_open(C.byref(h), index, ptr(stream) if stream else None))
        self.h = h
        self.index = index

    def close(self):
        if self.h:
            self.lib.wr_dev_close(self.h)
            self.h = None

    def sync(self):
        check(self.lib.wr_dev_sync(self.h))

    # --- raw device memory for tests that do not want torch --------------------
    def malloc(self, nbytes):
        p = C.c_void_p()
        check(self.lib.wr_dev_malloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, p):
        check(self.lib.wr_dev_free(self.h, C.c_void_p(p)))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(max(arr.nbytes, 4))
        check(self.lib.wr_dev_upload(self.h, C.c_void_p(p), ptr(arr), arr.nbytes))
        return p

    def download(self, p, count, dtype=np.float32):
        out = np.empty(count, dtype=dtype)
        check(self.lib.wr_dev_download(self.h, ptr(out), C.c_void_p(p), out.nbytes))
        return out

    # --- one kernel per reference block -----------------------------------------
    def mix(self, iq, phase, phase_step):
        """DownConverter::process on a host array; returns (mixed, new_phase)."""
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        n = iq.size // 2
        din = self.upload(iq)
        dout = self.malloc(max(iq.nbytes, 4))
        ph = C.c_uint(phase)
        check(self.lib.wr_mix(self.h, C.c_void_p(din), C.c_void_p(dout), n, C.byref(ph), phase_step))
        out = self.download(dout, 2 * n)
        self.free(din)
        self.free(dout)
        return out, ph.value

    def fir_decimate(self, x, channels, decimation, coeff, history_dev):
        """LowPass::process on a host array; history_dev is a device pointer kept by the caller."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        nframes = x.size // channels
        nout = (nframes // decimation) * channels
        din = self.upload(x)
        dout = self.malloc(max(nout * 4, 4))
        coeff = np.ascontiguousarray(coeff, dtype=np.float32)     # its length is LowPass::_firLength
        check(self.lib.wr_fir_decimate_n(self.h, C.c_void_p(din), nframes, channels, decimation, coeff.size,
                                         ptr(coeff), C.c_void_p(history_dev), C.c_void_p(dout)))
        out = self.download(dout, nout)
        self.free(din)
        self.free(dout)
        return out

    def demod(self, mode, iq, prev):
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        n = iq.size // 2
        din = self.upload(iq)
        dout = self.malloc(max(n * 4, 4))
        prev = np.array(prev, dtype=np.float32)
        check(self.lib.wr_demod(self.h, mode, C.c_void_p(din), n, ptr(prev), C.c_void_p(dout)))
        out = self.download(dout, n)
        self.free(din)
        self.free(dout)
        return out, prev


class Tuner:
    """wr_tuner: every Receiver chain of one tuner, evaluated by one launch sequence."""

    def __init__(self, dev, input_rate, max_channels, max_block_frames, nco=capi.WR_NCO_ROTATE):
        self.dev = dev
        self.lib = dev.lib
        h = C.c_void_p()
        check(self.lib.wr_tuner_create(C.byref(h), dev.h, input_rate, max_channels, max_block_frames, nco))
        self.h = h
        self.input_rate = input_rate

    def destroy(self):
        if self.h:
            self.lib.wr_tuner_destroy(self.h)
            self.h = None

    def add_receiver(self, if_hz, chan_passband, chan_rate, mode, audio_passband, audio_rate, fir_lengths=None,
                     stage2=None):
        """Receiver() + setFrontEnd(): the wiring of radio.cxx:62-90 with explicit parameters.
        fir_lengths: (channel, audio) LowPass::_firLength, powers of two: the channel filter up to 256
        (WR_FIR_FUSED_MAX; above 64: WR_NCO_EXACT the reference's own arithmetic, the other modes the ROTATE recurrence in L / 64 segments),
        the audio filter up to 256 as well (r05; default 64, 64).
        stage2: (fir_length, passband, out_rate) of a second channel LowPass in front of the demodulator; fir_length up to 256."""
        c = C.c_int()
        check(self.lib.wr_chan_add(self.h, C.byref(c)))
        ch = c.value
        check(self.lib.wr_chan_set_if(self.h, ch, if_hz))
        l1, l2 = fir_lengths if fir_lengths else (capi.WR_FIR_LENGTH, capi.WR_FIR_LENGTH)
        check(self.lib.wr_chan_set_filter_n(self.h, ch, 0, l1, chan_passband, chan_rate))
        if stage2:
            check(self.lib.wr_chan_set_filter_n(self.h, ch, 2, stage2[0], stage2[1], stage2[2]))
        check(self.lib.wr_chan_set_filter_n(self.h, ch, 1, l2, audio_passband, audio_rate))
        check(self.lib.wr_chan_set_mode(self.h, ch, mode))
        return ch

    def keep_stages(self, *stages):
        """Ask for intermediate stages (capi.WR_STAGE_DEMOD ...) to be kept for fetch()."""
        mask = 0
        for st in stages:
            mask |= 1 << st
        check(self.lib.wr_tuner_keep_stages(self.h, mask))

    def flush(self):
        """Launch a post stage (demod + audio filter) that is still waiting for the next submit."""
        check(self.lib.wr_tuner_flush(self.h))

    def blocks_per_launch(self, n):
        """Hold up to n back-to-back device blocks and launch them as one (the same bits, per group)."""
        check(self.lib.wr_tuner_set_blocks_per_launch(self.h, n))

    def streaming(self, enable=True):
        """Device blocks go to ONE persistent launch through a doorbell (wr_tuner_set_streaming): the same bits,
        no kernel launch per block.  flush() -- or anything else that touches the tuner -- closes the launch."""
        check(self.lib.wr_tuner_set_streaming(self.h, int(enable) if enable is not True else 1))

    def stream_info(self):
        """(live, launches opened, blocks taken) of the streaming mode"""
        live, launches, blocks = C.c_int(), C.c_ulonglong(), C.c_ulonglong()
        check(self.lib.wr_tuner_stream_info(self.h, C.byref(live), C.byref(launches), C.byref(blocks)))
        return bool(live.value), launches.value, blocks.value

    def stream_host_blocks(self):
        """blocks streamed out of page-locked host memory so far (wr_tuner_submit_u8(..., WR_HOST) while streaming)"""
        n = C.c_ulonglong()
        check(self.lib.wr_tuner_stream_host_blocks(self.h, C.byref(n)))
        return n.value

    def stream_long_blocks(self):
        """blocks (of launches that are over and checked) whose post stage ran in long runs of tiles: the host was ahead"""
        n = C.c_ulonglong()
        check(self.lib.wr_tuner_stream_long_blocks(self.h, C.byref(n)))
        return n.value

    def mark_launches(self, enable=True):
        """every launch that reads a submitted block stamps an event on completion (Ring.exchange_after waits for it)"""
        check(self.lib.wr_tuner_mark_launches(self.h, 1 if enable else 0))

    def seek(self, frame):
        """Every channel as if the stream started at `frame` (NCO phase closed-form): time sharding."""
        check(self.lib.wr_tuner_seek(self.h, C.c_ulonglong(frame)))

    def fetch_audio_all(self):
        """The last block's audio of every slot in one transfer: array [slots][frames]."""
        stride, frames, slots = C.c_size_t(), C.c_size_t(), C.c_uint()
        self.lib.wr_tuner_fetch_audio_all(self.h, None, 0, C.byref(stride), C.byref(frames), C.byref(slots))
        out = np.empty(max(1, slots.value * frames.value), dtype=np.float32)
        check(self.lib.wr_tuner_fetch_audio_all(self.h, ptr(out), out.size, C.byref(stride), C.byref(frames),
                                                C.byref(slots)))
        return out[: slots.value * frames.value].reshape(slots.value, frames.value)

    def slot(self, ch):
        s = C.c_int()
        check(self.lib.wr_chan_slot(self.h, ch, C.byref(s)))
        return s.value

    def audio_ring(self, depth):
        """Pinned host ring for the audio of every submit (0 = off)."""
        check(self.lib.wr_tuner_audio_ring(self.h, depth))

    def ring_acquire(self):
        """Oldest queued block: (audio[slots][frames] copy, seq).  Call ring_release() after."""
        p = C.POINTER(C.c_float)()
        stride, frames, slots, seq = C.c_size_t(), C.c_size_t(), C.c_uint(), C.c_ulonglong()
        check(self.lib.wr_tuner_audio_ring_acquire(self.h, C.byref(p), C.byref(stride), C.byref(frames),
                                                   C.byref(slots), C.byref(seq)))
        n = slots.value * stride.value
        a = np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.float32)
        return a.reshape(slots.value, stride.value)[:, : frames.value], seq.value

    def ring_release(self):
        check(self.lib.wr_tuner_audio_ring_release(self.h))

    def ring_stats(self):
        q, o = C.c_uint(), C.c_ulonglong()
        check(self.lib.wr_tuner_audio_ring_stats(self.h, C.byref(q), C.byref(o)))
        return q.value, o.value

    def submit_count(self):
        """The number the next submit's ring entry will carry (submits numbered so far, failed ones too)."""
        n = C.c_ulonglong()
        check(self.lib.wr_tuner_submit_count(self.h, C.byref(n)))
        return n.value

    def remove_receiver(self, ch):
        check(self.lib.wr_chan_remove(self.h, ch))

    def set_if(self, ch, if_hz):
        check(self.lib.wr_chan_set_if(self.h, ch, if_hz))

    def set_af_gain(self, ch, gain_db):
        check(self.lib.wr_chan_set_af_gain(self.h, ch, C.c_float(gain_db)))

    def set_squelch(self, ch, threshold_dbfs, enable=True):
        check(self.lib.wr_chan_set_squelch(self.h, ch, C.c_float(threshold_dbfs), 1 if enable else 0))

    def set_mode(self, ch, mode):
        check(self.lib.wr_chan_set_mode(self.h, ch, mode))

    def set_filter(self, ch, stage, passband, out_rate):
        check(self.lib.wr_chan_set_filter(self.h, ch, stage, passband, out_rate))

    def submit_host(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        check(self.lib.wr_tuner_submit(self.h, ptr(iq), iq.size // 2, capi.WR_HOST))
        self.dev.sync()   # the host array may be released by the caller

    def submit_device(self, dev_ptr, nframes):
        check(self.lib.wr_tuner_submit(self.h, ptr(dev_ptr), nframes, capi.WR_DEVICE))

    def submit_u8_host(self, raw):
        """A block in the RTL-SDR byte format in HOST memory (a numpy uint8 array, 2 bytes per frame)."""
        check(self.lib.wr_tuner_submit_u8(self.h, raw.ctypes.data_as(C.c_void_p), raw.size // 2, capi.WR_HOST))

    def last_staging(self):
        how = C.c_int()
        check(self.lib.wr_tuner_last_staging(self.h, C.byref(how)))
        return how.value

    def submit_u8_device(self, dev_ptr, nframes):
        """A block in the RTL-SDR byte format (io/rtlsdrtuner.cxx:106), already in device memory."""
        check(self.lib.wr_tuner_submit_u8(self.h, ptr(dev_ptr), nframes, capi.WR_DEVICE))

    def fetch(self, ch, stage, capacity):
        out = np.empty(max(capacity, 1), dtype=np.float32)
        n = C.c_size_t()
        check(self.lib.wr_chan_fetch(self.h, ch, stage, ptr(out), capacity, C.byref(n)))
        return out[: n.value].copy()

    def state(self, ch):
        ph = C.c_uint()
        prev = np.zeros(2, dtype=np.float32)
        check(self.lib.wr_chan_get_state(self.h, ch, C.byref(ph), ptr(prev)))
        return ph.value, prev

    def set_state(self, ch, phase, prev=None):
        p = None if prev is None else np.ascontiguousarray(prev, dtype=np.float32)
        check(self.lib.wr_chan_set_state(self.h, ch, phase, ptr(p)))

    def profile(self, enable):
        """False/0: off, True/1: every submit, n > 1: every n-th submit"""
        check(self.lib.wr_tuner_profile(self.h, int(enable)))

    def profile_read(self):
        """(launches, mean milliseconds) of the dominant kernel since the last read"""
        n = C.c_uint()
        ms = C.c_double()
        check(self.lib.wr_tuner_profile_read(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def audio_dev(self):
        a = C.c_void_p()
        stride = C.c_size_t()
        frames = C.c_size_t()
        check(self.lib.wr_tuner_audio_dev(self.h, C.byref(a), C.byref(stride), C.byref(frames)))
        return a.value, stride.value, frames.value


class Spectrum:
    """wr_spectrum: SpectrumSink (io/spectrumsink.h:44-68)."""

    def __init__(self, dev, fft_size, hop=0):
        self.dev = dev
        self.lib = dev.lib
        self.n = fft_size
        h = C.c_void_p()
        check(self.lib.wr_spectrum_create(C.byref(h), dev.h, fft_size, hop))
        self.h = h

    def destroy(self):
        if self.h:
            self.lib.wr_spectrum_destroy(self.h)
            self.h = None

    def push_host(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        check(self.lib.wr_spectrum_push(self.h, ptr(iq), iq.size // 2, capi.WR_HOST))

    def push_device(self, dev_ptr, nframes):
        check(self.lib.wr_spectrum_push(self.h, ptr(dev_ptr), nframes, capi.WR_DEVICE))

    def get_db(self):
        out = np.empty(self.n, dtype=np.float32)
        check(self.lib.wr_spectrum_get_db(self.h, ptr(out)))
        return out

    def get_bins(self):
        out = np.empty(2 * self.n, dtype=np.float32)
        check(self.lib.wr_spectrum_get_bins(self.h, ptr(out)))
        return out

    def waterfall_row(self, width, hold=0):
        db = np.empty(width, dtype=np.float32)
        pal = np.empty(width, dtype=np.uint8)
        check(self.lib.wr_spectrum_get_waterfall_row(self.h, width, hold, ptr(db), ptr(pal)))
        return db, pal

    def frames_done(self):
        n = C.c_ulong()
        check(self.lib.wr_spectrum_frames_done(self.h, C.byref(n)))
        return n.value

    def lazy_info(self):
        """(pushes kept for later because a streaming launch was open, frames transformed on demand)"""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        check(self.lib.wr_spectrum_lazy_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def batch_db(self, iq_dev, nframes_fft, db_dev):
        check(self.lib.wr_spectrum_batch_db(self.h, ptr(iq_dev), nframes_fft, ptr(db_dev)))


class Ring:
    """wr_ring: the halo ring of a time-sharded stream (BASELINE config 5) on RCCL -- one
    ncclSend / ncclRecv pair per exchange, on its own stream, handed to the device's stream by
    events.  `id_bytes`: what rank 0's Ring.make_id() returned, brought to every rank by the host."""

    @staticmethod
    def make_id():
        lib = capi.load()
        buf = (C.c_ubyte * lib.wr_ring_id_bytes())()
        check(lib.wr_ring_make_id(buf, len(buf)))
        return bytes(buf)

    @staticmethod
    def rccl_version():
        v = C.c_int()
        check(capi.load().wr_ring_version(C.byref(v)))
        return v.value

    def __init__(self, dev, id_bytes, rank, world):
        self.lib = capi.load()
        h = C.c_void_p()
        buf = (C.c_ubyte * len(id_bytes)).from_buffer_copy(id_bytes)
        check(self.lib.wr_ring_create(C.byref(h), dev.h, buf, len(id_bytes), rank, world))
        self.h, self.rank, self.world = h, rank, world

    def exchange(self, send_dev, recv_dev, nfloats):
        """enqueue: send_dev -> rank + 1, recv_dev <- rank - 1 (device pointers or torch tensors)"""
        check(self.lib.wr_ring_exchange(self.h, ptr(send_dev), ptr(recv_dev), nfloats))

    def exchange_after(self, tuner, send_dev, recv_dev, nfloats):
        """the same, ordered behind the tuner's launches so far (Tuner.mark_launches(True)) instead of behind the
        device's stream: for a halo that is input, received into a buffer only the tuner reads"""
        check(self.lib.wr_ring_exchange_after(self.h, tuner.h, ptr(send_dev), ptr(recv_dev), nfloats))

    def wait(self):
        """the device's stream waits for the last exchange (no host wait)"""
        check(self.lib.wr_ring_wait(self.h))

    def exchanges(self):
        n = C.c_ulonglong()
        check(self.lib.wr_ring_info(self.h, None, None, C.byref(n)))
        return n.value

    def destroy(self):
        if self.h:
            self.lib.wr_ring_destroy(self.h)
            self.h = None
