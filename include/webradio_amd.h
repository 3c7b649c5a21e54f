/*
 * webradio_amd.h -- C ABI of the MI355X (gfx950) backend for webradio's per-tuner
 * DSP hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI: its "plugin" interface
 * is the C++ virtual operator DspBlock (init/deinit/process, dsp/dspblock.h:82-84)
 * and the setters its callers use.  Each entry point below replaces the body of
 * one of those members; the host-side classes in webradio_amd/host/ keep the
 * reference's class names and signatures and forward to these functions, so
 * radio.cxx compiles against them unchanged (INTEGRATION.md shows the stub a
 * webradio maintainer would add to the upstream classes instead).
 *
 * Conventions
 *   - plain C, opaque handles, caller-owned buffers with explicit sizes
 *   - every function returns WR_OK (0) or a WR_ERR_* code; wr_last_error() gives a
 *     thread-local message.  The host classes map != WR_OK to `return false`
 *     from init()/process(), which is the reference's only error channel
 *     (dspblock.cxx:114-117,192-195).
 *   - "frames" are IQ pairs (one sample instant); IQ is interleaved float32
 *     [I0,Q0,I1,Q1,...] exactly like the reference's vector<sample_t> buffers.
 *   - pointers named *_dev are HIP device pointers on the context's device;
 *     pointers named *_host are ordinary host memory.
 *   - all work of one wr_dev is issued on one HIP stream given by the caller (e.g.
 *     torch.cuda.current_stream().cuda_stream); NULL is HIP's default stream.
 *     Functions taking host pointers synchronise that stream before returning;
 *     functions taking only device pointers are asynchronous.
 *   - there is NO CPU fallback: without a GPU wr_dev_open fails with
 *     WR_ERR_NODEV and nothing else can be created.
 *
 * Reference paths are relative to webradio's src/.
 */
#ifndef WEBRADIO_AMD_H_
#define WEBRADIO_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WR_ABI_VERSION   6       /* 2: wr_tuner_seek, *_n filters, af_gain/squelch, async uploads, ring_ready,
                                    blocks per launch added.  3: wr_ring_* (the halo ring of a time-sharded stream, incl.
                                    wr_ring_exchange_after / wr_tuner_mark_launches), wr_u8_to_f32_from_host, wr_dev_upload_ahead,
                                    wr_dev_wait_uploads_but added; channel filters of 128 / 256 taps accepted (WR_FIR_FUSED_MAX).
                                    4: wr_tune, wr_stage_windows_from_host, wr_tuner_last_staging added.
                                    5: wr_tuner_set_streaming, wr_tuner_stream_info, wr_block_kernel_calls added; audio filters and second
                                    channel stages of 128 / 256 taps accepted.
                                    6: wr_spectrum_lazy_info, wr_tuner_stream_host_blocks, wr_tuner_stream_long_blocks, wr_tuner_submit_count, WR_STREAM_MAX_BLOCKS added; wr_tuner_set_streaming
                                    takes 2 (byte blocks out of page-locked host memory stream too), wr_tuner_last_staging may say 3;
                                    wr_spectrum_push beside an open streaming launch keeps the frame and transforms it on demand.
                                    Nothing of an earlier version changed or removed */
#define WR_FIR_LENGTH    64      /* dsp/lowpass.cxx:39  FIR_LENGTH */
#define WR_TABLE_SIZE    65536   /* dsp/downconverter.cxx:35 LOOKUP_BITS 16 */

enum wr_status {
	WR_OK = 0,
	WR_ERR_ARG = 1,          /* bad argument (NULL, size, not a power of two, ...) */
	WR_ERR_HIP = 2,          /* a HIP runtime call or kernel launch failed */
	WR_ERR_STATE = 3,        /* call not legal in this state */
	WR_ERR_NOMEM = 4,
	WR_ERR_NODEV = 5,        /* no usable gfx950 device: the product has no CPU path */
	WR_ERR_RATE = 6          /* rates not integer related (dspblock.cxx:119-130) */
};

/* dsp/demodulator.h:40-46 enum Mode, same numbering */
enum wr_mode { WR_AM = 0, WR_FM = 1, WR_USB = 2, WR_LSB = 3 };

/* pipeline stages of one receiver chain (radio.cxx:68-76) that can be fetched */
enum wr_stage {
	WR_STAGE_CHAN_IQ = 1,    /* output of Receiver::_channelFilter  (2 ch) */
	WR_STAGE_DEMOD   = 2,    /* output of Receiver::_demodulator    (1 ch) */
	WR_STAGE_AUDIO   = 3     /* output of Receiver::_audioFilter    (1 ch) */
};

/* how the NCO sine/cosine of dsp/downconverter.cxx:100-101 is obtained */
enum wr_nco {
	WR_NCO_SPLIT = 0,        /* exact integer phase, sin/cos from two 256-entry
	                            LDS tables (coarse x fine angle addition); LO within
	                            3.5e-7 of the reference table entry */
	WR_NCO_EXACT = 1,        /* the reference's own 65536-entry sinf table, gathered from
	                            global memory, unfused multiply/add in the reference's
	                            order: channel-filter output is bit-identical */
	WR_NCO_ROTATE = 2        /* what the host runtime and bench.py use: the same table index sequence as the reference (exact integer
	                            phase), but each LO value is reached by turning the previous
	                            one by one of the two table angles the step allows, folded
	                            into the FIR as a Horner recurrence: no table access per tap.
	                            Channel IQ within 2e-7 of the bit-exact path for |x| <= 1 */
};

enum wr_where { WR_HOST = 0, WR_DEVICE = 1 };

typedef struct wr_dev      wr_dev;
typedef struct wr_tuner    wr_tuner;
typedef struct wr_spectrum wr_spectrum;
typedef struct wr_ring     wr_ring;

/* ------------------------------------------------------------------ misc -- */
int         wr_abi_version(void);
const char *wr_last_error(void);
int         wr_device_count(int *count);
/* Launch heuristics of the library, process-wide; they never change a result, only which of several
 * bit-identical kernel variants a launch takes.  Sets `key` to `value` (value < 0: back to the built-in /
 * environment default) and stores the value it had in *previous (optional).  For tests (both variants on
 * streams too short to reach a threshold) and for tuning; the reference has no counterpart. */
enum wr_tunable {
	WR_TUNE_DDC_NG2_MIN_PASSES = 1   /* k_tuner_ddc runs two lane groups per wave from this many passes of the grid on
	                                    (default 4, or $WR_DDC_NG2_MIN_PASSES at first use); 0: whenever the launch allows */
};
int         wr_tune(int key, long value, long *previous);

/* --------------------------------------------- host-side design helpers -- */
/* DownConverter::init / setIF: phaseStep = (int)((int64)hz * 2^31 / (int64)rate)
 * (dsp/downconverter.cxx:65,80) */
int wr_phase_step(int if_hz, unsigned int input_rate, int *phase_step);
/* DownConverter ctor table (dsp/downconverter.cxx:49-51); table[65536] */
int wr_sin_table(float *table_host);
/* LowPass::init window + LowPass::recalculate (dsp/lowpass.cxx:104-110,164-189):
 * 64 taps for `passband` Hz at `input_rate`; *maxbin_out (optional) receives the
 * integer cut-off bin of lowpass.cxx:167 */
int wr_lowpass_design(unsigned int passband, unsigned int input_rate,
                      float *coeff_host /* [64] */, unsigned int *maxbin_out);
/* the same with LowPass::_firLength = fir_length instead of the compiled-in 64: the
 * reference's FIXME "Make runtime variable" (dsp/lowpass.cxx:38-39) acted on -- every
 * expression of init()/recalculate() is already written in terms of _firLength.  A power of
 * two (the mask logic of lowpass.cxx:172,184) in [2, WR_FIR_MAX]. */
#define WR_FIR_MAX 1024
#define WR_FIR_FUSED_MAX 256     /* longest filter the per-tuner path takes (wr_chan_set_taps_n; r05: any of the three stages) */
int wr_lowpass_design_n(unsigned int fir_length, unsigned int passband, unsigned int input_rate,
                        float *coeff_host /* [fir_length] */, unsigned int *maxbin_out);
/* SpectrumSink::init window (io/spectrumsink.cxx:71-74); window[fft_size] */
int wr_spectrum_window(unsigned int fft_size, float *window_host);

/* ---------------------------------------------------------- device ctx -- */
/* Binds a device and a stream, uploads the NCO tables.  `hip_stream` is a
 * hipStream_t; NULL means HIP's default (null) stream. */
int wr_dev_open(wr_dev **dev, int device_index, void *hip_stream);
int wr_dev_close(wr_dev *dev);
int wr_dev_sync(wr_dev *dev);
void *wr_dev_stream(wr_dev *dev);
/* device memory for callers without another allocator (the C++ host classes) */
int wr_dev_malloc(wr_dev *dev, size_t bytes, void **ptr_dev);
int wr_dev_free(wr_dev *dev, void *ptr_dev);
int wr_dev_upload(wr_dev *dev, void *dst_dev, const void *src_host, size_t bytes);    /* sync */
int wr_dev_download(wr_dev *dev, void *dst_host, const void *src_dev, size_t bytes);
/* For callers that upload the same host buffer block after block (DspSource's output vector,
 * dspblock.h:118, 130-137): page-lock it once, then enqueue the copy and carry on -- the DMA runs
 * beside the caller's own work.  The buffer must not be written until wr_dev_wait_uploads()
 * (or wr_dev_sync) returns, and must be unregistered before it is freed.
 * Register memory that OWNS its pages -- a mapping of its own, as the allocator gives any large vector; page-aligned with
 * its last page to itself if it is carved out of something larger.  A registered range that shares a page with other
 * allocations, or whose address the allocator hands out again, can collide with the page locks the HIP runtime takes on
 * the fly for copies into pageable memory: on ROCm 7.2 that aborts the process (profiles/r04_abort_hunt.txt). */
int wr_dev_host_register(wr_dev *dev, void *host, size_t bytes);
int wr_dev_host_unregister(wr_dev *dev, void *host);
int wr_dev_upload_async(wr_dev *dev, void *dst_dev, const void *src_host, size_t bytes);
int wr_dev_wait_uploads(wr_dev *dev);  /* sync */
/* The same, leaving the `newest` (0..3) most recent uploads in flight: for a source that alternates between two (or more)
 * buffers -- it may refill buffer A as soon as everything but the upload out of buffer B has completed, and the host
 * then runs a block ahead of the GPU instead of in step with it. */
int wr_dev_wait_uploads_but(wr_dev *dev, unsigned int newest);
/* wr_dev_upload_async on the library's own upload stream: the copy runs beside what the device's stream is doing, and the
 * device's stream waits for it, so work enqueued after the call sees `dst_dev` filled.  A caller that ALTERNATES between two
 * `dst_dev` buffers gets the transfer of block b + 1 beside the kernels of block b: the copy waits only for the work that was
 * enqueued before the call after the last one that wrote the same `dst_dev` (rule for the caller: once the next block has been
 * uploaded into the other buffer, enqueue no more readers of this one).  `src_host` page-locked (wr_dev_host_register) for
 * the copy to be a DMA nobody waits for; counts as an upload in flight like wr_dev_upload_async. */
int wr_dev_upload_ahead(wr_dev *dev, void *dst_dev, const void *src_host, size_t bytes);

/* ------------------------------------------- one kernel per reference block -- */
/* DownConverter::process (dsp/downconverter.cxx:91-114).  Frame n uses phase
 * (*phase_io + n*phase_step) mod 2^31; on return *phase_io is the phase after the
 * block (the member DownConverter::phase).  Bit-identical to the reference. */
int wr_mix(wr_dev *dev, const float *in_dev, float *out_dev, size_t nframes,
           unsigned int *phase_io, int phase_step);

/* LowPass::process (dsp/lowpass.cxx:131-162) for `channels` interleaved channels.
 * history_dev holds (63*channels) floats: the last 63 input frames of the previous
 * call (zeros at start, lowpass.cxx:138-139); it is updated in place.
 * out_dev receives (nframes/decimation)*channels floats.  Bit-identical. */
int wr_fir_decimate(wr_dev *dev, const float *in_dev, size_t nframes, unsigned int channels,
                    unsigned int decimation, const float *coeff_host /* [64] */,
                    float *history_dev, float *out_dev);
/* the same for fir_length taps: history_dev holds (fir_length-1)*channels floats */
int wr_fir_decimate_n(wr_dev *dev, const float *in_dev, size_t nframes, unsigned int channels,
                      unsigned int decimation, unsigned int fir_length, const float *coeff_host /* [fir_length] */,
                      float *history_dev, float *out_dev);
/* how many stand-alone block kernels (wr_mix, wr_fir_decimate(_n), wr_demod) this process has run so far: 0 while every
 * Receiver rides a tuner batch */
unsigned long long wr_block_kernel_calls(void);

/* Demodulator::process (dsp/demodulator.cxx:77-115).  prev_io[2] = {prev_i,prev_q}
 * (host), updated on return. */
int wr_demod(wr_dev *dev, int mode, const float *in_dev, size_t nframes,
             float *prev_io /* [2] host */, float *out_dev);

/* RTL-SDR byte format to float, (u8 - 128)/128 (io/rtlsdrtuner.cxx:106) */
int wr_u8_to_f32(wr_dev *dev, const uint8_t *in_dev, float *out_dev, size_t count);
/* the same straight out of HOST memory page-locked with wr_dev_host_register -- a quarter of the float block's
 * bytes over PCIe.  Counts as an upload in flight: the host buffer must not be rewritten until wr_dev_wait_uploads
 * (or wr_dev_sync; wr_dev_wait_uploads_but for a source that alternates between buffers).
 * The bytes cross PCIe as a DMA copy on a stream of the library's own (into one of two internal raw buffers in turn), the
 * device's stream waits for the copy and converts: work enqueued after the call sees `out_dev` filled, in stream order
 * like any kernel's output, and the transfer of block b + 1 runs beside the kernels of block b. */
int wr_u8_to_f32_from_host(wr_dev *dev, const uint8_t *in_host_registered, float *out_dev, size_t count);
/* r04, sparse staging: of a block of `nframes` IQ frames in page-locked host memory (bytes when is_u8, else float32
 * pairs) bring over only what a tuner whose channel filters all have `length` taps and decimate by `period` reads --
 * frames [k * period - (length - 1), k * period] of every output frame k (dsp/lowpass.cxx:145-159), widened to 16-byte
 * boundaries -- plus the last `tail_frames` frames (the next block's filter history; a SpectrumSink's current frame), to
 * the SAME positions of the float block at `out_dev`; the frames in between are left as they are.  At BASELINE config 2
 * (length 64, period 400) that is a sixth of the block over PCIe.  The kernel reads the host memory itself, on the device's
 * stream (in order with the tuner's launches behind it); counts as an upload in flight like the call above.
 * Windows of any length up to 4096 frames (a window longer than a wave takes rounds).
 * A submit that follows must use the same `length` and `period` for ALL its receivers. */
int wr_stage_windows_from_host(wr_dev *dev, const void *in_host_registered, int is_u8, float *out_dev, size_t nframes,
                               unsigned int period, unsigned int length, size_t tail_frames);

/* ------------------------------------------------- fused per-tuner path -- */
/* One wr_tuner = one FrontEnd's tuner (radio.cxx:120-133) with all the Receivers
 * attached to it (radio.cxx:151-156).  Every receiver chain
 * DownConverter -> LowPass -> Demodulator -> LowPass (radio.cxx:68-82) of the tuner
 * is evaluated by ONE launch sequence per input block.
 * max_block_frames bounds nframes of wr_tuner_submit; max_channels <= WR_MAX_CHANNELS
 * (WR_ERR_ARG otherwise; the reference has no limit but its CPU, radio.cxx:151-156). */
#define WR_MAX_CHANNELS 4096
int wr_tuner_create(wr_tuner **tuner, wr_dev *dev, unsigned int input_rate,
                    unsigned int max_channels, size_t max_block_frames, int nco_mode);
int wr_tuner_destroy(wr_tuner *tuner);

/* Receiver() + FrontEnd::addReceiver (radio.cxx:62-90,151-156).  The new channel
 * starts with phase 0, zero histories, prev_i = prev_q = 0, IF 0, AM, and no
 * filters: it must be given both filters before the next submit. */
int wr_chan_add(wr_tuner *tuner, int *chan);
int wr_chan_remove(wr_tuner *tuner, int chan);
int wr_chan_count(wr_tuner *tuner, int *count);

/* Setters are STAGED and take effect at the next wr_tuner_submit (the reference
 * applies them mid-block from other threads without locking). */
/* DownConverter::setIF (dsp/downconverter.cxx:59-67) */
int wr_chan_set_if(wr_tuner *tuner, int chan, int if_hz);
/* LowPass::setPassband + setOutputSampleRate + init (dsp/lowpass.cxx:55-61,72-79,
 * 81-116).  stage 0 = channel filter (input = tuner rate, 2 ch), stage 1 = audio
 * filter (input = channel rate, 1 ch).  out_rate must divide the stage's input
 * rate (WR_ERR_RATE otherwise, dspblock.cxx:126-130).  Changing out_rate resets
 * the histories of the channel (as stop()/start() does, lowpass.cxx:118-129). */
int wr_chan_set_filter(wr_tuner *tuner, int chan, int stage, unsigned int passband,
                       unsigned int out_rate);
/* same with caller-designed taps */
int wr_chan_set_taps(wr_tuner *tuner, int chan, int stage, const float *coeff_host /* [64] */,
                     unsigned int decimation);
/* LowPass::_firLength as a run-time value (dsp/lowpass.cxx:38-39 FIXME, :102-110, :167-189 are
 * written in terms of it) INSIDE the fused path: fir_length a power of two in [2, 64].  A shorter
 * filter is the 64-tap filter with its oldest taps zero, bit for bit (the sum runs oldest sample
 * first, lowpass.cxx:150-158).  The CHANNEL filter (stage 0) may also have 128 or 256 taps
 * (WR_FIR_FUSED_MAX): receivers with such a filter form rate groups of their own, which keep the last L - 1
 * MIXED frames per channel as LowPass::block does (lowpass.cxx:138-142).  WR_NCO_EXACT: the reference's own
 * arithmetic -- table lookups, unfused products, oldest sample first -- bit-identical to the reference chain, at
 * the exact kernels' pace.  The other nco modes: the window as L / 64 segments of the ROTATE recurrence, within its
 * tolerance (1e-6), about L / 64 times the 64-tap kernel's time (lane groups whose channels do not share one filter:
 * the reference's arithmetic).  Never a full-rate mixer output.
 * r05: the AUDIO filter (stage 1) and the SECOND channel stage (stage 2) take 128 or 256 taps as well -- rate groups
 * keyed by those lengths too, 127 / 255 rows of history, k_tuner_demod + k_tuner_audio and k_tuner_iq2 with the
 * reference's order of additions: bit-identical to the reference chain in WR_NCO_EXACT, and in every mode as exact as
 * the channel IQ they are fed.  Longer filters (up to WR_FIR_MAX): wr_fir_decimate_n, block by block. */
int wr_chan_set_filter_n(wr_tuner *tuner, int chan, int stage, unsigned int fir_length,
                         unsigned int passband, unsigned int out_rate);
int wr_chan_set_taps_n(wr_tuner *tuner, int chan, int stage, const float *coeff_host /* [fir_length] */,
                       unsigned int fir_length, unsigned int decimation);
/* stage 2 (WR_FILTER_CHANNEL2): an optional SECOND channel LowPass between the first one and the
 * Demodulator -- several LowPass blocks in a row, which is how the reference's own means cut a
 * narrow channel out of a fast stream (one 64-tap stage gives bin 0 below fs / 128,
 * lowpass.cxx:167) -- evaluated inside the tuner's launch sequence.  Its input rate is stage 0's
 * output rate; set it after stage 0 and before stage 1 (the audio filter follows the last channel
 * stage).  WR_STAGE_CHAN_IQ then is ITS output: what the demodulator sees. */
#define WR_FILTER_CHANNEL  0
#define WR_FILTER_AUDIO    1
#define WR_FILTER_CHANNEL2 2
/* The two receiver controls the reference's REST interface names but never implements (af_gain and
 * squelch_threshold are reported as 0 with a FIXME, web/receiverhandler.cxx:112,118-119,127):
 *   af_gain   the audio sample times 10^(dB/20), in float, after the audio filter;
 *   squelch   an audio frame is muted (0.0f) when the mean power i^2 + q^2 of the decimation-many
 *             demodulator input frames behind it is below 10^(dBFS/10); enable = 0 opens it.
 * 0 dB and an open squelch leave every bit as it was. */
int wr_chan_set_af_gain(wr_tuner *tuner, int chan, float gain_db);
int wr_chan_set_squelch(wr_tuner *tuner, int chan, float threshold_dbfs, int enable);
/* Demodulator::setMode (dsp/demodulator.h:49) */
int wr_chan_set_mode(wr_tuner *tuner, int chan, int mode);
/* ask for WR_STAGE_CHAN_IQ / WR_STAGE_DEMOD to be kept for wr_chan_fetch (audio
 * always is).  A bitmask of (1 << stage). */
int wr_tuner_keep_stages(wr_tuner *tuner, unsigned int stage_mask);

/* streaming state of one channel: DownConverter::phase (downconverter.h:58),
 * Demodulator::prev_i/q (demodulator.h:60-61).  (The reference never sets the phase from
 * outside; under WR_NCO_ROTATE doing so also empties the channel filter's history, because
 * that mode keeps the NCO's turn into the next frame rather than its value.) */
int wr_chan_get_state(wr_tuner *tuner, int chan, unsigned int *phase, float *prev_iq /* [2] */);
int wr_chan_set_state(wr_tuner *tuner, int chan, unsigned int phase, const float *prev_iq);

/* DspSource::run for this tuner (dsp/dspblock.h:134, radio.cxx:56-59): push one
 * block of `nframes` IQ frames through every channel.  where = WR_HOST: iq is host
 * memory, copied to the device first; WR_DEVICE: iq is already in HBM and is read
 * in place (it must stay valid until the stream has passed this call). Async.
 * r04: a WR_HOST block in PAGE-LOCKED memory (wr_dev_host_register, hipHostMalloc) whose receivers all decimate alike
 * by at least twice their channel filter's length is staged sparsely -- only the frames under the taps and the block's
 * tail cross PCIe, read by a kernel on the device's stream (see wr_stage_windows_from_host); like any copy out of
 * page-locked memory it is asynchronous: the block must stay untouched until wr_dev_wait_uploads / wr_dev_sync.
 * $WR_HOST_SPARSE=0: the whole block, as before.  wr_tuner_last_staging says how the last WR_HOST block travelled:
 * 0 none yet, 1 copied whole, 2 staged sparsely, 3 (r06) streamed: a wr_tuner_submit_u8 block out of page-locked memory
 * under wr_tuner_set_streaming(tuner, 2) -- its bytes cross PCIe as a DMA copy on the library's upload stream and the
 * streaming launch's doorbell is rung by a stream memory operation behind them; nothing waits, and the block must stay
 * untouched until wr_dev_wait_uploads (as for any asynchronous upload).  wr_tuner_stream_host_blocks counts them.  The
 * WHOLE block crosses the link that way (sparse staging brings over a sixth at BASELINE config 2), and whatever else the
 * caller queues on the upload stream in front of a block (wr_dev_upload_ahead, wr_u8_to_f32_from_host) holds its doorbell
 * up: copies of 16 KB or less are copy kernels on this runtime and do not start beside an open launch. */
int wr_tuner_last_staging(wr_tuner *tuner, int *how);
int wr_tuner_submit(wr_tuner *tuner, const float *iq, size_t nframes, int where);
/* the same for a block in the RTL-SDR byte format (unsigned 8-bit interleaved IQ, what
 * RtlSdrTuner::dataReady receives, io/rtlsdrtuner.cxx:86-117): the (u8 - 128)/128
 * conversion of rtlsdrtuner.cxx:106 happens in the kernel's load stage, so a quarter of
 * the bytes cross PCIe and leave HBM.  Results are identical to converting first. */
int wr_tuner_submit_u8(wr_tuner *tuner, const uint8_t *iq_u8, size_t nframes, int where);

/* The demodulator + audio filter of a block (its "post stage") need not run inside the submit
 * that brought the block: where the kernel variant allows (WR_NCO_ROTATE, one channel filter per
 * lane group, audio decimation 1..6, demodulator output not kept) it is launched together with
 * the NEXT block's down-conversion, in the same kernel, so that a continuous stream costs one
 * launch per block.  Every call that reads results (wr_chan_fetch, wr_tuner_fetch_audio_all,
 * wr_tuner_audio_dev, wr_chan_get_state) or changes channel parameters launches a pending post
 * stage first, so nothing is ever observed out of order; wr_tuner_flush does only that (e.g. at
 * the end of a stream consumed through the audio ring).  WR_DEFER_POST=0 in the environment
 * turns the deferral off. */
int wr_tuner_flush(wr_tuner *tuner);

/* results of the last submit.  Frames per channel: CHAN_IQ nframes/d1 (x2 floats),
 * DEMOD nframes/d1, AUDIO nframes/d1/d2 -- the truncating arithmetic of
 * dspblock.cxx:177-178.  Synchronises the stream. */
int wr_chan_fetch(wr_tuner *tuner, int chan, int stage, float *out_host, size_t out_capacity,
                  size_t *count);
/* all channels' audio of the last submit, still on the device: audio of channel
 * slot s starts at (*audio_dev) + s * (*chan_stride); slot = wr_chan_slot(). */
int wr_tuner_audio_dev(wr_tuner *tuner, const float **audio_dev, size_t *chan_stride,
                       size_t *frames);
int wr_chan_slot(wr_tuner *tuner, int chan, int *slot);
/* every channel's audio of the last submit in ONE device-to-host copy (what the 256
 * AudioStreamManager sinks of a tuner consume, web/audiostream.cxx:65-73): channel slot s
 * lands at out_host + s * (*chan_stride); *slots_used rows are copied.  Synchronises. */
int wr_tuner_fetch_audio_all(wr_tuner *tuner, float *out_host, size_t out_capacity,
                             size_t *chan_stride, size_t *frames, unsigned int *slots_used);
/* ---- audio sink boundary: a ring of pinned host buffers ----
 * With a ring of `depth` >= 1 slots, ONE asynchronous device-to-host copy of all channels' audio
 * is queued into the next free slot right behind the kernel that produces it (with a deferred
 * post stage, see wr_tuner_flush: behind the next submit's launch, or behind the flush), and
 * nothing waits: the consumer side (the 256
 * AudioStreamManager sinks of a tuner, web/audiostream.cxx:65-73, or a recorder) takes the
 * blocks in order with acquire/release, typically from another thread or one block later,
 * so that the copy and the consumers overlap the next block's kernels.  Like the reference's
 * tuner ring (io/rtlsdrtuner.cxx:100-117) a full ring drops the NEW block's audio and counts
 * an overrun.  Blocks submitted while the tuner's channels do not all share one pair of
 * decimations are not queued (fetch per channel then).  depth 0 frees the ring.
 * r04: a deferred post stage writes its audio into the slot ITSELF (the slot is page-locked host memory mapped into the
 * device's address space; the kernel stores every sample there as well as in device memory), so the block is in the ring
 * when its launch has run and no copy is enqueued behind it ($WR_RING_DIRECT=0: the copy, as before). */
int wr_tuner_audio_ring(wr_tuner *tuner, unsigned int depth);
/* oldest block not yet released: waits for its copy, then channel slot s (wr_chan_slot) is at
 * (*audio_host) + s * (*chan_stride), *frames floats each; *seq counts submits from 0 (gaps =
 * overruns).  WR_ERR_STATE when nothing is queued.  The slot stays valid until released. */
int wr_tuner_audio_ring_acquire(wr_tuner *tuner, const float **audio_host, size_t *chan_stride,
                                size_t *frames, unsigned int *slots_used, unsigned long long *seq);
int wr_tuner_audio_ring_release(wr_tuner *tuner);
/* *ready = 1 when the oldest queued block's copy has landed: acquire would return without waiting */
int wr_tuner_audio_ring_ready(wr_tuner *tuner, int *ready);
int wr_tuner_audio_ring_stats(wr_tuner *tuner, unsigned int *queued, unsigned long long *overruns);
/* the number the NEXT wr_tuner_submit* of this tuner will carry (the `seq` of its ring entry): submits numbered so far,
 * whether or not they went through -- a caller that matches ring entries to its own blocks asks here instead of
 * keeping a count of its own beside the library's.  From the thread that submits. */
int wr_tuner_submit_count(wr_tuner *tuner, unsigned long long *submits);
/* LowPass::deinit + init (dsp/lowpass.cxx:118-129, 81-116): both filter histories of the
 * channel become empty again; NCO phase and Demodulator prev_i/q are kept (quirk Q5). */
int wr_chan_reset_history(wr_tuner *tuner, int chan);
/* Time sharding of ONE stream over several GPUs (BASELINE config 5; the reference has no
 * counterpart, its pipeline is one thread, radio.cxx:56-59): put every channel of the tuner in
 * the state it has when the stream STARTS at `frame` -- filter histories empty
 * (lowpass.cxx:138-139), Demodulator::prev_i/q zero (demodulator.cxx:60-70) -- except
 * DownConverter::phase, which takes its closed-form value after `frame` input frames from
 * phase 0 (downconverter.cxx:103).  A block [halo | chunk] submitted next reproduces the
 * sequential result for the chunk once the first halo/(D1*D2) audio frames are dropped
 * (webradio_amd/timeshard.py).  One call, no per-channel traffic -- and no launch: the next submit's launch
 * computes the phase itself and reads all-zero state sets, and a block's demodulator + audio filter that are
 * still waiting for that submit ride in it as usual (one launch per chunk).  Whatever else touches the tuner's
 * state in between (a getter, a setter, a kept demodulator stage, a second or a long channel-filter stage)
 * makes the seek real first. */
int wr_tuner_seek(wr_tuner *tuner, unsigned long long frame);
/* The halo ring of such a time-sharded stream (chunk c on rank c mod world, the halo of
 * every chunk -- the last H frames of the one before it, webradio_amd/timeshard.py: halo_frames --
 * from the ring neighbour).  What a LowPass carries from block to block in a member
 * (dsp/lowpass.cxx:133-142) here crosses GPUs: ONE ncclSend / ncclRecv pair per chunk in one group,
 * rank r -> r + 1, on RCCL directly (librccl.so.1 is looked up when the first wr_ring_* call
 * needs it; WR_ERR_NODEV if it is not there -- nothing else in the library depends on it).
 *   wr_ring_make_id   rank 0 makes the 128-byte rendezvous id (ncclGetUniqueId); the host gets the
 *                     bytes to the other ranks by its own means (a file, MPI, torch.distributed)
 *   wr_ring_create    every rank, collectively (ncclCommInitRank); one process per GPU
 *   wr_ring_exchange  enqueue one pair on the ring's own stream: send `nfloats` floats at send_dev
 *                     to rank + 1, receive as many from rank - 1 into recv_dev.  It is ordered
 *                     after everything the device's stream (wr_dev_open) holds at this moment and
 *                     returns at once -- issue it a round ahead: the halo is input, not a result
 *   wr_ring_wait      the device's stream waits (event to event, no host wait) for the last
 *                     exchange; then submit [halo | chunk] as usual
 * At world = 1 the neighbour is the rank itself (a device copy through RCCL). */
int wr_ring_id_bytes(void);
int wr_ring_version(int *version);                 /* ncclGetVersion of the RCCL that was loaded */
int wr_ring_make_id(void *id_host, size_t nbytes);
int wr_ring_create(wr_ring **ring, wr_dev *dev, const void *id_host, size_t nbytes, int rank, int world);
int wr_ring_exchange(wr_ring *ring, const float *send_dev, float *recv_dev, size_t nfloats);
/* The same, ordered behind the TUNER's launches so far instead of behind everything the device's stream holds: for a
 * halo that is input (already in device memory) received into a buffer only the tuner reads.  Needs
 * wr_tuner_mark_launches(tuner, 1): every launch that reads a submitted block then stamps an event with its own
 * completion signal, and the exchange waits for that -- nothing is put on the device's stream (an event record there
 * sits between two launches: 11 us a chunk).  The rule for the caller: `send_dev` is complete when the call is made,
 * `recv_dev` has no reader other than blocks already submitted to `tuner`.  Blocks the tuner still holds
 * (wr_tuner_set_blocks_per_launch) are launched by the call; blocks launched before marking was switched on are covered
 * by one record made when it is switched on; a tuner that does not mark its launches is WR_ERR_STATE. */
int wr_tuner_mark_launches(wr_tuner *tuner, int enable);
int wr_ring_exchange_after(wr_ring *ring, wr_tuner *tuner, const float *send_dev, float *recv_dev, size_t nfloats);
int wr_ring_wait(wr_ring *ring);
int wr_ring_info(wr_ring *ring, int *rank, int *world, unsigned long long *exchanges);
int wr_ring_destroy(wr_ring *ring);
/* scale applied to the audio as it is stored (default 1).  The MP3 encoder behind every
 * Receiver multiplies by 32768 before LAME (web/mp3encoder.cxx:65-72); a sink that wants
 * that format gets it from the audio kernel's store instead of a host loop. */
int wr_tuner_set_audio_scale(wr_tuner *tuner, float scale);

/* For callers whose blocks already sit in device memory (a resident recording, a capture ring):
 * up to `nblocks` consecutive WR_DEVICE submits whose buffers lie back to back, are equally long
 * and hold whole audio frames are HELD -- pointer and length, the caller keeps the memory alive --
 * and go out as ONE launch over the joined block when the last one arrives, when a submit does
 * not follow on, or when anything reads results or changes parameters (wr_tuner_flush, every
 * fetch and getter, every setter -- which so still takes effect at the boundary after the last
 * block submitted).  The outputs are the same bits (a frame does not
 * depend on where the stream is cut, DspBlock::run's only block-size effect being the truncation
 * of dspblock.cxx:177-178, which whole audio frames avoid); what changes is that they arrive per
 * group: the audio array, wr_chan_fetch and a ring entry then cover all blocks of the group, and
 * the tuner must have been created with max_block_frames >= nblocks * block.  The fixed costs of a
 * launch (start-up, the waves that finish early) are paid once per group: 32.8 -> 30.0 us per block
 * at BASELINE config 2 with nblocks = 4.  1 (the default) = off. */
int wr_tuner_set_blocks_per_launch(wr_tuner *tuner, unsigned int nblocks);

/* r05 -- the streaming launch.  The reference hands a block's output on within the run() that brought the block
 * (dsp/dspblock.cxx:169-212) and its tuner thread swaps the next block in whenever it has one
 * (io/rtlsdrtuner.cxx:265-285).  With `enable` != 0 the WR_DEVICE blocks of a tuner stop costing a kernel launch each:
 * the first opens ONE persistent launch, every following block of the same size and format rings its doorbell (a
 * descriptor and a counter in page-locked host memory -- wr_tuner_submit then returns after two stores), and the
 * launch takes the block up where it stands.  A block's demodulator and audio filter start the moment its last
 * channel-rate frame is out, whether or not another block follows, and its audio ring entry (wr_tuner_audio_ring)
 * becomes ready then -- no wr_tuner_flush needed, the launch stays open.  Everything else that touches the tuner (a
 * setter, a getter, wr_tuner_flush, a block of another size or out of host memory, a wait for the device's stream
 * through this library) closes the launch first: it finishes the blocks it was given and ends.  The results are the same
 * bits as with one launch per block.
 *   Taken up by: WR_NCO_ROTATE tuners with one rate group, at most 1024 channels on ONE 64-tap channel filter, no
 * second channel stage, an audio decimation of 1..6, 8 or 10, blocks of whole audio frames, no kept demodulator rows
 * (wr_tuner_keep_stages), no launch marks.  Any other submit goes the ordinary way, silently: wr_tuner_stream_info
 * tells which happened.
 *   Rules for the caller: a block's memory stays untouched until the NEXT block's audio is complete (or the launch is
 * closed and the stream waited for); before waiting for the device's stream by other means than this library
 * (hipStreamSynchronize on a stream handed to wr_dev_open, torch.cuda.synchronize()) call wr_tuner_flush -- an open
 * launch that nobody rings ends by itself only after half a second.  An open launch holds nearly every wave slot and
 * register of the GPU: calls on the same wr_dev close it first (a second tuner of the device that streams too takes
 * turns with it, a launch per block), but work sent to the GPU by other means -- another wr_dev, another process --
 * waits until the launch has ended; one streaming launch per GPU and process, a second context's tuner goes the
 * ordinary way meanwhile. */
#define WR_STREAM_MAX_BLOCKS 512u   /* blocks ONE streaming launch takes; the submit after them opens the next launch */
/* enable = 2 (r06): byte-format blocks out of page-locked HOST memory stream as well (wr_tuner_submit_u8(..., WR_HOST), see
 * wr_tuner_last_staging) */
int wr_tuner_set_streaming(wr_tuner *tuner, int enable);
/* `live`: a streaming launch is open right now; `launches`, `blocks`: opened / taken so far (any may be NULL) */
int wr_tuner_stream_info(wr_tuner *tuner, int *live, unsigned long long *launches, unsigned long long *blocks);
/* r06: of those blocks, how many came out of page-locked HOST memory (wr_tuner_submit_u8(..., WR_HOST), see wr_tuner_last_staging) */
int wr_tuner_stream_host_blocks(wr_tuner *tuner, unsigned long long *blocks);
/* r06: of the blocks of launches that are over and looked at (any fetch or sync after the close), how many had their demodulator + audio
 * filter run in LONG runs of tiles: the launch cuts a block's post stage into fewer, longer tasks when two further blocks are rung
 * already -- the host is ahead, the launch runs at the package's power limit and longer runs demodulate and read fewer rows twice --
 * and into short ones when the stream is host-paced or ends (a block's audio is out sooner).  Same bits either way. */
int wr_tuner_stream_long_blocks(wr_tuner *tuner, unsigned long long *blocks);

/* Profiling hook, the analogue of the reference's per-block profiler
 * (DspBlock::nsPerFrameOne, dsp/dspblock.h:69-75), with HIP events on the tuner's stream.
 * `enable` = 1: every submit's dominant launch (the fused mixer + channel filter) stamps its
 * own start and stop.  `enable` = n > 1: one event before the launch of every n-th submit and one
 * after the launch n - 1 submits later -- the mean per launch then includes the gaps between
 * launches and costs 1/n of the events' own few microseconds.  wr_tuner_profile_read
 * synchronises, returns the number of launches covered since the last read and their mean
 * duration in milliseconds, and resets the counters. */
int wr_tuner_profile(wr_tuner *tuner, int enable);
int wr_tuner_profile_read(wr_tuner *tuner, unsigned int *launches, double *mean_ms);

/* -------------------------------------------------------- SpectrumSink -- */
/* SpectrumSink::init (io/spectrumsink.cxx:60-77).  fft_size: power of two
 * (spectrumsink.cxx:53-56), 8 <= fft_size <= 1048576.  hop: frames between
 * successive transforms; 0 or fft_size = the reference's back-to-back frames
 * (spectrumsink.cxx:101-121); fft_size/2 = 50 % overlap (BASELINE config 3). */
int wr_spectrum_create(wr_spectrum **spec, wr_dev *dev, unsigned int fft_size, unsigned int hop);
int wr_spectrum_destroy(wr_spectrum *spec);
/* SpectrumSink::process (io/spectrumsink.cxx:88-123): append frames; every time a
 * frame is complete it is windowed and transformed.  Async for WR_DEVICE input.
 *   While a streaming launch is open on the device (wr_tuner_set_streaming) a WR_DEVICE block that holds a whole frame is
 * not transformed then and there -- that would close the launch every block: the newest frame (and what follows it) is
 * copied aside by the DMA engine and transformed when a getter below asks for it; only the most recent frame is
 * observable anyway (spectrumsink.cxx:93-94's FIXME, waterfallhandler.cxx:56-61 polls at 5 Hz).  A poll then costs one
 * closed launch, a block nothing.  wr_spectrum_lazy_info counts both. */
int wr_spectrum_push(wr_spectrum *spec, const float *iq, size_t nframes, int where);
int wr_spectrum_lazy_info(wr_spectrum *spec, unsigned long long *deferred_pushes, unsigned long long *resolves);
/* SpectrumSink::getSpectrum (io/spectrumsink.cxx:125-142): dB, fft-shifted, of the
 * most recent transform.  WR_ERR_STATE before the first complete frame (the
 * reference returns uninitialised memory there, quirk Q8). */
int wr_spectrum_get_db(wr_spectrum *spec, float *magnitudes_host /* [fft_size] */);
/* raw complex bins of the most recent transform (FFTW order), for tests */
int wr_spectrum_get_bins(wr_spectrum *spec, float *bins_host /* [2*fft_size] */);
int wr_spectrum_frames_done(wr_spectrum *spec, unsigned long *frames);
/* One waterfall row as the browser draws it (web/waterfallhandler.cxx:56-69 +
 * html/waterfall.js:92-109), computed on the device from the most recent transform:
 *   db_row_host[width]   (optional) dB per pixel column; non-finite values become -10000
 *                        (waterfallhandler.cxx:65-68)
 *   palette_host[width]  (optional) palette index clamp(floor((dB + 50) / 25 * 255), 0, 255)
 * `width` <= fft_size, fft_size % width == 0.  hold = 0: the bin the UI's overdraw leaves
 * visible in each column (the LAST of the fft_size/width bins that map to it);
 * hold = 1: the maximum of those bins (peak hold, loses no narrow carrier). */
int wr_spectrum_get_waterfall_row(wr_spectrum *spec, unsigned int width, int hold,
                                  float *db_row_host, uint8_t *palette_host);
/* transform `nframes_fft` whole frames laid out back to back at a fixed hop in
 * device memory and write dB rows (waterfall): frame f starts at iq_dev + 2*f*hop.
 * db_dev receives nframes_fft rows of fft_size floats (fft-shifted). Async. */
int wr_spectrum_batch_db(wr_spectrum *spec, const float *iq_dev, size_t nframes_fft,
                         float *db_dev);

#ifdef __cplusplus
}
#endif
#endif /* WEBRADIO_AMD_H_ */
