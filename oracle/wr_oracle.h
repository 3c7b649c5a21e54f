/*
 * wr_oracle.h -- CPU restatement of webradio's per-tuner DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (webradio_amd/) never links, loads or calls it.
 *
 * Every function cites the reference lines (relative to /root/reference/src) it
 * restates.  Plain C99, scalar, one thread, float arithmetic in the reference's
 * evaluation order; build with -O2 -ffp-contract=off (the reference is built
 * -O2 for baseline x86-64, i.e. without FMA contraction: src/Makefile.am:4).
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   wro_demod          pinned  -- checked against the real reference
 *                                 dsp/demodulator.cxx compiled into oracle/_ref.
 *   DspBlock scheduling pinned -- host runtime checked against the real
 *                                 dsp/dspblock.cxx in oracle/_ref.
 *   wro_sin_table / wro_phase_step / wro_mix / wro_lowpass_* / wro_fir_* /
 *   wro_spectrum_* / the whole Receiver chain
 *                      pinned since r04, with a qualifier -- downconverter.cxx, lowpass.cxx
 *                                 and spectrumsink.cxx call FFTW3, which this image lacks; the
 *                                 image's own FFTW3-API library (hipFFTW over rocFFT) serves
 *                                 those calls in oracle/_ref/libwr_ref_chain.so
 *                                 (oracle/ref_chain.cxx, `make ref_chain`), which runs where a
 *                                 GPU is.  Checked live on the GPU box
 *                                 (tests/test_gpu_reference_pin.py) and through the vectors it
 *                                 produced there (tests/golden/reference_chain.npz,
 *                                 tests/test_oracle_reference_chain.py): mixer bit-identical,
 *                                 taps <= 3e-8, chain <= 3e-8, spectrum <= 4e-4 dB
 *                                 (profiles/r04_reference_pin.txt).  The qualifier: the FFT under
 *                                 the reference is rocFFT, not FFTW3 itself (upstream pins no
 *                                 FFTW version either, configure.ac:32).
 */
#ifndef WR_ORACLE_H_
#define WR_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WRO_FIR_LENGTH   64          /* lowpass.cxx:39 */
#define WRO_FIR_MAX      1024        /* largest _firLength the _n variants take (a power of two, lowpass.cxx:172) */
#define WRO_TABLE_SIZE   65536       /* downconverter.cxx:35,49 (LOOKUP_BITS 16) */

enum wro_mode { WRO_AM = 0, WRO_FM = 1, WRO_USB = 2, WRO_LSB = 3 };   /* demodulator.h:40-46 */

/* ---- a1: DownConverter table + phase step (downconverter.cxx:49-51, :80) ---- */
void     wro_sin_table(float *table /* [65536] */);
int      wro_phase_step(int if_hz, unsigned int input_rate);

/* ---- a2: DownConverter::process (downconverter.cxx:91-114) ---- */
void     wro_mix(const float *table, unsigned int *phase, int phase_step,
                 const float *in, float *out, size_t nframes);

/* ---- a3: LowPass::init window + recalculate (lowpass.cxx:104-110, :164-189) ---- */
void     wro_lowpass_window(float *window /* [64] */);
unsigned wro_lowpass_maxbin(unsigned int passband, unsigned int input_rate);
void     wro_lowpass_design(unsigned int passband, unsigned int input_rate,
                            float *coeff /* [64] */);
/* the same three with LowPass::_firLength = L instead of the compiled-in 64 -- what the
 * reference's own code computes once its FIXME (lowpass.cxx:38 "Make runtime variable") is
 * acted on: every expression there is already written in terms of _firLength */
void     wro_lowpass_window_n(unsigned int L, float *window /* [L] */);
unsigned wro_lowpass_maxbin_n(unsigned int L, unsigned int passband, unsigned int input_rate);
void     wro_lowpass_design_n(unsigned int L, unsigned int passband, unsigned int input_rate,
                              float *coeff /* [L] */);

/* ---- a4: LowPass::process (lowpass.cxx:131-162) ---- */
typedef struct wro_fir {
	unsigned int channels;           /* inputChannels() */
	unsigned int decimation;         /* DspBlock::decimation() */
	unsigned int length;             /* LowPass::_firLength (64 unless wro_fir_init_n) */
	float        coeff[WRO_FIR_MAX];
	float       *block;              /* history + current block (lowpass.h:64) */
	size_t       block_len;          /* floats */
} wro_fir;
void     wro_fir_init(wro_fir *f, unsigned int channels, unsigned int decimation,
                      const float *coeff);
void     wro_fir_init_n(wro_fir *f, unsigned int channels, unsigned int decimation,
                        const float *coeff, unsigned int L);
void     wro_fir_free(wro_fir *f);
/* out must hold (in_floats/channels/decimation)*channels floats, as sized by
 * DspBlock::run (dspblock.cxx:177-184).  Returns output float count. */
size_t   wro_fir_process(wro_fir *f, const float *in, size_t in_floats, float *out);

/* ---- a5: Demodulator::process (demodulator.cxx:77-115) ---- */
int      wro_demod(int mode, float *prev_i, float *prev_q,
                   const float *in, float *out, size_t nframes);

/* ---- a6: SpectrumSink (spectrumsink.cxx:60-77, :88-123, :125-142) ---- */
typedef struct wro_spectrum {
	unsigned int fft_size;
	unsigned int inoffset;
	float       *inbuf;              /* fft_size complex */
	float       *outbuf;             /* fft_size complex */
	float       *window;
	unsigned long frames_done;
} wro_spectrum;
int      wro_spectrum_init(wro_spectrum *s, unsigned int fft_size);
void     wro_spectrum_free(wro_spectrum *s);
void     wro_spectrum_window(unsigned int fft_size, float *window);
void     wro_spectrum_process(wro_spectrum *s, const float *in, size_t nframes);
void     wro_spectrum_get(const wro_spectrum *s, float *magnitudes);
/* forward unnormalised complex DFT, the operation fftwf_plan_dft_1d(N,in,out,
 * FFTW_FORWARD) performs (spectrumsink.cxx:68,115); double accumulate, float out */
void     wro_fft_forward(unsigned int n, const float *in, float *out);
/* dB + fftshift of one already transformed frame (spectrumsink.cxx:127-140) */
void     wro_spectrum_db(unsigned int n, const float *outbuf, float *magnitudes);
/* waterfall row for the UI (web/waterfallhandler.cxx:56-69, html/waterfall.js:92-109) */
void     wro_waterfall_row(unsigned int n, const float *outbuf, unsigned int width, int hold,
                           float *db_row, unsigned char *palette);

/* ---- a7: one Receiver chain (radio.cxx:62-90): DownConverter -> LowPass ->
 *          Demodulator -> LowPass, with every intermediate materialised exactly
 *          as DspBlock::run does (dspblock.cxx:169-212).  This is what
 *          bench.py times as the CPU baseline. ---- */
typedef struct wro_receiver {
	unsigned int input_rate, chan_rate, audio_rate;
	unsigned int d1, d2;
	int          if_hz, phase_step, mode;
	unsigned int phase;
	float        prev_i, prev_q;
	wro_fir      chan_fir, audio_fir;
	float       *mixed, *chan_iq, *demod;      /* block-sized intermediates */
	size_t       mixed_len, chan_len, demod_len;
} wro_receiver;
int      wro_receiver_init(wro_receiver *r, unsigned int input_rate, int if_hz,
                           unsigned int chan_passband, unsigned int chan_rate,
                           int mode, unsigned int audio_passband,
                           unsigned int audio_rate);
void     wro_receiver_free(wro_receiver *r);
/* runs one tuner block through the chain; audio must hold nframes/d1/d2 floats.
 * Optional taps of the intermediates (may be NULL). Returns audio sample count. */
size_t   wro_receiver_run(wro_receiver *r, const float *table, const float *iq,
                          size_t nframes, float *audio, float *chan_iq_out,
                          float *demod_out);

/* Run `nrx` receivers (IF list) sequentially over `nblocks` blocks of the same
 * tuner buffer, as Radio::run does for one front end (radio.cxx:56-59); returns
 * elapsed CLOCK_MONOTONIC seconds.  Used for the cpu_baseline leg. */
double   wro_bench_receivers(unsigned int input_rate, const int *if_hz, unsigned int nrx,
                             unsigned int chan_passband, unsigned int chan_rate, int mode,
                             unsigned int audio_passband, unsigned int audio_rate,
                             const float *iq, size_t nframes, unsigned int nblocks,
                             float *audio_last /* [nrx * nframes/d1/d2] or NULL */);

/* The same on `nthreads` pipeline threads, each owning a disjoint subset of the receivers
 * (the all-cores CPU baseline; audio is discarded). */
double   wro_bench_receivers_mt(unsigned int input_rate, const int *if_hz, unsigned int nrx,
                                unsigned int chan_passband, unsigned int chan_rate, int mode,
                                unsigned int audio_passband, unsigned int audio_rate,
                                const float *iq, size_t nframes, unsigned int nblocks,
                                unsigned int nthreads);

/* RTL-SDR u8 -> float rule (io/rtlsdrtuner.cxx:106) */
void     wro_u8_to_float(const uint8_t *in, float *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* WR_ORACLE_H_ */
