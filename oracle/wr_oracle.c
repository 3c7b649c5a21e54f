/*
 * wr_oracle.c -- CPU restatement of webradio's per-tuner DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see wr_oracle.h for the rules and the pinning
 * status).  Reference paths are relative to /root/reference/src.
 *
 * Written from the reference's observable arithmetic, not copied: each function
 * restates, in plain scalar C, what the cited lines compute -- the same operand
 * types, the same evaluation order, the same integer truncations -- so that
 * float results are reproducible bit-for-bit where only our own arithmetic is
 * involved (mixer, FIR, AM/USB/LSB) and to libm precision elsewhere.
 */
#include "wr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define PHASE_BITS   31              /* downconverter.cxx:34 */
#define LOOKUP_BITS  16              /* downconverter.cxx:35 */

/* ------------------------------------------------------------------ a1 -- */

/* downconverter.cxx:49-51: sinTable[n] = sinf((float)n * 2 * M_PI / (float)65536).
 * (float)n * 2 is a float product; * M_PI promotes to double; the quotient is a
 * double that is narrowed to float by sinf's parameter. */
void wro_sin_table(float *table)
{
	for (unsigned int n = 0; n < WRO_TABLE_SIZE; n++) {
		double arg = (double)((float)n * 2) * M_PI / (double)(float)(1UL << LOOKUP_BITS);
		table[n] = sinf((float)arg);
	}
}

/* downconverter.cxx:65,80: (int)((int64)hz * (int64)2^31 / (int64)inputRate),
 * i.e. 64-bit signed division truncating toward zero (quirk Q4). */
int wro_phase_step(int if_hz, unsigned int input_rate)
{
	return (int)((int64_t)if_hz * (int64_t)(1UL << PHASE_BITS) / (int64_t)input_rate);
}

/* ------------------------------------------------------------------ a2 -- */

/* downconverter.cxx:91-114.  Per frame: table index from the phase BEFORE the
 * increment (truncation, Q3), cosine = table a quarter turn ahead, phase kept to
 * 31 bits, multiply by the conjugate of the LO. */
void wro_mix(const float *table, unsigned int *phase, int phase_step,
             const float *in, float *out, size_t nframes)
{
	unsigned int ph = *phase;
	for (size_t n = 0; n < nframes; n++) {
		unsigned int sinidx = ph >> (PHASE_BITS - LOOKUP_BITS);
		unsigned int cosidx = (sinidx + (1u << LOOKUP_BITS) / 4) & ((1u << LOOKUP_BITS) - 1);
		ph = (ph + (unsigned int)phase_step) & ((1u << PHASE_BITS) - 1);
		float i = in[2 * n], q = in[2 * n + 1];
		float c = table[cosidx], s = table[sinidx];
		out[2 * n]     = i * c + q * s;
		out[2 * n + 1] = q * c - i * s;
	}
	*phase = ph;
}

/* ------------------------------------------------------------------ a3 -- */

/* lowpass.cxx:104-110: Hamming window, pre-divided by the FIR length (which also
 * applies the inverse-DFT scale).  cosf's argument is a double expression
 * narrowed to float; 0.54 - 0.46*cosf() is evaluated in double and stored to
 * float; the division is float/float. */
void wro_lowpass_window(float *window)
{
	wro_lowpass_window_n(WRO_FIR_LENGTH, window);
}

void wro_lowpass_window_n(unsigned int L, float *window)
{
	for (unsigned int n = 0; n < L; n++) {
		double arg = 2 * M_PI * (double)(float)n / (double)(float)(L - 1);
		float w = (float)(0.54 - 0.46 * (double)cosf((float)arg));
		w /= (float)L;
		window[n] = w;
	}
}

/* lowpass.cxx:167: unsigned 32-bit, left to right: L * passband / rate / 2 (Q6:
 * the product wraps modulo 2^32, small passbands give 0 -> all-zero taps). */
unsigned wro_lowpass_maxbin(unsigned int passband, unsigned int input_rate)
{
	return wro_lowpass_maxbin_n(WRO_FIR_LENGTH, passband, input_rate);
}

unsigned wro_lowpass_maxbin_n(unsigned int L, unsigned int passband, unsigned int input_rate)
{
	return L * passband / input_rate / 2;
}

/* lowpass.cxx:164-189.  Brick-wall real symmetric spec (bins n and (L-n)&63 set
 * to 1 for n < maxbin, n = 0..32), unnormalised inverse complex DFT
 * (FFTW_BACKWARD: sign +1, lowpass.cxx:100), real part of bin (n+32)&63 times
 * the window.  The reference runs FFTW's single-precision 64-point transform,
 * absent here: we accumulate the DFT in double and narrow once, which sits
 * within ~1 float ulp of any correct float transform of this 0/1 input. */
void wro_lowpass_design(unsigned int passband, unsigned int input_rate, float *coeff)
{
	wro_lowpass_design_n(WRO_FIR_LENGTH, passband, input_rate, coeff);
}

void wro_lowpass_design_n(unsigned int L, unsigned int passband, unsigned int input_rate, float *coeff)
{
	const unsigned int mask = L - 1;
	unsigned int maxbin = wro_lowpass_maxbin_n(L, passband, input_rate);
	float spec[WRO_FIR_MAX];
	float window[WRO_FIR_MAX];

	memset(spec, 0, sizeof(spec));
	for (unsigned int n = 0; n < L / 2 + 1; n++)
		spec[n] = spec[(L - n) & mask] = (n < maxbin) ? 1.0f : 0.0f;

	wro_lowpass_window_n(L, window);

	for (unsigned int n = 0; n < L; n++) {
		unsigned int bin = (n + L / 2) & mask;
		double re = 0.0;
		for (unsigned int k = 0; k < L; k++) {
			/* exact argument reduction: (bin*k mod L) / L turns */
			unsigned int t = (bin * k) & mask;
			re += (double)spec[k] * cos(2.0 * M_PI * (double)t / (double)L);
		}
		coeff[n] = (float)re * window[n];
	}
}

/* ------------------------------------------------------------------ a4 -- */

void wro_fir_init(wro_fir *f, unsigned int channels, unsigned int decimation,
                  const float *coeff)
{
	wro_fir_init_n(f, channels, decimation, coeff, WRO_FIR_LENGTH);
}

void wro_fir_init_n(wro_fir *f, unsigned int channels, unsigned int decimation,
                    const float *coeff, unsigned int L)
{
	f->channels = channels;
	f->decimation = decimation;
	f->length = L;
	memset(f->coeff, 0, sizeof(f->coeff));
	memcpy(f->coeff, coeff, L * sizeof(float));
	f->block = NULL;
	f->block_len = 0;
}

void wro_fir_free(wro_fir *f)
{
	free(f->block);
	f->block = NULL;
	f->block_len = 0;
}

/* lowpass.cxx:131-162.
 *  - `block` is resized to history+input FIRST (new elements zero, a shrink
 *    truncates), THEN its last 63 frames are moved to the front and the new
 *    input appended (Q7: a block-size change therefore corrupts the history).
 *  - output frame k = sum over j = 0..63 of coeff[63-j] * block[k*D + j], the
 *    accumulator starting at 0.0f and adding one product at a time, oldest
 *    sample first, channels interleaved. */
size_t wro_fir_process(wro_fir *f, const float *in, size_t in_floats, float *out)
{
	const unsigned int L = f->length;
	const unsigned int ch = f->channels;
	const size_t hist = (size_t)ch * (L - 1);
	const size_t want = in_floats + hist;

	if (f->block_len != want) {
		float *nb = (float *)calloc(want ? want : 1, sizeof(float));
		size_t keep = f->block_len < want ? f->block_len : want;
		if (f->block && keep)
			memcpy(nb, f->block, keep * sizeof(float));
		free(f->block);
		f->block = nb;
		f->block_len = want;
	}
	/* copy(block.end()-hist, block.end(), block.begin()) -- ranges may overlap
	 * only when in_floats < hist; std::copy is a forward copy then. */
	if (in_floats >= hist)
		memcpy(f->block, f->block + f->block_len - hist, hist * sizeof(float));
	else
		for (size_t n = 0; n < hist; n++)
			f->block[n] = f->block[f->block_len - hist + n];
	memcpy(f->block + hist, in, in_floats * sizeof(float));

	size_t inframes = in_floats / ch;
	size_t outframes = inframes / f->decimation;          /* dspblock.cxx:177-178 */
	size_t instep = (size_t)ch * f->decimation;
	const float *src = f->block;
	for (size_t k = 0; k < outframes; k++) {
		const float *p = src;
		for (unsigned int c = 0; c < ch; c++)
			out[k * ch + c] = 0.0f;
		for (unsigned int j = 0; j < L; j++) {
			float h = f->coeff[L - 1 - j];
			for (unsigned int c = 0; c < ch; c++)
				out[k * ch + c] += h * (*p++);
		}
		src += instep;
	}
	return outframes * ch;
}

/* ------------------------------------------------------------------ a5 -- */

/* demodulator.cxx:77-115.  AM: sqrt of a float -> float overload.  FM:
 * atan2f(ii, qq) with ii = Re, qq = Im of z*conj(z_prev) (argument order as in
 * the reference, Q1), then / M_PI / 2.0 in double, narrowed on store (Q2).
 * prev_i/prev_q are updated for every mode. */
int wro_demod(int mode, float *prev_i, float *prev_q,
              const float *in, float *out, size_t nframes)
{
	float pi_ = *prev_i, pq_ = *prev_q;
	for (size_t n = 0; n < nframes; n++) {
		float i = in[2 * n], q = in[2 * n + 1];
		switch (mode) {
		case WRO_AM:
			out[n] = sqrtf(i * i + q * q);
			break;
		case WRO_FM: {
			float ii = i * pi_ + q * pq_;
			float qq = q * pi_ - i * pq_;
			out[n] = (float)((double)atan2f(ii, qq) / M_PI / 2.0);
			break;
		}
		case WRO_USB:
			out[n] = i + q;
			break;
		case WRO_LSB:
			out[n] = i - q;
			break;
		default:
			return 0;                              /* "Bad mode": process() false */
		}
		pi_ = i;
		pq_ = q;
	}
	*prev_i = pi_;
	*prev_q = pq_;
	return 1;
}

/* ------------------------------------------------------------------ a6 -- */

/* spectrumsink.cxx:73: window[n] = 0.54 - 0.46*cosf(2*M_PI*(float)n/(float)(N-1)),
 * double expression narrowed on store. */
void wro_spectrum_window(unsigned int fft_size, float *window)
{
	for (unsigned int n = 0; n < fft_size; n++) {
		double arg = 2 * M_PI * (double)(float)n / (double)(float)(fft_size - 1);
		window[n] = (float)(0.54 - 0.46 * (double)cosf((float)arg));
	}
}

/* Forward unnormalised complex DFT (what FFTW_FORWARD computes,
 * spectrumsink.cxx:68,115): X[k] = sum x[n] exp(-2*pi*i*n*k/N).  Iterative
 * radix-2 decimation in time, all arithmetic in double, twiddles from exact
 * integer angles, one narrowing to float at the end. */
void wro_fft_forward(unsigned int n, const float *in, float *out)
{
	double *re = (double *)malloc(sizeof(double) * n);
	double *im = (double *)malloc(sizeof(double) * n);
	unsigned int bits = 0;
	while ((1u << bits) < n)
		bits++;
	for (unsigned int i = 0; i < n; i++) {
		unsigned int r = 0;
		for (unsigned int b = 0; b < bits; b++)
			if (i & (1u << b))
				r |= 1u << (bits - 1 - b);
		re[r] = in[2 * i];
		im[r] = in[2 * i + 1];
	}
	for (unsigned int len = 2; len <= n; len <<= 1) {
		unsigned int half = len >> 1;
		for (unsigned int j = 0; j < half; j++) {
			double ang = -2.0 * M_PI * (double)j / (double)len;
			double wr = cos(ang), wi = sin(ang);
			for (unsigned int base = 0; base < n; base += len) {
				unsigned int a = base + j, b = a + half;
				double tr = re[b] * wr - im[b] * wi;
				double ti = re[b] * wi + im[b] * wr;
				re[b] = re[a] - tr;
				im[b] = im[a] - ti;
				re[a] += tr;
				im[a] += ti;
			}
		}
	}
	for (unsigned int i = 0; i < n; i++) {
		out[2 * i] = (float)re[i];
		out[2 * i + 1] = (float)im[i];
	}
	free(re);
	free(im);
}

int wro_spectrum_init(wro_spectrum *s, unsigned int fft_size)
{
	if (fft_size == 0 || (fft_size & (fft_size - 1)))
		return 0;                                  /* spectrumsink.cxx:53-56 */
	s->fft_size = fft_size;
	s->inoffset = 0;                               /* spectrumsink.cxx:67 */
	s->frames_done = 0;
	s->inbuf = (float *)calloc((size_t)fft_size * 2, sizeof(float));
	s->outbuf = (float *)calloc((size_t)fft_size * 2, sizeof(float));
	s->window = (float *)malloc(sizeof(float) * fft_size);
	wro_spectrum_window(fft_size, s->window);
	return 1;
}

void wro_spectrum_free(wro_spectrum *s)
{
	free(s->inbuf);
	free(s->outbuf);
	free(s->window);
	s->inbuf = s->outbuf = s->window = NULL;
}

/* spectrumsink.cxx:88-123: append IQ frames to inbuf; each time it fills, window
 * in place (float multiply), transform into outbuf, restart at offset 0. */
void wro_spectrum_process(wro_spectrum *s, const float *in, size_t nframes)
{
	while (nframes) {
		size_t blocksize = s->fft_size - s->inoffset;
		if (blocksize > nframes)
			blocksize = nframes;
		memcpy(s->inbuf + 2 * (size_t)s->inoffset, in, blocksize * 2 * sizeof(float));
		s->inoffset += (unsigned int)blocksize;
		if (s->inoffset == s->fft_size) {
			for (unsigned int n = 0; n < s->fft_size; n++) {
				s->inbuf[2 * n] *= s->window[n];
				s->inbuf[2 * n + 1] *= s->window[n];
			}
			wro_fft_forward(s->fft_size, s->inbuf, s->outbuf);
			s->inoffset = 0;
			s->frames_done++;
		}
		nframes -= blocksize;
		in += blocksize * 2;
	}
}

/* spectrumsink.cxx:125-142: db = 10*log10f(re*re + im*im) - 20*log10f((float)N),
 * all float, stored fft-shifted. */
void wro_spectrum_db(unsigned int n, const float *outbuf, float *magnitudes)
{
	float scaledb = 20 * log10f((float)n);
	for (unsigned int k = 0; k < n; k++) {
		float re = outbuf[2 * k], im = outbuf[2 * k + 1];
		float db = 10 * log10f(re * re + im * im);
		magnitudes[(k < n / 2) ? (k + n / 2) : (k - n / 2)] = db - scaledb;
	}
}

/* One waterfall row as the browser ends up drawing it: waterfallhandler.cxx:56-69 (dB row,
 * non-finite -> -10000) reduced to `width` pixel columns the way waterfall.js:92-109 paints
 * (each bin fills floor(bin*w)..+ceil(w), later bins overdraw earlier ones; hold = 1 keeps
 * the maximum instead) and mapped to a palette index floor((v + 50) / 25 * 255) clamped to
 * 0..255 in double arithmetic, like JavaScript. */
void wro_waterfall_row(unsigned int n, const float *outbuf, unsigned int width, int hold,
                       float *db_row, unsigned char *palette)
{
	float *db = (float *)malloc(sizeof(float) * n);
	unsigned int per = n / width;
	wro_spectrum_db(n, outbuf, db);
	for (unsigned int x = 0; x < width; x++) {
		float v = 0.0f;
		for (unsigned int i = 0; i < per; i++) {
			float d = db[x * per + i];
			if (!isfinite(d))
				d = -10000.0f;
			if (i == 0 || !hold || d > v)
				v = d;
		}
		if (db_row)
			db_row[x] = v;
		if (palette) {
			double c = floor(((double)v + 50.0) / 25.0 * 255.0);
			if (c < 0.0) c = 0.0;
			if (c > 255.0) c = 255.0;
			palette[x] = (unsigned char)c;
		}
	}
	free(db);
}

void wro_spectrum_get(const wro_spectrum *s, float *magnitudes)
{
	wro_spectrum_db(s->fft_size, s->outbuf, magnitudes);
}

/* ------------------------------------------------------------------ a7 -- */

/* radio.cxx:62-90 wiring, with the rate negotiation of DspBlock::start
 * (dspblock.cxx:119-130): decimations must be exact integer ratios. */
int wro_receiver_init(wro_receiver *r, unsigned int input_rate, int if_hz,
                      unsigned int chan_passband, unsigned int chan_rate,
                      int mode, unsigned int audio_passband, unsigned int audio_rate)
{
	float coeff[WRO_FIR_LENGTH];
	memset(r, 0, sizeof(*r));
	if (chan_rate == 0 || audio_rate == 0 || input_rate < chan_rate || chan_rate < audio_rate)
		return 0;
	r->input_rate = input_rate;
	r->chan_rate = chan_rate;
	r->audio_rate = audio_rate;
	r->d1 = input_rate / chan_rate;
	r->d2 = chan_rate / audio_rate;
	if (input_rate / r->d1 != chan_rate || chan_rate / r->d2 != audio_rate)
		return 0;                                  /* "Sample rates must be integer related" */
	r->if_hz = if_hz;
	r->phase_step = wro_phase_step(if_hz, input_rate);
	r->mode = mode;
	r->phase = 0;                                  /* downconverter.cxx:46 */
	r->prev_i = r->prev_q = 0.0f;                  /* demodulator.cxx:35 */
	wro_lowpass_design(chan_passband, input_rate, coeff);
	wro_fir_init(&r->chan_fir, 2, r->d1, coeff);
	wro_lowpass_design(audio_passband, chan_rate, coeff);
	wro_fir_init(&r->audio_fir, 1, r->d2, coeff);
	return 1;
}

void wro_receiver_free(wro_receiver *r)
{
	wro_fir_free(&r->chan_fir);
	wro_fir_free(&r->audio_fir);
	free(r->mixed);
	free(r->chan_iq);
	free(r->demod);
	r->mixed = r->chan_iq = r->demod = NULL;
}

static float *grow(float *p, size_t *len, size_t want)
{
	if (*len != want) {
		free(p);
		p = (float *)calloc(want ? want : 1, sizeof(float));
		*len = want;
	}
	return p;
}

size_t wro_receiver_run(wro_receiver *r, const float *table, const float *iq,
                        size_t nframes, float *audio, float *chan_iq_out,
                        float *demod_out)
{
	size_t k1 = nframes / r->d1;                   /* dspblock.cxx:177-178 per block */
	size_t k2 = k1 / r->d2;
	r->mixed = grow(r->mixed, &r->mixed_len, nframes * 2);
	r->chan_iq = grow(r->chan_iq, &r->chan_len, k1 * 2);
	r->demod = grow(r->demod, &r->demod_len, k1);

	wro_mix(table, &r->phase, r->phase_step, iq, r->mixed, nframes);
	wro_fir_process(&r->chan_fir, r->mixed, nframes * 2, r->chan_iq);
	wro_demod(r->mode, &r->prev_i, &r->prev_q, r->chan_iq, r->demod, k1);
	wro_fir_process(&r->audio_fir, r->demod, k1, audio);

	if (chan_iq_out)
		memcpy(chan_iq_out, r->chan_iq, k1 * 2 * sizeof(float));
	if (demod_out)
		memcpy(demod_out, r->demod, k1 * sizeof(float));
	return k2;
}

double wro_bench_receivers(unsigned int input_rate, const int *if_hz, unsigned int nrx,
                           unsigned int chan_passband, unsigned int chan_rate, int mode,
                           unsigned int audio_passband, unsigned int audio_rate,
                           const float *iq, size_t nframes, unsigned int nblocks,
                           float *audio_last)
{
	float *table = (float *)malloc(sizeof(float) * WRO_TABLE_SIZE);
	wro_receiver *rx = (wro_receiver *)calloc(nrx, sizeof(wro_receiver));
	struct timespec t0, t1;
	size_t k2 = 0;
	float *scratch = NULL;

	wro_sin_table(table);
	for (unsigned int c = 0; c < nrx; c++)
		if (!wro_receiver_init(&rx[c], input_rate, if_hz[c], chan_passband, chan_rate,
		                       mode, audio_passband, audio_rate)) {
			free(table);
			free(rx);
			return -1.0;
		}
	k2 = nframes / rx[0].d1 / rx[0].d2;
	scratch = (float *)malloc(sizeof(float) * (k2 ? k2 : 1));

	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (unsigned int b = 0; b < nblocks; b++)
		for (unsigned int c = 0; c < nrx; c++)
			wro_receiver_run(&rx[c], table, iq, nframes,
			                 audio_last ? audio_last + (size_t)c * k2 : scratch, NULL, NULL);
	clock_gettime(CLOCK_MONOTONIC, &t1);

	for (unsigned int c = 0; c < nrx; c++)
		wro_receiver_free(&rx[c]);
	free(rx);
	free(table);
	free(scratch);
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* The same with T pipeline threads, each owning a disjoint subset of the receivers of the one
 * tuner buffer (the reference pipeline is single-threaded, radio.cxx:56-59; its
 * receiver graphs are independent objects, so T of them run side by side).  Wall clock around
 * all threads; the graphs are built before the clock starts. */
struct wro_mt_job {
	wro_receiver *rx;
	unsigned int first, count;
	const float *table;
	const float *iq;
	size_t nframes;
	unsigned int nblocks;
	float *scratch;
};

static void *wro_mt_worker(void *arg)
{
	struct wro_mt_job *j = (struct wro_mt_job *)arg;
	for (unsigned int b = 0; b < j->nblocks; b++)
		for (unsigned int c = 0; c < j->count; c++)
			wro_receiver_run(&j->rx[j->first + c], j->table, j->iq, j->nframes, j->scratch, NULL, NULL);
	return NULL;
}

double wro_bench_receivers_mt(unsigned int input_rate, const int *if_hz, unsigned int nrx,
                              unsigned int chan_passband, unsigned int chan_rate, int mode,
                              unsigned int audio_passband, unsigned int audio_rate,
                              const float *iq, size_t nframes, unsigned int nblocks,
                              unsigned int nthreads)
{
	if (nthreads < 1)
		nthreads = 1;
	if (nthreads > nrx)
		nthreads = nrx;
	float *table = (float *)malloc(sizeof(float) * WRO_TABLE_SIZE);
	wro_receiver *rx = (wro_receiver *)calloc(nrx, sizeof(wro_receiver));
	struct wro_mt_job *jobs = (struct wro_mt_job *)calloc(nthreads, sizeof(struct wro_mt_job));
	pthread_t *th = (pthread_t *)calloc(nthreads, sizeof(pthread_t));
	struct timespec t0, t1;

	wro_sin_table(table);
	for (unsigned int c = 0; c < nrx; c++)
		if (!wro_receiver_init(&rx[c], input_rate, if_hz[c], chan_passband, chan_rate,
		                       mode, audio_passband, audio_rate)) {
			free(table);
			free(rx);
			free(jobs);
			free(th);
			return -1.0;
		}
	const size_t k2 = nframes / rx[0].d1 / rx[0].d2;
	for (unsigned int t = 0; t < nthreads; t++) {
		jobs[t].rx = rx;
		jobs[t].first = (unsigned int)((unsigned long long)nrx * t / nthreads);
		jobs[t].count = (unsigned int)((unsigned long long)nrx * (t + 1) / nthreads) - jobs[t].first;
		jobs[t].table = table;
		jobs[t].iq = iq;
		jobs[t].nframes = nframes;
		jobs[t].nblocks = nblocks;
		jobs[t].scratch = (float *)malloc(sizeof(float) * (k2 ? k2 : 1));
	}
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (unsigned int t = 0; t < nthreads; t++)
		pthread_create(&th[t], NULL, wro_mt_worker, &jobs[t]);
	for (unsigned int t = 0; t < nthreads; t++)
		pthread_join(th[t], NULL);
	clock_gettime(CLOCK_MONOTONIC, &t1);

	for (unsigned int t = 0; t < nthreads; t++)
		free(jobs[t].scratch);
	for (unsigned int c = 0; c < nrx; c++)
		wro_receiver_free(&rx[c]);
	free(rx);
	free(table);
	free(jobs);
	free(th);
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* io/rtlsdrtuner.cxx:106: ((float)byte - 128.0) / 128.0 (double, exact) */
void wro_u8_to_float(const uint8_t *in, float *out, size_t n)
{
	for (size_t k = 0; k < n; k++)
		out[k] = (float)(((double)(float)in[k] - 128.0) / 128.0);
}
