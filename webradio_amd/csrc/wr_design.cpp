/*
 * wr_design.cpp -- host-side, one-off design math of the backend: NCO tables,
 * phase step, FIR tap design, windows, FFT twiddles.  None of this is per-sample
 * work; the reference does the same things once in constructors / init() /
 * setters and so do we, on the host, then upload.
 *
 * Reference paths are relative to webradio's src/.
 */
#include "wr_internal.h"

#include <cmath>
#include <cstring>

namespace {
const double kPi = 3.14159265358979323846;   /* the value of M_PI the reference compiles in */
}

/* The DownConverter constructor's table (dsp/downconverter.cxx:49-51).  The kernels in
 * WR_NCO_EXACT mode gather from exactly this table, so it is produced the way the
 * reference produces it: float index doubled in float, scaled by M_PI/65536 in double,
 * narrowed, then libm sinf. */
void wrd_sin_table(float *table)
{
	const float entries = (float)WR_TABLE_SIZE;
	for (unsigned int n = 0; n < WR_TABLE_SIZE; ++n) {
		float twice = (float)n * 2;
		double angle = twice * kPi / entries;
		table[n] = sinf((float)angle);
	}
}

/* DownConverter::init / setIF (dsp/downconverter.cxx:65,80): signed 64-bit
 * hz * 2^31 / rate, C++ division (truncates toward zero), narrowed to int. */
/* sin(2 pi n / 65536) evaluated in double on the exact angle and narrowed once: the turns of
 * WR_NCO_ROTATE.  (The reference's table above carries the rounding of its float ARGUMENT, up
 * to 2.4e-7 per entry: harmless for one lookup, but a turn is applied up to 15 times in a row,
 * and |turn| != 1 compounds.) */
void wrd_sin_table_rounded(float *table)
{
	for (unsigned int n = 0; n < WR_TABLE_SIZE; ++n) {
		/* exact octant reduction keeps the symmetry sin^2 + cos^2 = 1 to the last bit of double */
		table[n] = (float)sin(2.0 * kPi * (double)n / (double)WR_TABLE_SIZE);
	}
	table[0] = 0.0f;
	table[WR_TABLE_SIZE / 2] = 0.0f;                       /* sin(pi): exactly zero, not 1.2e-16 */
}

int wrd_phase_step(int if_hz, unsigned int input_rate)
{
	long long num = (long long)if_hz * (1LL << 31);
	return (int)(num / (long long)input_rate);
}

/* dsp/lowpass.cxx:167 -- 32-bit unsigned arithmetic, evaluated left to right, so the
 * product wraps for passbands above 2^32/64 Hz and narrow passbands give bin 0. */
unsigned wrd_lowpass_maxbin(unsigned int L, unsigned int passband, unsigned int input_rate)
{
	unsigned int scaled = L * passband;               /* wraps modulo 2^32 like lowpass.cxx:167 */
	return scaled / input_rate / 2u;
}

/* LowPass::init window (dsp/lowpass.cxx:104-110) and LowPass::recalculate
 * (dsp/lowpass.cxx:164-189).
 *
 * The reference fills an L-bin real, even spectrum (L = _firLength, 64 as compiled) with ones
 * below `maxbin`, runs an unnormalised inverse DFT and keeps Re(impulse[(n+L/2)&(L-1)]) *
 * window[n].  For that 0/1 even spectrum the inverse DFT has the closed form of a Dirichlet
 * kernel,
 *      impulse[m] = e0 + 2*sum_{b=1}^{maxbin-1} cos(2*pi*b*m/L)      (+ Nyquist term)
 * which is what is evaluated here (in double, narrowed once).  Bins run to L/2 inclusive
 * (lowpass.cxx:173), so maxbin = L/2+1 would also switch the Nyquist bin on.  L is a power of
 * two (the mask logic of lowpass.cxx:172,184). */
void wrd_lowpass_design(unsigned int L, unsigned int passband, unsigned int input_rate, float *coeff)
{
	const unsigned int maxbin = wrd_lowpass_maxbin(L, passband, input_rate);

	for (unsigned int n = 0; n < L; ++n) {
		/* window sample, float arithmetic where the reference's is float */
		double warg = 2 * kPi * (double)(float)n / (double)(float)(L - 1);
		float w = (float)(0.54 - 0.46 * (double)cosf((float)warg));
		w /= (float)L;

		unsigned int m = (n + L / 2) & (L - 1);
		double acc = 0.0;
		if (maxbin > 0)
			acc += 1.0;                                   /* bin 0 */
		for (unsigned int b = 1; b < maxbin && b < L / 2; ++b) {
			unsigned int turn = (b * m) & (L - 1);        /* exact reduction of b*m/L turns */
			acc += 2.0 * cos(2.0 * kPi * (double)turn / (double)L);
		}
		if (maxbin > L / 2)                               /* bin L/2 switched on */
			acc += (m & 1u) ? -1.0 : 1.0;
		coeff[n] = (float)acc * w;
	}
}

/* SpectrumSink::init (io/spectrumsink.cxx:71-74): plain Hamming, no 1/N. */
void wrd_spectrum_window(unsigned int n, float *window)
{
	const float last = (float)(n - 1);
	for (unsigned int k = 0; k < n; ++k) {
		double arg = 2 * kPi * (double)(float)k / (double)last;
		window[k] = (float)(0.54 - 0.46 * (double)cosf((float)arg));
	}
}

/* WR_NCO_SPLIT tables.  The table index idx = phase >> 15 (16 bits) is split into a
 * coarse byte a and a fine byte b; sin/cos(2*pi*idx/65536) follow from the angle
 * addition of hi[a] = cis(2*pi*a/256) and lo[b] = cis(2*pi*b/65536). */
void wrd_split_tables(float *hi_cs, float *lo_cs)
{
	for (unsigned int k = 0; k < WR_SPLIT_N; ++k) {
		double ah = 2.0 * kPi * (double)k / 256.0;
		double al = 2.0 * kPi * (double)k / 65536.0;
		hi_cs[2 * k] = (float)cos(ah);
		hi_cs[2 * k + 1] = (float)sin(ah);
		lo_cs[2 * k] = (float)cos(al);
		lo_cs[2 * k + 1] = (float)sin(al);
	}
}

/* forward-transform twiddles exp(-2*pi*i*k/n), k < n/2, as (cos, -sin) pairs */
void wrd_twiddles(unsigned int n, float *tw)
{
	for (unsigned int k = 0; k < n / 2; ++k) {
		double a = 2.0 * kPi * (double)k / (double)n;
		tw[2 * k] = (float)cos(a);
		tw[2 * k + 1] = (float)(-sin(a));
	}
}
