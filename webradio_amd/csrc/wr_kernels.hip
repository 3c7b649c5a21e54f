/*
 * wr_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the webradio DSP hot path.
 *
 * Two tiers:
 *   1. one kernel per reference block (k_mix, k_fir, k_demod): used when a block
 *      stands alone in a graph; arithmetic is the reference's, operation for
 *      operation (compiled with -ffp-contract=off so nothing is fused behind our
 *      back), hence bit-identical to the CPU path.
 *   2. the fused per-tuner path (k_tuner_ddc -> k_tuner_demod -> k_tuner_audio):
 *      every receiver chain attached to one tuner in one launch sequence.  The
 *      full-rate mixer output, which the reference materialises per receiver
 *      (8 B x fs x channels), never exists: only the 64 input frames that reach
 *      a tap of each decimated output are mixed (lowpass.cxx:150-158 touches
 *      nothing else when decimation >= 64).
 *
 * Wavefronts are 64 wide.  In the fused kernels a LANE IS A RECEIVER CHANNEL: the
 * 64 lanes of a wave hold 64 channels of the same tuner, so the tuner samples are
 * wave-uniform (they arrive through the scalar cache into SGPRs, never through
 * VGPR loads) and per-channel state (phase, taps, accumulators) lives in VGPRs.
 *
 * Reference paths are relative to webradio's src/.
 */
#include "wr_internal.h"

typedef float v2f __attribute__((ext_vector_type(2)));

#define PHASE_FLAG_ACTIVE   1
#define PHASE_FLAG_HISTORY  2

/* ------------------------------------------------------------------------- */
/* tier 1: one kernel per reference block                                     */
/* ------------------------------------------------------------------------- */

/* DownConverter::process (dsp/downconverter.cxx:91-114).  Sample n of the block uses
 * phase (phase0 + n*step) mod 2^31 -- the closed form of the reference's running
 * accumulator -- table index phase >> 15, cosine a quarter table ahead. */
__global__ void __launch_bounds__(256)
k_mix(const float2 *__restrict__ in, float2 *__restrict__ out, size_t nframes,
      unsigned int phase0, unsigned int step, const float *__restrict__ table)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x; n < nframes; n += stride) {
		unsigned int ph = (phase0 + (unsigned int)n * step) & 0x7FFFFFFFu;
		unsigned int sinidx = ph >> 15;
		unsigned int cosidx = (sinidx + 16384u) & 65535u;
		float c = table[cosidx], s = table[sinidx];
		float2 x = in[n];
		float2 y;
		y.x = x.x * c + x.y * s;        /* not contracted: -ffp-contract=off */
		y.y = x.y * c - x.x * s;
		out[n] = y;
	}
}

/* LowPass::process (dsp/lowpass.cxx:131-162): out[k][c] = sum_j coeff[63-j] *
 * block[(k*D + j)][c], accumulated oldest sample first starting from 0.0f.
 * block = [63 history frames][input]; one thread per output float. */
__global__ void __launch_bounds__(256)
k_fir(const float *__restrict__ in, size_t outfloats, unsigned int channels, unsigned int decim,
      const float *__restrict__ coeff, const float *__restrict__ hist, float *__restrict__ out)
{
	__shared__ float taps[WR_FIR_LENGTH];
	if (threadIdx.x < WR_FIR_LENGTH)
		taps[threadIdx.x] = coeff[threadIdx.x];
	__syncthreads();
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < outfloats; o += stride) {
		size_t k = o / channels;
		unsigned int c = (unsigned int)(o - k * channels);
		size_t first = k * decim;               /* index into [hist|in] in frames */
		float acc = 0.0f;
		for (unsigned int j = 0; j < WR_FIR_LENGTH; ++j) {
			size_t f = first + j;
			float x = (f < WR_HIST) ? hist[f * channels + c] : in[(f - WR_HIST) * channels + c];
			acc = acc + taps[WR_FIR_LENGTH - 1 - j] * x;
		}
		out[o] = acc;
	}
}

/* next history = last 63 frames of [hist|in] (dsp/lowpass.cxx:138-142); written to
 * scratch first because for short blocks source and destination overlap */
__global__ void k_hist_build(const float *__restrict__ in, size_t nframes, unsigned int channels,
                             const float *__restrict__ hist, float *__restrict__ scratch)
{
	unsigned int total = WR_HIST * channels;
	for (unsigned int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
		size_t f = nframes + e / channels;      /* frame index in [hist|in] */
		unsigned int c = e % channels;
		scratch[e] = (f < WR_HIST) ? hist[f * channels + c] : in[(f - WR_HIST) * channels + c];
	}
}

__global__ void k_copy_f32(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		dst[i] = src[i];
}

/* the four detectors of Demodulator::process (dsp/demodulator.cxx:87-108) */
__device__ __forceinline__ float demod_one(int mode, float i, float q, float pi_, float pq_)
{
	switch (mode) {
	case WR_AM:
		/* correctly rounded like glibc's sqrtf: sqrt in double then one narrowing is exact
		 * for float inputs (53 >= 2*24 + 2) */
		return (float)sqrt((double)(i * i + q * q));
	case WR_FM: {
		float ii = i * pi_ + q * pq_;
		float qq = q * pi_ - i * pq_;
		/* atan2f(Re, Im) -- the reference's argument order -- then /M_PI/2.0 in double */
		return (float)((double)atan2f(ii, qq) / 3.14159265358979323846 / 2.0);
	}
	case WR_USB:
		return i + q;
	default:
		return i - q;
	}
}

__global__ void __launch_bounds__(256)
k_demod(int mode, const float2 *__restrict__ in, size_t nframes, float prev_i, float prev_q,
        float *__restrict__ out)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x; n < nframes; n += stride) {
		float2 z = in[n];
		float2 p = n ? in[n - 1] : make_float2(prev_i, prev_q);
		out[n] = demod_one(mode, z.x, z.y, p.x, p.y);
	}
}

/* io/rtlsdrtuner.cxx:106 */
__global__ void k_u8_to_f32(const uint8_t *__restrict__ in, float *__restrict__ out, size_t n)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		out[i] = ((float)in[i] - 128.0f) / 128.0f;
}

/* ------------------------------------------------------------------------- */
/* tier 2: fused per-tuner path                                               */
/* ------------------------------------------------------------------------- */

/* sin/cos of the NCO for left-aligned phase P (dsp/downconverter.cxx:100-101):
 * sinidx = P >> 16, cosidx = (sinidx + 16384) & 65535 = (P + 2^30) >> 16.
 * Returns (cos, sin). */
template <int NCO>
__device__ __forceinline__ v2f nco(unsigned int P, const float *__restrict__ table,
                                   const v2f *hi_l, const v2f *lo_l)
{
	if (NCO == WR_NCO_EXACT) {
		v2f cs;
		cs.y = table[P >> 16];
		cs.x = table[(P + 0x40000000u) >> 16];
		return cs;
	} else {
		const v2f a = hi_l[(P >> 24) << 5];              /* cis(2*pi*coarse/256) */
		const v2f b = lo_l[((P >> 16) & 255u) << 5];      /* cis(2*pi*fine/65536) */
		/* (ax + i ay)(bx + i by) as two packed ops */
		const v2f rot = {-a.y, a.x};
		return __builtin_elementwise_fma(rot, b.yy, a * b.xx);
	}
}

/* one tap: multiply the sample by conj(LO) (downconverter.cxx:109-110) and accumulate
 * coeff * mixed (lowpass.cxx:153-156).  EXACT keeps every rounding of the reference. */
template <int NCO>
__device__ __forceinline__ void mac(v2f xs, v2f cs, float hj, v2f &acc)
{
	if (NCO == WR_NCO_EXACT) {
		const float mi = xs.x * cs.x + xs.y * cs.y;
		const float mq = xs.y * cs.x - xs.x * cs.y;
		float ti, tq;                       /* products in asm: see the note below on pairs */
		asm("v_mul_f32 %0, %1, %2" : "=v"(ti) : "v"(hj), "v"(mi));
		asm("v_mul_f32 %0, %1, %2" : "=v"(tq) : "v"(hj), "v"(mq));
		acc.x = acc.x + ti;
		acc.y = acc.y + tq;
	} else {
		const v2f xr = {xs.y, -xs.x};
		const v2f m = __builtin_elementwise_fma(xr, cs.yy, xs * cs.xx);
		/* two scalar FMAs, spelled in asm so that the SLP vectoriser cannot turn them into
		 * one v_pk_fma_f32 whose broadcast tap operand would pin every one of the 64 taps
		 * to an aligned register PAIR (128 VGPRs, spills under the 1024-thread bound) */
		asm("v_fmac_f32 %0, %1, %2" : "+v"(acc.x) : "v"(hj), "v"(m.x));
		asm("v_fmac_f32 %0, %1, %2" : "+v"(acc.y) : "v"(hj), "v"(m.y));
	}
}

/*
 * k_tuner_ddc: DownConverter::process + channel LowPass::process for every channel
 * of a tuner (dsp/downconverter.cxx:91-114 feeding dsp/lowpass.cxx:131-162).
 *
 *   work unit  = (k, g): channel-rate output frame k for the 64 channel slots of
 *                lane group g.  One wave per unit; units are dealt round-robin to
 *                the waves of a persistent grid (one 1024-thread workgroup per CU
 *                in SPLIT mode, because the LDS tables take 128 KiB).
 *   per unit   : 64 taps.  Tap j touches input frame n = k*D1 - 63 + j, whose
 *                sample is wave-uniform.  Each lane advances its own left-aligned
 *                32-bit phase P (= reference phase << 1, so the 31-bit wrap of
 *                downconverter.cxx:103 is the natural 32-bit wrap), looks up
 *                sin/cos, rotates the sample and accumulates coeff[63-j] * mixed
 *                in the reference's order (oldest first).
 *   NCO lookup : EXACT  -> the reference's 65536-entry table, two global gathers,
 *                          unfused arithmetic: bit-identical channel IQ.
 *                SPLIT  -> idx16 = P >> 16 split into coarse/fine bytes; cis of each
 *                          from a 256-entry float2 table in LDS and one complex
 *                          multiply.  Each table is stored 32 times, copy r at
 *                          bank pair r, and lane l reads copy l & 31: a
 *                          ds_read_b64 is served per 32-lane half with every lane
 *                          on its own bank pair, so the gather is conflict-free
 *                          whatever the indices are.
 */
template <int NCO>
__global__ void __launch_bounds__(1024)
k_tuner_ddc(const float2 *__restrict__ cur, const float2 *__restrict__ hist, size_t k1,
            unsigned int d1, unsigned int slots,
            const unsigned int *__restrict__ phase, const unsigned int *__restrict__ step,
            const unsigned int *__restrict__ hist_step, const int *__restrict__ flags,
            const float *__restrict__ taps1, float2 *__restrict__ chan_iq,
            const float *__restrict__ table, const float2 *__restrict__ hi_cs,
            const float2 *__restrict__ lo_cs)
{
	extern __shared__ v2f lds_tables[];         /* SPLIT: [256][32] coarse then [256][32] fine */
	const unsigned int lane = threadIdx.x & 63u;
	const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned int waves_per_wg = blockDim.x >> 6;

	if (NCO == WR_NCO_SPLIT) {
		for (unsigned int e = threadIdx.x; e < WR_SPLIT_N * 32u; e += blockDim.x) {
			const float2 hv = hi_cs[e >> 5], lv = lo_cs[e >> 5];
			lds_tables[e] = (v2f){hv.x, hv.y};
			lds_tables[WR_SPLIT_N * 32u + e] = (v2f){lv.x, lv.y};
		}
		__syncthreads();
	}
	const v2f *hi_l = lds_tables + (lane & 31u);
	const v2f *lo_l = lds_tables + WR_SPLIT_N * 32u + (lane & 31u);

	const unsigned int groups = slots >> 6;
	const size_t units = k1 * groups;
	const size_t wave_global = (size_t)blockIdx.x * waves_per_wg + wave;
	const size_t wave_count = (size_t)gridDim.x * waves_per_wg;

	/* a wave keeps to one lane group where it can, so its taps stay in registers */
	unsigned int loaded_g = 0xFFFFFFFFu;
	float h[WR_FIR_LENGTH];
	unsigned int p0 = 0, st = 0, hst = 0;
	int fl = 0;

	for (size_t u = wave_global; u < units; u += wave_count) {
		/* g-major dealing: consecutive waves take consecutive k of the same group */
		const unsigned int g = (unsigned int)(u / k1);
		const size_t k = u - (size_t)g * k1;
		const unsigned int s = g * 64u + lane;
		if (g != loaded_g) {
#pragma unroll
			for (int j = 0; j < WR_FIR_LENGTH; ++j)
				h[j] = taps1[(size_t)j * slots + s];
			p0 = phase[s];
			st = step[s];
			hst = hist_step[s];
			fl = flags[s];
			loaded_g = g;
		}

		const long long n0 = (long long)k * d1 - WR_HIST;   /* input frame of tap j = 0 */
		v2f acc = {0.0f, 0.0f};

		if (n0 >= 0) {
			const float2 *x = cur + n0;                      /* wave-uniform */
			unsigned int P = p0 + (unsigned int)n0 * st;
#pragma unroll
			for (int jb = 0; jb < WR_FIR_LENGTH; jb += 8) {
#pragma unroll
				for (int jj = 0; jj < 8; ++jj) {
					const int j = jb + jj;
					const float2 xf = x[j];
					const v2f xs = {xf.x, xf.y};
					const v2f cs = nco<NCO>(P, table, hi_l, lo_l);
					P += st;
					mac<NCO>(xs, cs, h[WR_FIR_LENGTH - 1 - j], acc);
				}
				/* keep the scheduler from hoisting every lookup of the 64 taps at once */
				__builtin_amdgcn_sched_barrier(0);
			}
		} else {
			/* window reaches into the previous block (only the first ceil(63/D1) frames of
			 * a block): those frames were mixed with the phase sequence and the phase step
			 * of that block.  Rare, so taps come straight from memory, not registers. */
			const bool have_hist = (fl & PHASE_FLAG_HISTORY) != 0;
			for (int j = 0; j < WR_FIR_LENGTH; ++j) {
				const long long n = n0 + j;
				float2 xf;
				unsigned int P;
				if (n < 0) {
					xf = hist[WR_HIST + n];
					if (!have_hist)
						xf = make_float2(0.0f, 0.0f);
					P = p0 + (unsigned int)n * hst;
				} else {
					xf = cur[n];
					P = p0 + (unsigned int)n * st;
				}
				const v2f xs = {xf.x, xf.y};
				const v2f cs = nco<NCO>(P, table, hi_l, lo_l);
				const float hj = taps1[(size_t)(WR_FIR_LENGTH - 1 - j) * slots + s];
				mac<NCO>(xs, cs, hj, acc);
			}
		}
		if (fl & PHASE_FLAG_ACTIVE)
			chan_iq[k * slots + s] = make_float2(acc.x, acc.y);
	}
}

/* Demodulator::process for every channel (dsp/demodulator.cxx:77-115): thread
 * (k, s); the previous channel-rate frame is row k-1, or prev_iq for k = 0. */
__global__ void __launch_bounds__(256)
k_tuner_demod(const float2 *__restrict__ chan_iq, size_t k1, unsigned int slots,
              const int *__restrict__ mode, const int *__restrict__ flags,
              const float2 *__restrict__ prev_iq, float *__restrict__ dem)
{
	const size_t total = k1 * slots;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
		const size_t k = o / slots;
		const unsigned int s = (unsigned int)(o - k * slots);
		if (!(flags[s] & PHASE_FLAG_ACTIVE))
			continue;
		const float2 z = chan_iq[o];
		const float2 p = k ? chan_iq[o - slots] : prev_iq[s];
		dem[(size_t)(WR_HIST + k) * slots + s] = demod_one(mode[s], z.x, z.y, p.x, p.y);
	}
}

/* audio LowPass::process for every channel (dsp/lowpass.cxx:131-162, 1 channel):
 * dem already carries its 63 history rows, so row k2*D2 + j is tap j's sample.
 * A 64x64 tile goes through LDS so that audio[s][k2] is written in rows. */
__global__ void __launch_bounds__(256)
k_tuner_audio(const float *__restrict__ dem, size_t k2, unsigned int d2, unsigned int slots,
              const float *__restrict__ taps2, const int *__restrict__ flags,
              float *__restrict__ audio, size_t k2max)
{
	__shared__ float tile[64][65];
	const unsigned int lane = threadIdx.x & 63u;       /* slot within the group */
	const unsigned int row = threadIdx.x >> 6;         /* 0..3 */
	const unsigned int g = blockIdx.y;
	const unsigned int s = g * 64u + lane;
	const size_t kbase = (size_t)blockIdx.x * 64u;

	float h[WR_FIR_LENGTH];
#pragma unroll
	for (int j = 0; j < WR_FIR_LENGTH; ++j)
		h[j] = taps2[(size_t)j * slots + s];

	for (unsigned int r = row; r < 64u; r += 4u) {
		const size_t k = kbase + r;
		float acc = 0.0f;
		if (k < k2) {
			const float *x = dem + (k * d2) * slots + s;
#pragma unroll
			for (int j = 0; j < WR_FIR_LENGTH; ++j)
				acc = acc + h[WR_FIR_LENGTH - 1 - j] * x[(size_t)j * slots];
		}
		tile[r][lane] = acc;
	}
	__syncthreads();
	/* transposed write: thread (row, lane) writes slot (g*64 + r), time kbase + lane */
	for (unsigned int r = row; r < 64u; r += 4u) {
		const unsigned int so = g * 64u + r;
		const size_t k = kbase + lane;
		if (k < k2 && (flags[so] & PHASE_FLAG_ACTIVE))
			audio[(size_t)so * k2max + k] = tile[lane][r];
	}
}

/* end-of-block state update: phases advance by nframes steps, Demodulator prev_i/q
 * become the last channel-rate frame, history flag set */
__global__ void k_tuner_advance(unsigned int slots, unsigned int nframes_lo, size_t k1,
                                unsigned int *__restrict__ phase, const unsigned int *__restrict__ step,
                                unsigned int *__restrict__ hist_step, int *__restrict__ flags,
                                const float2 *__restrict__ chan_iq, float2 *__restrict__ prev_iq)
{
	unsigned int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= slots)
		return;
	int fl = flags[s];
	if (!(fl & PHASE_FLAG_ACTIVE))
		return;
	unsigned int stp = step[s];
	phase[s] = phase[s] + nframes_lo * stp;
	hist_step[s] = stp;
	flags[s] = fl | PHASE_FLAG_HISTORY;
	if (k1)
		prev_iq[s] = chan_iq[(k1 - 1) * slots + s];
}

/* dem history rows for the next block = last 63 rows of [hist rows | new rows] */
__global__ void k_dem_hist_build(const float *__restrict__ dem, size_t k1, unsigned int slots,
                                 float *__restrict__ scratch)
{
	size_t total = (size_t)WR_HIST * slots;
	for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
	     e += (size_t)gridDim.x * blockDim.x)
		scratch[e] = dem[k1 * slots + e];
}

/* strided row gather: dst[r*width + i] = src[r*row_stride + col_offset + i] */
__global__ void k_gather_rows(const float *__restrict__ src, size_t rows, size_t row_stride,
                              size_t col_offset, unsigned int width, float *__restrict__ dst)
{
	size_t total = rows * width;
	for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
	     e += (size_t)gridDim.x * blockDim.x) {
		size_t r = e / width;
		unsigned int i = (unsigned int)(e - r * width);
		dst[e] = src[r * row_stride + col_offset + i];
	}
}

/* ------------------------------------------------------------------------- */
/* launchers                                                                   */
/* ------------------------------------------------------------------------- */

static inline unsigned int grid_for(size_t n, unsigned int block, unsigned int cap)
{
	size_t g = (n + block - 1) / block;
	if (g < 1)
		g = 1;
	if (g > cap)
		g = cap;
	return (unsigned int)g;
}

hipError_t wrk_mix(hipStream_t st, const float *in, float *out, size_t nframes,
                   unsigned int phase, int step, const float *table_dev)
{
	if (!nframes)
		return hipSuccess;
	k_mix<<<grid_for(nframes, 256, 2048), 256, 0, st>>>((const float2 *)in, (float2 *)out, nframes,
	                                                    phase, (unsigned int)step, table_dev);
	return hipGetLastError();
}

hipError_t wrk_fir(hipStream_t st, const float *in, size_t nframes, unsigned int channels,
                   unsigned int decim, const float *coeff_dev, const float *hist_dev, float *out)
{
	size_t outfloats = (nframes / decim) * channels;
	if (!outfloats)
		return hipSuccess;
	k_fir<<<grid_for(outfloats, 256, 4096), 256, 0, st>>>(in, outfloats, channels, decim, coeff_dev,
	                                                      hist_dev, out);
	return hipGetLastError();
}

hipError_t wrk_hist_update(hipStream_t st, const float *in, size_t nframes, unsigned int channels,
                           float *hist_dev, float *scratch_dev)
{
	unsigned int total = WR_HIST * channels;
	k_hist_build<<<grid_for(total, 256, 64), 256, 0, st>>>(in, nframes, channels, hist_dev, scratch_dev);
	k_copy_f32<<<grid_for(total, 256, 64), 256, 0, st>>>(scratch_dev, hist_dev, total);
	return hipGetLastError();
}

hipError_t wrk_demod(hipStream_t st, int mode, const float *in, size_t nframes, float prev_i,
                     float prev_q, float *out)
{
	if (!nframes)
		return hipSuccess;
	k_demod<<<grid_for(nframes, 256, 2048), 256, 0, st>>>(mode, (const float2 *)in, nframes, prev_i,
	                                                      prev_q, out);
	return hipGetLastError();
}

hipError_t wrk_u8_to_f32(hipStream_t st, const uint8_t *in, float *out, size_t count)
{
	if (!count)
		return hipSuccess;
	k_u8_to_f32<<<grid_for(count, 256, 2048), 256, 0, st>>>(in, out, count);
	return hipGetLastError();
}

hipError_t wrk_tuner_ddc(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G,
                         const float *table_dev, const float *hi_dev, const float *lo_dev,
                         int num_cus)
{
	if (!L.k1 || !L.slots)
		return hipSuccess;
	const size_t units = L.k1 * (L.slots / 64);
	const unsigned int waves_per_wg = 16;
	unsigned int wgs = (unsigned int)((units + waves_per_wg - 1) / waves_per_wg);
	if (L.nco_mode == WR_NCO_EXACT) {
		/* no LDS: two workgroups per CU hide the gather latency */
		unsigned int cap = (unsigned int)num_cus * 2u;
		if (wgs > cap)
			wgs = cap;
		k_tuner_ddc<WR_NCO_EXACT><<<wgs, 1024, 0, st>>>(
			(const float2 *)L.cur, (const float2 *)L.hist, L.k1, L.d1, L.slots, G.phase, G.step,
			G.hist_step, G.flags, G.taps1, (float2 *)G.chan_iq, table_dev, (const float2 *)hi_dev,
			(const float2 *)lo_dev);
	} else {
		unsigned int cap = (unsigned int)num_cus;
		if (wgs > cap)
			wgs = cap;
		const size_t lds = (size_t)2 * WR_SPLIT_N * 32 * sizeof(float2);   /* 128 KiB */
		static bool attr_set = false;
		if (!attr_set) {
			hipError_t e = hipFuncSetAttribute((const void *)k_tuner_ddc<WR_NCO_SPLIT>,
			                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			if (e != hipSuccess)
				return e;
			attr_set = true;
		}
		k_tuner_ddc<WR_NCO_SPLIT><<<wgs, 1024, lds, st>>>(
			(const float2 *)L.cur, (const float2 *)L.hist, L.k1, L.d1, L.slots, G.phase, G.step,
			G.hist_step, G.flags, G.taps1, (float2 *)G.chan_iq, table_dev, (const float2 *)hi_dev,
			(const float2 *)lo_dev);
	}
	return hipGetLastError();
}

hipError_t wrk_tuner_demod(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G)
{
	size_t total = L.k1 * L.slots;
	if (!total)
		return hipSuccess;
	k_tuner_demod<<<grid_for(total, 256, 4096), 256, 0, st>>>((const float2 *)G.chan_iq, L.k1, L.slots,
	                                                          G.mode, G.flags, (const float2 *)G.prev_iq,
	                                                          G.dem);
	return hipGetLastError();
}

hipError_t wrk_tuner_audio(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G)
{
	if (!L.k2 || !L.slots)
		return hipSuccess;
	dim3 grid((unsigned int)((L.k2 + 63) / 64), L.slots / 64);
	k_tuner_audio<<<grid, 256, 0, st>>>(G.dem, L.k2, L.d2, L.slots, G.taps2, G.flags, G.audio, L.k2max);
	return hipGetLastError();
}

hipError_t wrk_tuner_advance(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G,
                             float *dem_scratch)
{
	if (!L.slots)
		return hipSuccess;
	k_tuner_advance<<<(L.slots + 255) / 256, 256, 0, st>>>(L.slots, (unsigned int)L.nframes, L.k1,
	                                                       G.phase, G.step, G.hist_step, G.flags,
	                                                       (const float2 *)G.chan_iq, (float2 *)G.prev_iq);
	size_t total = (size_t)WR_HIST * L.slots;
	k_dem_hist_build<<<grid_for(total, 256, 256), 256, 0, st>>>(G.dem, L.k1, L.slots, dem_scratch);
	k_copy_f32<<<grid_for(total, 256, 256), 256, 0, st>>>(dem_scratch, G.dem, total);
	return hipGetLastError();
}

hipError_t wrk_input_hist(hipStream_t st, const float *cur, size_t nframes, float *hist, float *scratch)
{
	return wrk_hist_update(st, cur, nframes, 2, hist, scratch);
}

hipError_t wrk_gather_rows(hipStream_t st, const float *src, size_t rows, size_t row_stride_floats,
                           size_t col_offset_floats, unsigned int width_floats, float *dst)
{
	size_t total = rows * width_floats;
	if (!total)
		return hipSuccess;
	k_gather_rows<<<grid_for(total, 256, 1024), 256, 0, st>>>(src, rows, row_stride_floats,
	                                                          col_offset_floats, width_floats, dst);
	return hipGetLastError();
}
