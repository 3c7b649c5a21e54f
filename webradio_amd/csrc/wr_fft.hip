/*
 * wr_fft.hip -- SpectrumSink on gfx950: Hamming window, forward complex FFT,
 * power -> dB with fftshift (io/spectrumsink.cxx:88-123 process, :125-142
 * getSpectrum).  Hand-written radix-2 Stockham transforms that live in LDS; no
 * rocFFT/hipFFT.
 *
 *   n <= 8192 : one workgroup per frame, the whole frame ping-pongs between two
 *               LDS buffers (2 x 64 KiB at n = 8192).
 *   n  > 8192 : four-step.  n = n1*n2, sample index n1*N2 + n2, bin k1 + N1*k2.
 *               pass 1: a workgroup takes CT adjacent columns n2, transforms them
 *                       over n1 in LDS, applies the window on load and the
 *                       inter-pass twiddle W_n^(n2*k1) on store -> work[k1][n2]
 *               pass 2: a workgroup takes RT adjacent rows k1, transforms them
 *                       over n2, writes bins (or dB, shifted) at k1 + N1*k2.
 *               For the 65536-point waterfall of BASELINE config 3 that is
 *               256 x 256 with one 512 KiB intermediate per frame, which stays in
 *               L2 / Infinity Cache between the passes.
 *
 * The window multiply is the reference's float multiply on the raw sample
 * (spectrumsink.cxx:109-112); the dB expression is the reference's float
 * expression (spectrumsink.cxx:127,137-138).
 */
#include "wr_internal.h"

#define FFT_THREADS 256

__device__ __forceinline__ float2 cmul(float2 a, float2 w)
{
	return make_float2(__builtin_fmaf(a.x, w.x, -a.y * w.y), __builtin_fmaf(a.x, w.y, a.y * w.x));
}

/* `ncols` independent length-`len` transforms, element (i, col) at buf[i*ncols + col].
 * Radix-2 Stockham autosort: natural order in, natural order out.  tw is the table
 * exp(-2*pi*i*k/(len*scale)), k < len*scale/2 (a table for a longer transform is the
 * same table sampled `scale` times finer).  Returns the buffer holding the result. */
__device__ float2 *lds_fft(float2 *a, float2 *b, unsigned int len, unsigned int ncols,
                           const float2 *__restrict__ tw, unsigned int scale, unsigned int tid)
{
	const unsigned int half = len >> 1;
	const unsigned int total = half * ncols;
	float2 *in = a, *out = b;
	for (unsigned int ns = 1; ns < len; ns <<= 1) {
		const unsigned int twstep = (half / ns) * scale;   /* len*scale / (2*ns) */
		for (unsigned int t = tid; t < total; t += FFT_THREADS) {
			const unsigned int j = t / ncols;
			const unsigned int col = t - j * ncols;
			const unsigned int k = j & (ns - 1);
			const float2 w = tw[k * twstep];
			const float2 v0 = in[j * ncols + col];
			const float2 v1 = cmul(in[(j + half) * ncols + col], w);
			const unsigned int j0 = ((j - k) << 1) + k;
			out[j0 * ncols + col] = make_float2(v0.x + v1.x, v0.y + v1.y);
			out[(j0 + ns) * ncols + col] = make_float2(v0.x - v1.x, v0.y - v1.y);
		}
		__syncthreads();
		float2 *tmp = in;
		in = out;
		out = tmp;
	}
	return in;
}

__device__ __forceinline__ float to_db(float2 v, float scaledb)
{
	/* spectrumsink.cxx:137-138: 10*log10f(re*re + im*im) - 20*log10f(N) */
	return 10.0f * log10f(v.x * v.x + v.y * v.y) - scaledb;
}

/* whole frame in LDS; grid.x = frame */
__global__ void __launch_bounds__(FFT_THREADS)
k_fft_single(const float2 *__restrict__ iq, size_t hop, unsigned int n,
             const float *__restrict__ window, const float2 *__restrict__ tw,
             float2 *__restrict__ bins, float *__restrict__ db, float scaledb)
{
	extern __shared__ float2 lds[];
	float2 *a = lds, *b = lds + n;
	const size_t frame = blockIdx.x;
	const float2 *x = iq + frame * hop;
	for (unsigned int i = threadIdx.x; i < n; i += FFT_THREADS) {
		const float2 v = x[i];
		const float w = window[i];
		a[i] = make_float2(v.x * w, v.y * w);
	}
	__syncthreads();
	const float2 *r = lds_fft(a, b, n, 1, tw, 1, threadIdx.x);
	for (unsigned int k = threadIdx.x; k < n; k += FFT_THREADS) {
		const float2 v = r[k];
		if (bins)
			bins[frame * n + k] = v;
		if (db)
			db[frame * n + ((k + (n >> 1)) & (n - 1))] = to_db(v, scaledb);
	}
}

/* four-step pass 1; grid = (n2/ct, frames) */
__global__ void __launch_bounds__(FFT_THREADS)
k_fft_pass1(const float2 *__restrict__ iq, size_t hop, unsigned int n1, unsigned int n2,
            unsigned int ct, const float *__restrict__ window, const float2 *__restrict__ tw_sub,
            unsigned int tw_sub_len, const float2 *__restrict__ tw_n, float2 *__restrict__ work)
{
	extern __shared__ float2 lds[];
	const unsigned int n = n1 * n2;
	float2 *a = lds, *b = lds + (size_t)n1 * ct;
	const size_t frame = blockIdx.y;
	const unsigned int col0 = blockIdx.x * ct;
	const float2 *x = iq + frame * hop;
	const unsigned int total = n1 * ct;
	for (unsigned int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const unsigned int r = e / ct, c = e - r * ct;
		const unsigned int idx = r * n2 + col0 + c;
		const float2 v = x[idx];
		const float w = window[idx];
		a[e] = make_float2(v.x * w, v.y * w);
	}
	__syncthreads();
	const float2 *res = lds_fft(a, b, n1, ct, tw_sub, tw_sub_len / n1, threadIdx.x);
	float2 *wout = work + frame * n;
	for (unsigned int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const unsigned int k1 = e / ct, c = e - k1 * ct;
		const unsigned int col = col0 + c;
		unsigned int m = col * k1;                      /* < n */
		float2 w;
		if (m < (n >> 1)) {
			w = tw_n[m];
		} else {
			const float2 t = tw_n[m - (n >> 1)];
			w = make_float2(-t.x, -t.y);
		}
		wout[(size_t)k1 * n2 + col] = cmul(res[e], w);
	}
}

/* four-step pass 2; grid = (n1/rt, frames) */
__global__ void __launch_bounds__(FFT_THREADS)
k_fft_pass2(const float2 *__restrict__ work, unsigned int n1, unsigned int n2, unsigned int rt,
            const float2 *__restrict__ tw_sub, unsigned int tw_sub_len,
            float2 *__restrict__ bins, float *__restrict__ db, float scaledb)
{
	extern __shared__ float2 lds[];
	const unsigned int n = n1 * n2;
	float2 *a = lds, *b = lds + (size_t)n2 * rt;
	const size_t frame = blockIdx.y;
	const unsigned int row0 = blockIdx.x * rt;
	const float2 *win = work + frame * n;
	const unsigned int total = n2 * rt;
	/* global read is contiguous along n2 within a row; LDS layout is [n2][rt] */
	for (unsigned int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const unsigned int r = e / n2, i = e - r * n2;
		a[i * rt + r] = win[(size_t)(row0 + r) * n2 + i];
	}
	__syncthreads();
	const float2 *in = lds_fft(a, b, n2, rt, tw_sub, tw_sub_len / n2, threadIdx.x);
	for (unsigned int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const unsigned int k2 = e / rt, r = e - k2 * rt;
		const unsigned int k = row0 + r + n1 * k2;
		const float2 v = in[e];
		if (bins)
			bins[frame * n + k] = v;
		if (db)
			db[frame * n + ((k + (n >> 1)) & (n - 1))] = to_db(v, scaledb);
	}
}

/* getSpectrum on stored bins (io/spectrumsink.cxx:125-142) */
__global__ void k_bins_to_db(const float2 *__restrict__ bins, unsigned int n, float *__restrict__ db,
                             float scaledb)
{
	for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
		db[(k + (n >> 1)) & (n - 1)] = to_db(bins[k], scaledb);
}

hipError_t wrk_bins_to_db(hipStream_t st, const float *bins, unsigned int n, float *db)
{
	const float scaledb = 20.0f * log10f((float)n);
	unsigned int grid = (n + 255) / 256;
	if (grid > 1024)
		grid = 1024;
	k_bins_to_db<<<grid, 256, 0, st>>>((const float2 *)bins, n, db, scaledb);
	return hipGetLastError();
}

hipError_t wrk_fft_frames(hipStream_t st, const WrFftPlan &P, const float *iq, size_t hop,
                          size_t nframes_fft, float *bins_out, float *db_out)
{
	if (!nframes_fft)
		return hipSuccess;
	const float scaledb = 20.0f * log10f((float)P.n);   /* spectrumsink.cxx:127 */
	hipError_t e;
	if (P.n2 == 1) {
		const size_t lds = (size_t)2 * P.n * sizeof(float2);
		if (lds > 48 * 1024) {
			e = hipFuncSetAttribute((const void *)k_fft_single,
			                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			if (e != hipSuccess)
				return e;
		}
		k_fft_single<<<(unsigned int)nframes_fft, FFT_THREADS, lds, st>>>(
			(const float2 *)iq, hop, P.n, P.window, (const float2 *)P.tw_n, (float2 *)bins_out,
			db_out, scaledb);
		return hipGetLastError();
	}
	const unsigned int tw_sub_len = P.n1 > P.n2 ? P.n1 : P.n2;
	unsigned int ct = 8192u / P.n1;
	if (ct > 16)
		ct = 16;
	unsigned int rt = 8192u / P.n2;
	if (rt > 16)
		rt = 16;
	const size_t lds1 = (size_t)2 * P.n1 * ct * sizeof(float2);
	const size_t lds2 = (size_t)2 * P.n2 * rt * sizeof(float2);
	e = hipFuncSetAttribute((const void *)k_fft_pass1, hipFuncAttributeMaxDynamicSharedMemorySize,
	                        (int)lds1);
	if (e != hipSuccess)
		return e;
	e = hipFuncSetAttribute((const void *)k_fft_pass2, hipFuncAttributeMaxDynamicSharedMemorySize,
	                        (int)lds2);
	if (e != hipSuccess)
		return e;
	size_t done = 0;
	while (done < nframes_fft) {
		size_t batch = nframes_fft - done;
		if (batch > P.work_frames)
			batch = P.work_frames;
		const float *src = iq + 2 * done * hop;
		dim3 g1(P.n2 / ct, (unsigned int)batch);
		k_fft_pass1<<<g1, FFT_THREADS, lds1, st>>>((const float2 *)src, hop, P.n1, P.n2, ct, P.window,
		                                           (const float2 *)P.tw_sub, tw_sub_len,
		                                           (const float2 *)P.tw_n, (float2 *)P.work);
		dim3 g2(P.n1 / rt, (unsigned int)batch);
		k_fft_pass2<<<g2, FFT_THREADS, lds2, st>>>(
			(const float2 *)P.work, P.n1, P.n2, rt, (const float2 *)P.tw_sub, tw_sub_len,
			bins_out ? (float2 *)(bins_out + 2 * done * P.n) : (float2 *)nullptr,
			db_out ? db_out + done * P.n : (float *)nullptr, scaledb);
		e = hipGetLastError();
		if (e != hipSuccess)
			return e;
		done += batch;
	}
	return hipSuccess;
}
