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

#include <cstdlib>

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

/* ------------------------------------------------------------------------------------
 * 65536-point fast path (BASELINE config 3): 256 x 256 four-step with the 256-point
 * sub-transforms done as 16 x 16 in REGISTERS -- each thread holds 16 points, does a
 * 16-point FFT, one exchange through LDS, another 16-point FFT.  One LDS round trip per
 * 256-point transform instead of eight (radix-2).
 * ------------------------------------------------------------------------------------ */

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

/* 16-point forward DFT in registers: decimation in frequency, then undo the bit reversal.
 * All indices are compile-time, so the loops dissolve into straight-line code and the
 * twiddles exp(-2*pi*i*j/len) into literals. */
__device__ __forceinline__ void fft16(float2 (&v)[16])
{
	const float c1 = 0.92387953251128673848f, s1 = 0.38268343236508978178f;   /* cos, sin(pi/8) */
	const float r2 = 0.70710678118654752440f;
	const float2 w16[8] = { {1.0f, 0.0f}, {c1, -s1}, {r2, -r2}, {s1, -c1},
	                        {0.0f, -1.0f}, {-s1, -c1}, {-r2, -r2}, {-c1, -s1} };
#pragma unroll
	for (int len = 16; len >= 2; len >>= 1) {
		const int half = len >> 1;
#pragma unroll
		for (int base = 0; base < 16; base += len) {
#pragma unroll
			for (int j = 0; j < half; ++j) {
				const float2 a = v[base + j], b = v[base + j + half];
				v[base + j] = cadd(a, b);
				const float2 d = csub(a, b);
				const int tw = j * (16 / len);             /* W_len^j = W_16^(j*16/len) */
				if (tw == 0)
					v[base + j + half] = d;
				else if (tw == 4)
					v[base + j + half] = make_float2(d.y, -d.x);          /* times -i */
				else
					v[base + j + half] = cmul(d, w16[tw]);
			}
		}
	}
	/* bit-reversed -> natural order (a register renaming) */
	float2 t;
#define SWAP16(a, b) t = v[a]; v[a] = v[b]; v[b] = t;
	SWAP16(1, 8) SWAP16(2, 4) SWAP16(3, 12) SWAP16(5, 10) SWAP16(7, 14) SWAP16(11, 13)
#undef SWAP16
}

/* W_256^m from the half-circle table (128 entries): W^(m+128) = -W^m */
__device__ __forceinline__ float2 w256(const float2 *__restrict__ tw, unsigned int m)
{
	const float2 w = tw[m & 127u];
	return (m & 128u) ? make_float2(-w.x, -w.y) : w;
}

typedef float nt_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 nt_load2(const float2 *p)
{
	const nt_v2f v = __builtin_nontemporal_load((const nt_v2f *)p);
	return make_float2(v.x, v.y);
}
__device__ __forceinline__ void nt_store2(float2 *p, float2 v)
{
	__builtin_nontemporal_store((nt_v2f){v.x, v.y}, (nt_v2f *)p);
}

/* write-through (sc1) stores: data that the NEXT launch reads (pass 1's intermediate) or nothing on the chip reads again
 * (the dB rows) leave the L2 as they are written, not at the end-of-kernel release.  r03, 121 frames on one box: pass 1
 * 25.6 -> 23.8 us, pass 2 16.9 -> 16.5 us (profiles/r03_fft_pass1_ablation.txt, section 4; FFT_PLAIN_STORE restores the old) */
__device__ __forceinline__ void wt_store2(float2 *p, float2 v)
{
#ifdef FFT_PLAIN_STORE
	*p = v;
#else
	union { float2 f; unsigned long long u; } cv;
	cv.f = v;
	__hip_atomic_store((unsigned long long *)p, cv.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void wt_store1(float *p, float v)
{
#ifdef FFT_PLAIN_STORE
	*p = v;
#else
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

#define F256_S 272        /* LDS row stride (float2): 16 x 16 plus 16 -> conflict-free exchanges */

/* pass 1: 16 adjacent columns n2 of FPW consecutive frames; thread (t = tid >> 4, c = tid & 15).
 * grid = (16, ceil(frames / FPW)).  work[k1][n2] = W_65536^(n2*k1) * sum_n1 w[n]x[n] W_256^(n1*k1).
 * FPW = 1: one frame per workgroup (small batches: the launch must fill the chip).  FPW = 4 (batches from
 * FFT64K_P1_LONG_MIN frames on): the window values a thread needs are the same 16 for every frame and stay in registers, as do
 * the two twiddle tables in LDS; the next frame's loads are issued before this frame's arithmetic; and at the reference's own
 * hop of half a frame, rows 128..255 of a frame are rows 0..127 of the next -- this thread already holds them.
 * (r03, 121 frames on one box: 25.0 -> 23.0 us; profiles/r03_fft_pass1_ablation.txt section 5.) */
template <unsigned int FPW>
__global__ void __launch_bounds__(256)
k_fft64k_pass1(const float2 *__restrict__ iq, size_t hop, const float *__restrict__ window,
               const float *__restrict__ window_p1, const float2 *__restrict__ tw256, const float2 *__restrict__ tw_n, float2 *__restrict__ work,
               unsigned int frames)
{
	__shared__ float2 ex[16 * F256_S];
	/* Twiddles from two 256-entry tables in LDS, W_65536^m = W_256^(m >> 8) * W_65536^(m & 255): the
	 * 65536 inter-pass twiddles of a frame were one 8-byte gather each from a 256 KB table in L2 --
	 * as many memory instructions as the samples themselves, and the slower half of this kernel
	 * (r02 profile: 40.8 us for 121 frames, against 19.8 us for pass 2) */
	__shared__ float2 thi[256], tlo[256];
	thi[threadIdx.x] = w256(tw256, threadIdx.x);
	tlo[threadIdx.x] = tw_n[threadIdx.x];
	const unsigned int c = threadIdx.x & 15u, t = threadIdx.x >> 4;
	const unsigned int col = blockIdx.x * 16u + c;
	const unsigned int f0 = blockIdx.y * FPW;
	float wv[16];
	float2 nx[16];                                         /* the frame about to be transformed, as loaded */
	{
		const float2 *x = iq + (size_t)f0 * hop;
#pragma unroll
		for (int a = 0; a < 16; ++a)
			nx[a] = x[(a * 16u + t) * 256u + col];
#ifdef FFT_P1_PLAIN_WINDOW
#pragma unroll
		for (int a = 0; a < 16; ++a)
			wv[a] = window[(a * 16u + t) * 256u + col];
#else
		const float4 *wp = (const float4 *)window_p1 + (size_t)blockIdx.x * 1024u + threadIdx.x;   /* [tile][a / 4][thread] */
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const float4 w4 = wp[q * 256u];
			wv[4 * q] = w4.x, wv[4 * q + 1] = w4.y, wv[4 * q + 2] = w4.z, wv[4 * q + 3] = w4.w;
		}
#endif
	}
#pragma unroll 1
	for (unsigned int fi = 0; fi < FPW; ++fi) {
		const unsigned int f = f0 + fi;
		if (f >= frames)
			break;
		float2 v[16];
#pragma unroll
		for (int a = 0; a < 16; ++a)
			v[a] = make_float2(nx[a].x * wv[a], nx[a].y * wv[a]);          /* spectrumsink.cxx:109-112 */
		if (FPW > 1 && fi + 1 < FPW && f + 1 < frames) {
			const float2 *xn = iq + (size_t)(f + 1) * hop;
			if (hop == 32768u) {
#pragma unroll
				for (int a = 0; a < 8; ++a)
					nx[a] = nx[a + 8];
#pragma unroll
				for (int a = 8; a < 16; ++a)
					nx[a] = xn[(a * 16u + t) * 256u + col];
			} else {
#pragma unroll
				for (int a = 0; a < 16; ++a)
					nx[a] = xn[(a * 16u + t) * 256u + col];
			}
		}
		fft16(v);                                          /* over a: k = 0..15, for n1 = a*16 + t */
		__syncthreads();                                   /* the tables; the frame before's reads of `ex` */
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			const float2 w = thi[(t * k) & 255u];          /* W_256^(t*k) */
			ex[k * F256_S + t * 16u + c] = (k == 0) ? v[k] : cmul(v[k], w);
		}
		__syncthreads();
#pragma unroll
		for (int b = 0; b < 16; ++b)
			v[b] = ex[t * F256_S + b * 16u + c];           /* this thread now owns k_low = t */
		fft16(v);                                          /* over b: k1 = t + 16*k_high */
		float2 *wout = work + (size_t)f * 65536u;
#pragma unroll
		for (int kh = 0; kh < 16; ++kh) {
			const unsigned int k1 = t + 16u * kh;
			const unsigned int m = col * k1;               /* < 65536 */
			const float2 w = cmul(thi[m >> 8], tlo[m & 255u]);
			/* intermediate laid out [column tile][k1][16]: this workgroup's 32 KiB are one contiguous run,
			 * and a pass-2 workgroup (16 rows k1) reads 2 KiB runs from each of the 16 tiles */
			wt_store2(&wout[(blockIdx.x * 256u + k1) * 16u + c], cmul(v[kh], w));
		}
	}
}

/* pass 2: 16 adjacent rows k1 of the intermediate; bins X[k1 + 256*k2].  Loads run with t
 * fastest (a row is contiguous), stores with the row index fastest (bins of neighbouring
 * k1 are contiguous).  grid = (16, frames). */
__global__ void __launch_bounds__(256)
k_fft64k_pass2(const float2 *__restrict__ work, const float2 *__restrict__ tw256,
               float2 *__restrict__ bins, float *__restrict__ db, float scaledb)
{
	__shared__ float2 ex[16 * 289];
	__shared__ float2 thi[256];
	thi[threadIdx.x] = w256(tw256, threadIdx.x);
	const unsigned int row0 = blockIdx.x * 16u;
	const float2 *win = work + (size_t)blockIdx.y * 65536u;
	float2 v[16];
	{
		const unsigned int t = threadIdx.x & 15u, r = threadIdx.x >> 4;
#pragma unroll
		for (int a = 0; a < 16; ++a)
#ifdef FFT_P2_NT
			v[a] = nt_load2(&win[(a * 256u + row0 + r) * 16u + t]);
#else
			v[a] = win[(a * 256u + row0 + r) * 16u + t];   /* column a*16 + t of row row0 + r (tiled, see pass 1) */
#endif
		fft16(v);
		__syncthreads();                               /* the table */
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			const float2 w = thi[(t * k) & 255u];
			ex[k * 289u + r * 17u + t] = (k == 0) ? v[k] : cmul(v[k], w);
		}
	}
	__syncthreads();
	{
		const unsigned int r = threadIdx.x & 15u, t = threadIdx.x >> 4;   /* t = k_low */
#pragma unroll
		for (int b = 0; b < 16; ++b)
			v[b] = ex[t * 289u + r * 17u + b];
		fft16(v);
		const size_t fbase = (size_t)blockIdx.y * 65536u;
#pragma unroll
		for (int kh = 0; kh < 16; ++kh) {
			const unsigned int k = row0 + r + 256u * (t + 16u * kh);
			if (bins)
				bins[fbase + k] = v[kh];
			if (db) {
#ifdef FFT_P2_NT
				__builtin_nontemporal_store(to_db(v[kh], scaledb), &db[fbase + ((k + 32768u) & 65535u)]);
#else
				wt_store1(&db[fbase + ((k + 32768u) & 65535u)], to_db(v[kh], scaledb));
#endif
			}
		}
	}
}

/* getSpectrum on stored bins (io/spectrumsink.cxx:125-142) */
__global__ void k_bins_to_db(const float2 *__restrict__ bins, unsigned int n, float *__restrict__ db,
                             float scaledb)
{
	for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
		db[(k + (n >> 1)) & (n - 1)] = to_db(bins[k], scaledb);
}

/* waterfall row: dB + fft-shift of the stored bins reduced to `width` pixel columns */
__global__ void k_waterfall_row(const float2 *__restrict__ bins, unsigned int n, unsigned int width, int hold,
                                float scaledb, float *__restrict__ db_row, uint8_t *__restrict__ palette)
{
	const unsigned int per = n / width;
	for (unsigned int x = blockIdx.x * blockDim.x + threadIdx.x; x < width; x += gridDim.x * blockDim.x) {
		float v = 0.0f;
		for (unsigned int i = 0; i < per; ++i) {
			const unsigned int shifted = x * per + i;                 /* index in the shifted row */
			const unsigned int k = (shifted + (n >> 1)) & (n - 1);     /* FFTW-order bin */
			float d = to_db(bins[k], scaledb);
			if (!isfinite(d))
				d = -10000.0f;                                        /* waterfallhandler.cxx:65-68 */
			if (i == 0 || !hold || d > v)
				v = d;                                                /* last one wins, or running max */
		}
		if (db_row)
			db_row[x] = v;
		if (palette) {
			/* waterfall.js:94-106, in double like JavaScript numbers */
			double c = floor(((double)v + 50.0) / 25.0 * 255.0);
			c = c < 0.0 ? 0.0 : (c > 255.0 ? 255.0 : c);
			palette[x] = (uint8_t)c;
		}
	}
}

hipError_t wrk_waterfall_row(hipStream_t st, const float *bins, unsigned int n, unsigned int width, int hold,
                             float *db_row, uint8_t *palette)
{
	const float scaledb = 20.0f * log10f((float)n);
	unsigned int grid = (width + 255) / 256;
	k_waterfall_row<<<grid, 256, 0, st>>>((const float2 *)bins, n, width, hold, scaledb, db_row, palette);
	return hipGetLastError();
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

/* batches of the 65536-point path from this many frames on take four frames per pass-1 workgroup (>= 384 workgroups) */
#define FFT64K_P1_LONG_MIN 96u
static int fft64k_p1_fpw(size_t batch)
{
	static int forced = -1;
	if (forced < 0) {
		const char *v = getenv("WR_FFT_P1_FPW");           /* 1, 2, 4: frames per pass-1 workgroup whatever the batch */
		forced = v ? atoi(v) : 0;
	}
	if (forced == 1 || forced == 2 || forced == 4)
		return forced;
	return batch >= FFT64K_P1_LONG_MIN ? 4 : 1;
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
	if (P.n1 == 256 && P.n2 == 256) {
		size_t done = 0;
		while (done < nframes_fft) {
			size_t batch = nframes_fft - done;
			if (batch > P.work_frames)
				batch = P.work_frames;
			dim3 grid(16, (unsigned int)batch);
			const int fpw = fft64k_p1_fpw(batch);
			if (fpw == 2)
				k_fft64k_pass1<2u><<<dim3(16, (unsigned int)((batch + 1) / 2)), 256, 0, st>>>(
					(const float2 *)(iq + 2 * done * hop), hop, P.window, P.window_p1, (const float2 *)P.tw_sub, (const float2 *)P.tw_n,
					(float2 *)P.work, (unsigned int)batch);
			else if (fpw == 4)
				k_fft64k_pass1<4u><<<dim3(16, (unsigned int)((batch + 3) / 4)), 256, 0, st>>>(
					(const float2 *)(iq + 2 * done * hop), hop, P.window, P.window_p1, (const float2 *)P.tw_sub, (const float2 *)P.tw_n,
					(float2 *)P.work, (unsigned int)batch);
			else
				k_fft64k_pass1<1u><<<grid, 256, 0, st>>>((const float2 *)(iq + 2 * done * hop), hop, P.window, P.window_p1,
				                                         (const float2 *)P.tw_sub, (const float2 *)P.tw_n, (float2 *)P.work,
				                                         (unsigned int)batch);
			k_fft64k_pass2<<<grid, 256, 0, st>>>(
				(const float2 *)P.work, (const float2 *)P.tw_sub,
				bins_out ? (float2 *)(bins_out + 2 * done * P.n) : (float2 *)nullptr,
				db_out ? db_out + done * P.n : (float *)nullptr, scaledb);
			e = hipGetLastError();
			if (e != hipSuccess)
				return e;
			done += batch;
		}
		return hipSuccess;
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
