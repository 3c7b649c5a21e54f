/*
 * wr_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the webradio DSP hot path.
 *
 * Two tiers:
 *   1. one kernel per reference block (k_mix, k_fir, k_demod): used when a block
 *      stands alone in a graph; arithmetic is the reference's, operation for
 *      operation (compiled with -ffp-contract=off so nothing is fused behind our
 *      back), hence bit-identical to the CPU path.
 *   2. the fused per-tuner path: k_tuner_ddc (NCO mix + channel filter of every receiver
 *      chain attached to one tuner) and the post stage (demodulator + audio filter:
 *      k_tuner_post, or post_role inside the NEXT block's k_tuner_ddc launch; k_tuner_demod +
 *      k_tuner_audio when the demodulator output itself is wanted).  The full-rate mixer
 *      output, which the reference materialises per receiver (8 B x fs x channels), never
 *      exists: only the 64 input frames that reach a tap of each decimated output are
 *      mixed (lowpass.cxx:150-158 touches nothing else when decimation >= 64).
 *
 * Wavefronts are 64 wide.  In the fused kernels a LANE IS A RECEIVER CHANNEL: the
 * 64 lanes of a wave hold 64 channels of the same tuner, so the tuner samples are
 * wave-uniform (one coalesced load per output frame, handed round through a per-wave LDS
 * window) and per-channel state (phase, taps, accumulators) lives in VGPRs.
 *
 * Reference paths are relative to webradio's src/.
 */
#include "wr_internal.h"

#include <atomic>
#include <cstdlib>

#include <hip/hip_ext.h>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2f lds_v2f;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f lds_v4f;

#define PHASE_FLAG_ACTIVE   1

/* loads through this address space are scalar (s_load) whenever the address is wave-uniform */
#define WR_CONSTANT __attribute__((address_space(4)))

#ifdef DDC_TIMELINE
/* development aid (tools/mkvariant.sh ... -DDDC_TIMELINE): per-wave s_memtime stamps of k_tuner_ddc */
#define TL_SLOTS 12
__device__ unsigned long long g_ddc_tl[16384 * TL_SLOTS];
extern "C" int wr_debug_timeline(unsigned long long *out, size_t n)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ddc_tl), n * sizeof(unsigned long long));
}
/* (slot 0 and 11 also take the constant 100 MHz clock: the shader clock the wave really ran at is
 * (memtime[11] - memtime[0]) / (realtime[11] - realtime[0]) x 100 MHz) */
__device__ unsigned long long g_ddc_rt[16384 * 2];
extern "C" int wr_debug_timeline_rt(unsigned long long *out, size_t n)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ddc_rt), n * sizeof(unsigned long long));
}
/* where the wave ran: HW_ID (gfx9: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]) and XCC_ID[3:0] */
__device__ unsigned int g_ddc_hw[16384 * 2];
extern "C" int wr_debug_timeline_hw(unsigned int *out, size_t n)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ddc_hw), n * sizeof(unsigned int));
}
extern "C" int wr_debug_timeline_reset(void)
{
	void *p = nullptr;
	if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_ddc_tl)) == hipSuccess)
		(void)hipMemset(p, 0, sizeof(g_ddc_tl));
	if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_ddc_rt)) == hipSuccess)
		(void)hipMemset(p, 0, sizeof(g_ddc_rt));
	if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_ddc_hw)) == hipSuccess)
		(void)hipMemset(p, 0, sizeof(g_ddc_hw));
	return 0;
}
#define TL(slot) do { if (lane == 0 && wid < 16384u && (slot) < TL_SLOTS) { g_ddc_tl[wid * TL_SLOTS + (slot)] = __builtin_amdgcn_s_memtime(); \
	if ((slot) == 0 || (slot) == 11) g_ddc_rt[wid * 2u + ((slot) ? 1u : 0u)] = __builtin_amdgcn_s_memrealtime(); \
	if ((slot) == 0) { g_ddc_hw[wid * 2u] = __builtin_amdgcn_s_getreg((31 << 11) | 4); g_ddc_hw[wid * 2u + 1u] = __builtin_amdgcn_s_getreg((31 << 11) | 20); } } } while (0)
/* the post-stage tenants of the same launch: wave start, stage phase done, filter done, wave end (of
 * the last tile of a run) */
__device__ unsigned long long g_post_tl[8192 * 4];
extern "C" int wr_debug_timeline_post(unsigned long long *out, size_t n)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_post_tl), n * sizeof(unsigned long long));
}
#define TLP(slot) do { const unsigned int pw_ = (bx + g * (A.ntiles + 1u)) * NROW + row; if (lane == 0 && pw_ < 8192u) g_post_tl[pw_ * 4u + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TL(slot) do { } while (0)
#define TLP(slot) do { } while (0)
#endif

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

/* LowPass::process (dsp/lowpass.cxx:131-162): out[k][c] = sum_j coeff[L-1-j] *
 * block[(k*D + j)][c], accumulated oldest sample first starting from 0.0f.
 * block = [L-1 history frames][input]; one thread per output float.  L = _firLength is a
 * run-time value here (the reference compiles in 64, lowpass.cxx:38-39). */
__global__ void __launch_bounds__(256)
k_fir(const float *__restrict__ in, size_t outfloats, unsigned int channels, unsigned int decim,
      unsigned int L, const float *__restrict__ coeff, const float *__restrict__ hist, float *__restrict__ out)
{
	__shared__ float taps[WR_FIR_MAX];
	for (unsigned int j = threadIdx.x; j < L; j += blockDim.x)
		taps[j] = coeff[j];
	__syncthreads();
	const unsigned int nh = L - 1u;
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < outfloats; o += stride) {
		size_t k = o / channels;
		unsigned int c = (unsigned int)(o - k * channels);
		size_t first = k * decim;               /* index into [hist|in] in frames */
		float acc = 0.0f;
		for (unsigned int j = 0; j < L; ++j) {
			size_t f = first + j;
			float x = (f < nh) ? hist[f * channels + c] : in[(f - nh) * channels + c];
			acc = acc + taps[L - 1u - j] * x;
		}
		out[o] = acc;
	}
}

/* next history = last L-1 frames of [hist|in] (dsp/lowpass.cxx:138-142); written to
 * scratch first because for short blocks source and destination overlap */
__global__ void k_hist_build(const float *__restrict__ in, size_t nframes, unsigned int channels, unsigned int nh,
                             const float *__restrict__ hist, float *__restrict__ scratch)
{
	unsigned int total = nh * channels;
	for (unsigned int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
		size_t f = nframes + e / channels;      /* frame index in [hist|in] */
		unsigned int c = e % channels;
		scratch[e] = (f < nh) ? hist[f * channels + c] : in[(f - nh) * channels + c];
	}
}

__global__ void k_copy_f32(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		dst[i] = src[i];
}

/* atan2f(y, x) / (2 pi), the FM discriminator's angle in cycles (dsp/demodulator.cxx:96-99:
 * atan2f(..) / M_PI / 2.0).  The library atan2f costs about 150 instructions a frame, which made
 * the demodulator compute-bound; this one is ~30: one reciprocal-multiply into [0, 1], an odd
 * minimax polynomial (degree 17, fitted to 6e-9; 1.0e-7 rad worst case evaluated in float),
 * and the octant/quadrant reflections done in cycles, where the constants 1/4 and 1/2 are
 * exact.  Signed zeros follow atan2f: (+0, -0) -> +1/2, (-0, +0) -> -0.  Worst difference
 * from the reference's correctly rounded atan2f path: 6e-8 cycles (tests allow 2.4e-7, two
 * ulp of the result range). */
__device__ __forceinline__ float fm_angle_cycles(float y, float x)
{
	const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
	const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
	/* mn / mx with one v_rcp_f32 (1 ulp) instead of the IEEE division sequence; operands are
	 * brought into the normal range first so that a vanishing signal cannot overflow the
	 * reciprocal (0/0 -> 0: atan2f(+-0, +-0) only depends on the signs) */
	const float sc = (mx < 0x1p-100f) ? 0x1p100f : 1.0f;
	const float r = (mx == 0.0f) ? 0.0f : (mn * sc) * __builtin_amdgcn_rcpf(mx * sc);
	const float t = r * r;
	float p = 0.002456712769344449f;
	p = __builtin_fmaf(p, t, -0.014401308260858059f);
	p = __builtin_fmaf(p, t, 0.03978113830089569f);
	p = __builtin_fmaf(p, t, -0.07234849780797958f);
	p = __builtin_fmaf(p, t, 0.1049894168972969f);
	p = __builtin_fmaf(p, t, -0.14161227643489838f);
	p = __builtin_fmaf(p, t, 0.19985906779766083f);
	p = __builtin_fmaf(p, t, -0.33332598209381104f);
	p = __builtin_fmaf(p, t, 0.9999998807907104f);
	float c = (p * r) * 0.15915494309189533577f;        /* atan(r) / (2 pi), in [0, 1/8] */
	if (ay > ax)
		c = 0.25f - c;
	if (__builtin_signbitf(x))
		c = 0.5f - c;
	return __builtin_copysignf(c, y);
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
		/* atan2f(Re, Im) -- the reference's argument order -- then / M_PI / 2.0 */
		return fm_angle_cycles(ii, qq);
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

/* The same, 16 bytes per lane and four such loads in flight per lane: a wave asks for 1 KiB at a time.  `in` may be
 * page-locked HOST memory read over PCIe (wr_u8_to_f32_from_host): with one byte per lane every wave-load was a 64-byte read
 * request and the link ran at 29 GB/s (279 us for the 8 MB of a 4 M-frame block, r03 rocprofv3; 170 us = 47 GB/s now).
 * PCIe-bound work needs few waves: the launch is kept to U8_X16_WGS workgroups (2 MB in flight), because a grid that fills
 * the chip with waves parked on PCIe reads leaves no wave slots to the kernels of the block before, which run beside it on
 * the device's own stream (a 32 MB device-to-device copy beside the full grid: 164 us instead of 13).
 * `n16` = count / 16, both pointers 16-byte aligned. */
#define U8_X16_WGS 128u
__global__ void __launch_bounds__(256) k_u8_to_f32_x16(const uint4 *__restrict__ in, float4 *__restrict__ out, size_t n16)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n16; i0 += 4 * stride) {
		uint4 v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u)
			if (i0 + u * stride < n16)
				v[u] = in[i0 + u * stride];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const size_t i = i0 + u * stride;
			if (i >= n16)
				break;
			const unsigned int w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				float4 o;
				o.x = ((float)(w[q] & 255u) - 128.0f) / 128.0f;
				o.y = ((float)((w[q] >> 8) & 255u) - 128.0f) / 128.0f;
				o.z = ((float)((w[q] >> 16) & 255u) - 128.0f) / 128.0f;
				o.w = ((float)(w[q] >> 24) - 128.0f) / 128.0f;
				out[4 * i + q] = o;
			}
		}
	}
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
		/* SPLIT: 32 interleaved copies of each table (see k_tuner_ddc); ROTATE looks the LO up
		 * only a few times per output frame and keeps one copy (a 64-lane gather from the
		 * global table costs the CU's L1 about a cycle per lane, an LDS gather a few cycles
		 * in all) */
		constexpr unsigned int SH = (NCO == WR_NCO_SPLIT) ? 5u : 0u;
		const v2f a = hi_l[(P >> 24) << SH];              /* cis(2*pi*coarse/256) */
		const v2f b = lo_l[((P >> 16) & 255u) << SH];      /* cis(2*pi*fine/65536) */
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

/* ROTATE NCO.  The reference's LO index advances by S = step >> 16 table entries per input
 * frame, plus one whenever the 16 fraction bits of the left-aligned phase carry
 * (downconverter.cxx:100-103: index = phase >> 15, phase += phase_step).  So the LO of frame m
 * is the LO of frame m-1 turned by one of just two angles, rot[0] = cis(2 pi S / 65536) or
 * rot[1] = cis(2 pi (S+1) / 65536), and
 *
 *     sum_m u[m] conj(LO[m])  =  conj(LO[n]) * ( ... ((u[n-63] r[n-62] + u[n-62]) r[n-61] + ...) r[n] + u[n] )
 *
 * with r[m] the turn INTO frame m: a Horner recurrence of one integer add-with-carry, two
 * selects and four FMAs per tap, no table access at all.  Which table entry every frame is
 * mixed with is exactly the reference's; only float rounding differs: a product of turns
 * stands where the reference has one table entry.  Two things keep that product honest:
 *   - the turns are cis(2 pi S / 65536) evaluated on the exact angle and rounded once (the
 *     device's `table_turn`), NOT pairs of entries of the reference's sinf table, whose float
 *     argument rounding (2.4e-7) makes |cos + i sin| != 1 -- harmless for one lookup, but a turn
 *     is applied dozens of times in a row (measured: 5e-7 -> 4.5e-8 worst error);
 *   - the 64 taps can be cut into ROT_Q segments of ROT_SEG taps, each its own recurrence
 *     closed with the LO value of its last frame (an anchor, one LDS table lookup): fewer turns
 *     between a frame and its anchor, and ROT_Q independent chains.  With exact turns one
 *     segment is enough (see ROT_SEG).
 *
 * rot_index(): table index of the turn into frame m >= 1 of the current block, for phase p0
 * at frame 0. */
__device__ __forceinline__ unsigned int rot_index(unsigned int p0, unsigned int st, unsigned int m)
{
	const unsigned int P = p0 + m * st;
	return (st >> 16) + (((P & 0xFFFFu) < (st & 0xFFFFu)) ? 1u : 0u);
}

__device__ __forceinline__ v2f rot_into(unsigned int p0, unsigned int st, unsigned int m,
                                        const float *__restrict__ table)
{
	const unsigned int idx = rot_index(p0, st, m);
	v2f r;
	r.y = table[idx & 0xFFFFu];
	r.x = table[(idx + 16384u) & 0xFFFFu];
	return r;
}

/* one Horner step A := A * r + u, and the closing multiplication by conj(LO); spelled once so
 * that the fast and the block-boundary path round identically */
__device__ __forceinline__ void horner_step(v2f &A, float rc, float rs, v2f u)
{
	const float tr = __builtin_fmaf(-A.y, rs, u.x);
	const float ti = __builtin_fmaf(A.x, rs, u.y);
	const float ar = __builtin_fmaf(A.x, rc, tr);
	const float ai = __builtin_fmaf(A.y, rc, ti);
	A.x = ar;
	A.y = ai;
}

/* y += A * conj(LO) */
__device__ __forceinline__ void horner_close(v2f &y, v2f A, v2f cs)
{
	const float yr = __builtin_fmaf(A.y, cs.y, __builtin_fmaf(A.x, cs.x, y.x));
	const float yi = __builtin_fmaf(-A.x, cs.y, __builtin_fmaf(A.y, cs.x, y.y));
	y.x = yr;
	y.y = yi;
}

/* s_setprio with a run-time (wave-uniform) level.  The SIMD's issue arbiter goes by priority, then
 * AGE: at equal priority the older waves of a SIMD take every VALU slot their tap loops can use and
 * the youngest wave does not even get its prologue issued until they stall (measured with
 * s_memtime stamps, C2: its state loads took 10 us, and it then ran its units alone on the SIMD at a
 * third of the issue rate after the others had finished -- a 40 % tail).  So a wave's priority
 * falls as it gets through its units: whoever is behind goes first. */
__device__ __forceinline__ void wave_prio(unsigned int level)
{
#ifndef DDC_NO_PRIO
	if (level >= 3u)
		__builtin_amdgcn_s_setprio(3);
	else if (level == 2u)
		__builtin_amdgcn_s_setprio(2);
	else if (level == 1u)
		__builtin_amdgcn_s_setprio(1);
	else
		__builtin_amdgcn_s_setprio(0);
#endif
}

#ifndef ROT_SEG
#define ROT_SEG 64                       /* measured on MI355X, C2: 16 -> 4.5e-8 / 40.9 us, 32 -> 6.7e-8 / 39.5 us,
                                            64 -> 1.2e-7 / 38.4 us (worst |IQ - bit-exact path| on +-0.4 signals / kernel) */
#endif
#define ROT_Q   (WR_FIR_LENGTH / ROT_SEG)
#define SLOW_CH 4                          /* taps per memory round of the block-boundary paths */

/* SPLIT NCO: issue the two LDS gathers for left-aligned phase P.  LDS byte address =
 * table base | index << 8 | (lane & 31) << 3; the index byte of P is dropped straight
 * into byte 1 of a copy of the base address with one SDWA move. */
__device__ __forceinline__ void gather_split(unsigned int P, unsigned int &ah, unsigned int &al,
                                             v2f &a, v2f &b)
{
	/* ah/al hold the base address; only their byte 1 is rewritten (UNUSED_PRESERVE) */
	asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3"
	    : "+v"(ah) : "v"(P));
	asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2"
	    : "+v"(al) : "v"(P));
	a = *(const lds_v2f *)(uintptr_t)ah;                         /* cis(2*pi*coarse/256) */
	b = *(const lds_v2f *)(uintptr_t)al;                         /* cis(2*pi*fine/65536) */
}

/*
 * k_tuner_ddc: DownConverter::process + channel LowPass::process for every channel
 * of a tuner (dsp/downconverter.cxx:91-114 feeding dsp/lowpass.cxx:131-162).
 *
 *   work unit  = (k, g): channel-rate output frame k for the 64 channel slots of
 *                lane group g.  One wave per unit; a wave of the persistent grid keeps one
 *                lane group and walks k with a fixed stride (launch geometry: DdcGeom).
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
/* One IQ frame of the tuner block: float32 pairs, or the RTL-SDR byte format converted
 * with the reference's rule (u8 - 128) / 128 (io/rtlsdrtuner.cxx:106) in the load stage --
 * a quarter of the bytes over PCIe and out of HBM, same values as converting first. */
__device__ __forceinline__ float2 input_frame(const float2 *__restrict__ cur, const uchar2 *__restrict__ cur_u8,
                                              size_t n)
{
	if (cur_u8) {
		const uchar2 b = cur_u8[n];
		return make_float2(((float)b.x - 128.0f) / 128.0f, ((float)b.y - 128.0f) / 128.0f);
	}
	return cur[n];
}

/*
 * k_tuner_post<D2>: Demodulator::process and the audio LowPass::process in one pass
 * (dsp/demodulator.cxx:77-115 feeding dsp/lowpass.cxx:131-162), for the common small audio
 * decimations and when nobody asked for the demodulator's own output (wr_tuner_keep_stages).
 * The demod rows of a tile are computed from the channel IQ rows straight into LDS: the
 * 4 B/frame/channel demod array makes no round trip through HBM and one launch goes away.
 * Rows a tile shares with its neighbour (the FIR overlap) are demodulated by both.
 *
 *   blockIdx.x <  ntiles : POST_TK audio frames x 64 slots.
 *        stage  : all 8 waves; thread (row, lane) takes rows row, row+8, ... of the tile.  Row rr
 *                 of [history | current] is dem_hist[rr] for rr < 63, else the demodulated
 *                 channel frame rr - 63 (its predecessor re-read from L1/L2, or prev_iq).
 *        filter : POST_TK / POST_B waves; thread (lane, group of POST_B consecutive frames):
 *                 the lane's 64 taps in registers, each staged row read from LDS ONCE and
 *                 applied to every frame of the group that uses it.  D2 is a template
 *                 parameter so that which tap meets which row is resolved at compile time.
 *                 Per frame the products are added oldest-first, unfused, as lowpass.cxx does.
 *   blockIdx.x == ntiles : what the block leaves behind per lane group, as k_tuner_demod does:
 *                 the last 63 demod outputs (audio filter history) and the last channel
 *                 frame (Demodulator::prev_i/q), into the other ping-pong set.
 */
#ifndef POST_TK
#define POST_TK 16u
#endif
#define POST_B 4u
#ifndef POST_LB
#define POST_LB 5u                       /* rows whose loads a wave of the stage phase has in flight together (measured
                                            at C2, us per block at 1 / 4 blocks per launch: 3 rows 34.7 / 30.5, 5 rows
                                            34.0 / 29.7, 6 rows 35.0 / 31.3, 9 rows 35.7 / 31.0 -- more rows, more
                                            registers: spills) */
#endif
#define POST_THREADS 512u
/* What leaves the audio filter for the sink: af_gain and squelch (the reference has the two fields
 * and no code behind them, receiverhandler.cxx:112,118-119,127), then the sink's scale.
 *   af_gain : a factor applied in float;
 *   squelch : the audio frame is muted (0.0f) when the mean power i*i + q*q of the d2 demodulator
 *             input frames behind it (frames k*d2 .. k*d2 + d2 - 1, summed in that order, divided by
 *             (float)d2) is below the threshold.
 * Factors of exactly 1 and thresholds of exactly 0 leave the value untouched. */
__device__ __forceinline__ float audio_out(float v, const float *__restrict__ gain, const float *__restrict__ squelch,
                                           const float2 *__restrict__ chan_iq, unsigned int slots, unsigned int so,
                                           size_t k, unsigned int d2, float scale)
{
	if (squelch) {
		const float thr = squelch[so];
		if (thr > 0.0f) {
			float p = 0.0f;
			for (unsigned int i = 0; i < d2; ++i) {
				const float2 z = chan_iq[(k * d2 + i) * slots + so];
				p = p + (z.x * z.x + z.y * z.y);
			}
			if (p / (float)d2 < thr)
				v = 0.0f;
		}
	}
	if (gain) {
		const float gn = gain[so];
		if (gn != 1.0f)
			v = v * gn;
	}
	return (scale == 1.0f) ? v : v * scale;
}

__device__ __forceinline__ float post_row(const float2 *__restrict__ chan_iq, unsigned int k1, unsigned int slots,
                                          unsigned int s, int m, const float2 *__restrict__ prev_iq,
                                          const float *__restrict__ dem_hist, size_t rr,
                                          const float2 *__restrict__ chan_prev = nullptr, unsigned int hist = WR_HIST)
{
	/* `hist`: rows of history in front of the block's = the audio filter's taps - 1 (63; 127 / 255 for 128 / 256 taps) */
	if (rr < hist) {
		if (!chan_prev)
			return dem_hist[rr * slots + s];
		/* (streaming launch) history row rr = the demodulated frame k1 - 63 + rr of the block before */
		const size_t kp = (size_t)k1 - hist + rr;
		const float2 z = chan_prev[kp * slots + s];
		const float2 zp = chan_prev[(kp - 1u) * slots + s];
		return demod_one(m, z.x, z.y, zp.x, zp.y);
	}
	const size_t kk = rr - hist;
	if (kk >= k1)
		return 0.0f;                                    /* beyond the block: never read by lowpass.cxx */
	const float2 z = chan_iq[kk * slots + s];
	const float2 zp = kk ? chan_iq[(kk - 1u) * slots + s] : chan_prev ? chan_prev[((size_t)k1 - 1u) * slots + s] : prev_iq[s];
	return demod_one(m, z.x, z.y, zp.x, zp.y);
}

/* One workgroup's share of the post stage: tile `bx` < ntiles of lane group `g`, or (bx == ntiles)
 * the group's end-of-block state.  HREGS: the filter's taps live in registers, one set per lane
 * (the stand-alone kernel); otherwise each tap is read from memory once per thread, when its
 * turn comes -- 64 fewer VGPRs, for the workgroups that run this inside k_tuner_ddc beside the
 * next block's DDC.  `stage` and `tile` are LDS: [NEED][64] and
 * [POST_TK][65] floats. */
template <unsigned int D2, bool HREGS, unsigned int NSEG = 1u>
__device__ __forceinline__ void post_role(const WrPostArgs &A, unsigned int bx, unsigned int g,
                                          float *stage, float *tile, int *modes)
{
	/* An audio filter of 128 / 256 taps (r05; LowPass::_firLength, lowpass.cxx:38-39) is NSEG = 2 / 4 SEGMENTS of 64
	 * taps over one staged window of (POST_TK - 1) D2 + 64 NSEG rows: segment q is the 64-tap filter
	 * coeff[(NSEG - 1 - q) * 64 ..] over the rows starting 64 q further on, its products added -- oldest row first, as
	 * lowpass.cxx:150-158 adds all L of them -- onto what the segments before left in the accumulators.  The history in
	 * front of the block is L - 1 rows.  (NSEG is a template parameter: the 64-tap instances, the ones that ride in the
	 * DDC launches, are compiled as they always were; k_tuner_post<D2, 2 | 4> is a launch of its own behind the DDC.) */
	constexpr unsigned int HIST = WR_FIR_LENGTH * NSEG - 1u;
	constexpr unsigned int NEED = (POST_TK - 1u) * D2 + WR_FIR_LENGTH * NSEG;
	constexpr unsigned int NROW = POST_THREADS / 64u;
	static_assert(NSEG == 1u || !HREGS, "the segmented filter reads its taps when their turn comes (measured at C2, 128 / 256 taps: "
	                                    "61 / 83 us per block; a segment's 64 taps fetched into registers first: 64 / 94)");
	const float2 *__restrict__ chan_iq = (const float2 *)A.chan_iq;
	const float2 *__restrict__ prev_iq = (const float2 *)A.prev_iq;
	const float2 *__restrict__ chan_prev = (const float2 *)A.chan_prev;   /* (uniform: a kernel argument) */
	const float *__restrict__ dem_hist = A.dem_hist;
	const float *__restrict__ taps2 = A.taps2;
	const unsigned int k1 = A.k1, slots = A.slots;
	const unsigned int lane = threadIdx.x & 63u;        /* slot within the group */
	const unsigned int row = threadIdx.x >> 6;
	const unsigned int s = g * 64u + lane;
	const int m = A.mode[s];                            /* < 0: idle slot */

	if (bx == A.ntiles) {
		if (m < 0)
			return;
		const size_t first = k1;                        /* the last 63 (L - 1) rows of [history | current] */
#pragma unroll
		for (unsigned int r = row; r < HIST; r += NROW)      /* unrolled: one memory round, not eight */
			A.dem_hist_next[(size_t)r * slots + s] = post_row(chan_iq, k1, slots, s, m, prev_iq, dem_hist, first + r, chan_prev, HIST);
		if (row == 0)
			((float2 *)A.prev_next)[s] = k1 ? chan_iq[(size_t)(k1 - 1u) * slots + s] : prev_iq[s];
		return;
	}

	TLP(0);
	/* the filter waves fetch their taps first: the latency hides behind the stage phase */
	float h[HREGS ? WR_FIR_LENGTH : 1];
	if (HREGS && row < POST_TK / POST_B) {
#pragma unroll
		for (int j = 0; j < (HREGS ? WR_FIR_LENGTH : 1); ++j)
			h[j] = taps2[(size_t)j * slots + s];
	}
	/* A.run consecutive tiles per workgroup: the last NEED - FRESH staged rows of a tile are the
	 * first rows of the next one and stay in LDS (moved down), so that only a run's first tile
	 * demodulates its 64 - D2 rows of overlap a second time (one tile per workgroup: 139 rows
	 * demodulated for every 80 new ones at D2 = 5) */
	if (row == 0)
		modes[lane] = m;                                /* for the audio write: no memory round trip there */
	constexpr unsigned int FRESH = POST_TK * D2;
	constexpr unsigned int CARRY = NEED - FRESH;
	constexpr unsigned int CARRY_PER = (CARRY * 64u + POST_THREADS - 1u) / POST_THREADS;
	for (unsigned int tt = 0; tt < A.run; ++tt) {
		const unsigned int tl = bx * A.run + tt;
		if (tl >= A.tiles)
			break;
		const size_t kbase = (size_t)tl * POST_TK;
		const size_t r0 = kbase * D2;
		unsigned int first = 0;
		if (tt) {
			/* (the barrier that ended the previous tile's filter phase is behind us: nobody reads
			 * the stage any more) */
			float keep[CARRY_PER];
#pragma unroll
			for (unsigned int i = 0; i < CARRY_PER; ++i) {
				const unsigned int e = threadIdx.x + i * POST_THREADS;
				keep[i] = (e < CARRY * 64u) ? stage[FRESH * 64u + e] : 0.0f;
			}
			__syncthreads();                            /* source and destination overlap when FRESH < CARRY */
#pragma unroll
			for (unsigned int i = 0; i < CARRY_PER; ++i) {
				const unsigned int e = threadIdx.x + i * POST_THREADS;
				if (e < CARRY * 64u)
					stage[e] = keep[i];
			}
			first = CARRY;
		}
		{
			/* a run of consecutive rows per wave: each channel frame is loaded once and stays in a
			 * register as the next row's predecessor */
			const unsigned int per = (NEED - first + NROW - 1u) / NROW;
			const unsigned int beg = first + row * per;
			const unsigned int end = (beg + per < NEED) ? beg + per : NEED;
			if (r0 > HIST && r0 + NEED <= (size_t)k1 + HIST) {
				/* every row is a frame of this block and so is its predecessor (all tiles but a
				 * block's first and last): the loads of POST_LB rows go out together, nothing to
				 * decide per row -- the stage phase is a chain of memory round trips, and one per
				 * row made the post workgroups the last to finish */
#ifdef POST_SMALL_READS
				/* (development, results wrong: every tile reads the same few hundred rows -- what the channel IQ's way back
				 * from memory costs, profiles/r06_power.txt) */
				const float2 *__restrict__ src = chan_iq + ((r0 - HIST) & 127u) * slots + s;
#else
				const float2 *__restrict__ src = chan_iq + (r0 - HIST) * slots + s;
#endif
				for (unsigned int r = beg; r < end; r += POST_LB) {
					float2 z[POST_LB + 1u];
					z[0] = src[((size_t)r - 1u) * slots];
#pragma unroll
					for (unsigned int i = 0; i < POST_LB; ++i) {
						const unsigned int ri = (r + i < end) ? r + i : end - 1u;
						z[i + 1u] = src[(size_t)ri * slots];
					}
#pragma unroll
					for (unsigned int i = 0; i < POST_LB; ++i)
						if (r + i < end)
							stage[(r + i) * 64u + lane] = (m >= 0) ? demod_one(m, z[i + 1u].x, z[i + 1u].y, z[i].x, z[i].y) : 0.0f;
				}
			} else {
				float2 zp = make_float2(0.0f, 0.0f);
				if (m >= 0 && beg < end) {
					const size_t rr = r0 + beg;
					if (rr == HIST)
						zp = chan_prev ? chan_prev[((size_t)k1 - 1u) * slots + s] : prev_iq[s];
					else if (rr > HIST && rr - HIST - 1u < k1)
						zp = chan_iq[(rr - HIST - 1u) * slots + s];
					else if (rr < HIST && chan_prev)
						zp = chan_prev[((size_t)k1 - HIST - 1u + rr) * slots + s];
				}
#pragma unroll 6
				for (unsigned int r = beg; r < end; ++r) {
					const size_t rr = r0 + r;
					float v = 0.0f;
					if (m >= 0) {
						if (rr < HIST && chan_prev) {
							/* (streaming launch) a history row = a demodulated frame of the block before, made again */
							const float2 z = chan_prev[((size_t)k1 - HIST + rr) * slots + s];
							v = demod_one(m, z.x, z.y, zp.x, zp.y);
							zp = z;                                 /* (row 62: the predecessor of the block's first frame) */
						} else if (rr < HIST) {
							v = dem_hist[rr * slots + s];
							if (rr + 1u == HIST)
								zp = prev_iq[s];                    /* the next row is the block's first frame */
						} else if (rr - HIST < k1) {
							const float2 z = chan_iq[(rr - HIST) * slots + s];
							v = demod_one(m, z.x, z.y, zp.x, zp.y);
							zp = z;
						}
					}
					stage[r * 64u + lane] = v;
				}
			}
		}
		__syncthreads();
		TLP(1);
		if (row < POST_TK / POST_B) {
			float acc[POST_B];
#pragma unroll
			for (unsigned int o = 0; o < POST_B; ++o)
				acc[o] = 0.0f;
			if (HREGS) {
				const float *x = stage + (row * POST_B * D2) * 64u + lane;
				/* rows outermost: each staged row is read from LDS once and meets the tap of every
				 * frame of the group that uses it (which tap: resolved at compile time) */
#pragma unroll
				for (unsigned int r = 0; r < (POST_B - 1u) * D2 + WR_FIR_LENGTH; ++r) {
					const float xv = x[r * 64u];
#pragma unroll
					for (unsigned int o = 0; o < POST_B; ++o) {
						if (r >= o * D2 && r - o * D2 < WR_FIR_LENGTH)
							acc[o] = acc[o] + h[HREGS ? WR_FIR_LENGTH - 1u - (r - o * D2) : 0] * xv;
					}
				}
			} else {
#pragma unroll 1
			for (unsigned int seg = 0; seg < NSEG; ++seg) {     /* (oldest rows first) */
				const float *x = stage + (row * POST_B * D2 + seg * WR_FIR_LENGTH) * 64u + lane;
				const unsigned int tap0 = (NSEG - 1u - seg) * WR_FIR_LENGTH;
				/* taps outermost, newest tap last (so that every frame still adds its products
				 * oldest row first): one memory read per tap, shared by the frames of the group; the
				 * rows come from LDS once per frame */
				if (g < 64u && ((A.uni2 >> g) & 1ull)) {
					/* every channel of the lane group has the same audio filter (radio.cxx:78-79: the
					 * usual case): its taps come through the scalar cache into SGPRs -- no vector
					 * memory round trips behind the next block's DDC gathers in this phase at all */
					const WR_CONSTANT float *tu = (const WR_CONSTANT float *)(A.taps2u + (size_t)g * (WR_FIR_LENGTH * NSEG) + tap0);
#pragma unroll
					for (int j = WR_FIR_LENGTH - 1; j >= 0; --j) {
						const float hj = tu[j];
#pragma unroll
						for (unsigned int o = 0; o < POST_B; ++o)
							acc[o] = acc[o] + hj * x[(o * D2 + (WR_FIR_LENGTH - 1u - (unsigned int)j)) * 64u];
					}
				} else {
#pragma unroll 16
					for (int j = WR_FIR_LENGTH - 1; j >= 0; --j) {
						const float hj = taps2[((size_t)tap0 + j) * slots + s];
#pragma unroll
						for (unsigned int o = 0; o < POST_B; ++o)
							acc[o] = acc[o] + hj * x[(o * D2 + (WR_FIR_LENGTH - 1u - (unsigned int)j)) * 64u];
					}
				}
			}                                           /* (segments) */
			}
#pragma unroll
			for (unsigned int o = 0; o < POST_B; ++o)
				tile[(row * POST_B + o) * 65u + lane] = acc[o];
		}
		__syncthreads();
		TLP(2);
		/* transposed write: POST_TK consecutive frames of one slot per POST_TK threads */
		for (unsigned int e = threadIdx.x; e < 64u * POST_TK; e += POST_THREADS) {
			const unsigned int sl = e / POST_TK, kk = e - sl * POST_TK;
			const unsigned int so = g * 64u + sl;
			const size_t k = kbase + kk;
			if (k < A.k2 && modes[sl] >= 0) {
				const float v = audio_out(tile[kk * 65u + sl], A.gain, A.squelch, chan_iq, slots, so, k, D2, A.scale);
				A.audio[(size_t)so * A.k2max + k] = v;
				if (A.audio_host)                       /* (uniform: a kernel argument) */
					A.audio_host[(size_t)so * A.host_stride + k] = v;
			} else if (k < A.k2 && A.audio_host) {
				/* an idle slot below slots_used: the device array holds zeros there (it is never written), the ring
				 * slot is page-locked memory that nobody cleared -- a consumer that walks every row reads zeros too */
				A.audio_host[(size_t)so * A.host_stride + k] = 0.0f;
			}
		}
		TLP(3);
	}
}

template <unsigned int D2, unsigned int NSEG = 1u>
__global__ void __launch_bounds__(POST_THREADS)
k_tuner_post(WrPostArgs A)
{
	constexpr unsigned int NEED = (POST_TK - 1u) * D2 + WR_FIR_LENGTH * NSEG;
	if constexpr (NSEG == 1u) {
		__shared__ float stage[NEED * 64u];
		__shared__ float tile[POST_TK * 65u];
		__shared__ int modes[64];
		post_role<D2, NSEG == 1u, NSEG>(A, blockIdx.x, blockIdx.y, stage, tile, modes);
	} else {
		/* (the window of a long filter: up to 406 rows = 104 KB at D2 = 10 and 256 taps -- dynamic LDS, launch_post) */
		extern __shared__ float post_lds[];
		float *stage = post_lds;
		float *tile = stage + NEED * 64u;
		int *modes = (int *)(tile + POST_TK * 65u);
		post_role<D2, NSEG == 1u, NSEG>(A, blockIdx.x, blockIdx.y, stage, tile, modes);
	}
}

/* LDS plan of k_tuner_ddc: [0, 128 KiB) the two replicated NCO tables (SPLIT only),
 * then one private 2 x 512 B sample window per wave (double buffered across units). */
#define DDC_TABLE_BYTES   (2u * WR_SPLIT_N * 32u * 8u)
#ifndef DDC_WAVES
#define DDC_WAVES         16u
#endif
#ifndef DDC_ROTATE_WGS_PER_CU
#define DDC_ROTATE_WAVES      8u
#define DDC_ROTATE_WGS_PER_CU 4u
#endif
#define DDC_LDS_BYTES     (DDC_TABLE_BYTES + DDC_WAVES * 2u * 512u)
#ifndef DDC_LTAPS_SMALL
#define DDC_LTAPS_SMALL 1                  /* the per-lane-taps ROTATE variant in 8-wave workgroups, four per CU, like the fast one */
#endif
#ifndef DDC_DEAL_WAYS
#define DDC_DEAL_WAYS 2u                   /* measured at C2, us per block at 1 / 4 blocks per launch: in order (and as
                                              many workgroups as make the units come out even) 37.6 / 32.8, 2 ways
                                              36.8 / 32.0, 4 ways 37.0 / 32.0, 8 ways 38.5 / 32.0 */
#endif

/* PD2 > 0: workgroups n_ddc.. of the grid run the post stage (audio decimation PD2) of the
 * PREVIOUS block -- see post_role and wr_capi.hip: the two have nothing to do with each other
 * except that they share the CUs, the post stage's latency-bound phases filling in between the
 * DDC's arithmetic.  Their dependency is the kernel boundary before this launch. */
/* r03: with DDC_ROLES the ROTATE kernel with folded taps splits its workgroups into roles:
 *   [0, n_ddc)               the persistent DDC waves: output frames kslow .. k1-1, whose 64-frame windows lie
 *                            inside this block -- ONE lean loop in the kernel itself, nothing else in its registers
 *   [n_ddc, n_ddc + n_bnd)   one wave per (lane group, frame k < kslow): the first ceil(63 / D1) frames of the
 *                            block, whose windows reach into the previous one (history rows, kept turns)
 *   [n_ddc + n_bnd, ...)     the previous block's post stage (PD2 != 0)
 * Everything but the lean loop is ddc_body, which the lean kernel reaches through a real call (its register
 * allocation is its own).  Kernels without roles (kslow = n_bnd = 0) ARE ddc_body: one generic loop. */
#ifndef DDC_ROLES
#define DDC_ROLES 1
#endif
#ifndef DDC_NG2
#define DDC_NG2 1                          /* two lane groups per wave of the lean loop where the launch allows */
#endif
#define DDC_PARAMS \
	const float2 *__restrict__ cur, const uchar2 *__restrict__ cur_u8, const float2 *__restrict__ hist, \
	float2 *__restrict__ hist_next, size_t nframes, size_t k1, unsigned int d1, unsigned int slots, unsigned int groups, \
	const unsigned int *__restrict__ phase, const unsigned int *__restrict__ step, \
	const float2 *__restrict__ hist_cs, const int *__restrict__ flags, \
	unsigned int *__restrict__ phase_next, float2 *__restrict__ hist_cs_next, \
	const float2 *__restrict__ hist_lo, float2 *__restrict__ hist_lo_next, \
	const float *__restrict__ taps1, const float4 *__restrict__ rot, const float *__restrict__ taps1u, \
	const int *__restrict__ tapsel, unsigned int kmax, float2 *__restrict__ chan_iq, \
	const float *__restrict__ table, const float2 *__restrict__ hi_cs, const float2 *__restrict__ lo_cs, \
	unsigned int n_ddc, const WrPostArgs &post, unsigned long long gmap0, unsigned long long gmap1, int whole, \
	unsigned int kslow, unsigned int n_bnd, unsigned int seek_on, unsigned int seek_lo
#define DDC_PASS \
	cur, cur_u8, hist, hist_next, nframes, k1, d1, slots, groups, phase, step, hist_cs, flags, phase_next, hist_cs_next, \
	hist_lo, hist_lo_next, taps1, rot, taps1u, tapsel, kmax, chan_iq, table, hi_cs, lo_cs, n_ddc, post, gmap0, gmap1, \
	whole, kslow, n_bnd, seek_on, seek_lo
/* the channel's phase at the block's first frame: kept from the block before, or -- the launch right after a
 * wr_tuner_seek (`seek_on`; the histories it reads are all-zero sets then) -- frame * step mod 2^32 in closed form
 * (downconverter.cxx:103), `seek_lo` = the frame's low 32 bits */
#ifdef DDC_NO_LAZY_SEEK                          /* (timing comparisons only) */
#define DDC_PHASE(s_) (phase[s_])
#else
#define DDC_PHASE(s_) (seek_on ? step[s_] * seek_lo : phase[s_])
#endif
#define DDC_ROLE_ALL      0                /* no roles: every frame, the state roll, the riding post stage */
#define DDC_ROLE_BOUNDARY 1                /* one block-boundary unit per wave (or the post stage, by workgroup index) */
#define DDC_ROLE_ROLL     2                /* the end-of-block state roll only */

template <int NCO, bool UTAPS, unsigned int PD2>
__device__ __forceinline__ void
ddc_body(DDC_PARAMS, v2f *lds, const int role)
{
	if (PD2 != 0u && role != DDC_ROLE_ROLL && blockIdx.x >= n_ddc + n_bnd) {
		/* latency-bound tenants: a chain of loads, barriers and short bursts of arithmetic */
		wave_prio(3u);
		constexpr unsigned int NEED = (POST_TK - 1u) * (PD2 ? PD2 : 1u) + WR_FIR_LENGTH;
		const unsigned int idx = blockIdx.x - n_ddc - n_bnd;
		if (threadIdx.x >= POST_THREADS)
			return;                                     /* (workgroups of the per-lane-taps variant have 16 waves) */
		float *stage = (float *)lds;
		post_role<(PD2 ? PD2 : 1u), false>(post, idx % (post.ntiles + 1u), idx / (post.ntiles + 1u), stage,
		                                    stage + NEED * 64u, (int *)(stage + NEED * 64u + POST_TK * 65u));
		return;
	}
	const unsigned int lane = threadIdx.x & 63u;
	const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned int waves_per_wg = blockDim.x >> 6;
	const unsigned int wid = blockIdx.x * waves_per_wg + wave;
	TL(0);                                                   /* wave starts */
	wave_prio(3u);                                           /* the prologue's loads go out at once */

	if (NCO == WR_NCO_SPLIT) {
		for (unsigned int e = threadIdx.x; e < WR_SPLIT_N * 32u; e += blockDim.x) {
			const float2 hv = hi_cs[e >> 5], lv = lo_cs[e >> 5];
			lds[e] = (v2f){hv.x, hv.y};
			lds[WR_SPLIT_N * 32u + e] = (v2f){lv.x, lv.y};
		}
		__syncthreads();
	} else if (NCO == WR_NCO_ROTATE) {
		for (unsigned int e = threadIdx.x; e < WR_SPLIT_N; e += blockDim.x) {
			const float2 hv = hi_cs[e], lv = lo_cs[e];
			lds[e] = (v2f){hv.x, hv.y};
			lds[WR_SPLIT_N + e] = (v2f){lv.x, lv.y};
		}
		__syncthreads();
	}
	const v2f *hi_l = (NCO == WR_NCO_ROTATE) ? lds : lds + (lane & 31u);
	const v2f *lo_l = (NCO == WR_NCO_ROTATE) ? lds + WR_SPLIT_N : lds + WR_SPLIT_N * 32u + (lane & 31u);
	/* the same two bases as LDS byte addresses (the dynamic LDS block starts at 0: it is
	 * the kernel's only LDS, so byte 1 of both is free for the table index) */
	const unsigned int a_hi = (unsigned int)(uintptr_t)hi_l;
	const unsigned int a_lo = (unsigned int)(uintptr_t)lo_l;
	/* this wave's sample windows: 64 x float2 each, [buffer][tap set][tap].  UTAPS: the window holds
	 * tap * sample, once per distinct channel filter of the lane group (`kmax` of them at most in
	 * this launch, WR_TAPSETS the limit); a lane reads the copy made with ITS filter. */
	const unsigned int nset = UTAPS ? kmax : 1u;
	v2f *win = lds + (NCO == WR_NCO_SPLIT ? DDC_TABLE_BYTES / 8u : NCO == WR_NCO_ROTATE ? 2u * WR_SPLIT_N : 0u)
	           + wave * (128u * nset);

	/* the tuner's next input history = last 63 frames of [hist | cur] (lowpass.cxx:138-142
	 * keeps them per LowPass; here once per tuner).  It goes to the OTHER history buffer,
	 * so no reader of `hist` in this launch is disturbed. */
	/* (`whole`: this launch covers every lane group of the rate group, or is the first of the
	 * two launches that between them do -- it then also rolls the state of ALL channels) */
	if (whole && role != DDC_ROLE_ROLL && blockIdx.x == 0 && wave == 0 && lane < WR_HIST) {
		const size_t f = nframes + lane;            /* frame index in [hist | cur] */
		hist_next[lane] = (f < WR_HIST) ? hist[f] : input_frame(cur, cur_u8, f - WR_HIST);
	}

	/* Units (g, k) are dealt so that a wave keeps ONE lane group for its whole life: wave w of
	 * the grid takes g = w mod groups and walks k = w / groups, + waves_per_group, ...  Its
	 * per-channel state (phase, step, taps, turns) is loaded once, before the loop -- a
	 * reload costs two dependent memory latencies, and with a plain round-robin over all
	 * units it came every few units.  Waves with the same k (neighbours in a workgroup) read
	 * the same window at about the same time: one HBM read, the rest L1/L2 hits. */
	const unsigned int k1u = (unsigned int)k1;
	const unsigned int nwaves = n_ddc * waves_per_wg;
	TL(1);                                                   /* tables in LDS, state rolled */
	/* ROTATE with per-lane taps keeps the taps of ONE lane group in LDS (16 KiB) instead of 64
	 * registers per lane: there a whole workgroup keeps to one group */
	constexpr bool LTAPS = (NCO == WR_NCO_ROTATE) && !UTAPS;
	unsigned int wpg = LTAPS ? (n_ddc / groups) * waves_per_wg : nwaves / groups;   /* >= 1: see the launcher */
	/* a wave of the block-boundary workgroups: ONE unit, frame k < kslow of one lane group */
	const bool bnd = role == DDC_ROLE_BOUNDARY;
	const unsigned int ub = bnd ? (blockIdx.x - n_ddc) * waves_per_wg + wave : 0u;
	/* `groups` lane groups take part in this launch; which ones: 16 one-byte entries */
	const unsigned int gl = bnd ? ub % groups : LTAPS ? blockIdx.x % groups : wid % groups;
	const unsigned int g = (unsigned int)(((gl < 8u ? gmap0 : gmap1) >> ((gl & 7u) * 8u)) & 255u);
	unsigned int k = bnd ? ub / groups : LTAPS ? (blockIdx.x / groups) * waves_per_wg + wave : wid / groups;
	if (role == DDC_ROLE_ROLL) {
		k = k1u;                                         /* no units: straight to the end-of-block state */
	} else if (bnd) {
		if (k >= kslow)
			k = k1u;
		wpg = k1u;                                       /* one unit, then out of the loop */
	} else if (k >= wpg) {
		k = k1u;                                         /* the few waves left over stay idle */
	} else if (!LTAPS && DDC_DEAL_WAYS > 1u) {
		/* A wave that starts at k does ceil((k1 - k) / wpg) units: the low starts one more than the
		 * high ones.  Neighbouring waves take their starts from DDC_DEAL_WAYS different parts of
		 * [0, wpg) in turn, so that every workgroup -- and with it every SIMD: its waves go round
		 * the four -- gets its share of the long and of the short ones.  (Dealt in order, the first
		 * workgroups of the grid had all the long waves, and whole CUs one round more than others.) */
		const unsigned int r = k % DDC_DEAL_WAYS, q = k / DDC_DEAL_WAYS;
		const unsigned int fl_ = wpg / DDC_DEAL_WAYS, rem = wpg % DDC_DEAL_WAYS;
		k = r * fl_ + (r < rem ? r : rem) + q;
	}
	const float *ltaps = (const float *)(lds + (NCO == WR_NCO_ROTATE ? 2u * WR_SPLIT_N : 0u) + waves_per_wg * 128u * nset);
	if (LTAPS) {
		float *lt = (float *)(lds + 2u * WR_SPLIT_N + waves_per_wg * 128u * nset);
		for (unsigned int e = threadIdx.x; e < WR_FIR_LENGTH * 64u; e += blockDim.x)
			lt[e] = taps1[(size_t)(e >> 6) * slots + g * 64u + (e & 63u)];
		__syncthreads();
	}

	float h[(UTAPS || LTAPS) ? 1 : WR_FIR_LENGTH];   /* per-lane taps in registers (SPLIT / EXACT) */
	float hlane[UTAPS ? WR_TAPSETS : 1];        /* UTAPS: lane j holds the tap of sample j, per tap set */
	unsigned int mysel = 0;                     /* UTAPS: which of the group's tap sets this lane's channel uses */
#pragma unroll
	for (int q = 0; q < (UTAPS ? WR_TAPSETS : 1); ++q)
		hlane[q] = 0.0f;
	unsigned int p0 = 0, st = 0;
	int fl = 0;
	unsigned int buf = 0;
	v2f rot0 = {1.0f, 0.0f}, rot1 = {1.0f, 0.0f};   /* ROTATE: the two possible turns per frame */

	/* sample `lane` of the window of unit (gg, kk) (frame index n0 + lane): from the current
	 * block, or from the previous block's last 63 frames.  (The tuner's history buffer starts
	 * out as zeros; whether a given channel may use it -- a receiver added later starts from an
	 * empty LowPass::block -- is decided per lane where the sample is consumed.) */
	auto window_sample = [&](unsigned int kk) -> float2 {
		const long long n = (long long)kk * d1 - WR_HIST + lane;
		return (n >= 0) ? input_frame(cur, cur_u8, (size_t)n) : hist[WR_HIST + n];
	};
	/* the window of the NEXT unit is fetched while this one is computed: its global-load
	 * latency would otherwise sit in front of every unit */
	float2 xnext = make_float2(0.0f, 0.0f);
#ifndef DDC_PREFETCH1
#define DDC_PREFETCH2 1                                 /* measured at C2: 38.1 -> 37.5 us */
#endif
#ifdef DDC_PREFETCH2
	float2 xnext2 = make_float2(0.0f, 0.0f);         /* the unit after the next: an HBM + TLB miss can outlast a unit */
#endif
	const unsigned int s = g * 64u + lane;
	if (k < k1u) {
		{
			/* one coalesced load each (WrGroupDev::rot, taps1u), all independent of one another */
			p0 = DDC_PHASE(s);
			st = step[s];
			fl = flags[s];
			if (NCO == WR_NCO_ROTATE) {
				const float4 r = rot[s];
				rot0 = (v2f){r.x, r.y};
				rot1 = (v2f){r.z, r.w};
			}
			if (UTAPS) {
				/* the lane group's channel filters: a few distinct ones at most (host guarantee) */
				mysel = (unsigned int)tapsel[s];
#pragma unroll
				for (int q = 0; q < (UTAPS ? WR_TAPSETS : 1); ++q)
					if ((unsigned int)q < nset)
						hlane[q] = taps1u[((size_t)g * WR_TAPSETS + q) * 64u + lane];
			} else {
#pragma unroll
				for (int j = 0; j < ((UTAPS || LTAPS) ? 1 : WR_FIR_LENGTH); ++j)
					h[j] = taps1[(size_t)j * slots + s];
			}
		}
		xnext = window_sample(k);
#ifdef DDC_PREFETCH2
		if (k + wpg < k1u)
			xnext2 = window_sample(k + wpg);
#endif
	}

	TL(2);                                                   /* per-channel state loaded (issued) */
	unsigned int tl_unit = 0;
	/* units this wave will do, and how far through them it is, in quarters (see wave_prio) */
	const unsigned int my_units = (k < k1u) ? (k1u - k + wpg - 1u) / wpg : 0u;
	for (; k < k1u; buf ^= 1u) {
		const unsigned int kn = k + wpg;              /* the unit after this one */
		{
			const unsigned int q = (4u * tl_unit) / (my_units ? my_units : 1u);   /* 0..3 */
			wave_prio(q >= 3u ? 0u : 2u - q + 0u);
		}

		/* ---- the window: sample j of this output frame goes to lane j, then to LDS.
		 * Every lane needs every sample; a tap reads its sample back with a broadcast
		 * ds_read_b64 whose address is a constant plus an immediate -- no VALU work, no
		 * scalar loads (they share lgkmcnt with the table gathers and return out of
		 * order), no 64-lane vector broadcast loads (8 B x 64 lanes of return bandwidth
		 * per tap). */
		const long long n0 = (long long)k * d1 - WR_HIST;   /* input frame of tap j = 0 */
		/* Order matters for the wait counters: (1) the window fetched during the previous unit
		 * is consumed, (2) ROTATE's segment anchors are looked up, (3) the next unit's window
		 * prefetch goes out last -- so that waiting for the anchors never waits for it. */
		{
			const float2 xf = xnext;
			if (UTAPS) {
#pragma unroll
				for (int q = 0; q < (UTAPS ? WR_TAPSETS : 1); ++q)
					if ((unsigned int)q < nset)
						win[(buf * nset + q) * 64u + lane] = (v2f){hlane[q] * xf.x, hlane[q] * xf.y};
			} else {
				win[buf * 64u + lane] = (v2f){xf.x, xf.y};
			}
		}
		v2f csq[ROT_Q];
		if (NCO == WR_NCO_ROTATE && n0 >= 0) {
			const unsigned int P0 = p0 + (unsigned int)n0 * st;
#pragma unroll
			for (int q = 0; q < ROT_Q; ++q)
				csq[q] = nco<NCO>(P0 + (unsigned int)(q * ROT_SEG + ROT_SEG - 1) * st, table, hi_l, lo_l);
		}
#ifdef DDC_PREFETCH2
		/* (the copy waits for the load issued one unit ago: the second slot buys little more than the
		 * first.  Alternating two slots in a loop unrolled by two -- no copy -- spills under the
		 * 64-VGPR bound and was slower, r02) */
		xnext = xnext2;
		if (kn + wpg < k1u)
			xnext2 = window_sample(kn + wpg);
#else
		if (kn < k1u)
			xnext = window_sample(kn);
#endif
		/* (per lane: lanes of channels with different filters read different copies of the window) */
		const v2f *wbase = win + (buf * nset + mysel) * 64u;
		const lds_v2f *w = (const lds_v2f *)wbase;
		v2f acc = {0.0f, 0.0f};

		if (n0 >= 0 && NCO == WR_NCO_SPLIT) {
			unsigned int P = p0 + (unsigned int)n0 * st;
			/* two-stage software pipeline over groups of NT taps: the table gathers of
			 * group t+1 are in flight while group t is multiplied out */
			constexpr int NT = 2;                    /* measured: 2 beats 4 and 8 by 1-2 % */
			v2f ta[2][NT], tb[2][NT];                /* gathered cis(coarse), cis(fine)       */
			v4f xw[2][NT / 2];                       /* window samples, two taps per 16 B read */
			unsigned int ah[NT], al[NT];             /* rotating address registers */
			const lds_v4f *w4 = (const lds_v4f *)wbase;
#pragma unroll
			for (int jj = 0; jj < NT; ++jj) {
				ah[jj] = a_hi;
				al[jj] = a_lo;
			}
			/* stage 0 */
#pragma unroll
			for (int jj = 0; jj < NT / 2; ++jj)
				xw[0][jj] = w4[jj];
#pragma unroll
			for (int jj = 0; jj < NT; ++jj) {
				gather_split(P, ah[jj], al[jj], ta[0][jj], tb[0][jj]);
				P += st;
			}
#pragma unroll
			for (int t = 0; t < WR_FIR_LENGTH / NT; ++t) {
				/* everything stage t+1 needs from LDS is issued before stage t is multiplied
				 * out; LDS returns in order, so stage t's operands only wait for older reads */
				if (t + 1 < WR_FIR_LENGTH / NT) {
#pragma unroll
					for (int jj = 0; jj < NT / 2; ++jj)
						xw[(t + 1) & 1][jj] = w4[(t + 1) * (NT / 2) + jj];
#pragma unroll
					for (int jj = 0; jj < NT; ++jj) {
						gather_split(P, ah[jj], al[jj], ta[(t + 1) & 1][jj], tb[(t + 1) & 1][jj]);
						P += st;
					}
				}
#pragma unroll
				for (int jj = 0; jj < NT; ++jj) {
					const int j = t * NT + jj;
					const v4f x2 = xw[t & 1][jj >> 1];
					const v2f xs = (jj & 1) ? (v2f){x2.z, x2.w} : (v2f){x2.x, x2.y};
					const v2f a = ta[t & 1][jj], b = tb[t & 1][jj];
					const float c = __builtin_fmaf(-a.y, b.y, a.x * b.x);
					const float sn = __builtin_fmaf(a.x, b.y, a.y * b.x);
					if (UTAPS) {
						/* xs = coeff * sample already: acc += xs * conj(LO) */
						acc.x = __builtin_fmaf(xs.y, sn, __builtin_fmaf(xs.x, c, acc.x));
						acc.y = __builtin_fmaf(-xs.x, sn, __builtin_fmaf(xs.y, c, acc.y));
					} else {
						const float mi = __builtin_fmaf(xs.y, sn, xs.x * c);
						const float mq = __builtin_fmaf(-xs.x, sn, xs.y * c);
						const float hj = h[(UTAPS || LTAPS) ? 0 : WR_FIR_LENGTH - 1 - j];
						acc.x = __builtin_fmaf(hj, mi, acc.x);
						acc.y = __builtin_fmaf(hj, mq, acc.y);
					}
				}
			}
		} else if (n0 >= 0 && NCO == WR_NCO_ROTATE) {
			const unsigned int P0 = p0 + (unsigned int)n0 * st;
			unsigned int F = P0 << 16;                /* the 16 fraction bits, left-aligned */
			const unsigned int fstep = st << 16;
			const lds_v4f *w4 = (const lds_v4f *)wbase;
			v2f A = {0.0f, 0.0f}, Aq[ROT_Q];
#pragma unroll
			for (int jp = 0; jp < WR_FIR_LENGTH / 2; ++jp) {
				const v4f x2 = w4[jp];
#pragma unroll
				for (int jj = 0; jj < 2; ++jj) {
					const int j = 2 * jp + jj;
					v2f u = jj ? (v2f){x2.z, x2.w} : (v2f){x2.x, x2.y};
					if (!UTAPS)
						u = u * ltaps[(WR_FIR_LENGTH - 1 - j) * 64 + lane];   /* one LDS word per tap */
					if (j % ROT_SEG == 0) {
						if (j)
							F += fstep;                 /* a segment starts afresh: no turn */
						A = u;
					} else {
						unsigned int F2;
						const bool carry = __builtin_uadd_overflow(F, fstep, &F2);
						F = F2;
						horner_step(A, carry ? rot1.x : rot0.x, carry ? rot1.y : rot0.y, u);
					}
					if (j % ROT_SEG == ROT_SEG - 1)
						Aq[j / ROT_SEG] = A;
				}
			}
			/* closed at the end, in segment order: nothing in the tap loop waits for memory */
#pragma unroll
			for (int q = 0; q < ROT_Q; ++q)
				horner_close(acc, Aq[q], csq[q]);
		} else if (NCO == WR_NCO_ROTATE) {
			/* window reaches into the previous block: the turns into frames <= 0 and the LO of
			 * anchors < 0 are the ones kept from back then (whatever the step was), the rest
			 * follow from the phase.  Same steps in the same order as above, so a frame's bits
			 * do not depend on where the block boundaries fall.
			 * Only the first ceil(63/D1) frames of a block come here, but the wave that gets one
			 * still has its share of ordinary units to do, so this path is on the kernel's
			 * critical path: everything it needs from memory is fetched SLOW_CH taps at a time,
			 * branch-free, before the arithmetic (one memory latency per chunk, not per tap). */
			v2f A = {0.0f, 0.0f};
			const float *hc = (const float *)hist_cs;
#pragma clang loop unroll(disable)
			for (int jb = 0; jb < WR_FIR_LENGTH; jb += SLOW_CH) {
				v2f r[SLOW_CH];
				float hj[UTAPS ? 1 : SLOW_CH];
#pragma unroll
				for (int jj = 0; jj < SLOW_CH; ++jj) {
					const int j = jb + jj;
					const long long n = n0 + j;
					if (!UTAPS)
						hj[UTAPS ? 0 : jj] = taps1[(size_t)(WR_FIR_LENGTH - 1 - j) * slots + s];
					/* (the turn into a segment's first frame is fetched too, and not used) */
					const unsigned int idx = rot_index(p0, st, (unsigned int)n);
					const bool old = n <= 0;                         /* wave-uniform */
					const size_t row = old ? (size_t)(WR_HIST - 1 + (n < -(long long)(WR_HIST - 1) ? -(long long)(WR_HIST - 1) : n)) : 0;
					const float *pc = old ? hc + 2 * (row * slots + s) : table + ((idx + 16384u) & 0xFFFFu);
					const float *ps = old ? pc + 1 : table + (idx & 0xFFFFu);
					r[jj] = (v2f){*pc, *ps};
				}
				const bool closes = (jb + SLOW_CH) % ROT_SEG == 0;   /* a segment ends with this chunk */
				const long long nl = n0 + jb + SLOW_CH - 1;
				const float2 olo = hist_lo[(size_t)(nl < 0 ? WR_HIST + nl : 0) * slots + s];
#pragma unroll
				for (int jj = 0; jj < SLOW_CH; ++jj) {
					const int j = jb + jj;
					v2f u = w[j];
					if (!UTAPS)
						u = u * hj[UTAPS ? 0 : jj];
					if (jj == 0 && jb % ROT_SEG == 0)
						A = u;
					else
						horner_step(A, r[jj].x, r[jj].y, u);
				}
				if (closes) {
					const v2f cs = (nl < 0) ? (v2f){olo.x, olo.y}
					                        : nco<NCO>(p0 + (unsigned int)nl * st, table, hi_l, lo_l);
					horner_close(acc, A, cs);
				}
			}
		} else if (n0 >= 0) {
			/* EXACT: the reference's table and the reference's roundings, tap by tap */
			unsigned int P = p0 + (unsigned int)n0 * st;
#pragma unroll
			for (int jb = 0; jb < WR_FIR_LENGTH; jb += 8) {
#pragma unroll
				for (int jj = 0; jj < 8; ++jj) {
					const int j = jb + jj;
					const v2f xs = w[j];
					const v2f cs = nco<NCO>(P, table, hi_l, lo_l);
					P += st;
					mac<NCO>(xs, cs, h[(UTAPS || LTAPS) ? 0 : WR_FIR_LENGTH - 1 - j], acc);
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		} else {
			/* window reaches into the previous block (only the first ceil(63/D1) frames of
			 * a block): those frames get the LO values they were mixed with back then.  The
			 * history rows are fetched SLOW_CH at a time before they are used (see above). */
#pragma clang loop unroll(disable)
			for (int jb = 0; jb < WR_FIR_LENGTH; jb += SLOW_CH) {
				float2 o[SLOW_CH];
				float hj[UTAPS ? 1 : SLOW_CH];
#pragma unroll
				for (int jj = 0; jj < SLOW_CH; ++jj) {
					const int j = jb + jj;
					const long long n = n0 + j;
					o[jj] = hist_cs[(size_t)(n < 0 ? WR_HIST + n : 0) * slots + s];
					if (!UTAPS)
						hj[UTAPS ? 0 : jj] = taps1[(size_t)(WR_FIR_LENGTH - 1 - j) * slots + s];
				}
#pragma unroll
				for (int jj = 0; jj < SLOW_CH; ++jj) {
					const int j = jb + jj;
					const long long n = n0 + j;
					const v2f xs = w[j];
					const v2f cs = (n < 0) ? (v2f){o[jj].x, o[jj].y}
					                       : nco<NCO>(p0 + (unsigned int)n * st, table, hi_l, lo_l);
					if (UTAPS) {
						/* the window is premultiplied; same operation order as the fast path, so a
						 * frame gives the same bits wherever the block boundaries fall */
						acc.x = __builtin_fmaf(xs.y, cs.y, __builtin_fmaf(xs.x, cs.x, acc.x));
						acc.y = __builtin_fmaf(-xs.x, cs.y, __builtin_fmaf(xs.y, cs.x, acc.y));
					} else {
						mac<NCO>(xs, cs, hj[UTAPS ? 0 : jj], acc);
					}
				}
			}
		}
		if (fl & PHASE_FLAG_ACTIVE) {
#ifdef DDC_PLAIN_STORE
			chan_iq[(size_t)k * slots + s] = make_float2(acc.x, acc.y);
#else
			/* Written once, read by the NEXT launch: a write-through (sc1) store streams the 20 MB of
			 * channel IQ out while the taps run.  As plain stores they sat dirty in the L2s until the
			 * end-of-kernel release wrote them back -- several microseconds in which nothing computes
			 * (MI355X_MICROARCH: + B / 6 TB/s per kernel boundary for B dirty bytes). */
			union { v2f f; unsigned long long u; } cv;
			cv.f = acc;
			__hip_atomic_store((unsigned long long *)&chan_iq[(size_t)k * slots + s], cv.u, __ATOMIC_RELAXED,
			                   __HIP_MEMORY_SCOPE_AGENT);
#endif
		}
		k = kn;
		TL(3u + tl_unit);                                    /* unit done (store issued) */
		++tl_unit;
	}
	wave_prio(0u);
	/* End-of-block state of every channel, written to the OTHER state set so that no reader
	 * of this launch is disturbed and the next block's launch depends on nothing but this one:
	 *   - DownConverter::phase advances by nframes steps (downconverter.cxx:103);
	 *   - the channel filter's history.  The reference keeps the last 63 MIXED frames per
	 *     receiver (lowpass.cxx:138-142).  Here the raw frames are kept once per tuner and,
	 *     per channel, the LO value (cos, sin) each of them was mixed with -- so the frames
	 *     can be re-mixed bit-identically next block whatever happens to the phase step in
	 *     between (setIF) and however short the blocks are.  A zero LO row = an empty
	 *     history (a fresh LowPass::block).
	 * Done by the LAST workgroups of the grid, after their units: they are the ones that get
	 * one unit fewer when the units do not divide evenly, and nothing in this launch waits for it
	 * (at the head of the kernel its dependent loads sat in front of the first unit). */
	const unsigned int roll_wgs = (WR_FIR_LENGTH * slots + blockDim.x - 1u) / blockDim.x;
	const unsigned int roll_first = n_ddc > roll_wgs ? n_ddc - roll_wgs : 0u;
	if (whole && blockIdx.x >= roll_first && blockIdx.x < n_ddc) {
		const unsigned int nlo = (unsigned int)nframes;
		const unsigned int gtid = (blockIdx.x - roll_first) * blockDim.x + threadIdx.x, gsz = (n_ddc - roll_first) * blockDim.x;
		for (unsigned int s = gtid; s < slots; s += gsz) {
			const bool act = (flags[s] & PHASE_FLAG_ACTIVE) != 0;
			phase_next[s] = act ? DDC_PHASE(s) + nlo * step[s] : DDC_PHASE(s);
		}
		for (unsigned int e = gtid; e < WR_HIST * slots; e += gsz) {
			const unsigned int r = e / slots, s = e - r * slots;
			const size_t f = nframes + r;                 /* frame index in [hist | cur] */
			v2f cs = {0.0f, 0.0f};
			if (NCO == WR_NCO_ROTATE) {
				/* ROTATE also keeps the LO values themselves: the segment anchors */
				v2f lo = {0.0f, 0.0f};
				if (flags[s] & PHASE_FLAG_ACTIVE) {
					if (f < WR_HIST) {
						const float2 o = hist_lo[f * slots + s];
						lo = (v2f){o.x, o.y};
					} else {
						lo = nco<NCO>(DDC_PHASE(s) + (unsigned int)(f - WR_HIST) * step[s], table, hi_l, lo_l);
					}
				}
				hist_lo_next[e] = make_float2(lo.x, lo.y);
			}
			if (flags[s] & PHASE_FLAG_ACTIVE) {
				if (f < WR_HIST) {
					const float2 o = hist_cs[f * slots + s];
					cs = (v2f){o.x, o.y};
				} else if (NCO == WR_NCO_ROTATE) {
					/* ROTATE keeps turns, not LO values: row r = the turn into frame r - 62 of the
					 * next block, i.e. into frame f - 62 >= 1 of this one.  (Row 62 is the turn
					 * into the next block's first frame: made with THIS block's step, as the
					 * reference adds phase_step right after using a frame.)  A zero row = nothing
					 * before that frame counts, which is also how a fresh channel starts. */
					cs = rot_into(DDC_PHASE(s), step[s], (unsigned int)(f - (WR_HIST - 1)), table);
				} else {
					cs = nco<NCO>(DDC_PHASE(s) + (unsigned int)(f - WR_HIST) * step[s], table, hi_l, lo_l);
				}
			}
			hist_cs_next[e] = make_float2(cs.x, cs.y);
		}
	}

	TL(11);
}

/* (The roles of the lean kernel that are not its loop are ddc_body INLINED at two places -- ahead of the
 * loop, for the workgroups that never enter it, and behind it with role = DDC_ROLE_ROLL, which folds
 * everything but the state roll away.  As a real call (tried) the kernel needs 936 bytes of stack per lane
 * for the 35 arguments and the call-clobbered registers: half a gigabyte of scratch for a full grid, which
 * the runtime then allocates and frees around EVERY dispatch -- 61 us per launch instead of 35.) */
#ifndef DDC_RD
#define DDC_RD 1                           /* lean loop: 16-byte window reads issued ahead of the taps, per stage */
#endif

/* NG: lane groups (recurrences) per wave of the lean loop.  The ROTATE tap is a chain -- every Horner step
 * waits for the one before it, and the carry travels add -> VCC -> select -> FMA -- so ONE recurrence per wave
 * needs eight waves per SIMD to reach 138 G wave-taps/s, while TWO independent ones per wave, one after the
 * other tap by tap, reach the issue ceiling (150-152 G, = the rate of plain independent FMAs) with as few as
 * two waves per SIMD (tools/ubench_tap.hip, profiles/r03_ubench_tap.txt).  The two recurrences are two lane
 * groups at the same output frame: they share the window and every one of its LDS reads.  NG = 2 takes an even
 * number of lane groups that all use ONE and the same channel filter (the host says: WrTunerLaunch::one_filter)
 * and 80 registers (6 waves per SIMD); otherwise NG = 1, as before. */
#ifndef DDC_NG2_WAVES_PER_EU
#define DDC_NG2_WAVES_PER_EU 6u
#endif
template <int NCO, bool UTAPS, unsigned int NG> struct DdcOcc {
	static constexpr unsigned int per_eu = (NG == 2u) ? DDC_NG2_WAVES_PER_EU
	                                       : (NCO == WR_NCO_ROTATE && (UTAPS || DDC_LTAPS_SMALL)) ? DDC_ROTATE_WGS_PER_CU * DDC_ROTATE_WAVES / 4u
	                                       : DDC_WAVES / 4u;
};

template <int NCO, bool UTAPS, unsigned int PD2, unsigned int NG>
__global__ void __launch_bounds__(DDC_WAVES * 64u) __attribute__((amdgpu_waves_per_eu(DdcOcc<NCO, UTAPS, NG>::per_eu)))
k_tuner_ddc(const float2 *__restrict__ cur, const uchar2 *__restrict__ cur_u8,
            const float2 *__restrict__ hist,
            float2 *__restrict__ hist_next, size_t nframes, size_t k1,
            unsigned int d1, unsigned int slots, unsigned int groups,
            const unsigned int *__restrict__ phase, const unsigned int *__restrict__ step,
            const float2 *__restrict__ hist_cs, const int *__restrict__ flags,
            unsigned int *__restrict__ phase_next, float2 *__restrict__ hist_cs_next,
            const float2 *__restrict__ hist_lo, float2 *__restrict__ hist_lo_next,
            const float *__restrict__ taps1, const float4 *__restrict__ rot, const float *__restrict__ taps1u,
            const int *__restrict__ tapsel, unsigned int kmax,
            float2 *__restrict__ chan_iq,
            const float *__restrict__ table, const float2 *__restrict__ hi_cs,
            const float2 *__restrict__ lo_cs, unsigned int n_ddc, WrPostArgs post,
            unsigned long long gmap0, unsigned long long gmap1, int whole, unsigned int kslow, unsigned int n_bnd,
            unsigned int seek_on, unsigned int seek_lo)
{
	extern __shared__ v2f lds[];                /* see DDC_LDS_BYTES */
	constexpr bool ROLES = DDC_ROLES && NCO == WR_NCO_ROTATE && UTAPS;
	static_assert(NG == 1u || (NG == 2u && ROLES), "two recurrences per wave: the lean loop only");
	if constexpr (!ROLES) {
		ddc_body<NCO, UTAPS, PD2>(DDC_PASS, lds, DDC_ROLE_ALL);
		return;
	}
	if (blockIdx.x >= n_ddc) {
		/* a block-boundary unit per wave, or (by workgroup index) the riding post stage */
		ddc_body<NCO, UTAPS, PD2>(DDC_PASS, lds, DDC_ROLE_BOUNDARY);
		return;
	}
	/* ---- the persistent waves: every window lies inside the block (n0 >= 0).  Nothing of the block-boundary
	 * path, the post stage or the state roll is live in this loop, so the registers are the tap loop's.
	 *   - A unit ENDS with the next unit's window going to LDS, then its own store, then the load for the
	 *     unit after the next: at the only point where the wave waits for vector memory, everything
	 *     outstanding was issued a whole unit (~3 us) ago -- whatever count the compiler waits for, it waits
	 *     for nothing.
	 *   - two window loads are in flight, in two register pairs that take turns (the loop is unrolled by
	 *     two: no copy of a load in flight), both LDS buffers have constant addresses, and the phase of a
	 *     unit's first frame advances by addition (wpg * d1 frames per unit). ---- */
	const unsigned int lane = threadIdx.x & 63u;
	const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned int waves_per_wg = blockDim.x >> 6;
	const unsigned int wid = blockIdx.x * waves_per_wg + wave;
	TL(0);
	wave_prio(3u);                                           /* the prologue's loads go out at once */
	for (unsigned int e = threadIdx.x; e < WR_SPLIT_N; e += blockDim.x) {
		const float2 hv = hi_cs[e], lv = lo_cs[e];
		lds[e] = (v2f){hv.x, hv.y};
		lds[WR_SPLIT_N + e] = (v2f){lv.x, lv.y};
	}
	__syncthreads();
	const v2f *hi_l = lds, *lo_l = lds + WR_SPLIT_N;
	const unsigned int nset = (NG == 2u) ? 1u : kmax;
	v2f *win = lds + 2u * WR_SPLIT_N + wave * (128u * nset);
	if (whole && blockIdx.x == 0 && wave == 0 && lane < WR_HIST) {
		const size_t f = nframes + lane;            /* frame index in [hist | cur]: the tuner's next input history */
		hist_next[lane] = (f < WR_HIST) ? hist[f] : input_frame(cur, cur_u8, f - WR_HIST);
	}
	const unsigned int k1u = (unsigned int)k1;
	const unsigned int nwaves = n_ddc * waves_per_wg;
	TL(1);
	const unsigned int gsets = groups / NG;                 /* sets of NG lane groups that share a wave */
	const unsigned int wpg = nwaves / gsets;                /* >= 1: see the launcher */
	const unsigned int gs = wid % gsets;
	unsigned int k = wid / gsets;
	if (k >= wpg) {
		k = k1u;                                         /* the few waves left over stay idle */
	} else {
		if (DDC_DEAL_WAYS > 1u) {                        /* long and short waves to every workgroup: see ddc_body */
			const unsigned int r = k % DDC_DEAL_WAYS, q = k / DDC_DEAL_WAYS;
			const unsigned int fl_ = wpg / DDC_DEAL_WAYS, rem = wpg % DDC_DEAL_WAYS;
			k = r * fl_ + (r < rem ? r : rem) + q;
		}
		k += kslow;                                      /* behind the block-boundary frames */
		if (k > k1u)
			k = k1u;
	}
	const unsigned int my_units = (k < k1u) ? (k1u - k + wpg - 1u) / wpg : 0u;
	unsigned int tl_unit = 0;
	if (k < k1u) {
		/* per recurrence: one coalesced load each (WrGroupDev::rot, taps1u), all independent of one another */
		unsigned int Pk[NG], fstep[NG], dP[NG], stv[NG];
		int fl[NG];
		v2f rot0[NG], rot1[NG];                          /* the two possible turns per frame */
		float2 *out[NG];
		unsigned int g0 = 0, s0 = 0;
#pragma unroll
		for (unsigned int c = 0; c < NG; ++c) {
			const unsigned int gl = gs * NG + c;
			const unsigned int g = (unsigned int)(((gl < 8u ? gmap0 : gmap1) >> ((gl & 7u) * 8u)) & 255u);
			const unsigned int s = g * 64u + lane;
			if (c == 0) {
				g0 = g;
				s0 = s;
			}
			const unsigned int p0 = DDC_PHASE(s);
			stv[c] = step[s];
			fl[c] = flags[s];
			const float4 r4 = rot[s];
			rot0[c] = (v2f){r4.x, r4.y};
			rot1[c] = (v2f){r4.z, r4.w};
			fstep[c] = stv[c] << 16;
			Pk[c] = p0 + (unsigned int)((size_t)k * d1 - WR_HIST) * stv[c];   /* phase of the unit's first window frame */
			dP[c] = wpg * d1 * stv[c];
			out[c] = chan_iq + s;
		}
		const unsigned int mysel = (NG == 2u) ? 0u : (unsigned int)tapsel[s0];   /* which of the group's tap sets this lane's channel uses */
		float hlane[WR_TAPSETS];                              /* lane j holds the tap of window sample j, per tap set */
#pragma unroll
		for (int q = 0; q < WR_TAPSETS; ++q)
			hlane[q] = ((unsigned int)q < nset) ? taps1u[((size_t)g0 * WR_TAPSETS + q) * 64u + lane] : 0.0f;
		const lds_v4f *w4[2];
		v2f *wst[2];
#pragma unroll
		for (int b = 0; b < 2; ++b) {
			w4[b] = (const lds_v4f *)(win + ((unsigned int)b * nset + mysel) * 64u);
			wst[b] = win + (unsigned int)b * nset * 64u + lane;
		}
		const size_t dn = (size_t)wpg * d1;
		/* window sample `lane` of unit kk is frame kk * d1 - 63 + lane >= 0, inside the block; past the
		 * wave's last unit the index stays where it is (a load nobody uses, and no branch around it) */
		const size_t nlast = (size_t)(k + (my_units - 1u) * wpg) * d1 - WR_HIST + lane;
		size_t nx = (size_t)k * d1 - WR_HIST + lane;
		/* (a frame travels as loaded -- two floats, or the two bytes of the RTL-SDR format in .x -- and is
		 * converted where it is used, (u8 - 128) / 128 as io/rtlsdrtuner.cxx:106: nothing waits at the load) */
		auto fetch = [&]() __attribute__((always_inline)) -> float2 {
			float2 v;
			if (cur_u8) {
				const uchar2 b2 = cur_u8[nx];
				v = make_float2(__builtin_bit_cast(float, (unsigned int)b2.x | ((unsigned int)b2.y << 8)), 0.0f);
			} else {
				v = cur[nx];
			}
			nx = (nx + dn <= nlast) ? nx + dn : nlast;
			return v;
		};
		auto to_lds = [&](const float2 raw, const int b) __attribute__((always_inline)) {
			float2 xf = raw;
			if (cur_u8) {
				const unsigned int bits = __builtin_bit_cast(unsigned int, raw.x);
				xf = make_float2(((float)(bits & 255u) - 128.0f) / 128.0f, ((float)((bits >> 8) & 255u) - 128.0f) / 128.0f);
			}
#pragma unroll
			for (int q = 0; q < WR_TAPSETS; ++q)
				if ((unsigned int)q < nset)
					wst[b][q * 64] = (v2f){hlane[q] * xf.x, hlane[q] * xf.y};
		};
		float2 xa = fetch();                              /* unit 0 */
		float2 xb = fetch();                              /* unit 1 */
		to_lds(xa, 0);
		xa = fetch();                                     /* unit 2 */
		TL(2);
		/* one unit from LDS buffer b; `xo` holds the NEXT unit's window sample (on its way or landed) */
		auto unit = [&](float2 &xo, const int b) __attribute__((always_inline)) {
			{
				const unsigned int q = (4u * tl_unit) / my_units;   /* 0..3: see wave_prio */
				wave_prio(q >= 3u ? 0u : 2u - q + 0u);
			}
			v2f csq[NG][ROT_Q];
			unsigned int F[NG];
			v2f acc[NG], A[NG], Aq[NG][ROT_Q];
#pragma unroll
			for (unsigned int c = 0; c < NG; ++c) {
#pragma unroll
				for (int q = 0; q < ROT_Q; ++q)
					csq[c][q] = nco<NCO>(Pk[c] + (unsigned int)(q * ROT_SEG + ROT_SEG - 1) * stv[c], table, hi_l, lo_l);
				F[c] = Pk[c] << 16;                   /* the 16 fraction bits, left-aligned */
				acc[c] = (v2f){0.0f, 0.0f};
				A[c] = (v2f){0.0f, 0.0f};
			}
			/* the window comes back DDC_RD x 16 bytes (2 taps each) ahead of the arithmetic */
			constexpr int RD = DDC_RD, NST = WR_FIR_LENGTH / 2 / RD;
			v4f xr[2][RD];
#pragma unroll
			for (int r = 0; r < RD; ++r)
				xr[0][r] = w4[b][r];
#pragma unroll
			for (int t = 0; t < NST; ++t) {
#ifndef DDC_ABL_NOLDS
				if (t + 1 < NST) {
#pragma unroll
					for (int r = 0; r < RD; ++r)
						xr[(t + 1) & 1][r] = w4[b][(t + 1) * RD + r];
				}
#endif
#pragma unroll
				for (int r = 0; r < RD; ++r) {
#ifdef DDC_ABL_NOLDS                                           /* (timing experiments only: results are wrong) */
					const v4f x2 = {xo.x, xo.y, xo.y, xo.x};
#else
					const v4f x2 = xr[t & 1][r];
#endif
#pragma unroll
					for (int jj = 0; jj < 2; ++jj) {
						const int j = 2 * (t * RD + r) + jj;
						const v2f u = jj ? (v2f){x2.z, x2.w} : (v2f){x2.x, x2.y};
#pragma unroll
						for (unsigned int c = 0; c < NG; ++c) {
							if (j % ROT_SEG == 0) {
								if (j)
									F[c] += fstep[c];       /* a segment starts afresh: no turn */
								A[c] = u;
							} else {
#ifdef DDC_ABL_NOSEL
								horner_step(A[c], rot0[c].x, rot0[c].y, u);
#else
								unsigned int F2;
								const bool carry = __builtin_uadd_overflow(F[c], fstep[c], &F2);
								F[c] = F2;
								horner_step(A[c], carry ? rot1[c].x : rot0[c].x, carry ? rot1[c].y : rot0[c].y, u);
#endif
							}
							if (j % ROT_SEG == ROT_SEG - 1)
								Aq[c][j / ROT_SEG] = A[c];
						}
					}
				}
				/* nothing crosses a stage: left alone, the scheduler hoists all 32 window reads of a unit to
				 * its top (128 registers of them) and spills them -- and the optimiser, for which the carries and
				 * selects of ALL taps depend on nothing but the phase, may compute them first and sink the
				 * multiply-adds to where the result is used: the recurrence is pinned stage by stage */
#pragma unroll
				for (unsigned int c = 0; c < NG; ++c)
					asm volatile("" : "+v"(A[c].x), "+v"(A[c].y), "+v"(F[c]));
				__builtin_amdgcn_sched_barrier(0);
			}
#pragma unroll
			for (unsigned int c = 0; c < NG; ++c) {
#pragma unroll
				for (int q = 0; q < ROT_Q; ++q)
					horner_close(acc[c], Aq[c][q], csq[c][q]);
				/* (the result is needed HERE: its only use is the store under `if (active)` below, and the
				 * optimiser would sink the whole recurrence into that branch, behind the window writes --
				 * leaving the unit's 32 window reads on their own at its top, 128 registers wide) */
				asm volatile("" : "+v"(acc[c].x), "+v"(acc[c].y));
			}
			/* the next unit's window (requested two units ago), this unit's result, the request after next */
			to_lds(xo, b ^ 1);
#pragma unroll
			for (unsigned int c = 0; c < NG; ++c) {
				if (fl[c] & PHASE_FLAG_ACTIVE) {
#ifdef DDC_PLAIN_STORE
					out[c][(size_t)k * slots] = make_float2(acc[c].x, acc[c].y);
#else
					union { v2f f; unsigned long long u; } cv;
					cv.f = acc[c];
					__hip_atomic_store((unsigned long long *)&out[c][(size_t)k * slots], cv.u, __ATOMIC_RELAXED,
					                   __HIP_MEMORY_SCOPE_AGENT);             /* write-through: see ddc_body */
#endif
				}
				Pk[c] += dP[c];
			}
			xo = fetch();
			k += wpg;
			TL(3u + tl_unit);
			++tl_unit;
		};
		while (k < k1u) {
			unit(xb, 0);
			if (k >= k1u)
				break;
			unit(xa, 1);
		}
	}
	wave_prio(0u);
	/* the end-of-block state of every channel: the LAST workgroups of the grid, after their units (they
	 * are the ones that get a unit fewer when the units do not divide evenly) */
	const unsigned int roll_wgs = (WR_FIR_LENGTH * slots + blockDim.x - 1u) / blockDim.x;
	const unsigned int roll_first = n_ddc > roll_wgs ? n_ddc - roll_wgs : 0u;
	if (whole && blockIdx.x >= roll_first)
		ddc_body<NCO, UTAPS, PD2>(DDC_PASS, lds, DDC_ROLE_ROLL);
	TL(11);
}

/*
 * k_tuner_ddc_long: DownConverter::process + a channel LowPass of MORE than 64 taps (LowPass::_firLength 128 or
 * 256: dsp/lowpass.cxx:38-39 "FIXME: Make runtime variable", :102-110, :131-162) inside the tuner's launch
 * sequence -- SURVEY 8f-4.  The fused kernels above are built around a 64-frame window; a longer filter is rare
 * and takes this plain kernel instead, which still never materialises the full-rate mixer output and still mixes
 * only the frames a tap reaches (L of every D1):
 *
 *   unit (k, lane group), one wave: y[k] = sum_{j < L} coeff[L-1-j] * block[k*D1 + j] in the reference's order
 *   (oldest first, unfused), where block = [the last L-1 MIXED frames of the previous block | this block's
 *   mixed frames] exactly as LowPass::block holds them (lowpass.cxx:138-142): frames of this block are mixed
 *   on the fly with the reference's table and operations (downconverter.cxx:100-110), the older ones come from
 *   `mixhist`, per channel -- so the result is bit-identical to the oracle's cascade in EVERY nco mode (the
 *   fast recurrences are not used here), whatever setIF did in between.
 *   The 64-frame pieces of the window go through LDS like the fused kernel's (lane j loads frame j).
 * k_ddc_long_roll: the end-of-block state -- phase advanced by nframes steps, the last L-1 mixed frames.
 */
/* units [u0, units) of the reference-arithmetic path, every `ustride`-th: one wave each (see above).  `winl`: this wave's
 * 64-frame window piece in LDS.  Everything a chunk of 8 taps needs from memory -- the table values, the taps, the mixed
 * frames of the previous block -- is requested before the chunk's arithmetic, unconditionally (a frame inside the block
 * loads a clamped history row it does not use and the other way round): one memory latency per chunk, not per tap. */
__device__ __forceinline__ void ddc_long_exact_units(const float2 *__restrict__ cur, const uchar2 *__restrict__ cur_u8,
                                                      size_t units, size_t u0, size_t ustride, unsigned int d1, unsigned int len,
                                                      unsigned int slots, unsigned int groups,
                                                      const unsigned int *__restrict__ phase, const unsigned int *__restrict__ step,
                                                      const int *__restrict__ flags, const float *__restrict__ taps,
                                                      const float2 *__restrict__ mixhist, const float *__restrict__ table,
                                                      float2 *__restrict__ chan_iq, v2f *winl)
{
	const unsigned int lane = threadIdx.x & 63u;
	const unsigned int hl = len - 1u;                    /* history frames per channel */
	for (size_t u = u0; u < units; u += ustride) {
		const unsigned int g = (unsigned int)(u % groups);
		const size_t k = u / groups;
		const unsigned int s = g * 64u + lane;
		const unsigned int p0 = phase[s], st = step[s];
		const bool active = (flags[s] & PHASE_FLAG_ACTIVE) != 0;
		v2f acc = {0.0f, 0.0f};
		for (unsigned int seg = 0; seg < len / 64u; ++seg) {
			const long long n0 = (long long)k * d1 - (long long)hl + 64ll * seg;    /* frame of this piece's first tap */
			const long long nl = n0 + lane;
			float2 xf = make_float2(0.0f, 0.0f);
			if (nl >= 0)
				xf = input_frame(cur, cur_u8, (size_t)nl);
			winl[lane] = (v2f){xf.x, xf.y};                  /* (one wave reads what it wrote: LDS is in order) */
			for (unsigned int jb = 0; jb < 64u; jb += 8u) {
				v2f cs[8];
				float hj[8];
				float2 m[8];
#pragma unroll
				for (unsigned int jj = 0; jj < 8u; ++jj) {
					const long long n = n0 + jb + jj;
					hj[jj] = taps[(size_t)(len - 1u - (64u * seg + jb + jj)) * slots + s];
					cs[jj] = nco<WR_NCO_EXACT>(p0 + (unsigned int)n * st, table, nullptr, nullptr);
					m[jj] = mixhist[(size_t)(n < 0 ? hl + n : 0) * slots + s];   /* the frame as it was mixed back then */
				}
#pragma unroll
				for (unsigned int jj = 0; jj < 8u; ++jj) {
					const long long n = n0 + jb + jj;
					if (n >= 0) {
						mac<WR_NCO_EXACT>(winl[jb + jj], cs[jj], hj[jj], acc);
					} else {
						float ti, tq;
						asm("v_mul_f32 %0, %1, %2" : "=v"(ti) : "v"(hj[jj]), "v"(m[jj].x));
						asm("v_mul_f32 %0, %1, %2" : "=v"(tq) : "v"(hj[jj]), "v"(m[jj].y));
						acc.x = acc.x + ti;
						acc.y = acc.y + tq;
					}
				}
			}
		}
		if (active)
			chan_iq[k * slots + s] = make_float2(acc.x, acc.y);
	}
}

__global__ void __launch_bounds__(256)
k_tuner_ddc_long(const float2 *__restrict__ cur, const uchar2 *__restrict__ cur_u8, size_t nframes, size_t k1,
                 unsigned int d1, unsigned int len, unsigned int slots, unsigned int groups,
                 const unsigned int *__restrict__ phase, const unsigned int *__restrict__ step,
                 const int *__restrict__ flags, const float *__restrict__ taps, const float2 *__restrict__ mixhist,
                 const float *__restrict__ table, float2 *__restrict__ chan_iq)
{
	__shared__ v2f winl[4][64];
	const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	ddc_long_exact_units(cur, cur_u8, k1 * groups, (size_t)blockIdx.x * 4u + wave, (size_t)gridDim.x * 4u, d1, len, slots, groups,
	                     phase, step, flags, taps, mixhist, table, chan_iq, winl[wave]);
}

/* The same channel filter with the ROTATE arithmetic, when every channel of a lane group has the same filter: the L-frame
 * window is L / 64 segments, each exactly a unit of k_tuner_ddc -- lane j folds tap L-1-(64 q + j) into window sample j, one
 * Horner chain per lane over the 64 samples with the turn selected by the phase fraction's carry, closed with the LO of the
 * segment's last frame (hi x lo tables) -- added up into one output.  Seven VALU instructions per channel-tap instead of two
 * table gathers: 128 taps in about twice the time of 64, within the ROTATE tolerance (|IQ - reference| <= 1e-6) instead of
 * bit-identical.  Frames of the previous block under the window (the first ceil((L - 1) / D1) outputs of a block) come
 * from `mixhist` as they were mixed back then (lowpass.cxx:138-142 keeps them; a retune in between does not touch them): a
 * plain real-tap sum, and the chain starts at the block's first frame.
 * One wave per (k, lane group); the wave's window in LDS, the next segment's frame requested before this segment's chain. */
/* NG: lane groups per wave (2 when the whole rate group has ONE filter and an even number of lane groups: the two
 * recurrences share the window and every one of its LDS reads, and fill each other's issue gaps -- see k_tuner_ddc) */
/* end-of-block state of a rate group with a long channel filter: the last len - 1 MIXED frames of every channel exactly as
 * LowPass::block keeps them (lowpass.cxx:138-142; the reference's arithmetic, downconverter.cxx:100-110) and the phase after
 * the block, into the other state set.  Element e of `total` = (len - 1) * slots; `first` / `stride`: this thread's share. */
__device__ __forceinline__ void
long_roll_body(const float2 *__restrict__ cur, const uchar2 *__restrict__ cur_u8, size_t nframes, unsigned int len,
               unsigned int slots, const unsigned int *__restrict__ phase, const unsigned int *__restrict__ step,
               const int *__restrict__ flags, unsigned int *__restrict__ phase_next,
               const float2 *__restrict__ mixhist, float2 *__restrict__ mixhist_next, const float *__restrict__ table,
               size_t first, size_t stride)
{
	const unsigned int hl = len - 1u;
	const size_t total = (size_t)hl * slots;
	for (size_t e = first; e < total; e += stride) {
		const unsigned int r = (unsigned int)(e / slots), s = (unsigned int)(e - (size_t)r * slots);
		const bool active = (flags[s] & PHASE_FLAG_ACTIVE) != 0;
		const long long f = (long long)nframes - (long long)hl + r;         /* frame of this block, or of an earlier one */
		float2 m = make_float2(0.0f, 0.0f);
		if (active) {
			if (f >= 0) {
				const float2 x = input_frame(cur, cur_u8, (size_t)f);
				const v2f cs = nco<WR_NCO_EXACT>(phase[s] + (unsigned int)f * step[s], table, nullptr, nullptr);
				m = make_float2(x.x * cs.x + x.y * cs.y, x.y * cs.x - x.x * cs.y);  /* downconverter.cxx:109-110 */
			} else {
				m = mixhist[(size_t)(hl + f) * slots + s];                        /* a block shorter than the history */
			}
		}
		mixhist_next[e] = m;
		if (r == 0)
			phase_next[s] = active ? phase[s] + (unsigned int)nframes * step[s] : phase[s];
	}
}

/* r04: workgroups of LONG_ROT_WAVES waves, and -- PD2 != 0 -- the post stage of the PREVIOUS block in extra workgroups
 * [n_ddc, gridDim.x) of the same launch, as k_tuner_ddc carries it (post_role: demodulator + audio filter with audio
 * decimation PD2; the two share nothing but the kernel boundary before the launch).  LDS is dynamic: the tables, windows and
 * tap segments of the DDC waves, or the post role's stage and tile, whichever a workgroup is. */
#define LONG_ROT_WAVES 8u
#define LONG_ROT_LDS   ((2u * WR_SPLIT_N + LONG_ROT_WAVES * 2u * 64u) * 8u + LONG_ROT_WAVES * 64u * 4u)
template <unsigned int NG, unsigned int PD2>
__global__ void __launch_bounds__(LONG_ROT_WAVES * 64u)
k_tuner_ddc_long_rot(const float2 *__restrict__ cur, const uchar2 *__restrict__ cur_u8, size_t k1,
                     unsigned int d1, unsigned int len, unsigned int slots, unsigned int groups,
                     const unsigned int *__restrict__ phase, const unsigned int *__restrict__ step,
                     const int *__restrict__ flags, const float4 *__restrict__ rot, const float *__restrict__ taps,
                     const float2 *__restrict__ hi_cs, const float2 *__restrict__ lo_cs, float2 *__restrict__ chan_iq,
                     const float2 *__restrict__ mixhist, unsigned int n_ddc, WrPostArgs post, unsigned int n_post,
                     size_t nframes, unsigned int *__restrict__ phase_next, float2 *__restrict__ mixhist_next,
                     const float *__restrict__ table)
{
	extern __shared__ float long_rot_lds[];
	if (blockIdx.x >= n_ddc + n_post) {
		/* the workgroups behind the post ones: the block's state roll (it reads this state set and writes the other: nothing
		 * the DDC waves of this launch touch) -- a launch of its own until r04 */
		const unsigned int w = blockIdx.x - n_ddc - n_post, nw = gridDim.x - n_ddc - n_post;
		long_roll_body(cur, cur_u8, nframes, len, slots, phase, step, flags, phase_next, mixhist, mixhist_next, table,
		               (size_t)w * blockDim.x + threadIdx.x, (size_t)nw * blockDim.x);
		return;
	}
	if (PD2 != 0u && blockIdx.x >= n_ddc) {
		wave_prio(3u);
		constexpr unsigned int NEED = (POST_TK - 1u) * (PD2 ? PD2 : 1u) + WR_FIR_LENGTH;
		const unsigned int idx = blockIdx.x - n_ddc;
		float *stage = long_rot_lds;
		post_role<(PD2 ? PD2 : 1u), false>(post, idx % (post.ntiles + 1u), idx / (post.ntiles + 1u), stage,
		                                    stage + NEED * 64u, (int *)(stage + NEED * 64u + POST_TK * 65u));
		return;
	}
	v2f *hi_l = (v2f *)long_rot_lds, *lo_l = hi_l + WR_SPLIT_N;
	v2f (*winl)[2][64] = (v2f (*)[2][64])(lo_l + WR_SPLIT_N);
	float (*hseg)[64] = (float (*)[64])(winl + LONG_ROT_WAVES);
	for (unsigned int e = threadIdx.x; e < WR_SPLIT_N; e += blockDim.x) {
		const float2 hv = hi_cs[e], lv = lo_cs[e];
		hi_l[e] = (v2f){hv.x, hv.y};
		lo_l[e] = (v2f){lv.x, lv.y};
	}
	__syncthreads();
	const unsigned int lane = threadIdx.x & 63u;
	const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned int segs = len / 64u, hl = len - 1u, gsets = groups / NG;
	const size_t units = k1 * gsets;
	for (size_t u = (size_t)blockIdx.x * LONG_ROT_WAVES + wave; u < units; u += (size_t)n_ddc * LONG_ROT_WAVES) {
		const unsigned int gs = (unsigned int)(u % gsets);
		const size_t k = u / gsets;
		unsigned int s[NG], p0[NG], st[NG], fstep[NG];
		float4 r4[NG];
		bool active[NG];
		v2f acc[NG];
		unsigned long long act = 0;
		unsigned int su = 0;
#pragma unroll
		for (unsigned int c = 0; c < NG; ++c) {
			s[c] = (gs * NG + c) * 64u + lane;
			active[c] = (flags[s[c]] & PHASE_FLAG_ACTIVE) != 0;
			/* the filter: that of the first active channel (the host checked that the others have the same) */
			const unsigned long long a = __ballot(active[c]);
			if (!act && a)
				su = (gs * NG + c) * 64u + (unsigned int)(__ffsll((long long)a) - 1);
			act |= a;
			p0[c] = phase[s[c]];
			st[c] = step[s[c]];
			fstep[c] = st[c] << 16;
			r4[c] = rot[s[c]];
			acc[c] = (v2f){0.0f, 0.0f};
		}
		if (!act)
			continue;
		const long long n0 = (long long)k * d1 - (long long)hl;      /* the window's first frame; < 0: in the previous block */
		long long nf = n0 + lane;
		float2 xn = nf >= 0 ? input_frame(cur, cur_u8, (size_t)nf) : make_float2(0.0f, 0.0f);
		float hn = taps[(size_t)(len - 1u - lane) * slots + su];
		for (unsigned int q = 0; q < segs; ++q) {
			const long long ns = n0 + 64ll * q;                       /* this segment's first frame */
			v2f *w = winl[wave][q & 1u];
			w[lane] = (v2f){hn * xn.x, hn * xn.y};                    /* (one wave reads what it wrote: LDS is in order) */
			if (ns < 0)
				hseg[wave][lane] = hn;
			if (q + 1u < segs) {
				nf = ns + 64 + lane;
				xn = nf >= 0 ? input_frame(cur, cur_u8, (size_t)nf) : make_float2(0.0f, 0.0f);
				hn = taps[(size_t)(len - 1u - (64u * (q + 1u) + lane)) * slots + su];
			}
			unsigned int j0 = 0;                                      /* the segment's first frame inside the block */
			if (ns < 0) {
				j0 = ns + 64 <= 0 ? 64u : (unsigned int)(-ns);
				for (unsigned int j = 0; j < j0; ++j) {               /* frames as they were mixed back then */
					const float h = hseg[wave][j];
#pragma unroll
					for (unsigned int c = 0; c < NG; ++c) {
						const float2 m = mixhist[(size_t)(hl + ns + j) * slots + s[c]];
						acc[c].x = __builtin_fmaf(h, m.x, acc[c].x);
						acc[c].y = __builtin_fmaf(h, m.y, acc[c].y);
					}
				}
				if (j0 == 64u)
					continue;
			}
			unsigned int F[NG];
			v2f A[NG];
#pragma unroll
			for (unsigned int c = 0; c < NG; ++c) {
				F[c] = (p0[c] + (unsigned int)(ns + j0) * st[c]) << 16;   /* the chain's first frame: its 16 fraction bits */
				A[c] = w[j0];
			}
			if (j0 == 0) {
#pragma unroll 8
				for (unsigned int j = 1; j < 64u; ++j) {
					const v2f uj = w[j];
#pragma unroll
					for (unsigned int c = 0; c < NG; ++c) {
						unsigned int F2;
						const bool carry = __builtin_uadd_overflow(F[c], fstep[c], &F2);
						F[c] = F2;
						horner_step(A[c], carry ? r4[c].z : r4[c].x, carry ? r4[c].w : r4[c].y, uj);
					}
				}
			} else {
				for (unsigned int j = j0 + 1u; j < 64u; ++j) {
					const v2f uj = w[j];
#pragma unroll
					for (unsigned int c = 0; c < NG; ++c) {
						unsigned int F2;
						const bool carry = __builtin_uadd_overflow(F[c], fstep[c], &F2);
						F[c] = F2;
						horner_step(A[c], carry ? r4[c].z : r4[c].x, carry ? r4[c].w : r4[c].y, uj);
					}
				}
			}
#pragma unroll
			for (unsigned int c = 0; c < NG; ++c) {
				const v2f cs = nco<WR_NCO_ROTATE>(p0[c] + (unsigned int)(ns + 63) * st[c], nullptr, hi_l, lo_l);
				horner_close(acc[c], A[c], cs);
			}
		}
#pragma unroll
		for (unsigned int c = 0; c < NG; ++c)
			if (active[c])
				chan_iq[(size_t)k * slots + s[c]] = make_float2(acc[c].x, acc[c].y);
	}
}

__global__ void __launch_bounds__(256)
k_ddc_long_roll(const float2 *__restrict__ cur, const uchar2 *__restrict__ cur_u8, size_t nframes, unsigned int len,
                unsigned int slots, const unsigned int *__restrict__ phase, const unsigned int *__restrict__ step,
                const int *__restrict__ flags, unsigned int *__restrict__ phase_next,
                const float2 *__restrict__ mixhist, float2 *__restrict__ mixhist_next, const float *__restrict__ table)
{
	long_roll_body(cur, cur_u8, nframes, len, slots, phase, step, flags, phase_next, mixhist, mixhist_next, table,
	               (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

hipError_t wrk_tuner_ddc_long(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G, unsigned int len,
                              const float *table_dev, int num_cus, bool rotate, bool rotate_one_filter, const float *hi_dev,
                              const float *lo_dev, const WrPostArgs *post, bool *post_taken)
{
	if (post_taken)
		*post_taken = false;
	if (!L.slots_used)
		return hipSuccess;
	const unsigned int groups = L.slots_used / 64u;
	/* `rotate` (a tolerance nco mode, one filter per lane group): only the frames whose window reaches into the previous
	 * block take the reference's arithmetic, the others k_tuner_ddc_long_rot */
	size_t k_exact = L.k1;
	bool exact_done = false, rolled = false;
	if (rotate && L.k1) {
		const size_t units_r = L.k1 * groups;
		exact_done = true;
		k_exact = 0;
		const bool two = rotate_one_filter && groups % 2u == 0;
		const hipEvent_t e0 = (L.ev_start && L.ev_stop) ? (hipEvent_t)L.ev_start : nullptr;
		const hipEvent_t e1 = (L.ev_start && L.ev_stop) ? (hipEvent_t)L.ev_stop : nullptr;
		/* the previous block's post stage rides in this launch where the kernel is built for its audio decimation */
		const bool ride = post && post->k1 && post->groups && wrk_tuner_post_supported(post->d2);
		unsigned int wgs_r = (unsigned int)(((two ? units_r / 2 : units_r) + LONG_ROT_WAVES - 1u) / LONG_ROT_WAVES);
		/* persistent: 32 waves per CU as before -- or 24 and one workgroup slot per CU left to the post workgroups, which
		 * come and go beside the DDC ones (as in k_tuner_ddc) instead of queueing up behind them */
		const unsigned int cap8 = (unsigned int)num_cus * (ride ? 3u : 4u);
		if (wgs_r > cap8)
			wgs_r = cap8;
		const unsigned int d2r = ride ? post->d2 : 0u;
		const unsigned int post_wgs = ride ? (post->ntiles + 1u) * post->groups : 0u;
		const WrPostArgs pa = ride ? *post : WrPostArgs();
		const size_t roll_total = (size_t)(len - 1u) * L.slots;
		const unsigned int roll_wgs = (unsigned int)((roll_total + LONG_ROT_WAVES * 64u - 1u) / (LONG_ROT_WAVES * 64u) > 256u
		                                             ? 256u : (roll_total + LONG_ROT_WAVES * 64u - 1u) / (LONG_ROT_WAVES * 64u));
		rolled = true;
		size_t lds = LONG_ROT_LDS;
		if (ride) {
			const size_t need = ((size_t)((POST_TK - 1u) * d2r + WR_FIR_LENGTH) * 64u + POST_TK * 65u + 64u) * sizeof(float);
			if (need > lds)
				lds = need;
			if (post_taken)
				*post_taken = true;
		}
#define LONG_ROT_LAUNCH(NG_, PD2_) \
		hipExtLaunchKernelGGL((k_tuner_ddc_long_rot<NG_, PD2_>), dim3(wgs_r + post_wgs + roll_wgs), dim3(LONG_ROT_WAVES * 64u), (uint32_t)lds, st, \
		                      e0, e1, 0u, (const float2 *)L.cur, (const uchar2 *)L.cur_u8, L.k1, L.d1, len, L.slots, groups, \
		                      (const unsigned int *)G.phase[L.sp], (const unsigned int *)G.step, (const int *)G.flags, \
		                      (const float4 *)G.rot, (const float *)G.taps1L, (const float2 *)hi_dev, (const float2 *)lo_dev, \
		                      (float2 *)G.chan_iq[L.cb], (const float2 *)G.mixhist[L.sp], wgs_r, pa, post_wgs, L.nframes, \
		                      G.phase[L.sp ^ 1], (float2 *)G.mixhist[L.sp ^ 1], table_dev)
#define LONG_ROT_BY_D2(NG_) \
		switch (d2r) { \
		case 0: LONG_ROT_LAUNCH(NG_, 0u); break; \
		case 1: LONG_ROT_LAUNCH(NG_, 1u); break; \
		case 2: LONG_ROT_LAUNCH(NG_, 2u); break; \
		case 3: LONG_ROT_LAUNCH(NG_, 3u); break; \
		case 4: LONG_ROT_LAUNCH(NG_, 4u); break; \
		case 5: LONG_ROT_LAUNCH(NG_, 5u); break; \
		case 6: LONG_ROT_LAUNCH(NG_, 6u); break; \
		case 8: LONG_ROT_LAUNCH(NG_, 8u); break; \
		case 10: LONG_ROT_LAUNCH(NG_, 10u); break; \
		default: return hipErrorInvalidValue; \
		}
		if (two) {
			LONG_ROT_BY_D2(2u)
		} else {
			LONG_ROT_BY_D2(1u)
		}
#undef LONG_ROT_BY_D2
#undef LONG_ROT_LAUNCH
	}
	const bool prof_exact = L.ev_start && L.ev_stop && !exact_done;
	const size_t units = exact_done ? 0 : k_exact * groups;
	if (units) {
		unsigned int wgs = (unsigned int)((units + 3) / 4);
		const unsigned int cap = (unsigned int)num_cus * 8u;
		if (wgs > cap)
			wgs = cap;
		if (prof_exact)
			/* profiling: the filter kernel's own start and end (the roll behind it is not in the bracket) */
			hipExtLaunchKernelGGL(k_tuner_ddc_long, dim3(wgs), dim3(256), 0, st, (hipEvent_t)L.ev_start, (hipEvent_t)L.ev_stop, 0u,
			                      (const float2 *)L.cur, (const uchar2 *)L.cur_u8, L.nframes, k_exact, L.d1, len, L.slots, groups,
			                      (const unsigned int *)G.phase[L.sp], (const unsigned int *)G.step, (const int *)G.flags,
			                      (const float *)G.taps1L, (const float2 *)G.mixhist[L.sp], table_dev, (float2 *)G.chan_iq[L.cb]);
		else
			k_tuner_ddc_long<<<wgs, 256, 0, st>>>((const float2 *)L.cur, (const uchar2 *)L.cur_u8, L.nframes, k_exact, L.d1, len,
			                                      L.slots, groups, G.phase[L.sp], G.step, G.flags, G.taps1L,
			                                      (const float2 *)G.mixhist[L.sp], table_dev, (float2 *)G.chan_iq[L.cb]);
	}
	if (rolled)
		return hipGetLastError();                           /* the ROTATE launch rolled the state itself */
	const size_t total = (size_t)(len - 1u) * L.slots;
	const unsigned int rwgs = (unsigned int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
	if (L.ev_start && L.ev_stop && !L.k1) {
		/* profiling a block too short for an output frame: the roll is all there is */
		hipExtLaunchKernelGGL(k_ddc_long_roll, dim3(rwgs), dim3(256), 0, st, (hipEvent_t)L.ev_start, (hipEvent_t)L.ev_stop, 0u,
		                      (const float2 *)L.cur, (const uchar2 *)L.cur_u8, L.nframes, len, L.slots,
		                      (const unsigned int *)G.phase[L.sp], (const unsigned int *)G.step, (const int *)G.flags,
		                      G.phase[L.sp ^ 1], (const float2 *)G.mixhist[L.sp], (float2 *)G.mixhist[L.sp ^ 1], table_dev);
		return hipGetLastError();
	}
	k_ddc_long_roll<<<rwgs, 256, 0, st>>>((const float2 *)L.cur, (const uchar2 *)L.cur_u8, L.nframes,
	                                                            len, L.slots, G.phase[L.sp], G.step, G.flags, G.phase[L.sp ^ 1],
	                                                            (const float2 *)G.mixhist[L.sp], (float2 *)G.mixhist[L.sp ^ 1],
	                                                            table_dev);
	return hipGetLastError();
}

/* Demodulator::process for every channel (dsp/demodulator.cxx:77-115): thread (k, s);
 * the previous channel-rate frame is row k-1, or prev_iq for k = 0.  The same launch
 * also finishes the block for every channel (what DspBlock::run leaves behind in the
 * members of the four blocks of a Receiver):
 *   - Demodulator::prev_i/q := last channel-rate frame        -> prev_next (ping-pong)
 *   - audio LowPass history := last 63 demod outputs          -> dem_next rows 0..62
 * All of these go to buffers no thread of this launch reads. */
#define DEM_RPT 4u             /* consecutive rows per thread: the previous frame stays in registers */
__global__ void __launch_bounds__(256)
k_tuner_demod(const float2 *__restrict__ chan_iq, unsigned int k1, unsigned int slots,
              const int *__restrict__ mode,
              const float2 *__restrict__ prev_iq, float2 *__restrict__ prev_next,
              float *__restrict__ dem, float *__restrict__ dem_next, unsigned int hist)
{
	/* `hist` = the audio filter's taps - 1: 63, or 127 / 255 (LowPass::_firLength, lowpass.cxx:38-39) */
	/* thread = (slot lane, row lane): 64 slots x 4 row lanes per workgroup, DEM_RPT rows each */
	const unsigned int lane = threadIdx.x & 63u;
	const unsigned int rl = threadIdx.x >> 6;
	const unsigned int s = blockIdx.x * 64u + lane;
	const unsigned int kbeg = (blockIdx.y * 4u + rl) * DEM_RPT;
	const int m = mode[s];                              /* < 0: idle slot */
	if (m >= 0 && kbeg < k1) {
		const unsigned int kend = (kbeg + DEM_RPT < k1) ? kbeg + DEM_RPT : k1;
		/* all loads first: the rows are independent, their latency should overlap */
		float2 z[DEM_RPT + 1];
		z[0] = kbeg ? chan_iq[(size_t)(kbeg - 1u) * slots + s] : prev_iq[s];
#pragma unroll
		for (unsigned int i = 0; i < DEM_RPT; ++i)
			z[i + 1] = (kbeg + i < kend) ? chan_iq[(size_t)(kbeg + i) * slots + s] : make_float2(0.0f, 0.0f);
		float2 p = z[0];
#pragma unroll
		for (unsigned int i = 0; i < DEM_RPT; ++i) {
			const unsigned int k = kbeg + i;
			if (k < kend) {
				const float v = demod_one(m, z[i + 1].x, z[i + 1].y, z[i].x, z[i].y);
				dem[(size_t)(hist + k) * slots + s] = v;
				if (k + hist >= k1)                     /* among the last `hist` rows */
					dem_next[(size_t)(k + hist - k1) * slots + s] = v;
				p = z[i + 1];
			}
		}
		if (kend == k1)
			prev_next[s] = p;
	}
	/* blocks shorter than the history: the older part of the next history comes from the
	 * current history rows (not written by this launch) */
	if (k1 < hist && blockIdx.y == 0 && m >= 0)
		for (unsigned int r = rl; r < hist - k1; r += 4u)
			dem_next[(size_t)r * slots + s] = dem[(size_t)(k1 + r) * slots + s];
}

/* audio LowPass::process for every channel (dsp/lowpass.cxx:131-162, 1 channel):
 * dem already carries its 63 history rows, so row k2*D2 + j is tap j's sample.
 * Workgroup = 64 channel slots x `tk` output frames.  The (tk-1)*D2 + 64 demod rows the
 * tile needs are staged in LDS once (each row is read by up to 64/D2 outputs), lanes
 * run over slots so both the global rows and the LDS rows are contiguous per wave, and
 * the finished 64 x tk tile is transposed through LDS so that audio[s][k2] (channel
 * major, what each audio sink consumes) is written in runs. */
#define AUD_ROWS 352u          /* staged demod rows: 352 x 64 x 4 B = 88 KiB */
#define AUD_TMAX 16u
#define AUD_THREADS 1024u
__global__ void __launch_bounds__(AUD_THREADS)
k_tuner_audio(const float *__restrict__ dem, size_t rows_valid, size_t k2, unsigned int d2,
              unsigned int tk, unsigned int slots, const float *__restrict__ taps2,
              const int *__restrict__ mode, float *__restrict__ audio, size_t k2max, float scale,
              const float *__restrict__ gain, const float *__restrict__ squelch, const float2 *__restrict__ chan_iq,
              unsigned int len)
{
	extern __shared__ float aud_lds[];      /* [AUD_ROWS][64] rows, [len][64] taps, [AUD_TMAX][65] out; len = 64, 128 or 256 taps */
	const unsigned int need = (tk - 1u) * d2 + len;
	float *stage = aud_lds;                             /* [need][64] */
	float *taps = aud_lds + need * 64u;
	float *tile = taps + len * 64u;
	const unsigned int lane = threadIdx.x & 63u;        /* slot within the group */
	const unsigned int row = threadIdx.x >> 6;          /* 0..15 */
	const unsigned int nrow = AUD_THREADS / 64u;
	const unsigned int g = blockIdx.y;
	const unsigned int s = g * 64u + lane;
	const size_t kbase = (size_t)blockIdx.x * tk;

	for (unsigned int j = row; j < len; j += nrow)
		taps[j * 64u + lane] = taps2[(size_t)j * slots + s];
	const size_t r0 = kbase * d2;
	for (unsigned int r = row; r < need; r += nrow) {
		const size_t rr = r0 + r;
		stage[r * 64u + lane] = (rr < rows_valid) ? dem[rr * slots + s] : 0.0f;
	}
	__syncthreads();
	for (unsigned int kk = row; kk < tk; kk += nrow) {
		const float *x = stage + (kk * d2) * 64u + lane;
		float acc = 0.0f;
		const float *hrev = taps + (len - 1u) * 64u + lane;
#pragma unroll 16
		for (unsigned int j = 0; j < len; ++j)
			acc = acc + hrev[-(int)(j * 64u)] * x[j * 64u];
		tile[kk * 65u + lane] = acc;
	}
	__syncthreads();
	/* transposed write: tk consecutive frames of one slot per tk threads */
	for (unsigned int e = threadIdx.x; e < 64u * tk; e += AUD_THREADS) {
		const unsigned int sl = e / tk, kk = e - sl * tk;
		const unsigned int so = g * 64u + sl;
		const size_t k = kbase + kk;
		if (k < k2 && mode[so] >= 0)
			audio[(size_t)so * k2max + k] = audio_out(tile[kk * 65u + sl], gain, squelch, chan_iq, slots, so, k, d2, scale);
	}
}

/* A second channel LowPass::process (dsp/lowpass.cxx:131-162, 2 channels) on the channel-rate IQ of
 * every receiver, between the DDC and the demodulator: out[k][s] = sum_j coeff[L - 1 - j] * X[k*D + j][s]
 * over X = [L - 1 history rows | the block's k1a first-stage rows], products added oldest row first,
 * unfused, I and Q apart (lowpass.cxx:150-158 walks the channels innermost).  L = 64, 128 or 256 taps.
 * r05: tiled through LDS.  Workgroup = 64 slots x `tk` output frames; per SEGMENT of 64 taps, oldest rows
 * first, the (tk - 1) D + 64 rows the tile's frames meet and the segment's 64 x 64 taps are staged once
 * (whole 512-byte / 256-byte rows per wave) and every wave runs its output frames (IQ2_B of them share a
 * tap read) over them, the sums carried from segment to segment in registers -- the order of additions
 * is the reference's.  (r02-r04: every thread read its 64 taps and 64 rows straight from L2: 20 / 42 / 84 us
 * per C2 block at 64 / 128 / 256 taps.)  blockIdx.x == tiles: the next history (last L - 1 rows). */
static hipError_t allow_lds(const void *fn, size_t bytes, bool (&done)[WR_MAX_DEVICES]);
#define IQ2_ROWS 128u           /* staged rows: 128 x 64 x 8 B = 64 KiB; + 16 KiB of taps: two workgroups per CU */
#define IQ2_B 2u                /* output frames per wave */
#define IQ2_TKMAX (8u * IQ2_B)
__global__ void __launch_bounds__(512)
k_tuner_iq2(const float2 *__restrict__ in, size_t k1a, unsigned int d1b, unsigned int slots,
            const float *__restrict__ taps, const int *__restrict__ mode,
            const float2 *__restrict__ hist, float2 *__restrict__ hist_next, float2 *__restrict__ out, unsigned int tiles,
            unsigned int len, unsigned int tk)
{
	extern __shared__ float iq2_lds[];                  /* [need][64] float2 rows, then [64][64] taps of the segment */
	const unsigned int nh = len - 1u;                   /* history rows: 63, or 127 / 255 for a stage of 128 / 256 taps */
	const unsigned int lane = threadIdx.x & 63u, row = threadIdx.x >> 6;
	const unsigned int s = blockIdx.y * 64u + lane;
	const int m = mode[s];
	auto xrow = [&](size_t r) -> float2 {              /* row r of [history | block] */
		return (r < nh) ? hist[r * slots + s] : in[(r - nh) * slots + s];
	};
	if (blockIdx.x == tiles) {
		if (m >= 0)
			for (unsigned int r = row; r < nh; r += 8u)
				hist_next[(size_t)r * slots + s] = xrow(k1a + r);
		return;
	}
	const unsigned int need = (tk - 1u) * d1b + WR_FIR_LENGTH;
	float2 *st = (float2 *)iq2_lds;
	float *tp = iq2_lds + (size_t)need * 128u;
	const size_t kout = k1a / d1b;
	const size_t k0 = (size_t)blockIdx.x * tk;
	const size_t rows_valid = (size_t)nh + k1a;
	float ai[IQ2_B], aq[IQ2_B];
#pragma unroll
	for (unsigned int b = 0; b < IQ2_B; ++b)
		ai[b] = aq[b] = 0.0f;
	const unsigned int nseg = len / WR_FIR_LENGTH;
	/* a thread's share of a segment -- 8 tap rows, up to 16 window rows -- is loaded into registers in one go (one
	 * memory round trip per segment, not one per row) and the NEXT segment's while this one is being summed */
	float tv[WR_FIR_LENGTH / 8u];
	float2 xv[IQ2_ROWS / 8u];
	auto fetch = [&](unsigned int seg) {
		const size_t r0 = k0 * d1b + (size_t)seg * WR_FIR_LENGTH;
		const unsigned int tap0 = (nseg - 1u - seg) * WR_FIR_LENGTH;
#pragma unroll
		for (unsigned int i = 0; i < WR_FIR_LENGTH / 8u; ++i)
			tv[i] = taps[((size_t)tap0 + row + 8u * i) * slots + s];
#pragma unroll
		for (unsigned int i = 0; i < IQ2_ROWS / 8u; ++i) {
			const unsigned int r = row + 8u * i;
			const size_t rr = r0 + r;
			xv[i] = (m >= 0 && r < need && rr < rows_valid) ? xrow(rr) : make_float2(0.0f, 0.0f);
		}
	};
	fetch(0);
	for (unsigned int seg = 0; seg < nseg; ++seg) {
		if (seg)
			__syncthreads();                            /* the segment before has been read */
#pragma unroll
		for (unsigned int i = 0; i < WR_FIR_LENGTH / 8u; ++i)
			tp[(row + 8u * i) * 64u + lane] = tv[i];
#pragma unroll
		for (unsigned int i = 0; i < IQ2_ROWS / 8u; ++i) {
			const unsigned int r = row + 8u * i;
			if (r < need)
				st[r * 64u + lane] = xv[i];
		}
		__syncthreads();
		if (seg + 1u < nseg)
			fetch(seg + 1u);
#pragma unroll 8
		for (unsigned int j = 0; j < WR_FIR_LENGTH; ++j) {
			const float c = tp[(WR_FIR_LENGTH - 1u - j) * 64u + lane];
#pragma unroll
			for (unsigned int b = 0; b < IQ2_B; ++b) {
				const unsigned int kk = row + 8u * b;   /* (rows beyond the tile's: within the staged window or not read) */
				if (kk < tk) {
					const float2 x = st[(kk * d1b + j) * 64u + lane];
					ai[b] = ai[b] + c * x.x;
					aq[b] = aq[b] + c * x.y;
				}
			}
		}
	}
	if (m < 0)
		return;
#pragma unroll
	for (unsigned int b = 0; b < IQ2_B; ++b) {
		const unsigned int kk = row + 8u * b;
		if (kk < tk && k0 + kk < kout)
			out[(k0 + kk) * slots + s] = make_float2(ai[b], aq[b]);
	}
}

hipError_t wrk_tuner_iq2(hipStream_t st, const WrGroupDev &G, unsigned int slots, unsigned int slots_used,
                         size_t k1a, unsigned int d1b, int cb, int p2)
{
	if (!slots_used || !d1b)
		return hipSuccess;
	/* as many output frames per tile as the staged rows allow (D = 5: 13; D = 10: 7; D >= 64: one) */
	unsigned int tk = IQ2_TKMAX;
	if ((IQ2_ROWS - WR_FIR_LENGTH) / d1b + 1u < tk)
		tk = (IQ2_ROWS - WR_FIR_LENGTH) / d1b + 1u;
	const unsigned int need = (tk - 1u) * d1b + WR_FIR_LENGTH;
	const size_t lds = ((size_t)need * 128u + 64u * 64u) * sizeof(float);
	static bool attr_done[WR_MAX_DEVICES];
	{
		hipError_t e = allow_lds((const void *)k_tuner_iq2, ((size_t)IQ2_ROWS * 128u + 64u * 64u) * sizeof(float), attr_done);
		if (e != hipSuccess)
			return e;
	}
	const unsigned int tiles = (unsigned int)((k1a / d1b + tk - 1) / tk);
	dim3 grid(tiles + 1u, slots_used / 64);
	k_tuner_iq2<<<grid, 512, lds, st>>>((const float2 *)G.chan_iq[cb], k1a, d1b, slots, G.taps1b, G.mode,
	                                    (const float2 *)G.iq2_hist[p2], (float2 *)G.iq2_hist[p2 ^ 1],
	                                    (float2 *)G.chan_iq2[cb], tiles, G.l1b, tk);
	return hipGetLastError();
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

/* hipFuncAttributeMaxDynamicSharedMemorySize is a per-device setting of the loaded code object:
 * a process that drives several GPUs (the host runtime gives every tuner its own) has to set
 * it once on each.  `done` is the call site's own flag array. */
static hipError_t allow_lds(const void *fn, size_t bytes, bool (&done)[WR_MAX_DEVICES])
{
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess)
		return e;
	if (dev < 0 || dev >= WR_MAX_DEVICES)
		return hipErrorInvalidDevice;
	if (done[dev])
		return hipSuccess;
	e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
	if (e == hipSuccess)
		done[dev] = true;
	return e;
}

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
                   unsigned int decim, unsigned int fir_length, const float *coeff_dev, const float *hist_dev,
                   float *out)
{
	size_t outfloats = (nframes / decim) * channels;
	if (!outfloats)
		return hipSuccess;
	k_fir<<<grid_for(outfloats, 256, 4096), 256, 0, st>>>(in, outfloats, channels, decim, fir_length, coeff_dev,
	                                                      hist_dev, out);
	return hipGetLastError();
}

hipError_t wrk_hist_update(hipStream_t st, const float *in, size_t nframes, unsigned int channels,
                           unsigned int fir_length, float *hist_dev, float *scratch_dev)
{
	unsigned int total = (fir_length - 1u) * channels;
	if (!total)
		return hipSuccess;
	k_hist_build<<<grid_for(total, 256, 64), 256, 0, st>>>(in, nframes, channels, fir_length - 1u, hist_dev,
	                                                      scratch_dev);
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

/* ---- sparse staging (r04): only the frames a decimating tuner's taps ever reach cross PCIe ------------------------------
 * A channel filter of `len` taps that decimates by `period` reads the input frames [k * period - (len - 1), k * period]
 * of every output frame k (lowpass.cxx:145-159: y[k] = sum_j coeff[len-1-j] * block[k * D + j], block = 63 frames of
 * history ++ the input) and nothing between those windows -- 64 of every 400 frames at BASELINE config 2, 64 of 4000 at
 * config 5 -- plus, for the next block's history and a SpectrumSink's current frame, a tail of the block.  k_stage_windows
 * copies exactly that from page-locked HOST memory (read by the kernel over PCIe) to the same positions of the staged
 * device block: the frames in between keep whatever the buffer held and nobody reads them.
 * A unit is one 16-byte chunk of the 16-byte-ALIGNED span that covers a window (what lies beside a window inside that span
 * is staged too, it is input all the same); NC chunks per window, 64 / NC windows per wave and iteration.
 *   u8  source: 8 frames per chunk, converted with the reference's rule (u8 - 128) / 128 (rtlsdrtuner.cxx:106)
 *   f32 source: 2 frames per chunk */
template <bool U8>
__global__ void __launch_bounds__(256) k_stage_windows(const uint8_t *__restrict__ src, float *__restrict__ dst, size_t nframes,
                                                        unsigned int period, unsigned int len, size_t k1, size_t tail_first,
                                                        unsigned int nc, unsigned int wpw)
{
	constexpr unsigned int FB = U8 ? 2u : 8u;           /* bytes per frame at the source */
	const unsigned int lane = threadIdx.x & 63u;
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const size_t total_bytes = nframes * FB;
	const unsigned int ncl = nc > 64u ? 64u : nc;       /* lanes a window gets (a window longer than 64 chunks: in rounds) */
	const unsigned int w = lane / ncl, c0 = lane - w * ncl;
	/* windows: wave-iteration i covers windows i * wpw ... */
	for (size_t k0 = wave * wpw; k0 < k1; k0 += nwaves * wpw)
	for (unsigned int c = c0; c < nc; c += ncl) {
		const size_t k = k0 + w;
		if (w >= wpw || k >= k1)
			continue;
		const long long f0 = (long long)k * period - (long long)(len - 1u);       /* first frame of the window */
		const size_t b0 = (size_t)(f0 < 0 ? 0 : f0) * FB, b1 = ((size_t)k * period + 1u) * FB;
		const size_t a = (b0 & ~(size_t)15) + (size_t)c * 16u;
		if (a >= b1 || a + 16u > total_bytes)
			continue;                                   /* (a block's last bytes: the tail pass below) */
		const uint4 v = *(const uint4 *)(src + a);
		if (U8) {
			const unsigned int q[4] = {v.x, v.y, v.z, v.w};
			float4 *o = (float4 *)(dst + a);            /* byte a of the source is float a of the block */
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				float4 r;
				r.x = ((float)(q[i] & 255u) - 128.0f) / 128.0f;
				r.y = ((float)((q[i] >> 8) & 255u) - 128.0f) / 128.0f;
				r.z = ((float)((q[i] >> 16) & 255u) - 128.0f) / 128.0f;
				r.w = ((float)(q[i] >> 24) - 128.0f) / 128.0f;
				o[i] = r;
			}
		} else {
			*(uint4 *)((uint8_t *)dst + a) = v;
		}
	}
	/* the tail [tail_first, nframes): frame by frame (its start need not be 16-byte aligned at the source) */
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
	for (size_t f = tail_first + tid; f < nframes; f += nthreads) {
		if (U8) {
			const uchar2 b = ((const uchar2 *)src)[f];
			((float2 *)dst)[f] = make_float2(((float)b.x - 128.0f) / 128.0f, ((float)b.y - 128.0f) / 128.0f);
		} else {
			((float2 *)dst)[f] = ((const float2 *)src)[f];
		}
	}
}

hipError_t wrk_stage_windows(hipStream_t st, const void *src_mapped, bool u8, float *dst, size_t nframes, unsigned int period,
                             unsigned int len, size_t tail_frames)
{
	if (!nframes)
		return hipSuccess;
	const unsigned int fb = u8 ? 2u : 8u;
	const unsigned int nc = (len * fb + 15u + 15u) / 16u;              /* chunks of the aligned span of a window */
	const unsigned int wpw = nc > 64u ? 1u : 64u / nc;
	const size_t k1 = nframes / period + ((nframes % period) ? 1u : 0u);  /* (a window that ends in the next block starts in this one) */
	/* (the window pass moves whole 16-byte chunks and leaves a block's last, partial one to the tail pass: the tail is at
	 * least those 8 frames, whatever the caller asked for) */
	if (tail_frames < 8)
		tail_frames = 8;
	const size_t tail_first = tail_frames >= nframes ? 0 : nframes - tail_frames;
	/* PCIe-bound: a few hundred workgroups keep megabytes of reads in flight; see k_u8_to_f32_x16 about parking more */
	const size_t waves = (k1 + wpw - 1) / wpw;
	unsigned int wgs = (unsigned int)((waves + 3) / 4);
	if (wgs > 512u)
		wgs = 512u;
	if (wgs < 1u)
		wgs = 1u;
	if (u8)
		k_stage_windows<true><<<wgs, 256, 0, st>>>((const uint8_t *)src_mapped, dst, nframes, period, len, k1, tail_first, nc, wpw);
	else
		k_stage_windows<false><<<wgs, 256, 0, st>>>((const uint8_t *)src_mapped, dst, nframes, period, len, k1, tail_first, nc, wpw);
	return hipGetLastError();
}

hipError_t wrk_u8_to_f32(hipStream_t st, const uint8_t *in, float *out, size_t count)
{
	if (!count)
		return hipSuccess;
	size_t head = 0;
	hipPointerAttribute_t pa;                           /* host memory read over PCIe: few workgroups (see k_u8_to_f32_x16) */
	const bool zc = hipPointerGetAttributes(&pa, in) == hipSuccess && pa.type == hipMemoryTypeHost;
	(void)hipGetLastError();
	if ((((uintptr_t)in | (uintptr_t)out) & 15u) == 0 && count >= 16) {
		head = count & ~(size_t)15;
		k_u8_to_f32_x16<<<grid_for(head / 16, 256, zc ? U8_X16_WGS : 2048u), 256, 0, st>>>((const uint4 *)in, (float4 *)out, head / 16);
	}
	if (head < count)
		k_u8_to_f32<<<grid_for(count - head, 256, 2048), 256, 0, st>>>(in + head, out + head, count - head);
	return hipGetLastError();
}

/* launch geometry of k_tuner_ddc<NCO, UTAPS> (waves per workgroup, persistent workgroups per CU):
 *   ROTATE, uniform taps : 8 x 4 -- 64 VGPRs, 20 KiB of LDS: 8 waves per SIMD hide the recurrence's
 *                                  dependent FMAs better than 4 (measured 35.6 -> 34.7 us)
 *   ROTATE, per-lane taps: 16 x 1 -- one lane group's taps (16 KiB) per workgroup
 *   SPLIT                : 16 x 1 -- the replicated tables take 128 KiB
 *   EXACT                : 16 x 2 -- small LDS footprint, two workgroups hide the gather latency */
/* include/webradio_amd.h: WR_TUNE_DDC_NG2_MIN_PASSES */
static std::atomic<long> g_ng2_min_passes{-1};
long wrk_tune_ng2_min_passes(long value, bool set)
{
	static const long builtin = [] {
		const char *e = getenv("WR_DDC_NG2_MIN_PASSES");
		return (long)((e && *e) ? strtoul(e, nullptr, 0) : 4ul);
	}();
	const long v = g_ng2_min_passes.load(std::memory_order_relaxed);
	const long before = v < 0 ? builtin : v;
	if (set)
		g_ng2_min_passes.store(value < 0 ? -1 : value, std::memory_order_relaxed);
	return before;
}

template <int NCO, bool UTAPS> struct DdcGeom {
	static constexpr unsigned int waves = (NCO == WR_NCO_ROTATE && (UTAPS || DDC_LTAPS_SMALL)) ? DDC_ROTATE_WAVES : DDC_WAVES;
	static constexpr unsigned int wgs_per_cu = (NCO == WR_NCO_ROTATE && (UTAPS || DDC_LTAPS_SMALL)) ? DDC_ROTATE_WGS_PER_CU
	                                           : (NCO == WR_NCO_EXACT) ? 2u : 1u;
};

/* `gsel`: bit g set = lane group g takes part in this launch (0 = all of them); `whole`: the launch
 * also rolls the per-channel state of every group (exactly one launch per block does) */
template <int NCO, bool UTAPS, unsigned int PD2, unsigned int NG = 1u>
static hipError_t launch_ddc(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G,
                             const float *table_dev, const float *hi_dev, const float *lo_dev, int num_cus,
                             const WrPostArgs *post, unsigned long long gsel = 0, bool whole = true)
{
	constexpr unsigned int W = DdcGeom<NCO, UTAPS>::waves;
	if constexpr (NG == 1u && DDC_ROLES && DDC_NG2 && NCO == WR_NCO_ROTATE && UTAPS) {
		/* two lane groups per wave (see k_tuner_ddc: NG) where the launch allows: an even number of lane groups, at
		 * most 16, all on one and the same channel filter */
		unsigned int n = 0;
		for (unsigned int g = 0; g < L.slots_used / 64; ++g)
			n += (!gsel || ((gsel >> g) & 1ull)) ? 1u : 0u;
		/* ... and enough output frames: a wave of the NG = 2 kernel does twice the work per pass of its loop, so
		 * with few passes per wave the grid ends ragged (BASELINE config 5: 5 065 frames per chunk, 2.5 passes per
		 * wave -- 48 us against 43 with one lane group per wave); from four passes on the shared reads win */
		const size_t waves2 = (size_t)num_cus * (DDC_NG2_WAVES_PER_EU * 4u / W - (PD2 ? 1u : 0u)) * W;
		const size_t min_passes = (size_t)wrk_tune_ng2_min_passes(0, false);      /* wr_tune: 0 = always */
		if (L.one_filter && n >= 2u && n <= 16u && (n & 1u) == 0u && L.k1 * (n / 2u) >= min_passes * waves2)
			return launch_ddc<NCO, UTAPS, PD2, 2u>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, post, gsel, whole);
	}
	const unsigned int allgroups = L.slots_used / 64;        /* <= 64: wr_tuner_create caps max_channels */
	unsigned long long gmap[2] = {0, 0};
	unsigned int ngroups = 0;
	unsigned long long rest = 0;                             /* selected groups beyond the 16 one launch maps */
	for (unsigned int g = 0; g < allgroups; ++g)
		if (!gsel || ((gsel >> g) & 1ull)) {
			if (ngroups < 16u) {
				gmap[ngroups >> 3] |= (unsigned long long)g << ((ngroups & 7u) * 8u);
				++ngroups;
			} else {
				rest |= 1ull << g;
			}
		}
	if (!ngroups)
		return hipSuccess;
	if (rest) {
		/* more than 1024 channels: the groups beyond the first 16 go out in launches of their own,
		 * which neither roll the state again nor carry the post stage or the profiling events */
		WrTunerLaunch L2 = L;
		L2.ev_start = L2.ev_stop = nullptr;
		hipError_t e = launch_ddc<NCO, UTAPS, 0>(st, L2, G, table_dev, hi_dev, lo_dev, num_cus, nullptr, rest, false);
		if (e != hipSuccess)
			return e;
	}
	/* tap sets in this launch: the largest count among the groups it covers (SPLIT: one, its
	 * replicated tables leave no room for more windows) */
	unsigned int kmax = 1;
	if (UTAPS && NCO == WR_NCO_ROTATE)
		for (unsigned int i = 0; i < ngroups; ++i) {
			const unsigned int g = (unsigned int)((gmap[i >> 3] >> ((i & 7u) * 8u)) & 255u);
			if (g < 64u && L.nsets[g] > kmax)
				kmax = L.nsets[g];
		}
	size_t lds = (NCO == WR_NCO_SPLIT) ? DDC_LDS_BYTES
	             : (W * 2u * 512u * kmax) + (NCO == WR_NCO_ROTATE ? 2u * WR_SPLIT_N * 8u : 0u)
	               + (NCO == WR_NCO_ROTATE && !UTAPS ? WR_FIR_LENGTH * 64u * 4u : 0u);
	/* NG = 2: 80 registers, 6 waves per SIMD: three 8-wave workgroups per CU */
	unsigned int wgs_per_cu = (NG == 2u) ? DDC_NG2_WAVES_PER_EU * 4u / W : DdcGeom<NCO, UTAPS>::wgs_per_cu;
	unsigned int post_wgs = 0;
	if (PD2 != 0u) {
		/* the post workgroups' stage + tile set the LDS size of every workgroup of the launch;
		 * POST_RESERVE workgroup slots per CU are left to them (they come and go, the DDC ones
		 * persist).  Measured at C2, kernel duration (r02: per-slot turns, priorities by progress,
		 * post workgroups at priority 3): 1 slot 38.2 us, 2 slots 39.7, 3 slots 49.8 (the DDC
		 * starves); DDC + post as two launches: 31.3 + 15.2 + gap. */
		constexpr unsigned int NEED = (POST_TK - 1u) * (PD2 ? PD2 : 1u) + WR_FIR_LENGTH;
		const size_t post_lds = ((size_t)NEED * 64u + POST_TK * 65u + 64u) * sizeof(float);
		if (post_lds > lds)
			lds = post_lds;
		unsigned int fit = (unsigned int)((160u * 1024u) / lds);
		if (fit < wgs_per_cu)
			wgs_per_cu = fit;
#ifndef POST_RESERVE
#define POST_RESERVE 1u
#endif
		wgs_per_cu = (wgs_per_cu > POST_RESERVE) ? wgs_per_cu - POST_RESERVE : 1u;
		post_wgs = (post->ntiles + 1u) * post->groups;
	}
	/* launched even for a block too short to yield a channel-rate frame: workgroup 0 still
	 * advances the NCO (downconverter.cxx:103 runs per input frame) and rolls the histories */
	/* (DDC_ROLES) the first ceil(63 / D1) output frames of the block reach into the previous one: they go to
	 * n_bnd workgroups of their own, one wave per (lane group, frame), and the persistent waves share the rest */
	constexpr bool ROLES = DDC_ROLES && NCO == WR_NCO_ROTATE && UTAPS;
	unsigned int kslow = 0, n_bnd = 0;
	if (ROLES && L.k1) {
		kslow = (WR_HIST + L.d1 - 1u) / L.d1;
		if (kslow > L.k1)
			kslow = (unsigned int)L.k1;
		n_bnd = (kslow * ngroups + W - 1u) / W;
	}
	const size_t units = (L.k1 - kslow) * (ngroups / NG);        /* per wave and pass of its loop: NG lane groups */
	unsigned int wgs = (unsigned int)((units + W - 1) / W);
	const unsigned int cap = (unsigned int)num_cus * wgs_per_cu;
	if (wgs > cap) {
#if DDC_DEAL_WAYS > 1u
		wgs = cap;                                     /* every slot: the kernel evens the units out (DDC_DEAL_WAYS) */
#else
		/* as many workgroups as make the units come out EVEN: with every slot filled (6 144 waves for
		 * C2's 40 000 units: 6.5 units per wave) half the waves take one unit more than the others
		 * and run the last round at half occupancy; 715 of the 768 workgroups give every wave seven */
		const size_t rounds = (units + (size_t)cap * W - 1) / ((size_t)cap * W);
		wgs = (unsigned int)((units + (size_t)W * rounds - 1) / ((size_t)W * rounds));
		if (wgs > cap)
			wgs = cap;
#endif
	}
	if (NCO == WR_NCO_ROTATE && !UTAPS) {
		/* per-lane taps live in LDS, one lane group per WORKGROUP: a whole number of
		 * workgroups per group, at least one */
		wgs = (wgs / ngroups) * ngroups;
		if (wgs < ngroups)
			wgs = ngroups;
	} else {
		/* every lane group needs at least one wave of its own (the kernel deals waves to groups) */
		const unsigned int min_wgs = (ngroups / NG + W - 1) / W;
		if (wgs < min_wgs)
			wgs = min_wgs;
	}
	static bool attr_done[WR_MAX_DEVICES];
	if (lds > 64 * 1024) {
		hipError_t e = allow_lds((const void *)k_tuner_ddc<NCO, UTAPS, PD2, NG>, lds, attr_done);
		if (e != hipSuccess)
			return e;
	}
	const WrPostArgs pa = post ? *post : WrPostArgs();
	if (L.ev_start || L.ev_stop) {
		/* profiling (both events; with two launches per rate group the first takes the start, the last the stop) or a caller that wants to wait for THIS launch from another stream (the stop event
		 * alone, wr_tuner_mark_launches): the launch stamps the events with the dispatch's own start and end, as
		 * rocprof sees them -- events recorded around it would add their own barrier packets */
		hipExtLaunchKernelGGL((k_tuner_ddc<NCO, UTAPS, PD2, NG>), dim3(wgs + n_bnd + post_wgs), dim3(W * 64u), (uint32_t)lds, st,
		                      (hipEvent_t)L.ev_start, (hipEvent_t)L.ev_stop, 0u,
		                      (const float2 *)L.cur, (const uchar2 *)L.cur_u8, (const float2 *)L.hist, (float2 *)L.hist_next,
		                      L.nframes, L.k1, L.d1, L.slots, ngroups, (const unsigned int *)G.phase[L.sp],
		                      (const unsigned int *)G.step, (const float2 *)G.hist_cs[L.sp], (const int *)G.flags,
		                      G.phase[L.sp ^ 1], (float2 *)G.hist_cs[L.sp ^ 1], (const float2 *)G.hist_lo[L.sp],
		                      (float2 *)G.hist_lo[L.sp ^ 1], (const float *)G.taps1, (const float4 *)G.rot, (const float *)G.taps1u,
		                      (const int *)G.tapsel, kmax, (float2 *)G.chan_iq[L.cb], table_dev,
		                      (const float2 *)hi_dev, (const float2 *)lo_dev, wgs, pa, gmap[0], gmap[1], whole ? 1 : 0, kslow, n_bnd, L.seeking ? 1u : 0u, L.seek_lo);
		return hipGetLastError();
	}
	k_tuner_ddc<NCO, UTAPS, PD2, NG><<<wgs + n_bnd + post_wgs, W * 64u, lds, st>>>(
		(const float2 *)L.cur, (const uchar2 *)L.cur_u8, (const float2 *)L.hist, (float2 *)L.hist_next, L.nframes,
		L.k1, L.d1,
		L.slots, ngroups, G.phase[L.sp], G.step, (const float2 *)G.hist_cs[L.sp], G.flags, G.phase[L.sp ^ 1],
		(float2 *)G.hist_cs[L.sp ^ 1], (const float2 *)G.hist_lo[L.sp], (float2 *)G.hist_lo[L.sp ^ 1], G.taps1,
		(const float4 *)G.rot, G.taps1u, G.tapsel, kmax, (float2 *)G.chan_iq[L.cb], table_dev,
		(const float2 *)hi_dev, (const float2 *)lo_dev, wgs, pa, gmap[0], gmap[1], whole ? 1 : 0, kslow, n_bnd, L.seeking ? 1u : 0u, L.seek_lo);
	return hipGetLastError();
}

/* the launch with the previous block's post stage riding: the audio decimation is a template parameter of the
 * post role (which tap meets which staged row is resolved at compile time) */
template <int NCO, bool UTAPS>
static hipError_t launch_ddc_riding(unsigned int d2, hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G,
                                    const float *table_dev, const float *hi_dev, const float *lo_dev, int num_cus,
                                    const WrPostArgs *post, unsigned long long gsel, bool whole)
{
	constexpr bool R = NCO == WR_NCO_ROTATE;
	switch (d2) {
	case 1: return launch_ddc<NCO, UTAPS, (R ? 1u : 0u)>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, post, gsel, whole);
	case 2: return launch_ddc<NCO, UTAPS, (R ? 2u : 0u)>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, post, gsel, whole);
	case 3: return launch_ddc<NCO, UTAPS, (R ? 3u : 0u)>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, post, gsel, whole);
	case 4: return launch_ddc<NCO, UTAPS, (R ? 4u : 0u)>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, post, gsel, whole);
	case 5: return launch_ddc<NCO, UTAPS, (R ? 5u : 0u)>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, post, gsel, whole);
	case 6: return launch_ddc<NCO, UTAPS, (R ? 6u : 0u)>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, post, gsel, whole);
	case 8: return launch_ddc<NCO, UTAPS, (R ? 8u : 0u)>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, post, gsel, whole);
	case 10: return launch_ddc<NCO, UTAPS, (R ? 10u : 0u)>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, post, gsel, whole);
	default: return hipErrorInvalidValue;      /* (wrk_tuner_post_supported keeps other decimations away) */
	}
}

template <int NCO>
static hipError_t launch_ddc_fast(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G, const float *table_dev,
                                  const float *hi_dev, const float *lo_dev, int num_cus, const WrPostArgs *post,
                                  bool *post_taken)
{
	/* Lane groups whose 64 channels share one channel filter take the uniform-taps kernel, the
	 * others the per-lane-taps one: one odd receiver costs its own lane group, not the tuner.
	 * (A launch maps 16 lane groups; launch_ddc sends further ones out in launches of their own.) */
	const unsigned int allgroups = L.slots_used / 64;
	const unsigned long long all = (allgroups >= 64u) ? ~0ull : ((1ull << allgroups) - 1ull);
	/* ROTATE takes a group with up to WR_TAPSETS distinct filters, SPLIT one with a single filter */
	const unsigned long long uni = ((NCO == WR_NCO_ROTATE) ? L.fewsets_mask : L.uniform_mask) & all;
	const unsigned long long odd = all & ~uni;
	bool whole = true;
	if (uni) {
		hipError_t e;
		const unsigned long long sel = (uni == all) ? 0ull : uni;
		/* two launches (uniform + odd groups): the FIRST stamps the start event, the LAST the stop event -- the
		 * completion mark of wr_tuner_mark_launches must not fire while the second launch still reads the block
		 * (r03 gave both to the first: a halo exchange ordered behind the mark could overwrite its input) */
		WrTunerLaunch L1 = L;
		if (odd)
			L1.ev_stop = nullptr;
		if (NCO == WR_NCO_ROTATE && post && post->k1 && post->groups && wrk_tuner_post_supported(post->d2)) {
			if (post_taken)
				*post_taken = true;
			e = launch_ddc_riding<NCO, true>(post->d2, st, L1, G, table_dev, hi_dev, lo_dev, num_cus, post, sel, true);
		} else {
			e = launch_ddc<NCO, true, 0>(st, L1, G, table_dev, hi_dev, lo_dev, num_cus, nullptr, sel, true);
		}
		if (e != hipSuccess)
			return e;
		whole = false;
	}
	if (odd || !uni) {
		WrTunerLaunch L2 = L;
		L2.ev_start = nullptr;                              /* the start event went to the first launch */
		const unsigned long long sel = (odd == all) ? 0ull : odd;
		if (!uni && NCO == WR_NCO_ROTATE && post && post->k1 && post->groups && wrk_tuner_post_supported(post->d2)) {
			/* no lane group left for the fast kernel (every group holds more than WR_TAPSETS channel
			 * filters): the previous block's post stage rides with this launch instead (its workgroups
			 * are as large as the per-lane-taps ones, 16 waves, of which the post stage uses eight) */
			if (post_taken)
				*post_taken = true;
			return launch_ddc_riding<NCO, false>(post->d2, st, L, G, table_dev, hi_dev, lo_dev, num_cus, post, sel, whole);
		}
		return launch_ddc<NCO, false, 0>(st, uni ? L2 : L, G, table_dev, hi_dev, lo_dev, num_cus, nullptr, sel, whole);
	}
	return hipSuccess;
}

hipError_t wrk_tuner_ddc(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G,
                         const float *table_dev, const float *hi_dev, const float *lo_dev,
                         int num_cus, const WrPostArgs *post, bool *post_taken)
{
	if (post_taken)
		*post_taken = false;
	if (!L.slots_used)
		return hipSuccess;
	if (L.nco_mode == WR_NCO_EXACT)
		return launch_ddc<WR_NCO_EXACT, false, 0>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, nullptr);
	if (L.nco_mode == WR_NCO_ROTATE)
		return launch_ddc_fast<WR_NCO_ROTATE>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, post, post_taken);
	return launch_ddc_fast<WR_NCO_SPLIT>(st, L, G, table_dev, hi_dev, lo_dev, num_cus, nullptr, nullptr);
}

hipError_t wrk_tuner_demod(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G)
{
	if (!L.k1 || !L.slots_used)
		return hipSuccess;
	const int p = L.parity;
	dim3 grid(L.slots_used / 64, (unsigned int)((L.k1 + 4 * DEM_RPT - 1) / (4 * DEM_RPT)));
	k_tuner_demod<<<grid, 256, 0, st>>>(
		(const float2 *)G.chan_iq[L.cb], (unsigned int)L.k1, L.slots, G.mode, (const float2 *)G.prev_iq[p],
		(float2 *)G.prev_iq[p ^ 1], G.dem[p], G.dem[p ^ 1], G.l2 - 1u);
	return hipGetLastError();
}

WrPostArgs wrk_post_args(const WrTunerLaunch &L, const WrGroupDev &G)
{
	const int p = L.parity;
	WrPostArgs A;
	A.chan_iq = G.chan_iq[L.cb];
	A.k1 = (unsigned int)L.k1;
	A.slots = L.slots;
	A.mode = G.mode;
	A.prev_iq = G.prev_iq[p];
	A.prev_next = G.prev_iq[p ^ 1];
	A.dem_hist = G.dem[p];
	A.dem_hist_next = G.dem[p ^ 1];
	A.k2 = L.k2;
	A.tiles = (unsigned int)((L.k2 + POST_TK - 1) / POST_TK);
	/* tiles per workgroup: as long a run as still leaves every CU a few workgroups */
	{
		const unsigned int all = A.tiles * (L.slots_used / 64u);
#ifdef POST_RUN
		A.run = POST_RUN;
#else
		/* measured at C2 (500 tiles a block), us per block: 1 block per launch, run 1 / 2 / 4:
		 * 38.3 / 37.1 / 38.0; 4 blocks per launch: 33.6 / 32.3 / 32.2 */
		A.run = all >= 1536u ? 4u : all >= 384u ? 2u : 1u;
#endif
	}
	A.ntiles = (A.tiles + A.run - 1u) / A.run;
	A.taps2 = G.taps2;
	A.taps2u = G.taps2u;
	A.uni2 = L.uniform2_mask;
	A.audio = G.audio;
	A.k2max = L.k2max;
	A.scale = L.audio_scale;
	A.d2 = L.d2;
	A.groups = L.slots_used / 64;
	A.gain = L.use_gain ? G.gain : nullptr;
	A.squelch = L.use_squelch ? G.squelch : nullptr;
	A.audio_host = nullptr;
	A.host_stride = 0;
	A.nseg = G.l2 / WR_FIR_LENGTH;                      /* 1; 2 / 4: an audio filter of 128 / 256 taps */
	return A;
}

template <unsigned int D2>
static hipError_t launch_post(hipStream_t st, const WrPostArgs &A)
{
	dim3 grid(A.ntiles + 1u, A.groups);
	/* (an audio filter of 128 / 256 taps: 2 / 4 segments of 64, post_role) */
	if (A.nseg == 2u || A.nseg == 4u) {
		const size_t lds = (((size_t)(POST_TK - 1u) * D2 + WR_FIR_LENGTH * A.nseg) * 64u + POST_TK * 65u + 64u) * sizeof(float);
		static bool attr2[WR_MAX_DEVICES], attr4[WR_MAX_DEVICES];
		const size_t lds_max = (((size_t)(POST_TK - 1u) * D2 + WR_FIR_LENGTH * 4u) * 64u + POST_TK * 65u + 64u) * sizeof(float);
		hipError_t e = A.nseg == 2u ? allow_lds((const void *)k_tuner_post<D2, 2u>, lds_max, attr2)
		                            : allow_lds((const void *)k_tuner_post<D2, 4u>, lds_max, attr4);
		if (e != hipSuccess)
			return e;
		if (A.nseg == 2u)
			k_tuner_post<D2, 2u><<<grid, POST_THREADS, lds, st>>>(A);
		else
			k_tuner_post<D2, 4u><<<grid, POST_THREADS, lds, st>>>(A);
	} else if (A.nseg == 1u)
		k_tuner_post<D2><<<grid, POST_THREADS, 0, st>>>(A);
	else
		return hipErrorInvalidValue;
	return hipGetLastError();
}

/* audio decimations k_tuner_post is instantiated for (anything else takes the two-kernel path): 1..6, and 8 and 10
 * -- 256 k -> 32 k (BASELINE config 1), 240 k -> 24 k, 480 k -> 48 k */
bool wrk_tuner_post_supported(unsigned int d2)
{
	return (d2 >= 1 && d2 <= 6) || d2 == 8 || d2 == 10;
}

hipError_t wrk_tuner_post_args(hipStream_t st, const WrPostArgs &A0)
{
	if (!A0.k1 || !A0.groups)
		return hipSuccess;
	/* on its own the post stage has the chip to itself and is a chain of latencies: one tile per
	 * workgroup, all of them in flight at once (runs of tiles are for the workgroups that ride in a
	 * DDC launch, where instructions are what is short); 15.4 against 18.2 us for a C2 block */
	WrPostArgs A = A0;
	{
		/* ... unless there are tiles enough to fill the chip several times over anyway (the flush behind a launch of
		 * several blocks): runs of tiles then save the rows two neighbouring tiles share from being loaded and
		 * demodulated twice (59 of 139 at D2 = 5) -- WR_POST_FLUSH_RUN overrides */
		static const int forced = getenv("WR_POST_FLUSH_RUN") ? atoi(getenv("WR_POST_FLUSH_RUN")) : 0;
		const unsigned int all = A.tiles * A.groups;
		A.run = forced > 0 ? (unsigned int)forced : all >= 4096u ? 2u : 1u;   /* C2, four blocks: 43.6 / 33.6 / 36.0 us at runs of 1 / 2 / 4 */
		if (forced <= 0 && A.nseg == 4u && all >= 384u)
			A.run = 2u;     /* 256 taps: a tile stages 331 rows for its 80 new ones (D2 = 5) and one workgroup fits a CU --
			                   C2 (500 tiles), us per block all in at runs of 1 / 2 / 4: 93 / 83 / 104 (128 taps: 62 / 69 / 79),
			                   profiles/r05_long_filter.txt */
	}
	A.ntiles = (A.tiles + A.run - 1u) / A.run;
	switch (A.d2) {
	case 1: return launch_post<1>(st, A);
	case 2: return launch_post<2>(st, A);
	case 3: return launch_post<3>(st, A);
	case 4: return launch_post<4>(st, A);
	case 5: return launch_post<5>(st, A);
	case 6: return launch_post<6>(st, A);
	case 8: return launch_post<8>(st, A);
	case 10: return launch_post<10>(st, A);
	default: return hipErrorInvalidValue;
	}
}

hipError_t wrk_tuner_post(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G)
{
	if (!L.k1 || !L.slots_used)
		return hipSuccess;
	return wrk_tuner_post_args(st, wrk_post_args(L, G));
}

hipError_t wrk_tuner_audio(hipStream_t st, const WrTunerLaunch &L, const WrGroupDev &G)
{
	if (!L.k2 || !L.slots_used)
		return hipSuccess;
	/* as many output frames per tile as the staged rows allow */
	const unsigned int len = G.l2;                      /* 64, or 128 / 256 taps (LowPass::_firLength) */
	unsigned int tk = AUD_TMAX;
	if (L.d2 > 1 && (AUD_ROWS - len) / L.d2 + 1u < tk)
		tk = (AUD_ROWS - len) / L.d2 + 1u;
	/* LDS sized to what this decimation needs, so that two workgroups fit a CU when D2 is small */
	const unsigned int need = (tk - 1u) * L.d2 + len;
	const size_t lds = ((size_t)need * 64u + (size_t)len * 64u + AUD_TMAX * 65u) * sizeof(float);
	static bool attr_done[WR_MAX_DEVICES];
	{
		const size_t lds_max = ((size_t)AUD_ROWS * 64u + (size_t)WR_FIR_FUSED_MAX * 64u + AUD_TMAX * 65u) * sizeof(float);   /* 156 KiB of 160 */
		hipError_t e = allow_lds((const void *)k_tuner_audio, lds_max, attr_done);
		if (e != hipSuccess)
			return e;
	}
	dim3 grid((unsigned int)((L.k2 + tk - 1) / tk), L.slots_used / 64);
	k_tuner_audio<<<grid, AUD_THREADS, lds, st>>>(G.dem[L.parity], (size_t)(len - 1u) + L.k1, L.k2, L.d2, tk, L.slots,
	                                      G.taps2, G.mode, G.audio, L.k2max, L.audio_scale,
	                                      L.use_gain ? G.gain : nullptr, L.use_squelch ? G.squelch : nullptr,
	                                      (const float2 *)G.chan_iq[L.cb], len);
	return hipGetLastError();
}

/* wr_tuner_seek: the state every channel has when the stream starts at `frame` -- empty filter
 * histories (a zero LO row = an empty LowPass::block), Demodulator::prev_i/q = 0 -- and the NCO phase
 * in closed form, frame * step mod 2^32 left-aligned (downconverter.cxx:103).  One small launch. */
__global__ void __launch_bounds__(256)
k_seek(unsigned int *__restrict__ phase, const unsigned int *__restrict__ step, unsigned long long frame,
       float2 *__restrict__ hist_cs, float2 *__restrict__ hist_lo, float2 *__restrict__ prev_iq,
       float *__restrict__ dem_hist, float2 *__restrict__ iq2_hist, unsigned int slots)
{
	const unsigned int gsz = gridDim.x * blockDim.x;
	const unsigned int rows = WR_HIST * slots;
	for (unsigned int e = blockIdx.x * blockDim.x + threadIdx.x; e < rows; e += gsz) {
		hist_cs[e] = make_float2(0.0f, 0.0f);
		hist_lo[e] = make_float2(0.0f, 0.0f);
		dem_hist[e] = 0.0f;
		iq2_hist[e] = make_float2(0.0f, 0.0f);
		if (e < slots) {
			phase[e] = (unsigned int)((unsigned long long)step[e] * frame);
			prev_iq[e] = make_float2(0.0f, 0.0f);
		}
	}
}

hipError_t wrk_seek(hipStream_t st, const WrGroupDev &G, unsigned int slots, int sp, int parity, int p2,
                    unsigned long long frame)
{
	const unsigned int rows = WR_HIST * slots;
	k_seek<<<(rows + 255u) / 256u, 256, 0, st>>>(G.phase[sp], G.step, frame, (float2 *)G.hist_cs[sp],
	                                             (float2 *)G.hist_lo[sp], (float2 *)G.prev_iq[parity], G.dem[parity],
	                                             (float2 *)G.iq2_hist[p2], slots);
	/* (an audio filter or a second channel stage of 128 / 256 taps keeps 127 / 255 rows: the rest of them) */
	hipError_t e = hipGetLastError();
	if (e == hipSuccess && G.l2 > WR_FIR_LENGTH)
		e = hipMemsetAsync(G.dem[parity], 0, (size_t)(G.l2 - 1u) * slots * sizeof(float), st);
	if (e == hipSuccess && G.l1b > WR_FIR_LENGTH && G.iq2_hist[p2])
		e = hipMemsetAsync(G.iq2_hist[p2], 0, (size_t)(G.l1b - 1u) * slots * 2u * sizeof(float), st);
	return e;
}

__global__ void k_input_hist(const float2 *__restrict__ cur, const uchar2 *__restrict__ cur_u8, size_t nframes,
                             const float2 *__restrict__ hist, float2 *__restrict__ hist_next)
{
	const unsigned int lane = threadIdx.x;
	if (lane < WR_HIST) {
		const size_t f = nframes + lane;
		hist_next[lane] = (f < WR_HIST) ? hist[f] : input_frame(cur, cur_u8, f - WR_HIST);
	}
}

hipError_t wrk_input_hist(hipStream_t st, const float *cur, const uint8_t *cur_u8, size_t nframes,
                          const float *hist, float *hist_next)
{
	k_input_hist<<<1, 64, 0, st>>>((const float2 *)cur, (const uchar2 *)cur_u8, nframes, (const float2 *)hist,
	                               (float2 *)hist_next);
	return hipGetLastError();
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

#include "wr_stream_kernel.inc"
