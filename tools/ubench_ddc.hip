// Ablation harness for the DDC inner loop (same structure as k_tuner_ddc<SPLIT>):
// template flags switch off one ingredient at a time.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2f lds_v2f;
#define L 64

template <bool GATHER, bool READLANE, bool SDWA, bool TAPS, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k(const float2 *__restrict__ cur, size_t k1, unsigned d1, const unsigned *__restrict__ phase,
  const unsigned *__restrict__ step, const float *__restrict__ taps1, float2 *__restrict__ out, unsigned slots)
{
	extern __shared__ v2f tab[];
	const unsigned lane = threadIdx.x & 63u;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	for (unsigned e = threadIdx.x; e < 2 * 256 * 32; e += blockDim.x)
		tab[e] = (v2f){1.0f, 1e-3f * (e & 255)};
	__syncthreads();
	unsigned a_hi = (lane & 31u) * 8u, a_lo = 65536u + (lane & 31u) * 8u;
	const unsigned groups = slots >> 6;
	const size_t units = k1 * groups;
	const size_t wg = (size_t)blockIdx.x * WAVES + wave, wc = (size_t)gridDim.x * WAVES;
	float h[L];
	unsigned loaded = ~0u, p0 = 0, st = 0;
	for (size_t u = wg; u < units; u += wc) {
		const unsigned g = (unsigned)(u / k1);
		const size_t kk = u - (size_t)g * k1;
		const unsigned s = g * 64u + lane;
		if (g != loaded) {
#pragma unroll
			for (int j = 0; j < L; ++j) h[j] = TAPS ? taps1[(size_t)j * slots + s] : 1.0f;
			p0 = phase[s]; st = step[s]; loaded = g;
		}
		const size_t n0 = kk * d1;
		unsigned P = p0 + (unsigned)n0 * st;
		const float2 xw = cur[n0 + lane];
		const int xwi = __builtin_bit_cast(int, xw.x), xwq = __builtin_bit_cast(int, xw.y);
		v2f acc = {0, 0};
		constexpr int NT = 4;
		v2f ta[2][NT], tb[2][NT];
		unsigned ah[NT], al[NT];
#pragma unroll
		for (int jj = 0; jj < NT; ++jj) { ah[jj] = a_hi; al[jj] = a_lo; }
		auto gather = [&](unsigned Pp, unsigned &rh, unsigned &rl, v2f &a, v2f &b) {
			if (SDWA) {
				asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(rh) : "v"(Pp));
				asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(rl) : "v"(Pp));
			} else {
				rh = ((Pp >> 24) << 8) + a_hi;
				rl = (((Pp >> 16) & 255u) << 8) + a_lo;
			}
			if (GATHER) { a = *(const lds_v2f *)rh; b = *(const lds_v2f *)rl; }
			else { a = (v2f){__builtin_bit_cast(float, rh), 1.0f}; b = (v2f){1.0f, __builtin_bit_cast(float, rl)}; }
		};
#pragma unroll
		for (int jj = 0; jj < NT; ++jj) { gather(P, ah[jj], al[jj], ta[0][jj], tb[0][jj]); P += st; }
#pragma unroll
		for (int t = 0; t < L / NT; ++t) {
			if (t + 1 < L / NT) {
#pragma unroll
				for (int jj = 0; jj < NT; ++jj) { gather(P, ah[jj], al[jj], ta[(t + 1) & 1][jj], tb[(t + 1) & 1][jj]); P += st; }
			}
#pragma unroll
			for (int jj = 0; jj < NT; ++jj) {
				const int j = t * NT + jj;
				float xi, xq;
				if (READLANE) {
					xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xwi, j));
					xq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xwq, j));
				} else { xi = xw.x; xq = xw.y; }
				const v2f a = ta[t & 1][jj], b = tb[t & 1][jj];
				const float c = __builtin_fmaf(-a.y, b.y, a.x * b.x);
				const float sn = __builtin_fmaf(a.x, b.y, a.y * b.x);
				const float mi = __builtin_fmaf(xq, sn, xi * c);
				const float mq = __builtin_fmaf(-xi, sn, xq * c);
				const float hj = h[L - 1 - j];
				acc.x = __builtin_fmaf(hj, mi, acc.x);
				acc.y = __builtin_fmaf(hj, mq, acc.y);
			}
		}
		out[kk * slots + s] = make_float2(acc.x, acc.y);
	}
}

template <typename F> static double timeit(F f, int reps = 5)
{
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	f(); (void)hipDeviceSynchronize();
	(void)hipEventRecord(a); for (int r = 0; r < reps; ++r) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3;
}

int main()
{
	const size_t n = 4000000, k1 = 10000; const unsigned d1 = 400, slots = 256;
	float2 *cur, *out; unsigned *phase, *step; float *taps;
	(void)hipMalloc(&cur, (n + 64) * 8); (void)hipMemset(cur, 0, (n + 64) * 8);
	(void)hipMalloc(&out, k1 * slots * 8); (void)hipMalloc(&phase, slots * 4); (void)hipMalloc(&step, slots * 4);
	(void)hipMalloc(&taps, 64 * slots * 4); (void)hipMemset(taps, 0, 64 * slots * 4);
	unsigned hs[256]; for (int i = 0; i < 256; ++i) hs[i] = 0x9E3779B9u * (i + 1);
	(void)hipMemcpy(step, hs, sizeof(hs), hipMemcpyHostToDevice); (void)hipMemcpy(phase, hs, sizeof(hs), hipMemcpyHostToDevice);
	const size_t lds = 131072;
#define RUN(G, R, S, T, W, label) { auto kk = k<G, R, S, T, W>; (void)hipFuncSetAttribute((const void *)kk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
	double us = timeit([&] { kk<<<256, W * 64, lds>>>(cur, k1, d1, phase, step, taps, out, slots); }); \
	hipError_t e = hipGetLastError(); printf("%-44s %8.1f us %s\n", label, us, e == hipSuccess ? "" : hipGetErrorString(e)); }
	RUN(true, true, true, true, 16, "full (gather+readlane+sdwa+taps), 16 waves");
	RUN(false, true, true, true, 16, "no LDS gather");
	RUN(true, false, true, true, 16, "no readlane");
	RUN(true, true, false, true, 16, "no sdwa (shift/add addresses)");
	RUN(true, true, true, false, 16, "no per-lane taps (h=1)");
	RUN(false, false, true, true, 16, "no gather, no readlane");
	RUN(false, false, false, false, 16, "ALU only");
	RUN(true, true, true, true, 8, "full, 8 waves/WG");
	RUN(true, true, true, true, 4, "full, 4 waves/WG");
	return 0;
}
