// Ablation harness for the DDC inner loop (same structure as k_tuner_ddc<SPLIT, UTAPS>):
// template flags switch off one ingredient at a time.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v2f lds_v2f;
typedef __attribute__((address_space(3))) v4f lds_v4f;
#define L 64

template <bool GATHER, bool WINDOW, bool SDWA, int NT, int WAVES, bool MATH>
__global__ void __launch_bounds__(WAVES * 64)
k(const float2 *__restrict__ cur, size_t k1, unsigned d1, const unsigned *__restrict__ phase,
  const unsigned *__restrict__ step, float2 *__restrict__ out, unsigned slots)
{
	extern __shared__ v2f lds[];
	const unsigned lane = threadIdx.x & 63u;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	for (unsigned e = threadIdx.x; e < 2 * 256 * 32; e += blockDim.x)
		lds[e] = (v2f){1.0f, 1e-3f * (e & 255)};
	__syncthreads();
	const unsigned a_hi = (lane & 31u) * 8u, a_lo = 65536u + (lane & 31u) * 8u;
	v2f *win = lds + 16384 + wave * 128u;
	const unsigned groups = slots >> 6;
	const size_t units = k1 * groups;
	const size_t wg = (size_t)blockIdx.x * WAVES + wave, wc = (size_t)gridDim.x * WAVES;
	unsigned loaded = ~0u, p0 = 0, st = 0, buf = 0;
	for (size_t u = wg; u < units; u += wc, buf ^= 1u) {
		const unsigned g = (unsigned)(u / k1);
		const size_t kk = u - (size_t)g * k1;
		const unsigned s = g * 64u + lane;
		if (g != loaded) { p0 = phase[s]; st = step[s]; loaded = g; }
		const size_t n0 = kk * d1;
		unsigned P = p0 + (unsigned)n0 * st;
		const float2 xf = cur[n0 + lane];
		win[buf * 64u + lane] = (v2f){xf.x, xf.y};
		const lds_v4f *w4 = (const lds_v4f *)(win + buf * 64u);
		v2f acc = {0, 0};
		v2f ta[2][NT], tb[2][NT];
		v4f xw[2][NT / 2];
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
		auto wload = [&](int idx) -> v4f {
			if (WINDOW) return w4[idx];
			return (v4f){xf.x, xf.y, xf.y, xf.x};
		};
#pragma unroll
		for (int jj = 0; jj < NT / 2; ++jj) xw[0][jj] = wload(jj);
#pragma unroll
		for (int jj = 0; jj < NT; ++jj) { gather(P, ah[jj], al[jj], ta[0][jj], tb[0][jj]); P += st; }
#pragma unroll
		for (int t = 0; t < L / NT; ++t) {
			if (t + 1 < L / NT) {
#pragma unroll
				for (int jj = 0; jj < NT / 2; ++jj) xw[(t + 1) & 1][jj] = wload((t + 1) * (NT / 2) + jj);
#pragma unroll
				for (int jj = 0; jj < NT; ++jj) { gather(P, ah[jj], al[jj], ta[(t + 1) & 1][jj], tb[(t + 1) & 1][jj]); P += st; }
			}
#pragma unroll
			for (int jj = 0; jj < NT; ++jj) {
				const v4f x2 = xw[t & 1][jj >> 1];
				const v2f xs = (jj & 1) ? (v2f){x2.z, x2.w} : (v2f){x2.x, x2.y};
				const v2f a = ta[t & 1][jj], b = tb[t & 1][jj];
				if (MATH) {
					const float c = __builtin_fmaf(-a.y, b.y, a.x * b.x);
					const float sn = __builtin_fmaf(a.x, b.y, a.y * b.x);
					acc.x = __builtin_fmaf(xs.y, sn, __builtin_fmaf(xs.x, c, acc.x));
					acc.y = __builtin_fmaf(-xs.x, sn, __builtin_fmaf(xs.y, c, acc.y));
				} else {
					acc.x += a.x + b.y + xs.x;
					acc.y += a.y + b.x + xs.y;
				}
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
	float2 *cur, *out; unsigned *phase, *step;
	(void)hipMalloc(&cur, (n + 64) * 8); (void)hipMemset(cur, 0, (n + 64) * 8);
	(void)hipMalloc(&out, k1 * slots * 8); (void)hipMalloc(&phase, slots * 4); (void)hipMalloc(&step, slots * 4);
	unsigned hs[256]; for (int i = 0; i < 256; ++i) hs[i] = 0x9E3779B9u * (i + 1);
	(void)hipMemcpy(step, hs, sizeof(hs), hipMemcpyHostToDevice); (void)hipMemcpy(phase, hs, sizeof(hs), hipMemcpyHostToDevice);
	const size_t lds = 131072 + 16 * 1024;
#define RUN(G, W, S, NT, WV, M, label) { auto kk = k<G, W, S, NT, WV, M>; (void)hipFuncSetAttribute((const void *)kk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
	double us = timeit([&] { kk<<<256, WV * 64, lds>>>(cur, k1, d1, phase, step, out, slots); }); \
	hipError_t e = hipGetLastError(); printf("%-52s %8.1f us %s\n", label, us, e == hipSuccess ? "" : hipGetErrorString(e)); }
	RUN(true, true, true, 4, 16, true, "full: gather+window+sdwa, NT=4, 16 waves");
	RUN(true, true, true, 8, 16, true, "full, NT=8");
	RUN(true, true, true, 2, 16, true, "full, NT=2");
	RUN(false, true, true, 4, 16, true, "no table gathers (address math kept)");
	RUN(true, false, true, 4, 16, true, "no LDS window reads");
	RUN(false, false, true, 4, 16, true, "no LDS at all");
	RUN(true, true, false, 4, 16, true, "shift/add addresses instead of SDWA");
	RUN(true, true, true, 4, 16, false, "LDS traffic only (adds instead of the 8 FMAs)");
	RUN(false, false, false, 4, 16, true, "ALU only (shift/add addr, no LDS)");
	RUN(true, true, true, 4, 8, true, "full, 8 waves/WG");
	return 0;
}
