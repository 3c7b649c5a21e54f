// Micro-benchmark for the GEMM formulation of the DDC (k_tuner_ddc, WR_NCO_GEMM): do the f32 matrix
// pipe and the VALU run side by side on gfx950, at the instruction mix that formulation needs?
//   main term   : v_mfma_f32_16x16x4_f32 (32 cycles per SIMD each) or v_mfma_f32_32x32x2_f32 (64)
//   correction  : per channel-tap  v_add_co_u32 (threshold + phase fraction -> carry), v_cndmask (mask the
//                 packed-half sample), 2 x v_dot2_f32_f16 (complex multiply-accumulate, f32 accumulators)
// One 16x16x4 MFMA covers 16 frames x 8 channels x 2 taps = 256 channel-taps = 4 wave-taps, i.e. 16
// correction instructions per MFMA: both pipes need the same 32 cycles per SIMD.
//   Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define ITERS 1024
#define NK 4               // frames in flight per wave in the correction (independent chains)

// MODE 0: correction only   1: MFMA 16x16x4 only   2: both, 16 VALU per MFMA   3: MFMA 32x32x2 only
// MODE 4: both with 32x32x2, 32 VALU per MFMA (same flops per VALU)      5: the ROTATE tap (7 VALU), for reference
template <int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_mix(float *out, unsigned seed, float fa, float fb)
{
	const unsigned t = threadIdx.x;
	unsigned F[NK];
	float are[NK], aim[NK];
	for (int k = 0; k < NK; ++k) {
		F[k] = (t * 2654435761u + k * 40503u) ^ seed;
		are[k] = 0.f;
		aim[k] = 0.f;
	}
	unsigned thr[4] = {t * 97u + seed, t * 193u + 1u, t * 389u + 7u, t * 769u + 3u};
	h2 E[4];
	for (int j = 0; j < 4; ++j)
		E[j] = (h2){(_Float16)(0.01f * (j + 1) + fa), (_Float16)(0.02f * (j + 1) + fb)};
	const h2 u = {(_Float16)fa, (_Float16)fb};
	const h2 zero = {(_Float16)0.f, (_Float16)0.f};
	v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
	v16f d0 = {0}, d1 = {0};
	float a = fa + t, b = fb - t;
	// ROTATE tap state
	float ar = t, ai = 1.f, rc = fa, rs = fb;
	unsigned G = t * 2654435761u;

	for (int i = 0; i < ITERS; ++i) {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			// one step = 4 taps x NK frames of correction (4 * NK * 4 = 64 VALU for NK = 4)
			if (MODE == 1 || MODE == 2) {
				// 4 MFMAs of 32 cycles for 64 VALU -> 16 per MFMA
				c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
			}
			if (MODE == 3 || MODE == 4)
				d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
			if (MODE == 0 || MODE == 2 || MODE == 4) {
#pragma unroll
				for (int m = 0; m < 2; ++m)
#pragma unroll
					for (int k = 0; k < NK; ++k) {
						unsigned tmp;
						const bool cy = __builtin_uadd_overflow(F[k], thr[(2 * j + m) & 3], &tmp);
						F[k] += 0x9E3779B9u * 0;          // keep F live without another instruction
						const h2 um = cy ? u : zero;
						are[k] = __builtin_amdgcn_fdot2(um, E[(m + j) & 3], are[k], false);
						aim[k] = __builtin_amdgcn_fdot2(um, E[(m + j + 1) & 3], aim[k], false);
					}
			}
			if (MODE == 1 || MODE == 2)
				c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
			if (MODE == 3 || MODE == 4)
				d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, d1, 0, 0, 0);
			if (MODE == 0 || MODE == 2 || MODE == 4) {
#pragma unroll
				for (int m = 2; m < 4; ++m)
#pragma unroll
					for (int k = 0; k < NK; ++k) {
						unsigned tmp;
						const bool cy = __builtin_uadd_overflow(F[k], thr[(2 * j + m) & 3], &tmp);
						const h2 um = cy ? u : zero;
						are[k] = __builtin_amdgcn_fdot2(um, E[(m + j) & 3], are[k], false);
						aim[k] = __builtin_amdgcn_fdot2(um, E[(m + j + 1) & 3], aim[k], false);
					}
			}
			if (MODE == 5) {
#pragma unroll
				for (int m = 0; m < 4; ++m) {
					asm volatile("v_add_co_u32 %0, vcc, %0, %5\n"
					             "v_cndmask_b32 %3, %6, %8, vcc\n"
					             "v_cndmask_b32 %4, %7, %9, vcc\n"
					             : "+v"(G), "+v"(ar), "+v"(ai), "=v"(rc), "=v"(rs)
					             : "v"(seed), "v"(fa), "v"(fb), "v"(a), "v"(b) : "vcc");
					float tr, ti;
					asm volatile("v_fma_f32 %0, -%3, %5, %6\n"
					             "v_fma_f32 %1, %2, %5, %6\n"
					             "v_fma_f32 %2, %2, %4, %0\n"
					             "v_fma_f32 %3, %3, %4, %1\n"
					             : "=&v"(tr), "=&v"(ti), "+v"(ar), "+v"(ai) : "v"(rc), "v"(rs), "v"(fa));
				}
			}
		}
		for (int k = 0; k < NK; ++k)
			F[k] += 0x9E3779B9u;
	}
	float s = c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w + d0[0] + d0[5] + d1[3] + ar + ai + (float)G;
	for (int k = 0; k < NK; ++k)
		s += are[k] + aim[k] + (float)F[k];
	out[blockIdx.x * blockDim.x + t] = s;
}

template <typename F>
static double timeit(F f, int reps = 5)
{
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	f();
	hipDeviceSynchronize();
	hipEventRecord(a);
	for (int r = 0; r < reps; ++r)
		f();
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	return ms / reps * 1e-3;
}

template <int WAVES>
static void run(int cus, int wgs_per_cu, float *out)
{
	const int wg = cus * wgs_per_cu;
	const double waves = (double)wg * WAVES;
	const double steps = (double)ITERS * 4;                 // per wave
	const double t0 = timeit([&] { k_mix<0, WAVES><<<wg, WAVES * 64>>>(out, 12345u, 0.5f, 0.25f); });
	const double t1 = timeit([&] { k_mix<1, WAVES><<<wg, WAVES * 64>>>(out, 12345u, 0.5f, 0.25f); });
	const double t2 = timeit([&] { k_mix<2, WAVES><<<wg, WAVES * 64>>>(out, 12345u, 0.5f, 0.25f); });
	const double t3 = timeit([&] { k_mix<3, WAVES><<<wg, WAVES * 64>>>(out, 12345u, 0.5f, 0.25f); });
	const double t4 = timeit([&] { k_mix<4, WAVES><<<wg, WAVES * 64>>>(out, 12345u, 0.5f, 0.25f); });
	const double t5 = timeit([&] { k_mix<5, WAVES><<<wg, WAVES * 64>>>(out, 12345u, 0.5f, 0.25f); });
	const double valu = waves * steps * 4 * NK * 4;          // correction wave-instructions
	const double wtaps = waves * steps * 4 * NK;            // wave-taps covered by the correction
	printf("%2d waves/WG x %d WG/CU = %2d waves/CU\n", WAVES, wgs_per_cu, WAVES * wgs_per_cu);
	printf("  correction only (4 VALU/tap)     : %.3f ms  %.3f T wave-inst/s  %.1f G wave-taps/s\n", t0 * 1e3, valu / t0 / 1e12, wtaps / t0 / 1e9);
	printf("  MFMA 16x16x4 f32 only            : %.3f ms  %.1f TFLOP/s\n", t1 * 1e3, waves * steps * 2 * 2048 / t1 / 1e12);
	printf("  both (16 VALU per 16x16x4)       : %.3f ms  %.3f T wave-inst/s  %.1f TFLOP/s  %.1f G wave-taps/s (sum of parts %.3f ms)\n",
	       t2 * 1e3, valu / t2 / 1e12, waves * steps * 2 * 2048 / t2 / 1e12, wtaps / t2 / 1e9, (t0 + t1) * 1e3);
	printf("  MFMA 32x32x2 f32 only            : %.3f ms  %.1f TFLOP/s\n", t3 * 1e3, waves * steps * 2 * 4096 / t3 / 1e12);
	printf("  both (32 VALU per 32x32x2)       : %.3f ms  %.3f T wave-inst/s  %.1f TFLOP/s\n", t4 * 1e3, valu / t4 / 1e12, waves * steps * 2 * 4096 / t4 / 1e12);
	printf("  ROTATE tap (7 VALU/tap), 4 taps  : %.3f ms  %.3f T wave-inst/s  %.1f G wave-taps/s\n", t5 * 1e3, waves * steps * 28 / t5 / 1e12, waves * steps * 4 / t5 / 1e9);
}

int main()
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	float *out;
	hipMalloc(&out, sizeof(float) * cus * 4 * 1024);
	run<8>(cus, 1, out);
	run<8>(cus, 2, out);
	run<16>(cus, 1, out);
	run<8>(cus, 4, out);
	printf("C2 needs 2.56 M wave-taps per launch\n");
	hipFree(out);
	return 0;
}
