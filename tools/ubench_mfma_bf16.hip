// Does the bf16 matrix pipe run BESIDE the VALU on gfx950 (the f32 MFMA does not: r02_ubench_mfma.txt)?
//   v_mfma_f32_32x32x16_bf16 (32 cycles per SIMD) alone, the 4-VALU correction mix alone, both interleaved.
//   Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_bf16.hip -o tools/ubench_mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define ITERS 1024
#define NK 4
// MODE 0: VALU only   1: MFMA bf16 only   2: both, VPM VALU per MFMA
template <int MODE, int WAVES, int VPM>
__global__ void __launch_bounds__(WAVES * 64) k(float *out, unsigned seed, float fa, float fb)
{
	const unsigned t = threadIdx.x;
	unsigned F[NK]; float are[NK], aim[NK];
	for (int i = 0; i < NK; ++i) { F[i] = (t * 2654435761u + i * 40503u) ^ seed; are[i] = aim[i] = 0.f; }
	unsigned thr[4] = {t * 97u + seed, t * 193u + 1u, t * 389u + 7u, t * 769u + 3u};
	h2 E[4];
	for (int j = 0; j < 4; ++j) E[j] = (h2){(_Float16)(0.01f * (j + 1) + fa), (_Float16)(0.02f * (j + 1) + fb)};
	const h2 u = {(_Float16)fa, (_Float16)fb}, zero = {(_Float16)0.f, (_Float16)0.f};
	bf8 a, b;
	for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(fa + i + t); b[i] = (__bf16)(fb - i); }
	v16f d0 = {0}, d1 = {0};
	for (int it = 0; it < ITERS; ++it) {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (MODE != 0) d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
			if (MODE != 1) {
#pragma unroll
				for (int v = 0; v < VPM / 4; ++v) {
					const int k = v % NK;
					unsigned tmp;
					const bool cy = __builtin_uadd_overflow(F[k], thr[(j + v) & 3], &tmp);
					const h2 um = cy ? u : zero;
					are[k] = __builtin_amdgcn_fdot2(um, E[(v + j) & 3], are[k], false);
					aim[k] = __builtin_amdgcn_fdot2(um, E[(v + j + 1) & 3], aim[k], false);
				}
			}
			if (MODE != 0) d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, d1, 0, 0, 0);
			if (MODE != 1) {
#pragma unroll
				for (int v = 0; v < VPM / 4; ++v) {
					const int k = (v + 2) % NK;
					unsigned tmp;
					const bool cy = __builtin_uadd_overflow(F[k], thr[(j + v + 1) & 3], &tmp);
					const h2 um = cy ? u : zero;
					are[k] = __builtin_amdgcn_fdot2(um, E[(v + j + 2) & 3], are[k], false);
					aim[k] = __builtin_amdgcn_fdot2(um, E[(v + j + 3) & 3], aim[k], false);
				}
			}
		}
		for (int i = 0; i < NK; ++i) F[i] += 0x9E3779B9u;
	}
	float s = d0[0] + d0[7] + d1[3] + d1[15];
	for (int i = 0; i < NK; ++i) s += are[i] + aim[i] + (float)F[i];
	out[blockIdx.x * blockDim.x + t] = s;
}
template <typename Fn> static double timeit(Fn f, int reps = 5)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	f(); hipDeviceSynchronize(); hipEventRecord(a);
	for (int r = 0; r < reps; ++r) f();
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	return ms / reps * 1e-3;
}
template <int WAVES, int VPM> static void run(int cus, int wgpc, float *out)
{
	const int wg = cus * wgpc;
	const double waves = (double)wg * WAVES, steps = (double)ITERS * 4;
	const double t0 = timeit([&] { k<0, WAVES, VPM><<<wg, WAVES * 64>>>(out, 1u, .5f, .25f); });
	const double t1 = timeit([&] { k<1, WAVES, VPM><<<wg, WAVES * 64>>>(out, 1u, .5f, .25f); });
	const double t2 = timeit([&] { k<2, WAVES, VPM><<<wg, WAVES * 64>>>(out, 1u, .5f, .25f); });
	const double valu = waves * steps * 2 * VPM, fl = waves * steps * 2 * 32768.0;
	printf("%2d waves/CU, %2d VALU per MFMA: VALU only %.3f ms (%.3f T inst/s) | bf16 MFMA only %.3f ms (%.0f TFLOP/s) | both %.3f ms "
	       "(%.3f T inst/s, %.0f TFLOP/s; sum of parts %.3f ms)\n", WAVES * wgpc, VPM, t0 * 1e3, valu / t0 / 1e12, t1 * 1e3, fl / t1 / 1e12,
	       t2 * 1e3, valu / t2 / 1e12, fl / t2 / 1e12, (t0 + t1) * 1e3);
}
int main()
{
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	float *out; hipMalloc(&out, sizeof(float) * p.multiProcessorCount * 4 * 1024);
	run<8, 16>(p.multiProcessorCount, 1, out);
	run<8, 16>(p.multiProcessorCount, 2, out);
	run<8, 16>(p.multiProcessorCount, 4, out);
	run<8, 8>(p.multiProcessorCount, 2, out);
	run<8, 32>(p.multiProcessorCount, 2, out);
	run<4, 16>(p.multiProcessorCount, 1, out);
	return 0;
}
