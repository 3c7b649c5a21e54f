// tools/ubench_pk.hip -- does v_pk_fma_f32 issue at the rate of v_fma_f32 on gfx950 (two FMAs for the slot of one)?
// 8 independent accumulator chains per lane, N iterations; waves per SIMD from the grid.  Prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b)
{
	v2f acc[8];
	for (int i = 0; i < 8; ++i)
		acc[i] = (v2f){(float)threadIdx.x + i, 1.0f};
	const v2f m = {a, a}, c = {b, b};
	long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				if (MODE == 0) {          // one v_pk_fma_f32
					asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(m), "v"(c));
				} else if (MODE == 1) {   // two v_fma_f32
					asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(a), "v"(b));
				} else if (MODE == 2) {   // one v_fma_f32
					asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].x) : "v"(a), "v"(b));
				} else if (MODE == 3) {   // v_pk_mul_f32
					asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(m));
				} else if (MODE == 4) {   // v_pk_add_f32
					asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(c));
				}
			}
		}
	}
	long long t1 = __builtin_readcyclecounter();
	float s = 0;
	for (int i = 0; i < 8; ++i)
		s += acc[i].x + acc[i].y;
	if (s == 12345.0f || (threadIdx.x == 0 && blockIdx.x == 0))
		out[blockIdx.x] = s + (float)(t1 - t0);
}

template <int MODE> void run(const char *name, int per_instr, float *d)
{
	const int iters = 2000;
	for (int wgs_per_cu = 1; wgs_per_cu <= 8; wgs_per_cu *= 2) {
		hipEvent_t e0, e1;
		hipEventCreate(&e0), hipEventCreate(&e1);
		const int grid = 256 * wgs_per_cu;
		k<MODE><<<grid, 256>>>(d, 10, 1.0001f, 0.5f);
		hipEventRecord(e0);
		k<MODE><<<grid, 256>>>(d, iters, 1.0001f, 0.5f);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		const double winstr = (double)grid * 4 * iters * 64 * per_instr;       // wave-instructions
		printf("%-16s %d waves/SIMD: %.3f T wave-instr/s  (%.3f T lane-FMA-slots x64)\n", name, wgs_per_cu, winstr / (ms * 1e-3) / 1e12,
		       winstr / (ms * 1e-3) / 1e12);
	}
}

int main()
{
	float *d;
	hipMalloc(&d, 1 << 20);
	run<2>("v_fma_f32", 1, d);
	run<1>("2 x v_fma_f32", 2, d);
	run<0>("v_pk_fma_f32", 1, d);
	run<3>("v_pk_mul_f32", 1, d);
	run<4>("v_pk_add_f32", 1, d);
	return 0;
}
