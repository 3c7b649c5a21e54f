// Micro-benchmark of the ROTATE tap (see k_tuner_ddc): which of its 7 VALU instructions set the
// pace on gfx950?  Each kernel runs 16 waves per CU (one 1024-thread workgroup), like the DDC.
//   Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_rot.hip -o ubench_rot
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 2048
#define UNR 16

// MODE 0: full tap (add_co, 2 cndmask, 4 fma)   1: fma only (Horner chain)   2: add_co + 2 cndmask only
// MODE 3: full tap with the selects replaced by v_fma (7 float ops)     4: 4 independent fma + 3 int (no VCC)
template <int MODE>
__global__ void __launch_bounds__(1024) k_rot(float *out, unsigned fs, float c0, float s0, float c1, float s1, float u)
{
	unsigned F = threadIdx.x * 2654435761u;
	float ar = threadIdx.x, ai = 1.0f, rc = c0, rs = s0;
	unsigned g = threadIdx.x;
	for (int i = 0; i < ITERS; ++i) {
#pragma unroll
		for (int k = 0; k < UNR; ++k) {
			if (MODE == 0) {
				asm volatile("v_add_co_u32 %0, vcc, %0, %5\n"
				             "v_cndmask_b32 %3, %6, %8, vcc\n"
				             "v_cndmask_b32 %4, %7, %9, vcc\n"
				             : "+v"(F), "+v"(ar), "+v"(ai), "=v"(rc), "=v"(rs)
				             : "v"(fs), "v"(c0), "v"(s0), "v"(c1), "v"(s1) : "vcc");
			} else if (MODE == 2) {
				asm volatile("v_add_co_u32 %0, vcc, %0, %3\n"
				             "v_cndmask_b32 %1, %4, %6, vcc\n"
				             "v_cndmask_b32 %2, %5, %7, vcc\n"
				             : "+v"(F), "=v"(rc), "=v"(rs)
				             : "v"(fs), "v"(c0), "v"(s0), "v"(c1), "v"(s1) : "vcc");
			} else if (MODE == 3) {
				asm volatile("v_fma_f32 %0, %0, %3, %4\n"
				             "v_fma_f32 %1, %0, %5, %3\n"
				             "v_fma_f32 %2, %0, %6, %4\n"
				             : "+v"(u), "=v"(rc), "=v"(rs) : "v"(c0), "v"(s0), "v"(c1), "v"(s1));
			} else if (MODE == 4) {
				asm volatile("v_add_u32 %0, %0, %3\n"
				             "v_xor_b32 %1, %0, %4\n"
				             "v_and_b32 %2, %0, %5\n"
				             : "+v"(g), "=v"(rc), "=v"(rs) : "v"(fs), "v"(c0), "v"(s0));
			}
			if (MODE != 2) {
				float tr, ti;
				asm volatile("v_fma_f32 %0, -%3, %5, %6\n"
				             "v_fma_f32 %1, %2, %5, %6\n"
				             "v_fma_f32 %2, %2, %4, %0\n"
				             "v_fma_f32 %3, %3, %4, %1\n"
				             : "=&v"(tr), "=&v"(ti), "+v"(ar), "+v"(ai) : "v"(rc), "v"(rs), "v"(u));
			}
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = ar + ai + rc + rs + (float)F + (float)g;
}

// two independent taps per iteration (two output frames of the same channel in flight in one wave)
__global__ void __launch_bounds__(1024) k_rot2(float *out, unsigned fs, float c0, float s0, float c1, float s1, float u)
{
	unsigned F = threadIdx.x * 2654435761u, G = F * 7u + 3u;
	float ar = threadIdx.x, ai = 1.0f, br = 2.0f, bi = threadIdx.x, rc, rs, qc, qs;
	for (int i = 0; i < ITERS; ++i) {
#pragma unroll
		for (int k = 0; k < UNR; ++k) {
			float tr, ti, vr, vi;
			asm volatile("v_add_co_u32 %0, vcc, %0, %14\n"
			             "v_cndmask_b32 %2, %15, %17, vcc\n"
			             "v_cndmask_b32 %3, %16, %18, vcc\n"
			             "v_add_co_u32 %1, vcc, %1, %14\n"
			             "v_cndmask_b32 %4, %15, %17, vcc\n"
			             "v_cndmask_b32 %5, %16, %18, vcc\n"
			             "v_fma_f32 %10, -%7, %3, %19\n"
			             "v_fma_f32 %12, -%9, %5, %19\n"
			             "v_fma_f32 %11, %6, %3, %19\n"
			             "v_fma_f32 %13, %8, %5, %19\n"
			             "v_fma_f32 %6, %6, %2, %10\n"
			             "v_fma_f32 %8, %8, %4, %12\n"
			             "v_fma_f32 %7, %7, %2, %11\n"
			             "v_fma_f32 %9, %9, %4, %13\n"
			             : "+v"(F), "+v"(G), "=&v"(rc), "=&v"(rs), "=&v"(qc), "=&v"(qs), "+v"(ar), "+v"(ai), "+v"(br), "+v"(bi),
			               "=&v"(tr), "=&v"(ti), "=&v"(vr), "=&v"(vi)
			             : "v"(fs), "v"(c0), "v"(s0), "v"(c1), "v"(s1), "v"(u) : "vcc");
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = ar + ai + br + bi + (float)F + (float)G;
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

int main()
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	float *out;
	hipMalloc(&out, sizeof(float) * cus * 2 * 1024);
	const char *names[5] = {"tap: add_co + 2 cndmask + 4 fma", "4 fma (Horner chain)", "add_co + 2 cndmask",
	                        "3 fma + 4 fma", "3 int (no VCC) + 4 fma"};
	const int ninst[5] = {7, 4, 3, 7, 7};
	for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
		const int wg = cus * wgs_per_cu;
		double t[5];
		t[0] = timeit([&] { k_rot<0><<<wg, 1024>>>(out, 12345u, 0.9f, 0.1f, 0.8f, 0.2f, 0.5f); });
		t[1] = timeit([&] { k_rot<1><<<wg, 1024>>>(out, 12345u, 0.9f, 0.1f, 0.8f, 0.2f, 0.5f); });
		t[2] = timeit([&] { k_rot<2><<<wg, 1024>>>(out, 12345u, 0.9f, 0.1f, 0.8f, 0.2f, 0.5f); });
		t[3] = timeit([&] { k_rot<3><<<wg, 1024>>>(out, 12345u, 0.9f, 0.1f, 0.8f, 0.2f, 0.5f); });
		t[4] = timeit([&] { k_rot<4><<<wg, 1024>>>(out, 12345u, 0.9f, 0.1f, 0.8f, 0.2f, 0.5f); });
		{
			double t2 = timeit([&] { k_rot2<<<wg, 1024>>>(out, 12345u, 0.9f, 0.1f, 0.8f, 0.2f, 0.5f); });
			const double taps2 = (double)wg * 16 * ITERS * UNR * 2;
			printf("%d WG/CU  %-34s: %.3f ms  %.3f G wave-taps/s  %.3f T wave-inst/s\n", wgs_per_cu,
			       "two interleaved taps per wave", t2 * 1e3, taps2 / t2 / 1e9, taps2 * 7 / t2 / 1e12);
		}
		for (int m = 0; m < 5; ++m) {
			const double taps = (double)wg * 16 * ITERS * UNR;     // wave-taps
			printf("%d WG/CU  %-34s: %.3f ms  %.3f G wave-taps/s  %.3f T wave-inst/s  (%.2f inst/clk/CU @2.4GHz)\n",
			       wgs_per_cu, names[m], t[m] * 1e3, taps / t[m] / 1e9, taps * ninst[m] / t[m] / 1e12,
			       taps * ninst[m] / t[m] / cus / 2.4e9);
		}
	}
	hipFree(out);
	return 0;
}
