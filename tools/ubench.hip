// Micro-benchmarks that size the DDC kernel's inner loop on gfx950:
//   (a) v_fma_f32 vs v_pk_fma_f32 issue rate, (b) conflict-free ds_read_b64 gather rate,
//   (c) SDWA byte-extract into an LDS address.  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4096

__global__ void __launch_bounds__(256) k_fma(float *out, float a, float b)
{
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	for (int i = 0; i < ITERS; ++i) {
		asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
		             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
		             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void __launch_bounds__(256) k_pkfma(float *out, float a, float b)
{
	v2f x0 = {(float)threadIdx.x, 1}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	v2f va = {a, a}, vb = {b, b};
	for (int i = 0; i < ITERS; ++i) {
		asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
		             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
		             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va), "v"(vb));
	}
	v2f s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// integer mix resembling the phase/address math: add, lshr, bfe, lshl_add, lshl_add
__global__ void __launch_bounds__(256) k_int(unsigned *out, unsigned st)
{
	unsigned p = threadIdx.x * 2654435761u, acc = 0;
	for (int i = 0; i < ITERS; ++i) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			unsigned a = ((p >> 24) << 8) + threadIdx.x;
			unsigned b = (((p >> 16) & 255u) << 8) + threadIdx.x;
			asm volatile("" : "+v"(a), "+v"(b));
			acc ^= a + b;
			p += st;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// conflict-free replicated-table gather: 2 x ds_read_b64 per "tap" + address math
template <int MODE>   // 0: replicated (lane-private bank pair), 1: shared single copy (conflicts)
__global__ void __launch_bounds__(1024) k_lds(float *out, unsigned st)
{
	extern __shared__ v2f tab[];
	for (unsigned e = threadIdx.x; e < 2 * 256 * 32; e += blockDim.x)
		tab[e] = (v2f){(float)e, 1.0f};
	__syncthreads();
	const unsigned lane = threadIdx.x & 31;
	unsigned p = threadIdx.x * 2654435761u;
	v2f acc = {0, 0};
	for (int i = 0; i < ITERS; ++i) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			unsigned ia, ib;
			if (MODE == 0) {
				ia = ((p >> 24) << 5) + lane;
				ib = 8192 + (((p >> 16) & 255u) << 5) + lane;
			} else {
				ia = (p >> 24);
				ib = 8192 + ((p >> 16) & 255u);
			}
			acc += tab[ia] + tab[ib];
			p += st;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
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
	printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
	float *out;
	hipMalloc(&out, sizeof(float) * cus * 8 * 1024);
	const int wg = cus * 8;   // 8 x 256 threads per CU = 32 waves/CU
	double t;
	t = timeit([&] { k_fma<<<wg, 256>>>(out, 1.0001f, 0.5f); });
	double inst = (double)wg * 4 * ITERS * 8;   // wave-instructions
	printf("v_fma_f32    : %.3f ms, %.2f wave-inst/clk/CU @2.4GHz, %.1f TFLOP/s\n", t * 1e3, inst / t / cus / 2.4e9, inst * 64 * 2 / t / 1e12);
	t = timeit([&] { k_pkfma<<<wg, 256>>>(out, 1.0001f, 0.5f); });
	printf("v_pk_fma_f32 : %.3f ms, %.2f wave-inst/clk/CU @2.4GHz, %.1f TFLOP/s\n", t * 1e3, inst / t / cus / 2.4e9, inst * 64 * 4 / t / 1e12);
	t = timeit([&] { k_int<<<wg, 256>>>((unsigned *)out, 12345u); });
	printf("int addr math: %.3f ms, %.2f taps/clk/CU (wave-taps)\n", t * 1e3, inst / t / cus / 2.4e9);
	hipFuncSetAttribute((const void *)k_lds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
	hipFuncSetAttribute((const void *)k_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
	double taps = (double)cus * 16 * ITERS * 8;  // wave-taps, each 2 ds_read_b64
	t = timeit([&] { k_lds<0><<<cus, 1024, 131072>>>(out, 12345u); });
	printf("lds replicated: %.3f ms, %.3f wave-taps/clk/CU (2 ds_read_b64 each)\n", t * 1e3, taps / t / cus / 2.4e9);
	t = timeit([&] { k_lds<1><<<cus, 1024, 131072>>>(out, 12345u); });
	printf("lds shared    : %.3f ms, %.3f wave-taps/clk/CU (2 ds_read_b64 each)\n", t * 1e3, taps / t / cus / 2.4e9);
	hipFree(out);
	return 0;
}
