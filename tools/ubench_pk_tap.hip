// tools/ubench_pk_tap.hip -- the ROTATE tap of k_tuner_ddc with its four v_fma_f32 as two v_pk_fma_f32 (op_sel picks the halves):
//   t   = (-ai, ar) * (rs, rs) + (ur, ui)        src0 = acc swapped, low half negated; src1 = r.hi twice
//   acc = ( ar, ai) * (rc, rc) + t               src1 = r.lo twice
// against the plain seven-instruction tap, two recurrences per wave (what the kernel runs), explicit registers.
#include <hip/hip_runtime.h>
#include <cstdio>

#define TAP_PLAIN(F, AR, AI, RC, RS, TR, TI)                                   \
	"v_add_co_u32 " F ", vcc, " F ", %0\n\t"                                   \
	"v_cndmask_b32 " RC ", %1, %3, vcc\n\t"                                    \
	"v_cndmask_b32 " RS ", %2, %4, vcc\n\t"                                    \
	"v_fma_f32 " TR ", -" AI ", " RS ", v98\n\t"                               \
	"v_fma_f32 " TI ", " AR ", " RS ", v99\n\t"                                \
	"v_fma_f32 " AR ", " AR ", " RC ", " TR "\n\t"                             \
	"v_fma_f32 " AI ", " AI ", " RC ", " TI "\n\t"

#define TAP_PK(F, ACC, RC, RS, R, T)                                           \
	"v_add_co_u32 " F ", vcc, " F ", %0\n\t"                                   \
	"v_cndmask_b32 " RC ", %1, %3, vcc\n\t"                                    \
	"v_cndmask_b32 " RS ", %2, %4, vcc\n\t"                                    \
	"v_pk_fma_f32 " T ", " ACC ", " R ", v[98:99] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t" \
	"v_pk_fma_f32 " ACC ", " ACC ", " R ", " T " op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"

#define CLOB "vcc", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", \
             "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119"

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, unsigned fs, float c0, float s0, float c1, float s1, int iters)
{
	asm volatile("v_mov_b32 v98, 0x3a000000\n\tv_mov_b32 v99, 0x3a800000\n\t"
	             "v_mov_b32 v100, 0x9e3779b1\n\tv_mul_lo_u32 v100, %0, v100\n\tv_mov_b32 v110, 0x85ebca6b\n\tv_mul_lo_u32 v110, %0, v110\n\t"
	             "v_cvt_f32_u32 v102, %0\n\tv_mov_b32 v103, 1.0\n\tv_cvt_f32_u32 v112, %0\n\tv_mov_b32 v113, 2.0\n\t"
	             :: "v"(threadIdx.x) : CLOB);
	for (int i = 0; i < iters; ++i) {
#pragma unroll
		for (int kk = 0; kk < 16; ++kk) {
			if (MODE == 0)
				asm volatile(TAP_PLAIN("v100", "v102", "v103", "v104", "v105", "v106", "v107")
				             TAP_PLAIN("v110", "v112", "v113", "v114", "v115", "v116", "v117")
				             :: "v"(fs), "v"(c0), "v"(s0), "v"(c1), "v"(s1) : CLOB);
			else
				asm volatile(TAP_PK("v100", "v[102:103]", "v104", "v105", "v[104:105]", "v[106:107]")
				             TAP_PK("v110", "v[112:113]", "v114", "v115", "v[114:115]", "v[116:117]")
				             :: "v"(fs), "v"(c0), "v"(s0), "v"(c1), "v"(s1) : CLOB);
		}
	}
	float r;
	asm volatile("v_add_f32 %0, v102, v103\n\tv_add_f32 %0, %0, v112\n\tv_add_f32 %0, %0, v113" : "=v"(r) :: CLOB);
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char *name, float *d)
{
	const int iters = 400;
	for (int w = 2; w <= 8; w += 2) {
		hipEvent_t e0, e1;
		(void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
		const int grid = 256 * w;
		k<MODE><<<grid, 256>>>(d, 0x01234567u, 0.999f, 0.01f, 0.998f, 0.02f, 10);
		(void)hipEventRecord(e0);
		k<MODE><<<grid, 256>>>(d, 0x01234567u, 0.999f, 0.01f, 0.998f, 0.02f, iters);
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		float ms;
		(void)hipEventElapsedTime(&ms, e0, e1);
		const double taps = (double)grid * 4 * iters * 16 * 2;
		printf("%-10s %d waves/SIMD: %.1f G wave-taps/s\n", name, w, taps / (ms * 1e-3) / 1e9);
	}
}

int main()
{
	float *d;
	(void)hipMalloc(&d, 256 * 8 * 256 * 4);
	float h[4];
	run<0>("plain", d);
	(void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
	printf("  plain: %.9g %.9g\n", h[0], h[1]);
	run<1>("packed", d);
	(void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
	printf("  packed: %.9g %.9g\n", h[0], h[1]);
	return 0;
}
