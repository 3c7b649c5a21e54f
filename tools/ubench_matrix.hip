// tools/ubench_matrix.hip -- VERDICT r05 item 5: the matrix-pipe formulation of the C2 tap sum, measured.  NOT a product path.
//
// The reference mixes frame n of a channel with table entry idx(n) = (P0 + n step) >> 16 (downconverter.cxx:100-103, left-aligned
// 32-bit phase) and sums 64 taps per channel-rate frame (lowpass.cxx:145-159).  With Pk the phase of a window's first frame,
// a = Pk & 0xFFFF, I = Pk >> 16, and m step = J_m 2^16 + b_m:   idx(m) = I + J_m + [a + b_m >= 2^16], so
//     y[c][k] = conj(E(I_ck)) ( A[c][k] + (conj(d) - 1) B[c][k] ),     d = cis(2 pi / 65536),
//     A[c][k] = sum_m M'[c][m] x[k][m]                    a plain [C x 64] x [64 x K] complex GEMM   (M' = h_m conj(cis(2 pi J_cm / 65536)))
//     B[c][k] = sum_{m : a_ck + b_cm >= 2^16} M'[c][m] x[k][m]          a THRESHOLDED sum: no contraction
// (DESIGN.md 3.1).  |conj(d) - 1| = 9.6e-5, so B may be computed in half precision.  This program, for one C2 block
// (256 channels, 10 000 channel-rate frames, D = 400):
//   * the reference arithmetic in f32 (table lookup per frame, unfused, reference order)                          -> y_exact
//   * A in f32, and A as the bf16 x 3 split an MFMA would compute (six cross products, f32 accumulation: emulated on the VALU,
//     not timed) -- and the rate the matrix pipe reaches on exactly the MFMAs that GEMM needs (register-resident operands: an
//     issue-bound LOWER bound of a real GEMM kernel's time), bf16 32x32x16 (x 6 products) and f32 32x32x2
//   * B on the VALU with lanes = frames, the window as 64 packed half pairs in registers, M' and the thresholds in scalar
//     registers: v_add_co + v_cndmask + 2 v_dot2_f32_f16 = 4 instructions per tap (timed), and in f32 (7 per tap, timed)
//   * max |y - y_exact| of every combination.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_matrix.hip -o tools/ubench_matrix
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define C_CH 256
#define K1 10000
#define D1 400
#define TAPS 64
#define CPW 16                      /* channels a wave of the B kernel walks */
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Chan { unsigned step; };

// ---- the reference arithmetic: lanes = frames ----
__global__ void k_exact(const float2 *__restrict__ x, const float *__restrict__ table, const float *__restrict__ h,
                        const unsigned *__restrict__ step, float2 *__restrict__ y)
{
	const unsigned k = blockIdx.x * 64 + (threadIdx.x & 63), c = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (k >= K1)
		return;
	const unsigned st = step[c];
	unsigned P = (k * D1) * st;
	float ai = 0.0f, aq = 0.0f;
	for (int m = 0; m < TAPS; ++m) {
		const float2 s = x[(size_t)k * D1 + m];
		const float sn = table[P >> 16], cs = table[((P >> 16) + 16384u) & 65535u];
		const float mi = s.x * cs + s.y * sn, mq = s.y * cs - s.x * sn;          /* downconverter.cxx:109-110 */
		ai = ai + h[m] * mi;                                                       /* lowpass.cxx:153-156, oldest first */
		aq = aq + h[m] * mq;
		P += st;
	}
	y[(size_t)c * K1 + k] = make_float2(ai, aq);
}

// ---- A: f32, and the bf16 x 3 split emulated ----
__device__ __forceinline__ float bf16_round(float v) { return (float)(__bf16)v; }
template <int SPLIT>
__global__ void k_A(const float2 *__restrict__ x, const float2 *__restrict__ M, float2 *__restrict__ A)
{
	const unsigned k = blockIdx.x * 64 + (threadIdx.x & 63), c = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (k >= K1)
		return;
	float ar = 0.0f, ai = 0.0f;
	for (int m = 0; m < TAPS; ++m) {
		const float2 s = x[(size_t)k * D1 + m], w = M[c * TAPS + m];
		if (SPLIT == 0) {
			ar = __builtin_fmaf(w.x, s.x, ar); ar = __builtin_fmaf(-w.y, s.y, ar);
			ai = __builtin_fmaf(w.x, s.y, ai); ai = __builtin_fmaf(w.y, s.x, ai);
		} else {
			float a[4] = {w.x, w.y, s.x, s.y}, p[4][3];
			for (int i = 0; i < 4; ++i) {
				p[i][0] = bf16_round(a[i]);
				p[i][1] = bf16_round(a[i] - p[i][0]);
				p[i][2] = bf16_round(a[i] - p[i][0] - p[i][1]);
			}
			/* u v ~ sum over split pairs (i, j) with i + j <= 2: six products, each exact in f32 */
			auto mul = [&](int u, int v, float acc, float sign) {
				for (int i = 0; i < 3; ++i)
					for (int j = 0; i + j < 3; ++j)
						acc = __builtin_fmaf(sign * p[u][i], p[v][j], acc);
				return acc;
			};
			ar = mul(0, 2, ar, 1.0f); ar = mul(1, 3, ar, -1.0f);
			ai = mul(0, 3, ai, 1.0f); ai = mul(1, 2, ai, 1.0f);
		}
	}
	A[(size_t)c * K1 + k] = make_float2(ar, ai);
}

// ---- the windows as packed halves, [tap][frame] ----
__global__ void k_pack(const float2 *__restrict__ x, h2 *__restrict__ x16)
{
	const unsigned k = blockIdx.x * 64 + (threadIdx.x & 63), m = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (k >= K1)
		return;
	const float2 s = x[(size_t)k * D1 + m];
	x16[(size_t)m * K1 + k] = (h2){(_Float16)s.x, (_Float16)s.y};
}

// ---- B in half precision: 4 VALU instructions per tap ----
template <int CPWT>
__global__ void __launch_bounds__(256) k_B16(const h2 *__restrict__ x16, const unsigned *__restrict__ step, const unsigned *__restrict__ thr,
                                             const h2 *__restrict__ Mre, const h2 *__restrict__ Mim, float2 *__restrict__ B)
{
	const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned k = blockIdx.x * 64 + lane, c0 = (blockIdx.y * 4 + wave) * CPWT;
	const unsigned kk = k < K1 ? k : K1 - 1;
	h2 w[TAPS];
#pragma unroll
	for (int m = 0; m < TAPS; ++m)
		w[m] = x16[(size_t)m * K1 + kk];
	const h2 zero = {(_Float16)0.0f, (_Float16)0.0f};
	/* two channels side by side: two independent chains, and a carry is consumed two instructions after it is made */
	for (unsigned c = c0; c < c0 + CPWT; c += 2) {
		const unsigned F0 = (kk * D1 * step[c]) << 16, F1 = (kk * D1 * step[c + 1]) << 16;   /* the 16 fraction bits, left-aligned */
		float br0 = 0.0f, bi0 = 0.0f, br1 = 0.0f, bi1 = 0.0f;
#pragma unroll
		for (int m = 0; m < TAPS; ++m) {
			unsigned t;
			const bool cy0 = __builtin_uadd_overflow(F0, thr[c * TAPS + m], &t);
			const bool cy1 = __builtin_uadd_overflow(F1, thr[(c + 1) * TAPS + m], &t);
			const h2 x0 = cy0 ? w[m] : zero, x1 = cy1 ? w[m] : zero;
			br0 = __builtin_amdgcn_fdot2(x0, Mre[c * TAPS + m], br0, false);
			bi0 = __builtin_amdgcn_fdot2(x0, Mim[c * TAPS + m], bi0, false);
			br1 = __builtin_amdgcn_fdot2(x1, Mre[(c + 1) * TAPS + m], br1, false);
			bi1 = __builtin_amdgcn_fdot2(x1, Mim[(c + 1) * TAPS + m], bi1, false);
		}
		if (k < K1) {
			B[(size_t)c * K1 + k] = make_float2(br0, bi0);
			B[(size_t)(c + 1) * K1 + k] = make_float2(br1, bi1);
		}
	}
}

// ---- B in f32: 7 per tap ----
__global__ void __launch_bounds__(256) k_B32(const float2 *__restrict__ x, const unsigned *__restrict__ step, const unsigned *__restrict__ thr,
                                             const float2 *__restrict__ M, float2 *__restrict__ B)
{
	const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned k = blockIdx.x * 64 + lane, c0 = (blockIdx.y * 4 + wave) * CPW;
	const unsigned kk = k < K1 ? k : K1 - 1;
	float2 w[TAPS];
#pragma unroll
	for (int m = 0; m < TAPS; ++m)
		w[m] = x[(size_t)kk * D1 + m];
	for (unsigned c = c0; c < c0 + CPW; ++c) {
		const unsigned F = (kk * D1 * step[c]) << 16;
		float br = 0.0f, bi = 0.0f;
#pragma unroll
		for (int m = 0; m < TAPS; ++m) {
			unsigned t;
			const bool cy = __builtin_uadd_overflow(F, thr[c * TAPS + m], &t);
			const float xr = cy ? w[m].x : 0.0f, xi = cy ? w[m].y : 0.0f;
			const float2 mm = M[c * TAPS + m];
			br = __builtin_fmaf(mm.x, xr, br); br = __builtin_fmaf(-mm.y, xi, br);
			bi = __builtin_fmaf(mm.x, xi, bi); bi = __builtin_fmaf(mm.y, xr, bi);
		}
		if (k < K1)
			B[(size_t)c * K1 + k] = make_float2(br, bi);
	}
}

// ---- y = conj(E(I)) (A + (conj(d) - 1) B) ----
__global__ void k_combine(const float2 *__restrict__ A, const float2 *__restrict__ B, const float *__restrict__ table,
                          const unsigned *__restrict__ step, float dr, float di, float2 *__restrict__ y)
{
	const unsigned k = blockIdx.x * 64 + (threadIdx.x & 63), c = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (k >= K1)
		return;
	const unsigned I = (k * D1 * step[c]) >> 16;
	const float sn = table[I], cs = table[(I + 16384u) & 65535u];
	const float2 a = A[(size_t)c * K1 + k], b = B[(size_t)c * K1 + k];
	const float tr = a.x + (dr * b.x - di * b.y), ti = a.y + (dr * b.y + di * b.x);
	y[(size_t)c * K1 + k] = make_float2(tr * cs + ti * sn, ti * cs - tr * sn);
}

// ---- the matrix pipe on exactly the MFMAs the A GEMM needs: operands in registers (an issue-bound lower bound) ----
template <int KIND>       /* 0: bf16 32x32x16, 1: f32 32x32x2 */
__global__ void __launch_bounds__(256) k_mfma(float *out, int per_wave, float fa)
{
	v16f d0 = {0}, d1 = {0};
	bf8 a, b;
	for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(fa + i + threadIdx.x); b[i] = (__bf16)(fa - i); }
	float af = fa + threadIdx.x, bfv = fa * 0.5f;
	for (int i = 0; i < per_wave; i += 2) {
		if (KIND == 0) {
			d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
			d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, d1, 0, 0, 0);
		} else {
			d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bfv, d0, 0, 0, 0);
			d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bfv, af, d1, 0, 0, 0);
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d0[5] + d1[3] + d1[15];
}

template <typename Fn> static double time_us(Fn f, int reps = 200)
{
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	/* (an idle MI355X starts a kernel stream at a reduced clock and takes ~50 ms of continuous load to ramp up,
	 * profiles/r02_clock_ramp.txt: 100 ms of the same launches first) */
	hipEvent_t w0, w1;
	CHECK(hipEventCreate(&w0)); CHECK(hipEventCreate(&w1));
	CHECK(hipEventRecord(w0));
	for (;;) {
		for (int i = 0; i < 50; ++i) f();
		CHECK(hipEventRecord(w1)); CHECK(hipEventSynchronize(w1));
		float ms; CHECK(hipEventElapsedTime(&ms, w0, w1));
		if (ms > 100.0f) break;
	}
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	for (int r = 0; r < reps; ++r) f();
	CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
	float ms;
	CHECK(hipEventElapsedTime(&ms, a, b));
	return ms / reps * 1e3;
}

static double max_err(const std::vector<float2> &a, const std::vector<float2> &b)
{
	double e = 0;
	for (size_t i = 0; i < a.size(); ++i) {
		e = fmax(e, fabs((double)a[i].x - b[i].x));
		e = fmax(e, fabs((double)a[i].y - b[i].y));
	}
	return e;
}

int main()
{
	const size_t NX = (size_t)K1 * D1 + TAPS;
	std::vector<float2> x(NX);
	std::vector<float> table(65536), h(TAPS);
	for (int n = 0; n < 65536; ++n)
		table[n] = sinf((float)(n * 2.0 * M_PI / 65536.0));                  /* downconverter.cxx:49-51 */
	/* LowPass::recalculate for passband 6.4 MHz at 100 Msps (maxbin 2), lowpass.cxx:164-197; h[m] = coeff[63 - m] */
	{
		double coeff[TAPS];
		for (int n = 0; n < TAPS; ++n) {
			const int t = (n + 32) & 63;
			const double imp = 1.0 + 2.0 * cos(2.0 * M_PI * t / 64.0);
			const float w = (float)((0.54 - 0.46 * cosf((float)(2.0 * M_PI * (float)n / 63.0f))) / 64.0);
			coeff[n] = imp * w;
		}
		for (int m = 0; m < TAPS; ++m)
			h[m] = (float)coeff[63 - m];
	}
	std::vector<unsigned> step(C_CH), thr((size_t)C_CH * TAPS);
	std::vector<float2> M((size_t)C_CH * TAPS);
	std::vector<h2> Mre(M.size()), Mim(M.size());
	for (int c = 0; c < C_CH; ++c) {
		const long long f = -39843750LL + 312500LL * c;
		const int ps = (int)(f * 2147483648LL / 100000000LL);                 /* downconverter.cxx:80 */
		step[c] = (unsigned)ps << 1;
		for (int m = 0; m < TAPS; ++m) {
			const unsigned ms = (unsigned)m * step[c];
			const unsigned J = ms >> 16, b = ms & 0xFFFFu;
			thr[(size_t)c * TAPS + m] = b << 16;
			const double ang = 2.0 * M_PI * (double)J / 65536.0;
			const double wr = h[m] * cos(ang), wi = -h[m] * sin(ang);          /* h conj(cis) */
			M[(size_t)c * TAPS + m] = make_float2((float)wr, (float)wi);
			Mre[(size_t)c * TAPS + m] = (h2){(_Float16)wr, (_Float16)(-wi)};   /* (xr, xi) . (wr, -wi) = Re */
			Mim[(size_t)c * TAPS + m] = (h2){(_Float16)wi, (_Float16)wr};      /* (xr, xi) . (wi, wr) = Im */
		}
	}
	/* a C2-like stream: FM carriers on every 4th channel, 0.5 / 64 each, and noise at -40 dBFS */
	{
		unsigned lcg = 12345;
		for (size_t n = 0; n < NX; ++n) {
			if (n % D1 >= TAPS) {                                   /* (only the frames under the taps are ever read) */
				x[n] = make_float2(0.0f, 0.0f);
				continue;
			}
			double re = 0, im = 0;
			for (int q = 0; q < 64; ++q) {
				const int c = 4 * q;
				const double f = -39843750.0 + 312500.0 * c, fm = 300.0 + 10.0 * c;
				const double p = 2.0 * M_PI * fmod(f * (double)n / 1e8, 1.0) + 5.0 * sin(2.0 * M_PI * fm * (double)n / 1e8);
				re += cos(p); im += sin(p);
			}
			lcg = lcg * 1664525u + 1013904223u; const double n1 = ((double)(lcg >> 8) - 8388608.0) / 8388608.0;
			lcg = lcg * 1664525u + 1013904223u; const double n2 = ((double)(lcg >> 8) - 8388608.0) / 8388608.0;
			x[n] = make_float2((float)(re * 0.5 / 64 + 0.01 * n1), (float)(im * 0.5 / 64 + 0.01 * n2));
		}
	}
	float2 *dx, *dM, *dA, *dA3, *dB16, *dB32, *dy, *dye; float *dt, *dh, *dout; unsigned *dstep, *dthr; h2 *dx16, *dMre, *dMim;
	const size_t NY = (size_t)C_CH * K1;
	CHECK(hipMalloc(&dx, NX * 8)); CHECK(hipMalloc(&dM, M.size() * 8)); CHECK(hipMalloc(&dA, NY * 8)); CHECK(hipMalloc(&dA3, NY * 8));
	CHECK(hipMalloc(&dB16, NY * 8)); CHECK(hipMalloc(&dB32, NY * 8)); CHECK(hipMalloc(&dy, NY * 8)); CHECK(hipMalloc(&dye, NY * 8));
	CHECK(hipMalloc(&dt, 65536 * 4)); CHECK(hipMalloc(&dh, TAPS * 4)); CHECK(hipMalloc(&dstep, C_CH * 4)); CHECK(hipMalloc(&dthr, thr.size() * 4));
	CHECK(hipMalloc(&dx16, (size_t)TAPS * K1 * 4)); CHECK(hipMalloc(&dMre, M.size() * 4)); CHECK(hipMalloc(&dMim, M.size() * 4));
	CHECK(hipMalloc(&dout, 4096 * 256 * 4));
	CHECK(hipMemcpy(dx, x.data(), NX * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(dt, table.data(), 65536 * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dh, h.data(), TAPS * 4, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(dstep, step.data(), C_CH * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dthr, thr.data(), thr.size() * 4, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(dMre, Mre.data(), M.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dMim, Mim.data(), M.size() * 4, hipMemcpyHostToDevice));
	const dim3 gk((K1 + 63) / 64, C_CH / 4), gb((K1 + 63) / 64, C_CH / (4 * CPW));
	const double dl = 2.0 * M_PI / 65536.0;
	const float dr = (float)(cos(dl) - 1.0), di = (float)(-sin(dl));           /* conj(d) - 1 */

	k_exact<<<gk, 256>>>(dx, dt, dh, dstep, dye);
	k_A<0><<<gk, 256>>>(dx, dM, dA);
	k_A<1><<<gk, 256>>>(dx, dM, dA3);
	k_pack<<<dim3((K1 + 63) / 64, TAPS / 4), 256>>>(dx, dx16);
	k_B16<CPW><<<gb, 256>>>(dx16, dstep, dthr, dMre, dMim, dB16);
	k_B32<<<gb, 256>>>(dx, dstep, dthr, dM, dB32);
	CHECK(hipDeviceSynchronize());
	std::vector<float2> ye(NY), yv(NY);
	CHECK(hipMemcpy(ye.data(), dye, NY * 8, hipMemcpyDeviceToHost));
	double ymax = 0;
	for (size_t i = 0; i < NY; ++i) ymax = fmax(ymax, fmax(fabs(ye[i].x), fabs(ye[i].y)));
	printf("one C2 block: %d channels x %d channel-rate frames x %d taps; max |y_exact| = %.4f\n", C_CH, K1, TAPS, ymax);
	struct { const char *name; float2 *A, *B; } combos[] = {
		{"A f32          + B f32 ", dA, dB32}, {"A f32          + B half", dA, dB16},
		{"A bf16 x 3 (6) + B f32 ", dA3, dB32}, {"A bf16 x 3 (6) + B half", dA3, dB16},
	};
	for (auto &cb : combos) {
		k_combine<<<gk, 256>>>(cb.A, cb.B, dt, dstep, dr, di, dy);
		CHECK(hipMemcpy(yv.data(), dy, NY * 8, hipMemcpyDeviceToHost));
		printf("  %s : max |y - y_exact| = %.3e\n", cb.name, max_err(yv, ye));
	}
	{
		/* what the B term is worth: y with B left out */
		CHECK(hipMemset(dy, 0, NY * 8));
		float2 *dz; CHECK(hipMalloc(&dz, NY * 8)); CHECK(hipMemset(dz, 0, NY * 8));
		k_combine<<<gk, 256>>>(dA, dz, dt, dstep, dr, di, dy);
		CHECK(hipMemcpy(yv.data(), dy, NY * 8, hipMemcpyDeviceToHost));
		printf("  A f32 alone (B = 0: an exact-phase LO)  : max |y - y_exact| = %.3e\n", max_err(yv, ye));
	}
	const double tpack = time_us([&] { k_pack<<<dim3((K1 + 63) / 64, TAPS / 4), 256>>>(dx, dx16); });
	const double tb16 = time_us([&] { k_B16<CPW><<<gb, 256>>>(dx16, dstep, dthr, dMre, dMim, dB16); });
	const double tb16_2 = time_us([&] { k_B16<2><<<dim3((K1 + 63) / 64, C_CH / 8), 256>>>(dx16, dstep, dthr, dMre, dMim, dB16); });
	const double tb16_4 = time_us([&] { k_B16<4><<<dim3((K1 + 63) / 64, C_CH / 16), 256>>>(dx16, dstep, dthr, dMre, dMim, dB16); });
	const double tb16_8 = time_us([&] { k_B16<8><<<dim3((K1 + 63) / 64, C_CH / 32), 256>>>(dx16, dstep, dthr, dMre, dMim, dB16); });
	const double tb16_32 = time_us([&] { k_B16<32><<<dim3((K1 + 63) / 64, C_CH / 128), 256>>>(dx16, dstep, dthr, dMre, dMim, dB16); });
	const double tb32 = time_us([&] { k_B32<<<gb, 256>>>(dx, dstep, dthr, dM, dB32); });
	const double tcomb = time_us([&] { k_combine<<<gk, 256>>>(dA, dB16, dt, dstep, dr, di, dy); });
	printf("timed (us per block, HIP events, 200 launches back to back behind 100 ms of the same launches):\n");
	printf("  windows -> packed halves [tap][frame]            %7.2f\n", tpack);
	printf("  B, half precision (4 VALU per tap)               %7.2f   = %.2f T wave-instructions/s on the taps\n", tb16,
	       (double)C_CH * K1 / 64 * TAPS * 4 / (tb16 * 1e-6) / 1e12);
	printf("     ... with 2 / 4 / 8 / 32 channels per wave instead of %d: %.2f / %.2f / %.2f / %.2f\n", CPW, tb16_2, tb16_4, tb16_8, tb16_32);
	printf("  B, f32 (7 VALU per tap)                          %7.2f\n", tb32);
	printf("  combine (anchor lookup, 2 complex multiply-adds) %7.2f\n", tcomb);
	hipDeviceProp_t p;
	CHECK(hipGetDeviceProperties(&p, 0));
	/* the A GEMM as real [512 x 128] x [128 x 10000]: 32 x 32 output tiles x K / 16 (bf16) or K / 2 (f32) MFMAs each */
	const double tiles = 16.0 * ceil(K1 / 32.0);
	for (int wpc = 4; wpc <= 16; wpc *= 2) {
		const int wgs = p.multiProcessorCount * wpc / 4;
		const double waves = wgs * 4.0;
		const int per_bf = (int)ceil(tiles * (128 / 16) * 6 / waves / 2) * 2, per_f32 = (int)ceil(tiles * (128 / 2) / waves / 2) * 2;
		const double t0 = time_us([&] { k_mfma<0><<<wgs, 256>>>(dout, per_bf, 0.5f); });
		const double t1 = time_us([&] { k_mfma<1><<<wgs, 256>>>(dout, per_f32, 0.5f); });
		printf("  A's MFMAs alone, %2d waves per CU: bf16 x 3 split (6 products, %d v_mfma_f32_32x32x16_bf16 per wave) %6.2f us = %.0f TFLOP/s;"
		       "  f32 (%d v_mfma_f32_32x32x2_f32 per wave) %6.2f us = %.0f TFLOP/s\n", wpc, per_bf, t0, waves * per_bf * 32768.0 / (t0 * 1e-6) / 1e12,
		       per_f32, t1, waves * per_f32 * 4096.0 / (t1 * 1e-6) / 1e12);
	}
	return 0;
}
