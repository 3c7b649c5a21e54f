// tools/ubench_fetch.hip -- what does rocprofv3's FETCH_SIZE count for the load shapes of this repo's kernels?
// (tools/make_traffic.py doubled FETCH_SIZE for the FFT passes -- wide coalesced streaming reads, which
// gfx950 tallies at 64 of their 128 bytes, MI355X_MICROARCH.md "HBM" -- and took it as counted for the DDC's 512-byte window
// loads.  Which reading is right for which shape is measured here on KNOWN byte counts.)
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o tools/ubench_fetch
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/ubench_fetch
//
// Every kernel reads `bytes` bytes exactly once out of a 1 GiB buffer (four times the Infinity Cache) and prints the count
// it asked for; the harness (tools/fetch_calibration.sh) divides the counter by it.
//   k_win8    the DDC's shape: a wave reads ONE 512-byte window (8 bytes per lane), windows 3200 bytes apart
//             (64 of every 400 frames of 8 bytes: BASELINE config 2)
//   k_win8u   the same windows of a byte-format block: 128 bytes per wave (2 bytes per lane), 800 bytes apart
//   k_seq8    8 bytes per lane, the whole buffer front to back (the post stage's channel-IQ rows: 512 bytes per wave-load)
//   k_seq16   16 bytes per lane, front to back (the FFT passes' float4 loads; the guide's calibration shape)
//   k_seq4    4 bytes per lane, front to back
//   k_st8<false|true>, k_st4   WRITE_SIZE: the whole buffer written once, 8 bytes per lane plain / write-through (the DDC's
//             channel-IQ rows), 4 bytes per lane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_win8(const float2 *__restrict__ p, size_t nwin, size_t stride_frames, float *sink)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const unsigned lane = threadIdx.x & 63u;
	float acc = 0.0f;
	for (size_t w = wave; w < nwin; w += nwaves) {
		const float2 v = p[w * stride_frames + lane];
		acc += v.x + v.y;
	}
	if (acc == 123456.789f)
		*sink = acc;
}

__global__ void __launch_bounds__(256) k_win8u(const uchar2 *__restrict__ p, size_t nwin, size_t stride_frames, float *sink)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const unsigned lane = threadIdx.x & 63u;
	float acc = 0.0f;
	for (size_t w = wave; w < nwin; w += nwaves) {
		const uchar2 v = p[w * stride_frames + lane];
		acc += (float)v.x + (float)v.y;
	}
	if (acc == 123456.789f)
		*sink = acc;
}

template <typename T>
__global__ void __launch_bounds__(256) k_seq(const T *__restrict__ p, size_t n, float *sink)
{
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
	float acc = 0.0f;
	for (size_t i = tid; i < n; i += nt) {
		const T v = p[i];
		acc += *(const float *)&v;
	}
	if (acc == 123456.789f)
		*sink = acc;
}

/* stores: 8 bytes per lane, plain and write-through (the DDC's channel-IQ rows), and 4 bytes per lane (the audio) */
template <bool SC1>
__global__ void __launch_bounds__(256) k_st8(unsigned long long *__restrict__ p, size_t n)
{
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
	for (size_t i = tid; i < n; i += nt) {
		if (SC1)
			__hip_atomic_store(&p[i], (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else
			p[i] = (unsigned long long)i;
	}
}
__global__ void __launch_bounds__(256) k_st4(float *__restrict__ p, size_t n)
{
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
	for (size_t i = tid; i < n; i += nt)
		p[i] = (float)i;
}

int main()
{
	const size_t bytes = (size_t)1 << 30;
	char *buf;
	float *sink;
	CHECK(hipMalloc((void **)&buf, bytes));
	CHECK(hipMalloc((void **)&sink, 4));
	CHECK(hipMemset(buf, 1, bytes));
	CHECK(hipDeviceSynchronize());
	const int grid = 2048;
	for (int rep = 0; rep < 3; ++rep) {
		const size_t nwin = bytes / 3200;               /* windows of 64 frames, 400 frames apart, 8 bytes a frame */
		k_win8<<<grid, 256>>>((const float2 *)buf, nwin, 400, sink);
		const size_t nwinu = bytes / 800;               /* the same windows of a byte-format block */
		k_win8u<<<grid, 256>>>((const uchar2 *)buf, nwinu, 400, sink);
		k_seq<float2><<<grid, 256>>>((const float2 *)buf, bytes / 8, sink);
		k_seq<float4><<<grid, 256>>>((const float4 *)buf, bytes / 16, sink);
		k_seq<float><<<grid, 256>>>((const float *)buf, bytes / 4, sink);
		k_st8<false><<<grid, 256>>>((unsigned long long *)buf, bytes / 8);
		k_st8<true><<<grid, 256>>>((unsigned long long *)buf, bytes / 8);
		k_st4<<<grid, 256>>>((float *)buf, bytes / 4);
		CHECK(hipDeviceSynchronize());
		if (rep == 0)
			printf("bytes asked for per launch: k_win8 %zu  k_win8u %zu  k_seq<float2> %zu  k_seq<float4> %zu  k_seq<float> %zu\n",
			       nwin * 512, nwinu * 128, bytes, bytes, bytes);
	}
	CHECK(hipFree(buf));
	CHECK(hipFree(sink));
	return 0;
}
