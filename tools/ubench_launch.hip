// Launch floor on gfx950: duration of EMPTY kernels of various shapes (what a kernel of k_tuner_ddc's
// geometry costs before it does anything).  Read the durations with rocprofv3 --kernel-trace.
//   Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o tools/ubench_launch
#include <hip/hip_runtime.h>
#include <cstdio>
template <int T> __global__ void __launch_bounds__(T) k_empty(float *out) { extern __shared__ float l[]; if (out == (float *)1) out[0] = l[threadIdx.x]; }
template <int T> static void go(int wgs, int lds, float *o)
{
	hipFuncSetAttribute((const void *)k_empty<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int i = 0; i < 3; ++i) k_empty<T><<<wgs, T, lds>>>(o);
	hipDeviceSynchronize();
	hipEventRecord(a);
	for (int i = 0; i < 200; ++i) k_empty<T><<<wgs, T, lds>>>(o);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	printf("%5d WGs x %4d threads, %3d KB LDS: %.2f us per launch back to back\n", wgs, T, lds / 1024, ms / 200 * 1e3);
}
int main()
{
	float *o = nullptr;
	for (int lds : {0, 12 * 1024, 40 * 1024}) {
		go<64>(1, lds, o); go<256>(256, lds, o); go<256>(2048, lds, o); go<256>(4096, lds, o);
		go<512>(512, lds, o); go<512>(1024, lds, o); go<512>(1536, lds, o);
		go<1024>(256, lds, o); go<1024>(512, lds, o);
	}
	return 0;
}
