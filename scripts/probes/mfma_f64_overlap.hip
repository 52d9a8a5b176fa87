// Does FP64 MFMA work overlap FP64 VALU work on gfx950? One loop trip = 3 MFMAs (a 16x16 tile of 10-term quadratic
// forms) and 68 dependent-free v_fma_f64 (the exp + sum work of that tile). Times: both, VALU part only, MFMA part only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>   // 1: fma only, 2: mfma only, 3: both
__global__ __launch_bounds__(256) void k(double* out, int iters)
{
	double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, y = 1.0 + 1e-9 * threadIdx.x;
	double f[17];
#pragma unroll
	for (int i = 0; i < 17; i++) f[i] = i + a;
	double4_t acc = {0, 0, 0, 0};
	for (int it = 0; it < iters; it++) {
		if (MODE & 2) {
			double4_t c = {a, b, a, b};
			c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
			c = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c, 0, 0, 0);
			c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c, 0, 0, 0);
			acc += c;
		}
		if (MODE & 1) {
#pragma unroll
			for (int r = 0; r < 4; r++) {
#pragma unroll
				for (int i = 0; i < 17; i++) f[i] = fma(f[i], y, a);
			}
		}
	}
	double s = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
	for (int i = 0; i < 17; i++) s += f[i];
	out[(size_t) blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K> double run(K kern, int blocks, int iters, double* out)
{
	hipEvent_t t0, t1;
	hipEventCreate(&t0); hipEventCreate(&t1);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
	hipDeviceSynchronize();
	hipEventRecord(t0, 0);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
	hipEventRecord(t1, 0);
	hipEventSynchronize(t1);
	float ms = 0;
	hipEventElapsedTime(&ms, t0, t1);
	return ms;
}

int main()
{
	double* out;
	hipMalloc(&out, (size_t) 256 * 16 * 256 * 8);
	const int iters = 20000, blocks = 256 * 4;   // 4 waves per SIMD
	double t1 = run(k<1>, blocks, iters, out), t2 = run(k<2>, blocks, iters, out), t3 = run(k<3>, blocks, iters, out);
	printf("4 waves/SIMD, %d trips: VALU only %.2f ms, MFMA only %.2f ms, both %.2f ms (sum %.2f, max %.2f)\n", iters, t1, t2, t3, t1 + t2, t1 > t2 ? t1 : t2);
	return 0;
}
