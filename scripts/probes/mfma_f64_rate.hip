// Probe behind DESIGN.md §4 "MFMA: not used": issue rate of v_mfma_f64_16x16x4_f64 against v_fma_f64 on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_probe scripts/probes/mfma_f64_rate.hip && /tmp/mfma_probe
// Result on MI355X (profiles/r01_mfma_f64_probe.txt): 47 TFLOP/s through the matrix core, 53 TFLOP/s through v_fma_f64.
// Every wave runs a chain-free loop (4 independent accumulators); one wave per SIMD and four waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma(double* out, int iters)
{
	double4_t acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
	double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
	for (int i = 0; i < iters; i++) {
		acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
		acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
		acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc2, 0, 0, 0);
		acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, acc3, 0, 0, 0);
	}
	double4_t s = acc0 + acc1 + acc2 + acc3;
	out[(size_t) blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void k_fma(double* out, int iters)
{
	double x = threadIdx.x * 1e-3, y = 1.0 + 1e-9 * threadIdx.x;
	double a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
	for (int i = 0; i < iters; i++) {
		a0 = fma(a0, y, x); a1 = fma(a1, y, x); a2 = fma(a2, y, x); a3 = fma(a3, y, x);
		a4 = fma(a4, y, x); a5 = fma(a5, y, x); a6 = fma(a6, y, x); a7 = fma(a7, y, x);
	}
	out[(size_t) blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <class K>
double run(K k, int blocks, int iters, double* out)
{
	hipEvent_t t0, t1;
	hipEventCreate(&t0); hipEventCreate(&t1);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
	hipDeviceSynchronize();
	hipEventRecord(t0, 0);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
	hipEventRecord(t1, 0);
	hipEventSynchronize(t1);
	float ms = 0;
	hipEventElapsedTime(&ms, t0, t1);
	return ms * 1e-3;
}

int main()
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	double* out;
	hipMalloc(&out, (size_t) cus * 16 * 256 * 8);
	const int iters = 20000;
	for (int wgs_per_cu : {1, 4}) {
		const int blocks = cus * wgs_per_cu;
		double tm = run(k_mfma, blocks, iters, out);
		double tf = run(k_fma, blocks, iters, out);
		// one 16x16x4 MFMA = 2 * 16 * 16 * 4 flop per wave; one wave64 FMA = 2 * 64 flop
		double mfma_flops = (double) blocks * 4 * iters * 4 * 2048.0 / tm;
		double fma_flops  = (double) blocks * 4 * iters * 8 * 128.0 / tf;
		printf("%d CUs, %d waves/SIMD: v_mfma_f64_16x16x4_f64 %.1f TFLOP/s (%.1f cycles per MFMA per SIMD at 2.4 GHz), v_fma_f64 %.1f TFLOP/s\n",
		       cus, wgs_per_cu, mfma_flops * 1e-12, tm * 2.4e9 / ((double) wgs_per_cu * iters * 4), fma_flops * 1e-12);
	}
	return 0;
}
