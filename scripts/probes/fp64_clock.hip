// Probe behind DESIGN.md §4 "the FP64 ceiling": what the chip sustains in a loop of independent v_fma_f64 at 1, 2, 4 and 8
// waves per SIMD, AND the clock it holds while doing so — the in-kernel clock of MI355X_MICROARCH.md, DVFS give-back item 6:
// delta s_memtime (shader clock) / delta s_memrealtime (100 MHz) around the loop, median over the waves, after two seconds of
// back-to-back launches. With the clock known, "53 TFLOP/s" reads either as a clock (cycles per FMA at the 4-cycle cadence)
// or as stalls.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_clock scripts/probes/fp64_clock.hip && ./fp64_clock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_fma(double* out, long long* stamps, int iters, double y0)
{
	double x = threadIdx.x * 1e-3 + 0.25, y = y0 + 1e-9 * threadIdx.x;
	double a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
	const long long c0 = clock64(), r0 = wall_clock64();
	for (int i = 0; i < iters; i++) {
		a0 = fma(a0, y, x); a1 = fma(a1, y, x); a2 = fma(a2, y, x); a3 = fma(a3, y, x);
		a4 = fma(a4, y, x); a5 = fma(a5, y, x); a6 = fma(a6, y, x); a7 = fma(a7, y, x);
	}
	const long long c1 = clock64(), r1 = wall_clock64();
	out[(size_t) blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	if ((threadIdx.x & 63) == 0) {
		const size_t w = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6);
		stamps[2 * w] = c1 - c0;
		stamps[2 * w + 1] = r1 - r0;
	}
}

int main()
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	double* out;
	long long* st;
	const int maxblocks = cus * 8;
	hipMalloc(&out, (size_t) maxblocks * 256 * 8);
	hipMalloc(&st, (size_t) maxblocks * 4 * 2 * 8);
	const int iters = 20000;
	printf("%s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
	for (int wgs_per_cu : {1, 2, 4, 8}) {
		const int blocks = cus * wgs_per_cu;
		// two seconds of back-to-back launches first: the clock the chip settles at under this load
		const auto t0 = std::chrono::steady_clock::now();
		int warm = 0;
		while (std::chrono::steady_clock::now() - t0 < std::chrono::seconds(2)) {
			for (int q = 0; q < 8; q++) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, st, iters, 0.999999);
			hipDeviceSynchronize();
			warm += 8;
		}
		hipEvent_t e0, e1;
		hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, st, iters, 0.999999);
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		std::vector<long long> h((size_t) blocks * 8);
		hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
		std::vector<double> clk, cyc;
		for (int w = 0; w < blocks * 4; w++) {
			if (h[2 * w + 1] > 0) clk.push_back((double) h[2 * w] / (double) h[2 * w + 1] * 100e6);
			cyc.push_back((double) h[2 * w]);
		}
		std::sort(clk.begin(), clk.end());
		std::sort(cyc.begin(), cyc.end());
		const double clock = clk.empty() ? 0 : clk[clk.size() / 2], cycles = cyc[cyc.size() / 2];
		const double flops = (double) blocks * 4 * iters * 8 * 128.0 / (ms * 1e-3);
		// a SIMD holds wgs_per_cu waves: cycles per v_fma_f64 per SIMD = the wave's loop cycles / (its FMAs * waves per SIMD)
		printf("%d waves/SIMD (%d launches warm): %.1f TFLOP/s by events (%.3f ms); in-kernel clock %.0f MHz (median of %zu waves; min %.0f max %.0f); "
		       "%.2f shader cycles per wave-level v_fma_f64 per SIMD -> %.1f TFLOP/s at that clock if the cadence were 4\n",
		       wgs_per_cu, warm, flops * 1e-12, ms, clock * 1e-6, clk.size(), clk.empty() ? 0 : clk.front() * 1e-6, clk.empty() ? 0 : clk.back() * 1e-6,
		       cycles / ((double) iters * 8 * wgs_per_cu), (double) cus * 4 * clock / 4.0 * 128.0 * 1e-12);
	}
	return 0;
}
