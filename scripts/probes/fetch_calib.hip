// Calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ[_32B] on gfx950 for the access patterns of this library (the guide:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern"). Four kernels of known
// byte counts over a 1 GiB table (far beyond the 256 MiB Infinity Cache), each launched 4 times:
//   k_stream8    every thread reads consecutive doubles                  (8 B per lane: the old plane layout)
//   k_stream16   every thread reads consecutive double2                  (16 B per lane)
//   k_rec80      every thread reads its own 80-byte record, five 16-byte loads, records consecutive across the lanes
//   k_gather80   every thread reads ONE 80-byte record at a pseudo-random index (two 64-byte sectors, one or two 128-byte lines)
// Run under  rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace  and compare with the bytes printed.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_stream8(const double* __restrict__ t, size_t n, double* out)
{
	double acc = 0;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) acc += t[i];
	if (acc == 1.2345e300) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_stream16(const double2* __restrict__ t, size_t n, double* out)
{
	double acc = 0;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) { const double2 v = t[i]; acc += v.x + v.y; }
	if (acc == 1.2345e300) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_rec80(const double2* __restrict__ t, size_t nrec, double* out)
{
	double acc = 0;
	for (size_t r = (size_t) blockIdx.x * 256 + threadIdx.x; r < nrec; r += (size_t) gridDim.x * 256) {
		const double2* q = t + r * 5;
		const double2 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
		acc += a.x + b.y + c.x + d.y + e.x;
	}
	if (acc == 1.2345e300) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_gather80(const double2* __restrict__ t, size_t nrec, size_t reads, double* out)
{
	double acc = 0;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < reads; i += (size_t) gridDim.x * 256) {
		// a fixed pseudo-random permutation-like map (odd multiplier modulo a power of two): every record at most once
		const size_t r = (i * 2654435761ull + 12345ull) & (nrec - 1);
		const double2* q = t + r * 5;
		const double2 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
		acc += a.x + b.y + c.x + d.y + e.x;
	}
	if (acc == 1.2345e300) out[0] = acc;
}

int main()
{
	const size_t bytes = (size_t) 1 << 30;
	double* t;
	double* out;
	hipMalloc(&t, bytes + 1024);
	hipMalloc(&out, 64);
	hipMemset(t, 0, bytes);
	const size_t nrec_pow2 = (size_t) 1 << 23;          // 8 Mi records of 80 B = 640 MiB
	const size_t reads = (size_t) 1 << 21;              // 2 Mi of them: 160 MiB of records
	const int blocks = 256 * 8;
	for (int rep = 0; rep < 4; rep++) {
		hipLaunchKernelGGL(k_stream8, dim3(blocks), dim3(256), 0, 0, t, bytes / 8, out);
		hipLaunchKernelGGL(k_stream16, dim3(blocks), dim3(256), 0, 0, (const double2*) t, bytes / 16, out);
		hipLaunchKernelGGL(k_rec80, dim3(blocks), dim3(256), 0, 0, (const double2*) t, bytes / 80, out);
		hipLaunchKernelGGL(k_gather80, dim3(blocks), dim3(256), 0, 0, (const double2*) t, nrec_pow2, reads, out);
		hipDeviceSynchronize();
	}
	printf("k_stream8 %zu bytes | k_stream16 %zu bytes | k_rec80 %zu bytes | k_gather80 %zu records: %zu bytes of records, %zu bytes of 64-byte sectors (2 each), "
	       "%zu bytes of 128-byte lines (1.625 each on average: 5 of the 8 offsets straddle)\n",
	       bytes, bytes, (bytes / 80) * 80, reads, reads * 80, reads * 128, (size_t) (reads * 128 * 1.625));
	return 0;
}
