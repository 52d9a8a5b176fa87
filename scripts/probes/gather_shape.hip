// Probe: what does a gathered 80-byte record cost by the SHAPE of the access? ("requests, not bytes")
//   k_thread_per_record   lane l reads record idx[l] with five 16-byte loads (five wave instructions, each touching 64 lines)
//   k_lanes_per_record    8 lanes per record, lane j of the group reads the j-th 16 bytes (j < 5; one instruction per 8 records,
//                         whose lanes' addresses coalesce into the record's one or two lines)
//   k_store_*             the same two shapes for 96-byte rows written at consecutive positions
// Random records of a 640 MiB table; 2 Mi records per launch; time by HIP events, launches repeated.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_thread_per_record(const double2* __restrict__ t, const unsigned* __restrict__ idx, size_t n, double* out)
{
	double acc = 0;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
		const double2* q = t + (size_t) idx[i] * 5;
		const double2 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
		acc += a.x + b.y + c.x + d.y + e.x;
	}
	if (acc == 1.2345e300) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_lanes_per_record(const double2* __restrict__ t, const unsigned* __restrict__ idx, size_t n, double* out)
{
	double acc = 0;
	const int j = threadIdx.x & 7;
	for (size_t g = ((size_t) blockIdx.x * 256 + threadIdx.x) >> 3; g < n; g += ((size_t) gridDim.x * 256) >> 3) {
		const double2* q = t + (size_t) idx[g] * 5;
		if (j < 5) { const double2 a = q[j]; acc += a.x + a.y; }
	}
	if (acc == 1.2345e300) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_store_thread_per_row(double2* __restrict__ o, size_t n)
{
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
		double2* q = o + i * 6;
		const double2 v = make_double2((double) i, 1.0);
#pragma unroll
		for (int u = 0; u < 6; u++) q[u] = v;
	}
}

__global__ __launch_bounds__(256) void k_store_lanes_per_row(double2* __restrict__ o, size_t n)
{
	const int j = threadIdx.x & 7;
	for (size_t g = ((size_t) blockIdx.x * 256 + threadIdx.x) >> 3; g < n; g += ((size_t) gridDim.x * 256) >> 3) {
		if (j < 6) o[g * 6 + j] = make_double2((double) g, 1.0);
	}
}

template <class F>
float timeit(F f)
{
	hipEvent_t e0, e1;
	(void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
	f();
	(void) hipDeviceSynchronize();
	(void) hipEventRecord(e0, 0);
	for (int r = 0; r < 5; r++) f();
	(void) hipEventRecord(e1, 0);
	(void) hipEventSynchronize(e1);
	float ms = 0;
	(void) hipEventElapsedTime(&ms, e0, e1);
	return ms / 5;
}

int main()
{
	const size_t nrec = (size_t) 1 << 23, n = (size_t) 1 << 21;
	double2* t; unsigned* idx; double* out; double2* o;
	(void) hipMalloc(&t, nrec * 80);
	(void) hipMalloc(&idx, n * 4);
	(void) hipMalloc(&out, 64);
	(void) hipMalloc(&o, n * 96);
	(void) hipMemset(t, 0, nrec * 80);
	unsigned* h = new unsigned[n];
	for (size_t i = 0; i < n; i++) h[i] = (unsigned) ((i * 2654435761ull + 12345ull) & (nrec - 1));
	(void) hipMemcpy(idx, h, n * 4, hipMemcpyHostToDevice);
	const int blocks = 256 * 4;
	const float a = timeit([&] { hipLaunchKernelGGL(k_thread_per_record, dim3(blocks), dim3(256), 0, 0, t, idx, n, out); });
	const float b = timeit([&] { hipLaunchKernelGGL(k_lanes_per_record, dim3(blocks), dim3(256), 0, 0, t, idx, n, out); });
	const float c = timeit([&] { hipLaunchKernelGGL(k_store_thread_per_row, dim3(blocks), dim3(256), 0, 0, o, n); });
	const float d = timeit([&] { hipLaunchKernelGGL(k_store_lanes_per_row, dim3(blocks), dim3(256), 0, 0, o, n); });
	printf("%zu random 80-byte records, 4 workgroups per CU: thread per record %.3f ms (%.1f G records/s) | 8 lanes per record %.3f ms (%.1f G records/s)\n",
	       n, a, n / a * 1e-6, b, n / b * 1e-6);
	printf("%zu consecutive 96-byte rows written: thread per row %.3f ms (%.1f G rows/s) | 8 lanes per row %.3f ms (%.1f G rows/s)\n", n, c, n / c * 1e-6, d, n / d * 1e-6);
	return 0;
}
