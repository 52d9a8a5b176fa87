// Probe: what one packet between two dependent kernels costs on this device / runtime. hipcc --offload-arch=gfx950 -O2 -o hop_cost hop_cost.hip
// Every case is a loop of N iterations on stream A (and B), timed by the host over the whole loop; reported per iteration.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_small(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1e-9; }
__global__ void k_big(double* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0000001 + 1e-9; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv)
{
	const int N = 2000;
	const unsigned flags = (argc > 1) ? (unsigned) strtoul(argv[1], nullptr, 0) : hipEventDisableTiming;
	hipStream_t A, B;
	CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
	hipEvent_t e1, e2, old;
	CK(hipEventCreateWithFlags(&e1, flags)); CK(hipEventCreateWithFlags(&e2, flags)); CK(hipEventCreateWithFlags(&old, flags));
	double *p, *q;
	const int n = 1 << 24;   // 128 MB: k_big leaves the L2s dirty
	CK(hipMalloc(&p, (size_t) n * 8)); CK(hipMalloc(&q, (size_t) n * 8));
	CK(hipMemset(p, 0, (size_t) n * 8)); CK(hipMemset(q, 0, (size_t) n * 8));
	CK(hipEventRecord(old, B)); CK(hipStreamSynchronize(B));
	auto run = [&](const char* name, auto body) {
		for (int i = 0; i < 50; i++) body();
		CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
		auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < N; i++) body();
		CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
		double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
		printf("%-78s %8.2f us per iteration\n", name, us);
		return us;
	};
	auto small = [&](hipStream_t s, double* x) { hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, s, x, 256); };
	auto big = [&](hipStream_t s, double* x) { hipLaunchKernelGGL(k_big, dim3(1024), dim3(256), 0, s, x, n / 8); };   // 16 MB read + written
	printf("event flags 0x%x\n", flags);
	run("A: small", [&] { small(A, p); });
	run("A: small, small", [&] { small(A, p); small(A, p); });
	run("A: small, record(e1), small", [&] { small(A, p); hipEventRecord(e1, A); small(A, p); });
	run("A: small, wait(old event of B), small", [&] { small(A, p); hipStreamWaitEvent(A, old, 0); small(A, p); });
	run("A: small, record(e1); B: wait(e1), small, record(e2); A: wait(e2), small", [&] { small(A, p); hipEventRecord(e1, A); hipStreamWaitEvent(B, e1, 0); small(B, p); hipEventRecord(e2, B); hipStreamWaitEvent(A, e2, 0); small(A, p); });
	run("A: big", [&] { big(A, p); });
	run("A: big, small", [&] { big(A, p); small(A, p); });
	run("A: big, record(e1), small", [&] { big(A, p); hipEventRecord(e1, A); small(A, p); });
	run("A: big, wait(old event of B), small", [&] { big(A, p); hipStreamWaitEvent(A, old, 0); small(A, p); });
	run("A: big, record(e1); B: wait(e1), small", [&] { big(A, p); hipEventRecord(e1, A); hipStreamWaitEvent(B, e1, 0); small(B, q); hipEventRecord(e2, B); hipStreamWaitEvent(A, e2, 0); });
	return 0;
}
