#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
// hypothesis: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15], D[row = (lane >> 4) + 4 * i][col = lane & 15]
__global__ void k(const double* A, const double* B, double* D)   // A 16x4 row-major, B 4x16 row-major, D 16x16
{
	const int l = threadIdx.x;
	double a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
	double4_t c = {0, 0, 0, 0};
	c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
	for (int i = 0; i < 4; i++) D[((l >> 4) + 4 * i) * 16 + (l & 15)] = c[i];
}
int main()
{
	double hA[64], hB[64], hD[256], ref[256];
	for (int i = 0; i < 64; i++) { hA[i] = 0.5 + i * 0.37; hB[i] = 1.0 - i * 0.11; }
	for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) { double s = 0; for (int k = 0; k < 4; k++) s += hA[r * 4 + k] * hB[k * 16 + c]; ref[r * 16 + c] = s; }
	double *dA, *dB, *dD;
	hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
	hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
	hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
	double e = 0; for (int i = 0; i < 256; i++) e = fmax(e, fabs(hD[i] - ref[i]));
	printf("max abs error vs reference layout: %g\n", e);
	return 0;
}
