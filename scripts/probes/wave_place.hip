// Probe: where the waves of a 256-thread workgroup land. hipcc --offload-arch=gfx950 -O2 -o bin/wave_place wave_place.hip
// Every wave records HW_ID (wave slot, SIMD, CU, SH, SE) and XCC_ID at its start; the kernel holds 128 registers and 35 KB of
// LDS per workgroup like k_emit_prune (four workgroups per CU) and spins ~20 us so that a CU's four slots are taken together.
// Questions: is wave w of every workgroup on SIMD w? do co-resident workgroups hold distinct wave slots on a SIMD, and are those
// slots 0..3? (phd_correct.h: which wave takes the odd pass through the Kalman path)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 4) void k_probe(unsigned int* out, long long* t, int spin)
{
	__shared__ double pad[4400];   // 35 KB
	const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
	const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, all 32 bits
	const unsigned int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID[3:0]
	const long long t0 = wall_clock64();
	pad[tid] = (double) tid;
	__syncthreads();
	double acc = pad[(tid * 7) & 255];
	// 128 registers: a block of values that stays live
	double v[56];
#pragma unroll
	for (int i = 0; i < 56; i++) v[i] = acc + i;
	while (wall_clock64() - t0 < spin) {
#pragma unroll
		for (int i = 0; i < 56; i++) v[i] = v[i] * 1.0000001 + 1e-9;
	}
#pragma unroll
	for (int i = 0; i < 56; i++) acc += v[i];
	if (lane == 0) {
		out[(blockIdx.x * 4 + wv) * 2] = hw;
		out[(blockIdx.x * 4 + wv) * 2 + 1] = xcc;
		t[blockIdx.x * 4 + wv] = t0;
	}
	if (acc == 12345.678) out[0] = 0;
}

int main()
{
	const int G = 2048;
	unsigned int* d; long long* dt;
	CK(hipMalloc(&d, G * 4 * 2 * 4)); CK(hipMalloc(&dt, G * 4 * 8));
	hipLaunchKernelGGL(k_probe, dim3(G), dim3(256), 0, 0, d, dt, 2000);   // 20 us of the 100 MHz counter
	CK(hipDeviceSynchronize());
	std::vector<unsigned int> h(G * 8); std::vector<long long> ht(G * 4);
	CK(hipMemcpy(h.data(), d, G * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht.data(), dt, G * 32, hipMemcpyDeviceToHost));
	long long tmin = ht[0];
	for (auto x : ht) tmin = x < tmin ? x : tmin;
	int simd_is_wave = 0, perm = 0, same_slot = 0;
	std::map<int, int> slot_hist, rot_hist;
	for (int b = 0; b < G; b++) {
		int mask = 0, ok = 1, slot0 = -1, sl = 1;
		for (int w = 0; w < 4; w++) {
			const unsigned int hw = h[(b * 4 + w) * 2];
			const int slot = hw & 15, simd = (hw >> 4) & 3;
			mask |= 1 << simd;
			ok &= simd == w;
			if (w == 0) slot0 = slot; else sl &= slot == slot0;
			slot_hist[slot]++;
		}
		simd_is_wave += ok; perm += mask == 15; same_slot += sl;
	}
	printf("workgroups %d: wave w on SIMD w in %d, four distinct SIMDs in %d, the same slot on all four SIMDs in %d\n", G, simd_is_wave, perm, same_slot);
	printf("wave slots used:");
	for (auto& kv : slot_hist) printf(" %d: %d", kv.first, kv.second);
	printf("\n");
	// co-resident workgroups of the first round (started within 5 us of the first): per CU, the slots of their SIMD-0 waves
	std::map<long long, std::vector<int>> cu;
	for (int b = 0; b < G; b++) {
		if (ht[b * 4] - tmin > 500) continue;
		for (int w = 0; w < 4; w++) {
			const unsigned int hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1];
			if (((hw >> 4) & 3) != 0) continue;
			const long long key = ((long long) (xcc & 15) << 16) | ((hw >> 8) & 0xff);   // XCC, SE / SH / CU
			cu[key].push_back(((hw & 15) << 16) | b);
		}
	}
	int ncu = 0, distinct_mod4 = 0, four = 0;
	for (auto& kv : cu) {
		ncu++;
		int m = 0;
		for (int x : kv.second) m |= 1 << ((x >> 16) & 3);
		four += kv.second.size() == 4;
		distinct_mod4 += (int) kv.second.size() == __builtin_popcount(m);
		if (ncu <= 6) {
			printf("  xcc %lld cu 0x%02llx:", kv.first >> 16, kv.first & 0xff);
			for (int x : kv.second) printf(" wg %d slot %d", x & 0xffff, x >> 16);
			printf("\n");
		}
	}
	printf("first round: %d CUs, %d with four workgroups, %d whose SIMD-0 slots are distinct mod 4\n", ncu, four, distinct_mod4);
	// first 16 workgroups: raw fields
	for (int b = 0; b < 12; b++) {
		printf("wg %d:", b);
		for (int w = 0; w < 4; w++) {
			const unsigned int hw = h[(b * 4 + w) * 2];
			printf("  [slot %u simd %u cu 0x%02x xcc %u]", hw & 15, (hw >> 4) & 3, (hw >> 8) & 0xff, h[(b * 4 + w) * 2 + 1] & 15);
		}
		printf("\n");
	}
	return 0;
}
