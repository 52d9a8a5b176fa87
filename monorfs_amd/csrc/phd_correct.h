// phd_correct.h — PredictConditional + CorrectConditional (+ the MinWeight cut of PruneModel) as two kernels (one
// workgroup per particle):
//
//   k_sweep<ZB>     (phd_sweep.h)  every (component, measurement) pair once: the Explored density of
//                   PredictConditional (:793-819, births), the per-component measurement quantities (:857-870), the
//                   misdetection copies (:837-840) and the weight sums of CorrectConditional (:886-890); the pairs
//                   that can reach MinWeight are queued.
//   k_emit_finish   the queued pairs: w' = PD w q / (kappa + weightsum) (:899) in the reference's own arithmetic
//                   and, for those that reach MinWeight, the Kalman update (:895-897) m' = m + K nu,
//                   P' = (I - K H) P.
//
// The Kalman algebra stays out of the pair loop: a version with everything in one kernel needed 252 VGPRs.
#pragma once
#include "phd_device.h"


// component c of the predicted mixture = prior slab entry or a birth (mean from the explore kernel)
__device__ __forceinline__ void load_predicted(const DevParams& prm, const StepBufs& a, const MixView& vin, int p, int n, int c,
                                               double& w, double m[3], double P[6])
{
	if (c < n) load_comp(vin.rec + (in_base(a, p) + c) * MIX_REC, w, m, P);
	else {
		const double* bm = a.born_mean + ((size_t) p * a.Mcap + (c - n)) * 3;
		w = prm.birthw;
		m[0] = bm[0]; m[1] = bm[1]; m[2] = bm[2];
#pragma unroll
		for (int t = 0; t < 6; t++) P[t] = prm.birthP[t];
	}
}

// =================================================================================================
#define EMIT_LIST 512   // four per-wave lists of 128 pairs waiting for the Kalman path
#define EMIT_LDS_DOUBLES (EXPTAB_N + 256 + (EMIT_LIST + 4) / 2)   // etab, ldenom[256] | ints: list[4][128], npair
__device__ __forceinline__ void emit_finish_body(const DevParams& prm, const StepBufs& a, double* pool)
{
	double* const etab = pool;
	double* const ldenom = pool + EXPTAB_N;
	int* const list = (int*) (pool + EXPTAB_N + 256);
	int& s_npair = list[EMIT_LIST];
	const int p = a.p0 + blockIdx.x, tid = threadIdx.x;
	const int M = a.M;
	const MixView vin = bank_view(a, SEL_IN);
	const Bank bin = bank_of(a, SEL_IN);
	const int n = vin.count[p], np = n + a.born_count[p];
	const int nmis = a.emit_count[p];
	// the candidate queue: four segments, one per wave of k_sweep
	const int segcap = a.candcap >> 2;
	const int nc0 = a.cand_count[(size_t) p * 4], nc1 = a.cand_count[(size_t) p * 4 + 1], nc2 = a.cand_count[(size_t) p * 4 + 2],
	          nc3 = a.cand_count[(size_t) p * 4 + 3];
	const bool overflow = nc0 > segcap || nc1 > segcap || nc2 > segcap || nc3 > segcap;
	const int ncand = nc0 + nc1 + nc2 + nc3;
	exp_tab_init(etab, tid);
	if (tid == 0) s_npair = 0;
	__syncthreads();   // every thread has read emit_count before thread 0 rewrites it
	const PoseD pose = load_pose(bin.poses + (size_t) p * 7);
	double rq[9];
	conj_matrix(pose, rq);
	const int2*   cands = (const int2*) a.cand + (size_t) p * a.candcap;
	const double* denom = a.denom + (size_t) p * a.Mcap;
	// the pair (component c, measurement k), already inside the radius gate: weight in the reference's arithmetic,
	// and the Kalman update when it reaches MinWeight
	auto pair = [&](int c, int k) {
		PHD_REF_ARITH
		double w, m[3], P[6];
		load_predicted(prm, a, vin, p, n, c, w, m, P);
		CompMeas q;
		comp_measure(prm, pose, rq, m, P, q);
		const double n0 = a.z[k * 3] - q.zh[0], n1 = a.z[k * 3 + 1] - q.zh[1], n2 = a.z[k * 3 + 2] - q.zh[2];
		const double d2  = quad_gen(q.Sinv, n0, n1, n2);
		const double qz  = q.qmult * exp_neg(-0.5 * d2, etab);   // mc.Evaluate(z)
		const double wgt = (q.pd * w) * qz / denom[k];           // :899
		if (wgt < prm.minw) return;
		const int slot = nmis + atomicAdd(&s_npair, 1);
		if (slot >= a.ecap) return;
		double K[9], Pn[6], mn[3];
		kalman_gain(q, K);
		kalman_cov(q, K, P, Pn);
#pragma unroll
		for (int t = 0; t < 3; t++) mn[t] = m[t] + (K[t * 3] * n0 + K[t * 3 + 1] * n1 + K[t * 3 + 2] * n2);   // :896
		const size_t e = (size_t) p * a.ecap + slot;
		a.emit_w[e]   = wgt;
		a.emit_idx[e] = np + k * np + c;   // position in the reference's `corrected` list: after the np copies, z-major
		store_comp(a.emit_rec + e * MIX_REC, wgt, mn, Pn);   // the update as a component record, as the banks hold them
	};
	if (!overflow) {
		// Most queued pairs fail once the real denominator is known. Their exponent x = log(PD w q) travels with
		// them as a float32: x - log(denom) < log(MinWeight) by more than the float32 rounding settles it; the
		// others are compacted into a list so that the heavy path runs on full lanes. Every wave takes the queue segment
		// its counterpart in k_sweep wrote, with a list of its own: no workgroup barrier in the loop.
		const int wv = tid >> 6, lane = tid & 63;
		int* const wlist = list + wv * 128;   // up to 63 waiting + 64 new entries
		for (int k = tid; k < M; k += 256) ldenom[k] = log(denom[k]);
		__syncthreads();
		const double lminw = log(prm.minw);
		const int ncw = (wv == 0) ? nc0 : ((wv == 1) ? nc1 : ((wv == 2) ? nc2 : nc3));
		const int2* seg = cands + (size_t) wv * segcap;
		int nl = 0;   // wave-uniform: entries waiting in the list
		for (int j0 = 0; j0 < ncw; j0 += 64) {
			const int j = j0 + lane;
			bool keep = false;
			int  code = 0;
			if (j < ncw) {
				const int2 cd = seg[j];
				code = cd.x;
				const double x = (double) __int_as_float(cd.y);
				keep = !(x - ldenom[code & 255] < lminw - 1e-3 - 1e-6 * fabs(x));
			}
			const unsigned long long bal = ballot64(keep);
			if (keep) wlist[nl + __popcll(bal & lanemask_lt())] = code;
			nl += __popcll(bal);
			if (nl >= 64) {
				lds_fence();
				__builtin_amdgcn_wave_barrier();
				const int cd = wlist[lane];
				const int rest = (lane < nl - 64) ? wlist[64 + lane] : 0;
				pair(cd >> 8, cd & 255);
				lds_fence();
				__builtin_amdgcn_wave_barrier();
				if (lane < nl - 64) wlist[lane] = rest;
				nl -= 64;
			}
		}
		lds_fence();
		__builtin_amdgcn_wave_barrier();
		if (lane < nl) {
			const int cd = wlist[lane];
			pair(cd >> 8, cd & 255);
		}
	}
	else {
		// a segment of the queue overflowed (more than a quarter of a wave's pairs are candidates): every pair, gate included
		for (int j = tid; j < np * M; j += 256) {
			const int c = j / M, k = j - c * M;
			double w, m[3], P[6], x[3];
			load_predicted(prm, a, vin, p, n, c, w, m, P);
			const double z[3] = {a.z[k * 3], a.z[k * 3 + 1], a.z[k * 3 + 2]};
			measure_to_map(prm, pose, z, x);
			const double e0 = x[0] - m[0], e1 = x[1] - m[1], e2 = x[2] - m[2];
			if (e0 * e0 + e1 * e1 + e2 * e2 <= prm.g2_correct) pair(c, k);
		}
	}
	__syncthreads();
	if (tid == 0) {
		if (nmis + s_npair > a.ecap) atomicOr(a.flags, PHD_FLAG_EMIT_OVERFLOW);
		a.emit_count[p] = min(nmis + s_npair, a.ecap);
	}
}

#ifndef PHD_EF_WAVES
#define PHD_EF_WAVES 3   // waves per SIMD the register allocation of k_emit_finish aims at (tuning: scripts/ab_variants.sh)
#endif
__global__ __launch_bounds__(256, PHD_EF_WAVES) void k_emit_finish(const DevParams prm, const StepBufs a)
{
	__shared__ __align__(16) double pool[EMIT_LDS_DOUBLES];
	emit_finish_body(prm, a, pool);
}
