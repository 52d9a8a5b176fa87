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

#define EMIT_LIST 1024 // pairs k_emit_finish gathers before it runs the Kalman path on them

// component c of the predicted mixture = prior slab entry or a birth (mean from the explore kernel)
__device__ __forceinline__ void load_predicted(const DevParams& prm, const StepBufs& a, const MixView& vin, int p, int n, int c,
                                               double& w, double m[3], double P[6])
{
	if (c < n) {
		const size_t i = in_base(a, p) + c;
		w = vin.w[i];
#pragma unroll
		for (int t = 0; t < 3; t++) m[t] = vin.m[t][i];
#pragma unroll
		for (int t = 0; t < 6; t++) P[t] = vin.P[t][i];
	}
	else {
		const double* bm = a.born_mean + ((size_t) p * a.Mcap + (c - n)) * 3;
		w = prm.birthw;
		m[0] = bm[0]; m[1] = bm[1]; m[2] = bm[2];
#pragma unroll
		for (int t = 0; t < 6; t++) P[t] = prm.birthP[t];
	}
}

// =================================================================================================
#define EMIT_LDS_DOUBLES (EXPTAB_N + 256 + (EMIT_LIST + 4) / 2)   // etab, ldenom[256] | ints: list[EMIT_LIST], npair, nlist
__device__ __forceinline__ void emit_finish_body(const DevParams& prm, const StepBufs& a, double* pool)
{
	double* const etab = pool;
	double* const ldenom = pool + EXPTAB_N;
	int* const list = (int*) (pool + EXPTAB_N + 256);
	int& s_npair = list[EMIT_LIST];
	int& s_nlist = list[EMIT_LIST + 1];
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
		double* r = a.emit_rec + e * 9;
		r[0] = mn[0]; r[1] = mn[1]; r[2] = mn[2];
#pragma unroll
		for (int t = 0; t < 6; t++) r[3 + t] = Pn[t];
	};
	if (!overflow) {
		// Most queued pairs fail once the real denominator is known. Their exponent x = log(PD w q) travels with
		// them as a float32: x - log(denom) < log(MinWeight) by more than the float32 rounding settles it; the
		// others are compacted into a list so that the waves run the heavy path on full lanes.
		for (int k = tid; k < M; k += 256) ldenom[k] = log(denom[k]);
		if (tid == 0) s_nlist = 0;
		__syncthreads();
		const double lminw = log(prm.minw);
		for (int j0 = 0; j0 < ncand; j0 += 256) {
			const int j = j0 + tid;
			bool keep = false;
			int  code = 0;
			if (j < ncand) {
				// entry j of the four segments laid end to end
				const int js = (j < nc0) ? j : ((j < nc0 + nc1) ? j - nc0 + segcap : ((j < nc0 + nc1 + nc2) ? j - nc0 - nc1 + 2 * segcap : j - nc0 - nc1 - nc2 + 3 * segcap));
				const int2 cd = cands[js];
				code = cd.x;
				const double x = (double) __int_as_float(cd.y);
				keep = !(x - ldenom[code & 255] < lminw - 1e-3 - 1e-6 * fabs(x));
			}
			// (at most 256 new entries per trip: the list is drained whenever fewer than that are free)
			if (keep) list[atomicAdd(&s_nlist, 1)] = code;
			__syncthreads();
			if (s_nlist > EMIT_LIST - 256 || j0 + 256 >= ncand) {
				const int nl = s_nlist;
				for (int i = tid; i < nl; i += 256) pair(list[i] >> 8, list[i] & 255);
				__syncthreads();
				if (tid == 0) s_nlist = 0;
				__syncthreads();
			}
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
