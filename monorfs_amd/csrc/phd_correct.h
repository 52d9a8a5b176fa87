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
// k_emit_finish (round 5): the queued pairs, one COMPONENT per lane.
// Everything dear in the Kalman update depends on the component alone — h(m), H, S^-1, the multiplier, the detection
// probability, and with them the gain K = P H^T S^-1 and the updated covariance P' = (I - K H) P (PHDNavigator.cs:857-870,
// :895, :897): ~950 dependent FP64 instructions in the reference's arithmetic. Only the innovation, the weight and the updated
// mean (:896, :899; ~70 instructions) depend on the measurement. The queue of k_sweep holds the candidates of one component —
// one visit of its pair loop — next to each other, so a lane takes a RUN of queue entries with one component: the dear part
// once, the cheap part per entry. And the lanes are pooled over the workgroup: every wave scans its own segment of the queue
// (float32 pre-filter on the exponent, as before) and appends what it keeps to ONE list, the runs' heads to a second one; the
// heads are then worked off 256 at a time. Before (a lane per PAIR, a list per wave): 4 waves x 3 batches of ~1000
// instructions for ~570 pairs in ~400 runs on config B, the third batch a fifth full, all four workgroups of a CU in this issue-
// bound phase together (23 cycles per instruction); now ~7 wave-batches of ~1150.
// The values are the same bits (the same functions of the same inputs); the emitted list comes out in another order,
// which k_prune_merge does not see (it orders by weight and canonical index).
#define EMIT_CAP 2048   // queue entries scanned per chunk (512 per wave); what is kept of them fits the two lists
#define EMIT_RUN 4      // measurements a lane takes with one component (a longer run is cut: its wave would wait for it)
#define EMIT_LDS_DOUBLES (EXPTAB_N + 5 * 256 + EMIT_CAP + 4)   // etab, ldenom[256], denom[256], z[256][3] | ints: list[EMIT_CAP], lead[EMIT_CAP], npair, nlist, nlead
// LEAN (the fused launch k_emit_prune, compiled at the prune's 128 registers): the component's mean and covariance are read AGAIN
// behind the gain instead of living through the 3 x 3 inverse; where the registers are not short (k_emit_finish at three waves
// per SIMD, the one-launch chain) the second read only adds a trip to the chain.
// SPREAD (the one-launch chain: one workgroup per CU, one wave per SIMD): the runs are dealt to the four waves in turn, so that all
// four SIMDs work on a short list; on a full machine a wave takes 64 neighbouring runs (fewest wave-passes through the Kalman path:
// the SIMDs are shared with three other workgroups in the same phase).
template <bool LEAN = false, bool SPREAD = false>
__device__ __forceinline__ void emit_finish_body(const DevParams& prm, const StepBufs& a, double* pool)
{
	double* const etab = pool;
	double* const ldenom = pool + EXPTAB_N;
	double* const sden = ldenom + 256;            // kappa + weightsum[z], and the measurements: what the run loop reads per entry
	double* const sz = sden + 256;
	int* const list = (int*) (pool + EXPTAB_N + 5 * 256);
	int* const lead = list + EMIT_CAP;
	int& s_npair = lead[EMIT_CAP];
	int& s_nlist = lead[EMIT_CAP + 1];
	int& s_nlead = lead[EMIT_CAP + 2];
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
#ifdef PHD_STAMPS   // diagnostic build, PHD_STAMP_KERNEL=7: wave 0's cycles in the Kalman path, its rounds, the runs and the queue length
	const long long em_t0 = clock64();
	long long em_pair = 0, em_nb = 0, em_runs = 0, em_setup = 0, em_scan = 0;
#endif
	exp_tab_init(etab, tid);
	if (tid == 0) s_npair = 0;
	const double* denom = a.denom + (size_t) p * a.Mcap;
	for (int k = tid; k < M; k += 256) {
		const double dk = denom[k];
		sden[k] = dk; ldenom[k] = log(dk);
		sz[k * 3] = a.z[k * 3]; sz[k * 3 + 1] = a.z[k * 3 + 1]; sz[k * 3 + 2] = a.z[k * 3 + 2];
	}
	__syncthreads();   // every thread has read emit_count before thread 0 rewrites it
	PoseD pose = load_pose(bin.poses + (size_t) p * 7);   // (the same in every lane: scalar registers)
	pose.t[0] = uniform_d(pose.t[0]); pose.t[1] = uniform_d(pose.t[1]); pose.t[2] = uniform_d(pose.t[2]);
	pose.qw = uniform_d(pose.qw); pose.qx = uniform_d(pose.qx); pose.qy = uniform_d(pose.qy); pose.qz = uniform_d(pose.qz);
	const int2*   cands = (const int2*) a.cand + (size_t) p * a.candcap;
	// Component c and its run of measurements: list[start .. start + len) (kf >= 0: the one measurement kf instead — the
	// fallback below). comp_measure (phd_device.h) written out in its own order of operations, so that what is dead can go early:
	// the detection probability as soon as h(m) is known.
	auto component_run = [&](int c, const int* run, int len, int kf) {
		PHD_REF_ARITH
		double w, m[3], P[6];
		load_predicted(prm, a, vin, p, n, c, w, m, P);
		double zh[3], H[9], PH[9], Sinv[9], qmult, pdw;
		{
			double l[3];
			measure_perfect(prm, pose, m, zh, l);
			pdw = detection_probability_m(prm, zh) * w;
			// (the rotation matrix is made HERE, per run, from the quaternion in scalar registers: ~20 instructions against nine
			// doubles kept — spilled, in the fused launch — across everything else; the empty asm keeps the compiler from hoisting it)
			PoseD pq = pose;
			asm volatile("" : "+s"(pq.qw), "+s"(pq.qx), "+s"(pq.qy), "+s"(pq.qz));
			double rq[9];
			conj_matrix(pq, rq);
			jacobian_l(prm, l, rq, H);
		}
		{
			const double Pf[9] = {P[0], P[1], P[2], P[1], P[3], P[4], P[2], P[4], P[5]};
#pragma unroll
			for (int i = 0; i < 3; i++) {
#pragma unroll
				for (int j = 0; j < 3; j++) {
					double sacc = 0;
#pragma unroll
					for (int e = 0; e < 3; e++) sacc += Pf[i * 3 + e] * H[j * 3 + e];
					PH[i * 3 + j] = sacc;
				}
			}
		}
		{
			double S[9], det;
#pragma unroll
			for (int i = 0; i < 3; i++) {
#pragma unroll
				for (int j = 0; j < 3; j++) {
					double sacc = 0;
#pragma unroll
					for (int e = 0; e < 3; e++) sacc += H[i * 3 + e] * PH[e * 3 + j];
					S[i * 3 + j] = sacc + prm.R[i * 3 + j];
				}
			}
			inv_gen3(S, Sinv, det);
			qmult = PHD_INV_2PI / sqrt(fabs(det));
		}
		// K = P H^T S^-1 (:895), P' = (I - K H) P (:897): the component's, whatever the measurement
		double K[9], Pn[6];
#pragma unroll
		for (int i = 0; i < 3; i++) {
#pragma unroll
			for (int j = 0; j < 3; j++) {
				double sacc = 0;
#pragma unroll
				for (int e = 0; e < 3; e++) sacc += PH[i * 3 + e] * Sinv[e * 3 + j];
				K[i * 3 + j] = sacc;
			}
		}
		{
			double IKH[9];
#pragma unroll
			for (int i = 0; i < 3; i++) {
#pragma unroll
				for (int j = 0; j < 3; j++) {
					double sacc = 0;
#pragma unroll
					for (int e = 0; e < 3; e++) sacc += K[i * 3 + e] * H[e * 3 + j];
					IKH[i * 3 + j] = ((i == j) ? 1.0 : 0.0) - sacc;
				}
			}
			if (LEAN) {   // (the component's record once more — two lines the lane has in its cache — behind everything that needed H)
				asm volatile("" ::: "memory");
				load_predicted(prm, a, vin, p, n, c, w, m, P);
			}
			const double Pf[9] = {P[0], P[1], P[2], P[1], P[3], P[4], P[2], P[4], P[5]};
			const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
			for (int t = 0; t < 6; t++) {
				double sacc = 0;
#pragma unroll
				for (int e = 0; e < 3; e++) sacc += IKH[ia[t] * 3 + e] * Pf[e * 3 + ib[t]];
				Pn[t] = sacc;
			}
		}
		for (int r = 0; r < len; r++) {
			const int k = (kf >= 0) ? kf : (run[r] & 255);
			const double n0 = sz[k * 3] - zh[0], n1 = sz[k * 3 + 1] - zh[1], n2 = sz[k * 3 + 2] - zh[2];
			const double d2  = quad_gen(Sinv, n0, n1, n2);
			const double qz  = qmult * exp_neg(-0.5 * d2, etab);   // mc.Evaluate(z)
			const double wgt = pdw * qz / sden[k];                 // :899
			if (wgt < prm.minw) continue;
			const int slot = nmis + atomicAdd(&s_npair, 1);
			if (slot >= a.ecap) continue;
			double mn[3];
#pragma unroll
			for (int t = 0; t < 3; t++) mn[t] = m[t] + (K[t * 3] * n0 + K[t * 3 + 1] * n1 + K[t * 3 + 2] * n2);   // :896
			const size_t e = (size_t) p * a.ecap + slot;
			a.emit_w[e]   = wgt;
			a.emit_idx[e] = np + k * np + c;   // position in the reference's `corrected` list: after the np copies, z-major
			store_comp(a.emit_rec + e * MIX_REC, wgt, mn, Pn);   // the update as a component record, as the banks hold them
		}
	};
	if (!overflow) {
		// Most queued pairs fail once the real denominator is known. Their exponent x = log(PD w q) travels with them as a
		// float32: x - log(denom) < log(MinWeight) by more than the float32 rounding settles it; the others go to the list.
		const int wv = tid >> 6, lane = tid & 63;
		const double lminw = log(prm.minw);
		const int ncw = (wv == 0) ? nc0 : ((wv == 1) ? nc1 : ((wv == 2) ? nc2 : nc3));
		const int ncmax = max(max(nc0, nc1), max(nc2, nc3));
		const int2* seg = cands + (size_t) wv * segcap;
		// (measured and dropped: a quarter of the lists per wave with scalar counters — no LDS atomic, no shuffle of the bases in
		// the scan —: the scan 22.6 k -> 21.3 k cycles of the body's 80 k, 48 bytes of scratch instead of 12; eight groups of the
		// queue requested ahead instead of one: the body 86 k -> 112 k cycles in the stamped build — where every guarded load was
		// followed by its own s_waitcnt: the compiler had sunk the float -> double conversion into the guard —; with unconditional
		// loads, all eight in one trip, the product build: config B 0.608 -> 0.610 ms per step, S 3.45 -> 3.40. The scan is not
		// waiting for the queue: its ~200 instructions per group issue at the pace the three other workgroups' Kalman paths leave.)
#ifdef PHD_STAMPS
		em_setup = clock64() - em_t0;
#endif
		for (int ch0 = 0; ch0 < ncmax; ch0 += EMIT_CAP / 4) {   // (workgroup-uniform; one trip unless a wave queued more than 512 pairs)
#ifdef PHD_STAMPS
			const long long em_s0 = clock64();
#endif
			if (tid == 0) { s_nlist = 0; s_nlead = 0; }
			__syncthreads();   // (also: ldenom is written)
			const int cend = min(ncw, ch0 + EMIT_CAP / 4);
			int2 nxt = (ch0 + lane < cend) ? seg[ch0 + lane] : make_int2(0, 0);   // (the queue is read one group of 64 ahead)
			for (int j0 = ch0; j0 < cend; j0 += 64) {
				const int j = j0 + lane;
				bool keep = false;
				int  code = 0;
				const int2 cd = nxt;
				nxt = (j + 64 < cend) ? seg[j + 64] : make_int2(0, 0);
				if (j < cend) {
					code = cd.x;
					const double x = (double) __int_as_float(cd.y);
					keep = !(x - ldenom[code & 255] < lminw - 1e-3 - 1e-6 * fabs(x));
				}
				const unsigned long long bal = ballot64(keep);
				if (bal) {   // (wave-uniform)
					// a run's head: the first kept lane of the group, or a kept lane whose component is not that of the kept lane before it
					const unsigned long long below = bal & lanemask_lt();
					const int pl = below ? 63 - __clzll((long long) below) : lane;
					const int pcode = __shfl(code, pl, 64);
					const bool head0 = keep && (!below || (pcode >> 8) != (code >> 8));
					// ... or the EMIT_RUN-th kept lane behind such a head: a long run is cut, the component's part computed again
					// (a wave works its runs off in the time of its longest one)
					const unsigned long long hb0 = ballot64(head0);
					const unsigned long long upme = hb0 & (lanemask_lt() | (1ull << lane));
					const int hl = upme ? 63 - __clzll((long long) upme) : 0;          // the run's first lane
					const int inrun = __popcll(below & ~((1ull << hl) - 1ull));       // kept lanes of the run before this one
					const bool head = keep && (head0 || (inrun % EMIT_RUN) == 0);
					const unsigned long long hb = ballot64(head);
					const int first = __ffsll((long long) bal) - 1;
					int base = 0, hbase = 0;
					if (lane == first) { base = atomicAdd(&s_nlist, __popcll(bal)); hbase = atomicAdd(&s_nlead, __popcll(hb)); }
					base = __shfl(base, first, 64); hbase = __shfl(hbase, first, 64);
					const int pos = base + __popcll(below);
					if (keep) list[pos] = code;
					if (head) {
						const unsigned long long above = (lane < 63) ? (hb >> (lane + 1)) << (lane + 1) : 0ull;   // heads behind this lane
						const unsigned long long upto = above ? ((1ull << (__ffsll((long long) above) - 1)) - 1ull) : ~0ull;   // lanes before the next head
						const int len = __popcll(bal & upto & ~below);
						lead[hbase + __popcll(hb & lanemask_lt())] = (pos << 8) | len;   // (len <= 64, pos < EMIT_CAP)
					}
				}
			}
			__syncthreads();
			const int nlead = s_nlead;
#ifdef PHD_STAMPS
			const long long em_p0 = clock64();
			em_scan += em_p0 - em_s0;
#endif
			for (int t0 = SPREAD ? 0 : wv * 64; t0 < nlead; t0 += 256) {   // (wave-uniform: a wave without a head left skips the Kalman path)
				const int t = SPREAD ? t0 + lane * 4 + wv : t0 + lane;
				if (t < nlead) {
					const int hd = lead[t];
					const int* run = list + (hd >> 8);
					component_run(run[0] >> 8, run, hd & 255, -1);
				}
#ifdef PHD_STAMPS
				em_nb++;
#endif
			}
#ifdef PHD_STAMPS
			em_pair += clock64() - em_p0; em_runs += nlead;
#endif
			__syncthreads();   // (the lists are rewritten by the next chunk)
		}
#ifdef PHD_STAMPS
		if (tid == 0 && a.stamps && a.stamp_kernel == 7) {
			double* o = a.stamps + (size_t) p * 16;
			o[0] = 0; o[1] = (double) (clock64() - em_t0); o[2] = (double) em_pair; o[3] = (double) em_nb; o[4] = (double) em_runs; o[5] = (double) (nc0 + nc1 + nc2 + nc3);
			o[6] = (double) em_setup; o[7] = (double) em_scan;
		}
#endif
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
			if (e0 * e0 + e1 * e1 + e2 * e2 <= prm.g2_correct) component_run(c, nullptr, 1, k);
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
