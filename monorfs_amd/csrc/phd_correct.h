// phd_correct.h — PredictConditional + CorrectConditional (+ the MinWeight cut of PruneModel) as four
// kernels with small live state each (one workgroup per particle):
//
//   k_explore<ZB>   PredictConditional (PHDNavigator.cs:793-819): Explored(model, MeasureToMap(z)) for every
//                   measurement against the PRIOR map; unexplored measurements become births.
//   k_measure       per component of the predicted mixture (prior + births): h(m), H, P H^T, S^-1, PD
//                   (:857-870) written as 18 planes of per-component quantities, and the misdetection copies
//                   w (1 - PD) (:837-840) that survive MinWeight.
//   k_correct<ZB>   every (component, measurement) pair, measurement per lane, the component broadcast from an
//                   LDS tile: sweep 0 accumulates weightsum[z] over the near components (:886-890), sweep 1
//                   finds the pairs whose w' = PD w q / (kappa + weightsum) reaches MinWeight (:899) and
//                   queues them.
//   k_emit_finish   Kalman update of the queued pairs (:895-897): m' = m + K nu, P' = (I - K H) P.
//
// Splitting keeps the hot pair loops free of the register-hungry per-component algebra (the fused version
// needed 252 VGPRs and spilled ~100 SGPRs); the price is the 18-plane scratch, written once and read twice.
#pragma once
#include "phd_device.h"

#define CM_PLANES 18   // zh[3], Sinv[9], qmult, pdw, m[3], dcut

// component c of the predicted mixture = prior slab entry or a birth (mean from the explore kernel)
__device__ __forceinline__ void load_predicted(const DevParams& prm, const StepBufs& a, const MixView& vin, int p, int n, int c,
                                               double& w, double m[3], double P[6])
{
	if (c < n) {
		const size_t i = (size_t) p * a.cap + c;
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
template <int ZB>
__global__ __launch_bounds__(256) void k_explore(const DevParams prm, const StepBufs a)
{
	constexpr int MP = ZB * 64;
	__shared__ double zmap[3 * MP];        // MeasureToMap(z)
	__shared__ double part[4 * MP];        // per-wave partial densities
	__shared__ double tile[TILE * 10];     // [TILE][10]: gauss_record
	__shared__ double etab[EXPTAB_N];
	__shared__ int    born[MP];

	const int p = a.p0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int M = a.M;
	const MixView vin = bank_view(a, SEL_IN);
	const Bank& bin  = a.bank[a.sel[SEL_IN]];
	const Bank& bout = a.bank[a.sel[SEL_OUT]];
	const int n = vin.count[p];
	const size_t sb = (size_t) p * a.cap;
	const PoseD pose = load_pose(bin.poses + (size_t) p * 7);
	// the particle keeps its pose and (until the reweight kernel runs) its weight in the output bank
	if (tid < 7) bout.poses[(size_t) p * 7 + tid] = bin.poses[(size_t) p * 7 + tid];
	if (tid == 7) bout.weights[p] = bin.weights[p];
	exp_tab_init(etab, tid);
	for (int k = tid; k < MP; k += 256) {
		double z[3] = {0, 0, 1}, x[3] = {0, 0, 0};
		if (k < M) {
			z[0] = a.z[k * 3]; z[1] = a.z[k * 3 + 1]; z[2] = a.z[k * 3 + 2];
			measure_to_map(prm, pose, z, x);
		}
		zmap[k * 3] = x[0]; zmap[k * 3 + 1] = x[1]; zmap[k * 3 + 2] = x[2];
	}
	__syncthreads();
	double wx[ZB], wy[ZB], wz[ZB], acc[ZB];
	bool   zv[ZB];
#pragma unroll
	for (int b = 0; b < ZB; b++) {
		int k = b * 64 + lane;
		zv[b] = k < M;
		wx[b] = zmap[k * 3]; wy[b] = zmap[k * 3 + 1]; wz[b] = zmap[k * 3 + 2];
		acc[b] = 0;
	}
	const double g2 = prm.g2_explore, thr = prm.expl_thr;

	bool wavedone = false;
	for (int c0 = 0; c0 < n; c0 += TILE) {
		int c = c0 + tid;
		if (c < n) {
			double P[6], Pi[6], det;
#pragma unroll
			for (int t = 0; t < 6; t++) P[t] = vin.P[t][sb + c];
			inv_sym3(P, Pi, det);
			const double m[3] = {vin.m[0][sb + c], vin.m[1][sb + c], vin.m[2][sb + c]};
			gauss_record(vin.w[sb + c], m, Pi, PHD_INV_2PI / sqrt(fabs(det)), tile + tid * 10);
		}
		__syncthreads();
		const int cend = min(TILE, n - c0);
		// w * N(x; m, P) of component cc at this lane's measurements, inside the radius gate (Map.cs:214-217)
		auto visit = [&](int cc) {
			const double* tt = tile + cc * 10;
#pragma unroll
			for (int b = 0; b < ZB; b++) {
				double d0 = wx[b] - tt[0], d1 = wy[b] - tt[1], d2 = wz[b] - tt[2];
				double sq = d0 * d0 + d1 * d1 + d2 * d2;
				double v  = exp_neg(gauss_logw(tt, d0, d1, d2), etab);
				if (zv[b] && sq <= g2) acc[b] += v;
			}
		};
		for (int cc = wv; cc < cend && !wavedone; cc += 8) {
			visit(cc);
			if (cc + 4 < cend) visit(cc + 4);
			// every term is >= 0: once this wave's partial sum of a measurement reaches the threshold the full
			// sum does too, so a wave whose measurements are all explored can stop (a NaN keeps it going)
			bool open = false;
#pragma unroll
			for (int b = 0; b < ZB; b++) open |= zv[b] && !(acc[b] >= thr);
			wavedone = __ballot(open) == 0;
		}
		if (__syncthreads_and(wavedone)) break;
	}
#pragma unroll
	for (int b = 0; b < ZB; b++) part[wv * MP + b * 64 + lane] = acc[b];
	__syncthreads();
	for (int k = tid; k < MP; k += 256) {
		double dens = part[k] + part[MP + k] + part[2 * MP + k] + part[3 * MP + k];
		born[k] = (k < M) && !(dens >= thr);   // !Explored (:808, :958)
	}
	__syncthreads();
	if (tid == 0) {   // births keep measurement order (:814-816)
		int nb = 0;
		for (int k = 0; k < M; k++) {
			if (born[k]) {
				a.born_k[(size_t) p * a.Mcap + nb] = k;
				a.born_mean[((size_t) p * a.Mcap + nb) * 3]     = zmap[k * 3];
				a.born_mean[((size_t) p * a.Mcap + nb) * 3 + 1] = zmap[k * 3 + 1];
				a.born_mean[((size_t) p * a.Mcap + nb) * 3 + 2] = zmap[k * 3 + 2];
				nb++;
			}
		}
		a.born_count[p] = nb;
	}
}

// =================================================================================================
__global__ __launch_bounds__(256) void k_measure(const DevParams prm, const StepBufs a)
{
	__shared__ int s_cnt;
	const int p = a.p0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63;
	const MixView vin = bank_view(a, SEL_IN);
	const Bank& bin = a.bank[a.sel[SEL_IN]];
	const int n = vin.count[p], np = n + a.born_count[p];
	const PoseD pose = load_pose(bin.poses + (size_t) p * 7);
	double rq[9];
	conj_matrix(pose, rq);
	if (tid == 0) s_cnt = 0;
	__syncthreads();
	const size_t cstride = a.cmplane;                       // doubles per plane
	double* cm = a.cm + (size_t) p * a.cmcap;
	for (int c0 = 0; c0 < np; c0 += 256) {
		const int  c = c0 + tid;
		const bool valid = c < np;
		double w = 0, m[3] = {0, 0, 0}, P[6] = {1, 0, 0, 1, 0, 1}, wm = 0;
		bool mis = false;
		if (valid) {
			load_predicted(prm, a, vin, p, n, c, w, m, P);
			CompMeas q;
			comp_measure(prm, pose, rq, m, P, q);
			const double pdw = q.pd * w;
#pragma unroll
			for (int t = 0; t < 3; t++) cm[(size_t) t * cstride + c] = q.zh[t];
#pragma unroll
			for (int t = 0; t < 9; t++) cm[(size_t) (3 + t) * cstride + c] = q.Sinv[t];
			cm[(size_t) 12 * cstride + c] = q.qmult;
			cm[(size_t) 13 * cstride + c] = pdw;
#pragma unroll
			for (int t = 0; t < 3; t++) cm[(size_t) (14 + t) * cstride + c] = m[t];
			// no pair can reach MinWeight unless  PD w mult exp(-d2/2) >= MinWeight * kappa
			double dc = 2.0 * (log(pdw * q.qmult) - prm.emit_log_floor) + 1.0;
			cm[(size_t) 17 * cstride + c] = isinf(prm.emit_log_floor) ? INFINITY : dc;
			wm  = (1 - q.pd) * w;        // component.Reweight((1 - PD) w), :838-839
			mis = !(wm < prm.minw);
		}
		unsigned long long bal = __ballot(mis);
		if (bal) {
			int base = 0, first = __ffsll((long long) bal) - 1;
			if (lane == first) base = atomicAdd(&s_cnt, __popcll(bal));
			base = __shfl(base, first, 64);
			if (mis) {
				int slot = base + __popcll(bal & lanemask_lt());
				if (slot < a.ecap) {
					size_t e = (size_t) p * a.ecap + slot;
					a.emit_w[e]   = wm;
					a.emit_idx[e] = c;
					double* r = a.emit_rec + e * 9;
					r[0] = m[0]; r[1] = m[1]; r[2] = m[2];
#pragma unroll
					for (int t = 0; t < 6; t++) r[3 + t] = P[t];
				}
			}
		}
	}
	__syncthreads();
	if (tid == 0) {
		int ne = s_cnt;
		if (ne > a.ecap) { atomicOr(a.flags, PHD_FLAG_EMIT_OVERFLOW); ne = a.ecap; }
		a.emit_count[p] = ne;
	}
}

// =================================================================================================
template <int ZB>
__global__ __launch_bounds__(256) void k_correct(const DevParams prm, const StepBufs a)
{
	constexpr int MP = ZB * 64;
	__shared__ double zs[3 * MP], zmap[3 * MP];
	__shared__ double part[4 * MP], denom[MP];
	__shared__ double tile[TILE * CM_PLANES];   // [TILE][18]
	__shared__ double etab[EXPTAB_N];
	__shared__ int    s_npair, s_ncand;

	const int p = a.p0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int M = a.M;
	const MixView vin = bank_view(a, SEL_IN);
	const Bank& bin = a.bank[a.sel[SEL_IN]];
	const int np = vin.count[p] + a.born_count[p];
	const PoseD pose = load_pose(bin.poses + (size_t) p * 7);
	exp_tab_init(etab, tid);
	if (tid == 0) { s_npair = 0; s_ncand = 0; }
	for (int k = tid; k < MP; k += 256) {
		double z[3] = {0, 0, 1}, x[3] = {0, 0, 0};
		if (k < M) {
			z[0] = a.z[k * 3]; z[1] = a.z[k * 3 + 1]; z[2] = a.z[k * 3 + 2];
			measure_to_map(prm, pose, z, x);
		}
		zs[k * 3] = z[0]; zs[k * 3 + 1] = z[1]; zs[k * 3 + 2] = z[2];
		zmap[k * 3] = x[0]; zmap[k * 3 + 1] = x[1]; zmap[k * 3 + 2] = x[2];
	}
	__syncthreads();
	double zx[ZB], zy[ZB], zr[ZB], wx[ZB], wy[ZB], wz[ZB], wsum[ZB];
	bool   zv[ZB];
#pragma unroll
	for (int b = 0; b < ZB; b++) {
		int k = b * 64 + lane;
		zv[b] = k < M;
		zx[b] = zs[k * 3]; zy[b] = zs[k * 3 + 1]; zr[b] = zs[k * 3 + 2];
		wx[b] = zmap[k * 3]; wy[b] = zmap[k * 3 + 1]; wz[b] = zmap[k * 3 + 2];
		wsum[b] = 0;
	}
	const double g2 = prm.g2_correct, minw = prm.minw;
	const size_t cstride = a.cmplane;
	const double* cm = a.cm + (size_t) p * a.cmcap;
	int2*   pairs_ck = (int2*) (a.pair_ck + (size_t) p * a.ecap);
	double* pairs_w  = a.pair_w + (size_t) p * a.ecap;
	int*    cands    = a.cand + (size_t) p * a.candcap;   // (component << 8 | measurement) of the pairs worth a second look

	for (int sweep = 0; sweep < 2; sweep++) {
		if (sweep == 1 && s_ncand <= a.candcap) break;   // the queue held every candidate: handled below
		for (int c0 = 0; c0 < np; c0 += TILE) {
			const int c = c0 + tid;
			if (c < np) {
				double* tt = tile + tid * CM_PLANES;
#pragma unroll
				for (int t = 0; t < CM_PLANES; t++) tt[t] = cm[(size_t) t * cstride + c];
			}
			__syncthreads();
			const int cend = min(TILE, np - c0);
			if (sweep == 0) {
				// weightsum[z] += PD w q(z) over the components near MeasureToMap(z) (:882-890)
				auto visit = [&](int cc) {
					const double* tt = tile + cc * CM_PLANES;
					double Si[9];
#pragma unroll
					for (int t = 0; t < 9; t++) Si[t] = tt[3 + t];
#pragma unroll
					for (int b = 0; b < ZB; b++) {
						double e0 = wx[b] - tt[14], e1 = wy[b] - tt[15], e2 = wz[b] - tt[16];
						double sq = e0 * e0 + e1 * e1 + e2 * e2;
						double d2 = quad_gen(Si, zx[b] - tt[0], zy[b] - tt[1], zr[b] - tt[2]);
						double v  = tt[13] * (tt[12] * exp_neg(-0.5 * d2, etab));   // PD w * mc.Evaluate(z)
						const bool near = zv[b] && sq <= g2;
						if (near) wsum[b] += v;
						// only a pair with d2 <= dcut can reach MinWeight: remember it for the second pass
						const bool cand = near && d2 <= tt[17];
						unsigned long long bal = __ballot(cand);
						if (bal) {
							int base = 0, first = __ffsll((long long) bal) - 1;
							if (lane == first) base = atomicAdd(&s_ncand, __popcll(bal));
							base = __shfl(base, first, 64);
							if (cand) {
								int slot = base + __popcll(bal & lanemask_lt());
								if (slot < a.candcap) cands[slot] = ((c0 + cc) << 8) | (b * 64 + lane);
							}
						}
					}
				};
				int cc = wv;
				for (; cc + 4 < cend; cc += 8) { visit(cc); visit(cc + 4); }
				if (cc < cend) visit(cc);
			}
			else {
				for (int cc = wv; cc < cend; cc += 4) {
					const double* tt = tile + cc * CM_PLANES;
					double Si[9];
#pragma unroll
					for (int t = 0; t < 9; t++) Si[t] = tt[3 + t];
					const double dc = tt[17];
#pragma unroll
					for (int b = 0; b < ZB; b++) {
						double e0 = wx[b] - tt[14], e1 = wy[b] - tt[15], e2 = wz[b] - tt[16];
						double sq = e0 * e0 + e1 * e1 + e2 * e2;
						double d2 = quad_gen(Si, zx[b] - tt[0], zy[b] - tt[1], zr[b] - tt[2]);
						bool cand = zv[b] && sq <= g2 && (d2 <= dc);
						if (__ballot(cand)) {
							double q   = tt[12] * exp_neg(-0.5 * d2, etab);
							double wgt = tt[13] * q / denom[b * 64 + lane];   // :899
							bool   em  = cand && !(wgt < minw);
							unsigned long long bal = __ballot(em);
							if (bal) {
								int base = 0, first = __ffsll((long long) bal) - 1;
								if (lane == first) base = atomicAdd(&s_npair, __popcll(bal));
								base = __shfl(base, first, 64);
								if (em) {
									int slot = base + __popcll(bal & lanemask_lt());
									if (slot < a.ecap) {
										pairs_ck[slot] = make_int2(c0 + cc, b * 64 + lane);
										pairs_w[slot]  = wgt;
									}
								}
							}
						}
					}
				}
			}
			__syncthreads();
		}
		if (sweep == 0) {
#pragma unroll
			for (int b = 0; b < ZB; b++) part[wv * MP + b * 64 + lane] = wsum[b];
			__syncthreads();
			for (int k = tid; k < MP; k += 256) {
				denom[k] = prm.kappa + (part[k] + part[MP + k] + part[2 * MP + k] + part[3 * MP + k]);
			}
			__syncthreads();
		}
	}
	// second pass over the queued candidates only: w' = PD w q / (kappa + weightsum) >= MinWeight (:899)
	if (s_ncand <= a.candcap) {
		const int ncand = s_ncand;
		for (int j = tid; j < ncand; j += 256) {
			const int code = cands[j], c = code >> 8, k = code & 255;
			double Si[9];
#pragma unroll
			for (int t = 0; t < 9; t++) Si[t] = cm[(size_t) (3 + t) * cstride + c];
			const double d2 = quad_gen(Si, zs[k * 3] - cm[c], zs[k * 3 + 1] - cm[cstride + c], zs[k * 3 + 2] - cm[2 * cstride + c]);
			const double q   = cm[(size_t) 12 * cstride + c] * exp_neg(-0.5 * d2, etab);
			const double wgt = cm[(size_t) 13 * cstride + c] * q / denom[k];
			if (!(wgt < minw)) {
				int slot = atomicAdd(&s_npair, 1);
				if (slot < a.ecap) {
					pairs_ck[slot] = make_int2(c, k);
					pairs_w[slot]  = wgt;
				}
			}
		}
		__syncthreads();
	}
	if (tid == 0) a.pair_count[p] = s_npair;
}

// =================================================================================================
__global__ __launch_bounds__(256) void k_emit_finish(const DevParams prm, const StepBufs a)
{
	const int p = a.p0 + blockIdx.x, tid = threadIdx.x;
	const MixView vin = bank_view(a, SEL_IN);
	const Bank& bin = a.bank[a.sel[SEL_IN]];
	const int n = vin.count[p], np = n + a.born_count[p];
	const int nmis = a.emit_count[p];
	int npair = a.pair_count[p];
	const bool overflow = npair > a.ecap || nmis + npair > a.ecap;
	npair = min(npair, a.ecap - nmis);
	__syncthreads();   // every thread has read emit_count before thread 0 rewrites it
	const PoseD pose = load_pose(bin.poses + (size_t) p * 7);
	double rq[9];
	conj_matrix(pose, rq);
	const int2*   pairs_ck = (const int2*) (a.pair_ck + (size_t) p * a.ecap);
	const double* pairs_w  = a.pair_w + (size_t) p * a.ecap;
	for (int j = tid; j < npair; j += 256) {
		const int2 ck = pairs_ck[j];
		double w, m[3], P[6];
		load_predicted(prm, a, vin, p, n, ck.x, w, m, P);
		CompMeas q;
		comp_measure(prm, pose, rq, m, P, q);
		double K[9], Pn[6], mn[3];
		kalman_gain(q, K);
		kalman_cov(q, K, P, Pn);
		const double n0 = a.z[ck.y * 3] - q.zh[0], n1 = a.z[ck.y * 3 + 1] - q.zh[1], n2 = a.z[ck.y * 3 + 2] - q.zh[2];
#pragma unroll
		for (int t = 0; t < 3; t++) mn[t] = m[t] + (K[t * 3] * n0 + K[t * 3 + 1] * n1 + K[t * 3 + 2] * n2);   // :896
		const size_t e = (size_t) p * a.ecap + nmis + j;
		a.emit_w[e]   = pairs_w[j];
		a.emit_idx[e] = np + ck.y * np + ck.x;   // position in the reference's `corrected` list: after the np copies, z-major
		double* r = a.emit_rec + e * 9;
		r[0] = mn[0]; r[1] = mn[1]; r[2] = mn[2];
#pragma unroll
		for (int t = 0; t < 6; t++) r[3 + t] = Pn[t];
	}
	if (tid == 0) {
		if (overflow) atomicOr(a.flags, PHD_FLAG_EMIT_OVERFLOW);
		a.emit_count[p] = nmis + npair;
	}
}
