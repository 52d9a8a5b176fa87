// phd_correct.h — PredictConditional + CorrectConditional (+ the MinWeight cut of PruneModel) as four
// kernels with small live state each (one workgroup per particle):
//
//   k_explore<ZB>   PredictConditional (PHDNavigator.cs:793-819): Explored(model, MeasureToMap(z)) for every
//                   measurement against the PRIOR map; unexplored measurements become births.
//   k_measure       per component of the predicted mixture (prior + births): h(m), S^-1, PD (:857-870) reduced to
//                   the 10 planes the pair sweep needs (h(m), the folded quadratic form, the log of PD w times the
//                   multiplier), and the misdetection copies w (1 - PD) (:837-840) that survive MinWeight.
//   k_correct<ZB>   every (component, measurement) pair, measurement per lane, the component broadcast from an
//                   LDS tile: weightsum[z] over the near components (:886-890); pairs that can reach MinWeight
//                   are queued.
//   k_emit_finish   the queued pairs: w' = PD w q / (kappa + weightsum) (:899) in the reference's own arithmetic
//                   and, for those that reach MinWeight, the Kalman update (:895-897) m' = m + K nu,
//                   P' = (I - K H) P.
//
// Splitting keeps the hot pair loop free of the register-hungry per-component algebra (the fused version
// needed 252 VGPRs and spilled ~100 SGPRs); the price is the plane scratch, written once and read once.
#pragma once
#include "phd_device.h"

#define CM_PLANES 10   // zh[3], G[6], lw: PD w N(z; zh, S) = exp(lw + d^T G d), d = z - zh
#define CM_TILE   13   // what the pair sweep stages per component: the planes and the mean
#define EMIT_LIST 1024 // pairs k_emit_finish gathers before it runs the Kalman path on them

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
			// G = -(S^-1 + S^-T) / 4 folded for the upper-triangle sum of gauss_logw
			cm[(size_t) 3 * cstride + c] = -0.5 * q.Sinv[0];
			cm[(size_t) 4 * cstride + c] = -0.5 * (q.Sinv[1] + q.Sinv[3]);
			cm[(size_t) 5 * cstride + c] = -0.5 * (q.Sinv[2] + q.Sinv[6]);
			cm[(size_t) 6 * cstride + c] = -0.5 * q.Sinv[4];
			cm[(size_t) 7 * cstride + c] = -0.5 * (q.Sinv[5] + q.Sinv[7]);
			cm[(size_t) 8 * cstride + c] = -0.5 * q.Sinv[8];
			cm[(size_t) 9 * cstride + c] = log(pdw * q.qmult);
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
	__shared__ double part[4 * MP];
	__shared__ double tile[TILE * CM_TILE];   // [TILE][13]
	__shared__ double etab[EXPTAB_N];
	__shared__ int    s_ncand;

	const int p = a.p0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int M = a.M;
	const MixView vin = bank_view(a, SEL_IN);
	const Bank& bin = a.bank[a.sel[SEL_IN]];
	const int n = vin.count[p], np = n + a.born_count[p];
	const size_t sb = (size_t) p * a.cap;
	const PoseD pose = load_pose(bin.poses + (size_t) p * 7);
	exp_tab_init(etab, tid);
	if (tid == 0) s_ncand = 0;
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
	const double g2 = prm.g2_correct;
	const size_t cstride = a.cmplane;
	const double* cm = a.cm + (size_t) p * a.cmcap;
	int2* cands = (int2*) a.cand + (size_t) p * a.candcap;   // (component << 8 | measurement, exponent as float32) of the pairs worth a second look

	// weightsum[z] += PD w q(z) over the components near MeasureToMap(z) (:882-890); a pair can only reach MinWeight
	// when PD w q(z) >= MinWeight kappa, i.e. its exponent reaches emit_log_floor (half a unit of margin for the
	// rounding of the folded form): those are queued for k_emit_finish
	const double xcut = prm.emit_log_floor - 0.5;
	for (int c0 = 0; c0 < np; c0 += TILE) {
		const int c = c0 + tid;
		if (c < np) {
			double* tt = tile + tid * CM_TILE;
#pragma unroll
			for (int t = 0; t < CM_PLANES; t++) tt[t] = cm[(size_t) t * cstride + c];
			if (c < n) {
#pragma unroll
				for (int t = 0; t < 3; t++) tt[10 + t] = vin.m[t][sb + c];
			}
			else {
				const double* bm = a.born_mean + ((size_t) p * a.Mcap + (c - n)) * 3;
				tt[10] = bm[0]; tt[11] = bm[1]; tt[12] = bm[2];
			}
		}
		__syncthreads();
		const int cend = min(TILE, np - c0);
		auto visit = [&](int cc) {
			const double* tt = tile + cc * CM_TILE;
#pragma unroll
			for (int b = 0; b < ZB; b++) {
				double e0 = wx[b] - tt[10], e1 = wy[b] - tt[11], e2 = wz[b] - tt[12];
				double sq = e0 * e0 + e1 * e1 + e2 * e2;
				double x  = gauss_logw(tt, zx[b] - tt[0], zy[b] - tt[1], zr[b] - tt[2]);
				double v  = exp_neg(x, etab);   // PD w * mc.Evaluate(z)
				const bool near = zv[b] && sq <= g2;
				if (near) wsum[b] += v;
				const bool cand = near && x >= xcut;
				unsigned long long bal = __ballot(cand);
				if (bal) {
					int base = 0, first = __ffsll((long long) bal) - 1;
					if (lane == first) base = atomicAdd(&s_ncand, __popcll(bal));
					base = __shfl(base, first, 64);
					if (cand) {
						int slot = base + __popcll(bal & lanemask_lt());
						if (slot < a.candcap) cands[slot] = make_int2(((c0 + cc) << 8) | (b * 64 + lane), __float_as_int((float) x));
					}
				}
			}
		};
		int cc = wv;
		for (; cc + 4 < cend; cc += 8) { visit(cc); visit(cc + 4); }
		if (cc < cend) visit(cc);
		__syncthreads();
	}
#pragma unroll
	for (int b = 0; b < ZB; b++) part[wv * MP + b * 64 + lane] = wsum[b];
	__syncthreads();
	for (int k = tid; k < M; k += 256) {
		a.denom[(size_t) p * a.Mcap + k] = prm.kappa + (part[k] + part[MP + k] + part[2 * MP + k] + part[3 * MP + k]);
	}
	if (tid == 0) a.cand_count[p] = s_ncand;
}

// =================================================================================================
__global__ __launch_bounds__(256) void k_emit_finish(const DevParams prm, const StepBufs a)
{
	__shared__ double etab[EXPTAB_N];
	__shared__ int    s_npair;
	const int p = a.p0 + blockIdx.x, tid = threadIdx.x;
	const int M = a.M;
	const MixView vin = bank_view(a, SEL_IN);
	const Bank& bin = a.bank[a.sel[SEL_IN]];
	const int n = vin.count[p], np = n + a.born_count[p];
	const int nmis = a.emit_count[p];
	const int ncand = a.cand_count[p];
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
	if (ncand <= a.candcap) {
		// Most queued pairs fail once the real denominator is known. Their exponent x = log(PD w q) travels with
		// them as a float32: x - log(denom) < log(MinWeight) by more than the float32 rounding settles it; the
		// others are compacted into a list so that the waves run the heavy path on full lanes.
		__shared__ double ldenom[256];
		__shared__ int    list[EMIT_LIST], s_nlist;
		for (int k = tid; k < M; k += 256) ldenom[k] = log(denom[k]);
		if (tid == 0) s_nlist = 0;
		__syncthreads();
		const double lminw = log(prm.minw);
		for (int j0 = 0; j0 < ncand; j0 += 256) {
			const int j = j0 + tid;
			bool keep = false;
			int  code = 0;
			if (j < ncand) {
				const int2 cd = cands[j];
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
		// the queue overflowed (more than a quarter of all pairs are candidates): every pair, gate included
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
