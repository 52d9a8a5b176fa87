// phd_prune.h — k_prune_merge: PruneModel (PHDNavigator.cs:913-948) for one particle per workgroup.
//
// The reference sorts the corrected mixture by weight, keeps the first min(MaxQuantity, #w >= MinWeight)
// entries and then, in weight order, lets every still-present entry i absorb every later entry k with
// (m_i - m_k)^T P_i^-1 (m_i - m_k) < MergeThreshold^2 (Gaussian.AreClose, Gaussian.cs:243-246), replacing
// the set by its moment match (Gaussian.Merge, Gaussian.cs:297-347).
//
// Only the question "is i still present" is sequential. The kernel therefore splits the work:
//   A. rank     : stable descending order by counting (every thread ranks its entries against LDS key tiles)
//   B. pairs    : every closeness test close_i(k), k > i, in parallel — row i per thread, the candidate's
//                 P_i^-1 in registers, the other means broadcast from LDS; a Euclidean bound
//                 |d|^2 > T^2 trace(P_i)  =>  d^T P_i^-1 d >= |d|^2 / lambda_max(P_i) > T^2
//                 skips the quadratic form for far pairs. Each row keeps its first 7 close entries.
//   C. resolve  : one wave walks the rows in weight order with the "absorbed" bits spread over its lanes
//                 (integer work only); rows with more than 7 close entries are re-tested by the 64 lanes.
//   D. merge    : one thread per surviving row accumulates the raw moments of its set in list order
//                 (leader first, members by rank — the reference's summation order) and writes the
//                 result at its position among the survivors.
#pragma once
#include "phd_device.h"

#define PRUNE_KEYTILE 1024
#define PRUNE_NBR 7

struct PruneLds {
	int sw, sm, order, x, scan;   // offsets in doubles
	int bytes;
};

__host__ __device__ inline PruneLds prune_lds(int cutcap)
{
	PruneLds l;
	int cc = (cutcap + 1) & ~1;
	l.sw    = 0;
	l.sm    = l.sw + cc;
	l.order = l.sm + 3 * cc;                 // int[cc]
	l.x     = l.order + cc / 2;              // key tile (1024 doubles + 1024 ints)  |  nbr u64[2*cc] + owner int[cc]
	int keytile = PRUNE_KEYTILE + PRUNE_KEYTILE / 2;
	int rest    = 2 * cc + cc / 2;
	l.scan  = l.x + (keytile > rest ? keytile : rest);
	l.bytes = (l.scan + 132) * 8;            // int[264]
	return l;
}

__global__ __launch_bounds__(256) void k_prune_merge(const DevParams prm, const StepBufs a, int cutcap)
{
	extern __shared__ __align__(16) double smem[];
	const PruneLds lay = prune_lds(cutcap);
	const int cc = (cutcap + 1) & ~1;
	double* sw    = smem + lay.sw;                         // [cut] sorted weights
	double* sm    = smem + lay.sm;                         // [3][cc] sorted means
	int*    order = (int*) (smem + lay.order);             // [cut] emit slot of rank r
	double* kw    = smem + lay.x;                          // key tile: weights
	int*    ki    = (int*) (kw + PRUNE_KEYTILE);           //           canonical indices
	unsigned long long* nbr = (unsigned long long*) (smem + lay.x);   // [cut][2]: count + up to 7 close later rows
	int*    owner = (int*) (nbr + 2 * cc);                 // [cut] row that absorbed k (k itself for a survivor)
	int*    scan  = (int*) (smem + lay.scan);              // [264]

	const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const MixView vout = bank_view(a, SEL_OUT);
	const int ne = a.emit_count[p];
	const size_t eb = (size_t) p * a.ecap;
	const int cut = min(min(prm.maxq, ne), cutcap);   // weightcut: every emitted weight is already >= MinWeight

	// ---- A. rank by (weight desc, canonical index asc) == the reference's sort made stable
	for (int g0 = 0; g0 < ne; g0 += 1024) {           // this thread's entries g0 + u*256 + tid, u < 4
		double we[4];
		int    ie[4], rank[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			int e = g0 + u * 256 + tid;
			we[u] = (e < ne) ? a.emit_w[eb + e] : 0.0;
			ie[u] = (e < ne) ? a.emit_idx[eb + e] : 0x7fffffff;
			rank[u] = 0;
		}
		for (int t0 = 0; t0 < ne; t0 += PRUNE_KEYTILE) {
			__syncthreads();
			for (int j = tid; j < PRUNE_KEYTILE && t0 + j < ne; j += 256) {
				kw[j] = a.emit_w[eb + t0 + j];
				ki[j] = a.emit_idx[eb + t0 + j];
			}
			__syncthreads();
			int jend = min(PRUNE_KEYTILE, ne - t0);
			for (int j = 0; j < jend; j++) {
				double wj = kw[j];
				int    ij = ki[j];
#pragma unroll
				for (int u = 0; u < 4; u++) rank[u] += (wj > we[u]) || (wj == we[u] && ij < ie[u]);
			}
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			int e = g0 + u * 256 + tid;
			if (e < ne && rank[u] < cut) {
				order[rank[u]] = e;
				sw[rank[u]]    = we[u];
			}
		}
	}
	__syncthreads();
	for (int r = tid; r < cut; r += 256) {
		const double* rec = a.emit_rec + (eb + order[r]) * 9;
		sm[r] = rec[0]; sm[cc + r] = rec[1]; sm[2 * cc + r] = rec[2];
	}
	__syncthreads();   // the key tile is dead from here on: nbr / owner take its place

	// ---- B. all closeness tests, row per thread
	for (int rb = 0; rb * 256 < cut; rb++) {
		const int  i = rb * 256 + tid;
		const bool valid = i < cut;
		double Pi[6] = {1, 0, 0, 1, 0, 1}, m0 = 0, m1 = 0, m2 = 0, bound = -1.0;
		if (valid) {
			const double* rec = a.emit_rec + (eb + order[i]) * 9;
			double P[6], det;
#pragma unroll
			for (int t = 0; t < 6; t++) P[t] = rec[3 + t];
			inv_sym3(P, Pi, det);
			m0 = sm[i]; m1 = sm[cc + i]; m2 = sm[2 * cc + i];
			// Sylvester: only a positive definite P_i admits the Euclidean bound
			bool pd = P[0] > 0 && (P[0] * P[3] - P[1] * P[1]) > 0 && det > 0;
			bound = pd ? prm.merge_thr2 * (P[0] + P[3] + P[5]) * (1.0 + 1e-6) : INFINITY;
		}
		unsigned long long lo = 0, hi = 0;
		int cnt = 0;
		const int kstart = rb * 256 + wv * 64 + 1;   // rows of this wave are >= kstart - 1
		for (int k = kstart; k < cut; k++) {
			double d0 = m0 - sm[k], d1 = m1 - sm[cc + k], d2 = m2 - sm[2 * cc + k];
			double sq = d0 * d0 + d1 * d1 + d2 * d2;
			if (valid && k > i && sq <= bound) {
				if (quad_sym(Pi, d0, d1, d2) < prm.merge_thr2) {   // Gaussian.SquareMahalanobis(b.Mean) < threshold^2
					if (cnt < 3) lo |= (unsigned long long) k << (16 * (cnt + 1));
					else if (cnt < PRUNE_NBR) hi |= (unsigned long long) k << (16 * (cnt - 3));
					cnt++;
				}
			}
		}
		if (valid) {
			nbr[2 * i]     = lo | (unsigned long long) min(cnt, 0xffff);
			nbr[2 * i + 1] = hi;
			owner[i]       = -1;
		}
	}
	__syncthreads();

	// ---- C. who survives: sequential in rank order, one wave, absorbed bits in registers
	if (wv == 0) {
		unsigned int absorbed = 0;                  // bit s of lane l <-> row s*64 + l
		const int nslots = (cut + 63) >> 6;
		unsigned long long nlo = (cut > 0) ? nbr[0] : 0, nhi = (cut > 0) ? nbr[1] : 0;
		for (int i = 0; i < cut; i++) {
			const unsigned long long lo = nlo, hi = nhi;
			if (i + 1 < cut) { nlo = nbr[2 * (i + 1)]; nhi = nbr[2 * (i + 1) + 1]; }   // independent of the state: prefetched
			unsigned int om = (unsigned int) __builtin_amdgcn_readlane((int) absorbed, i & 63);
			if ((om >> (i >> 6)) & 1u) continue;
			if (lane == 0) owner[i] = i;
			const int cnt = (int) (lo & 0xffff);
			if (cnt <= PRUNE_NBR) {
				for (int c = 0; c < cnt; c++) {
					int k = (int) (((c < 3) ? (lo >> (16 * (c + 1))) : (hi >> (16 * (c - 3)))) & 0xffff);
					unsigned int km = (unsigned int) __builtin_amdgcn_readlane((int) absorbed, k & 63);
					if (!((km >> (k >> 6)) & 1u)) {
						if (lane == (k & 63)) absorbed |= 1u << (k >> 6);
						if (lane == 0) owner[k] = i;
					}
				}
			}
			else {
				// more close rows than the list holds: re-test row i against every later row, 64 at a time
				const double* rec = a.emit_rec + (eb + order[i]) * 9;
				double P[6], Pi[6], det;
#pragma unroll
				for (int t = 0; t < 6; t++) P[t] = rec[3 + t];
				inv_sym3(P, Pi, det);
				double m0 = sm[i], m1 = sm[cc + i], m2 = sm[2 * cc + i];
				for (int s = i >> 6; s < nslots; s++) {
					int k = s * 64 + lane;
					if (k > i && k < cut && !((absorbed >> s) & 1u)) {
						if (quad_sym(Pi, m0 - sm[k], m1 - sm[cc + k], m2 - sm[2 * cc + k]) < prm.merge_thr2) {
							absorbed |= 1u << s;
							owner[k] = i;
						}
					}
				}
			}
		}
	}
	__syncthreads();

	// ---- D. output position of every survivor (exclusive scan of the survivor flags) and the merges
	int nsurv_before = 0;
	for (int r0 = 0; r0 < cut; r0 += 256) {
		const int  i = r0 + tid;
		const bool surv = i < cut && owner[i] == i;
		unsigned long long bal = __ballot(surv);
		if (lane == 0) scan[wv] = __popcll(bal);
		__syncthreads();
		int base = nsurv_before;
		for (int q = 0; q < wv; q++) base += scan[q];
		const int total = scan[0] + scan[1] + scan[2] + scan[3];
		__syncthreads();
		if (surv) {
			const int pos = base + __popcll(bal & lanemask_lt());
			const double* rec = a.emit_rec + (eb + order[i]) * 9;
			// Gaussian.Merge (Gaussian.cs:329-346): raw moments, the candidate first, then its set in list order
			double w = sw[i], m0 = rec[0], m1 = rec[1], m2 = rec[2];
			double W = 0.0 + w;
			double M0 = 0.0 + w * m0, M1 = 0.0 + w * m1, M2 = 0.0 + w * m2;
			double C0 = 0.0 + w * (rec[3] + m0 * m0), C1 = 0.0 + w * (rec[4] + m0 * m1), C2 = 0.0 + w * (rec[5] + m0 * m2);
			double C3 = 0.0 + w * (rec[6] + m1 * m1), C4 = 0.0 + w * (rec[7] + m1 * m2), C5 = 0.0 + w * (rec[8] + m2 * m2);
			auto absorb = [&](int k) {
				const double* rk = a.emit_rec + (eb + order[k]) * 9;
				double wk = sw[k], k0 = rk[0], k1 = rk[1], k2 = rk[2];
				W += wk;
				M0 += wk * k0; M1 += wk * k1; M2 += wk * k2;
				C0 += wk * (rk[3] + k0 * k0); C1 += wk * (rk[4] + k0 * k1); C2 += wk * (rk[5] + k0 * k2);
				C3 += wk * (rk[6] + k1 * k1); C4 += wk * (rk[7] + k1 * k2); C5 += wk * (rk[8] + k2 * k2);
			};
			const unsigned long long lo = nbr[2 * i], hi = nbr[2 * i + 1];
			const int cnt = (int) (lo & 0xffff);
			if (cnt <= PRUNE_NBR) {
				for (int c = 0; c < cnt; c++) {
					int k = (int) (((c < 3) ? (lo >> (16 * (c + 1))) : (hi >> (16 * (c - 3)))) & 0xffff);
					if (owner[k] == i) absorb(k);
				}
			}
			else {
				for (int k = i + 1; k < cut; k++) {
					if (owner[k] == i) absorb(k);
				}
			}
			double ow, o0, o1, o2, oP[6];
			if (W < 1e-15) {   // Gaussian.cs:339-341
				ow = 0.0; o0 = m0; o1 = m1; o2 = m2;
				oP[0] = 1e12; oP[1] = 0; oP[2] = 0; oP[3] = 1e12; oP[4] = 0; oP[5] = 1e12;
			}
			else {
				ow = W; o0 = M0 / W; o1 = M1 / W; o2 = M2 / W;
				oP[0] = C0 / W - o0 * o0; oP[1] = C1 / W - o0 * o1; oP[2] = C2 / W - o0 * o2;
				oP[3] = C3 / W - o1 * o1; oP[4] = C4 / W - o1 * o2; oP[5] = C5 / W - o2 * o2;
			}
			const size_t ob = (size_t) p * a.cap + pos;
			vout.w[ob] = ow;
			vout.m[0][ob] = o0; vout.m[1][ob] = o1; vout.m[2][ob] = o2;
#pragma unroll
			for (int t = 0; t < 6; t++) vout.P[t][ob] = oP[t];
		}
		nsurv_before += total;
	}
	if (tid == 0) vout.count[p] = nsurv_before;
}
