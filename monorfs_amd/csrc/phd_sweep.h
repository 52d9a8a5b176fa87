// phd_sweep.h — PredictConditional and the weight sums of CorrectConditional in one pass over the prior mixture.
//
// Both steps visit every (prior component, measurement) pair around the same vector e = MeasureToMap(z) - m:
//   Explored (PHDNavigator.cs:956-959)      density sum_c w_c N(x_z; m_c, P_c) over |e|^2 <= (3 DDT) gate   -> births
//   CorrectConditional (:882-890)           weightsum[z] = sum_c PD_c w_c N(z; h(m_c), S_c) over |e|^2 <= DDT gate
// so k_sweep evaluates both Gaussians per visit and shares e, |e|^2, the staging and the loop. Per prior component it
// stages two gauss_logw records, both built on the spot from (w, m, P): the map-space one (m, -P^-1/2 folded,
// log(w mult_P)) by the first wave(s) and the measurement-space one (h(m), -(S^-1 + S^-T)/4 folded, log(PD w mult_S); :857-870)
// by the others, which also emit the component's misdetection copy w (1 - PD) (:837-840) when it reaches MinWeight.
// The births it finds are measured, emitted and added to the weight sums in its tail. Nothing per component goes
// through HBM between the prior mixture and the emitted list.
#pragma once
#include "phd_device.h"

#ifndef SW_FIRST1
#define SW_FIRST1 128   // length of the first tile of the one-block kernel (64 or SW_TILE1)
#endif
#ifndef SW_TILE1
#define SW_TILE1  128   // prior components staged per LDS tile in the one-block kernel (up to 64 measurements): 128 or 192. With 192
#endif                  // three waves share the dear part of the staging (and the kernel's LDS just fits four workgroups per CU):
                        // measured, no change (0.2007 vs 0.2003 ms) — the kernel is bound by the instructions it issues, not by the
                        // longest wave. The two- and four-block kernels (more partial-sum arrays) are 128 either way.
#define SW_REC    20    // doubles per staged component: zh[3] G[6] lw | m[3] Gm[6] lwm  (two gauss_logw records)
#define SW_UMAX   16    // measurements still unexplored when the density part leaves the pair loop (see the sweep)

// =================================================================================================
// LDS of the sweep (doubles): the bodies of a step's kernels take their arrays from a pool handed in by the kernel, so that
// k_particle_chain can run them back to back in ONE pool (phd_kernels.h)
template <int ZB>
struct SweepLds {
	static constexpr int MP = ZB * 64;
	static constexpr int T = (ZB == 1) ? SW_TILE1 : 128;   // components per tile
	static constexpr int NMAP = 256 - T;                   // threads that build the map-space records (the first waves), ...
	static constexpr int MAPEACH = T / NMAP;               // ... this many components each
	static_assert(T % 64 == 0 && T < 256 && T % NMAP == 0, "tile: whole waves, and the map-space threads share it evenly");
	static constexpr int TD = (T * SW_REC > MP * 13) ? T * SW_REC : MP * 13;
	static constexpr int zs = 0, zmap = zs + 3 * MP, part = zmap + 3 * MP, part2 = part + 4 * MP, tile = part2 + 4 * MP,
	                     etab = tile + TD, du = etab + EXPTAB_N, ints = du + SW_UMAX;   // ints: born[MP], ulist[SW_UMAX], nb, nmis, nu
	static constexpr int doubles = ints + (MP + SW_UMAX + 4) / 2;
};

// HALF (with ZB = 1, at most 32 measurements — the common frame of the reference's own scenes): lanes 32-63 hold the
// measurements of lanes 0-31 again and take the component four further on, so a visit of the pair loop covers two
// components and the loop is half as long; the two halves' partial sums of a measurement meet in the reductions.
template <int ZB, bool HALF = false>
__device__ __forceinline__ void sweep_body(const DevParams& prm, const StepBufs& a, double* pool)
{
	static_assert(!HALF || ZB == 1, "HALF is a layout of the one-block kernel");
	using L = SweepLds<ZB>;
	constexpr int MP = L::MP;
	double* const zs = pool + L::zs;
	double* const zmap = pool + L::zmap;
	double* const part = pool + L::part;
	double* const part2 = pool + L::part2;
	double* const tile = pool + L::tile;                  // [L::T][20] prior components | [births][13] in the tail
	double* const etab = pool + L::etab;
	double* const s_du = pool + L::du;
	int* const born = (int*) (pool + L::ints);
	int* const s_ulist = born + MP;
	int& s_nb = born[MP + SW_UMAX];
	int& s_nmis = born[MP + SW_UMAX + 1];
	int& s_nu = born[MP + SW_UMAX + 2];

	const int p = a.p0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int M = a.M;
#ifdef PHD_STAMPS
	// diagnostic build (PHD_STAMP_KERNEL=6): the clock the chip holds under this kernel — shader cycles (s_memtime) against
	// the constant 100 MHz counter (s_memrealtime) over the workgroup's lifetime, and over its pair loops alone
	const long long st_c0 = clock64(), st_r0 = wall_clock64();
	long long st_pairc = 0, st_pairr = 0;
#endif
	const int klane = HALF ? (lane & 31) : lane;          // this lane's measurement (within its block of 64)
	const int hoff  = HALF ? 4 * (lane >> 5) : 0;         // ... and how far behind the wave's component its own is
	const bool owner = !HALF || lane < 32;                // the lane that speaks for the measurement in the reductions
	const MixView vin = bank_view(a, SEL_IN);
	const Bank bin = bank_of(a, SEL_IN);
	const int n = vin.count[p];
	const size_t sb = in_base(a, p);   // the particle's prior mixture in the INMIX bank
	const Bank bout = bank_of(a, SEL_OUT);
	PoseD pose = load_pose(bin.poses + (size_t) p * 7);
	pose.t[0] = uniform_d(pose.t[0]); pose.t[1] = uniform_d(pose.t[1]); pose.t[2] = uniform_d(pose.t[2]);
	pose.qw = uniform_d(pose.qw); pose.qx = uniform_d(pose.qx); pose.qy = uniform_d(pose.qy); pose.qz = uniform_d(pose.qz);
	// the particle keeps its pose and (until the reweight kernel runs) its weight in the output bank
	if (tid < 7) bout.poses[(size_t) p * 7 + tid] = bin.poses[(size_t) p * 7 + tid];
	if (tid == 7) bout.weights[p] = bin.weights[p];
	// (the rotation matrix of the Jacobian is rebuilt from the quaternion — scalar registers — where a staging phase needs it: ~20
	// scalar-operand instructions per tile against nine doubles that were kept across the pair loops, i.e. spilled: the kernel
	// is at its 106 scalar and 128 vector registers)
	auto rotation = [&](double* rq) {
		PoseD pq = pose;
		asm volatile("" : "+s"(pq.qw), "+s"(pq.qx), "+s"(pq.qy), "+s"(pq.qz));   // (keeps the compiler from hoisting the products out of the tile loop)
		conj_matrix(pq, rq);
	};
	exp_tab_init(etab, tid);
	if (tid == 0) { s_nb = 0; s_nmis = 0; s_nu = 0; }
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
	double zx[ZB], zy[ZB], zr[ZB], wx[ZB], wy[ZB], wz[ZB], wsum[ZB], dens[ZB];
	bool   zv[ZB];
#pragma unroll
	for (int b = 0; b < ZB; b++) {
		zv[b] = b * 64 + klane < M;
		wsum[b] = 0; dens[b] = 0;
	}
	// this lane's measurements, from LDS into registers: done anew behind every staging phase (which needs the registers
	// for the 3 x 3 algebra of a component) rather than once, so that the values are not live across it
	auto load_z = [&]() {
#pragma unroll
		for (int b = 0; b < ZB; b++) {
			const int k = b * 64 + klane;
			zx[b] = zs[k * 3]; zy[b] = zs[k * 3 + 1]; zr[b] = zs[k * 3 + 2];
			wx[b] = zmap[k * 3]; wy[b] = zmap[k * 3 + 1]; wz[b] = zmap[k * 3 + 2];
		}
	};
	const double g2c = prm.g2_correct, g2e = prm.g2_explore, thr = prm.expl_thr;
	// (component << 8 | measurement, exponent as float32) of the pairs worth a second look. Every wave appends to a
	// segment of its own (a quarter of the queue) and counts in a scalar register: no atomic, no LDS round trip in the loop.
	const int segcap = a.candcap >> 2;
	int2* cands = (int2*) a.cand + (size_t) p * a.candcap + (size_t) wv * segcap;
	int ncand_w = 0;   // wave-uniform

	// A pair can only reach MinWeight when PD w q(z) >= MinWeight (kappa + weightsum[z]) (:899). The sum is not known
	// yet, but its terms are >= 0 and this lane's own partial sum only grows: a pair below MinWeight (kappa + partial sum so
	// far) e^-1/2 — the factor is the margin for the rounding of the folded form — cannot make it. The others are queued
	// for k_emit_finish, which knows the whole sum.
	const double cfac = prm.minw * 0.6065306597126334, ckap = cfac * prm.kappa;
	// the weight-sum part of a visit: component record tt = [zh(3) G(6) lw], squared distance in map space sq
	// (the masks are built from the bare compares — a ballot of a compound condition costs two vector instructions — and
	// combined in scalar registers; zvm[b] = the lanes that hold a measurement)
	unsigned long long zvm[ZB];
#pragma unroll
	for (int b = 0; b < ZB; b++) zvm[b] = ballot64(zv[b]);
	// (cval: this lane's component exists — always, except for the upper half of a HALF visit at the end of a tile)
	// (a visit none of whose 64 pairs lies inside the radius gate contributes exact zeros: it is skipped — the reference only
	// ever evaluates gated pairs, PHDNavigator.cs:882 — unless the handle is in all-pairs mode, SURVEY §8d's benchmark mode,
	// where every pair is evaluated and the gate only masks; the results are the same bits either way)
	const bool allpairs = a.all_pairs != 0;
	auto weigh = [&](const double* tt, int b, double sq, int c, bool cval) {
		const unsigned long long nm = ballot64(sq <= g2c) & (HALF ? ballot64(zv[b] && cval) : zvm[b]);
		if (nm || allpairs) {   // (wave-uniform)
			const double x = gauss_logw(tt, zx[b] - tt[0], zy[b] - tt[1], zr[b] - tt[2]);
			const double v = exp_pair(x, etab, zv[b] && sq <= g2c && (!HALF || cval));   // PD w * mc.Evaluate(z); 0 outside the gate
			wsum[b] += v;
			const unsigned long long bal = ballot64(v >= fma(cfac, wsum[b], ckap)) & nm;
			if (bal) {
				if ((bal >> lane) & 1ull) {
					const int slot = ncand_w + __popcll(bal & lanemask_lt());
					if (slot < segcap) cands[slot] = make_int2((c << 8) | (b * 64 + klane), __float_as_int((float) x));
				}
				ncand_w += __popcll(bal);
			}
		}
	};

	// the four waves' partial sums of measurement k (HALF: of both lane halves)
	auto sumk = [&](const double* arr, int k) {
		const double lo = arr[k] + arr[MP + k] + arr[2 * MP + k] + arr[3 * MP + k];
		if (!HALF) return lo;
		return lo + (arr[k + 32] + arr[MP + k + 32] + arr[2 * MP + k + 32] + arr[3 * MP + k + 32]);
	};

	// ---- prior components: Explored density and weight sums together. Every term of the density is >= 0, so a measurement
	// is explored as soon as the sum so far reaches the threshold — within a tile or two for all but the few that will
	// be born: a component within five sigma does it. Once at most SW_UMAX measurements are still open the
	// density part leaves the pair loop (where it would keep costing all 64 lanes an evaluation per component) and
	// the open measurements are summed the other way round: component per lane, one open measurement at a time, a
	// wave reduction per tile. The terms are the same; only their order of addition differs.
	bool compact = false;   // block-uniform
	constexpr int SW_TILE = L::T, SW_NMAP = L::NMAP, SW_MAPEACH = L::MAPEACH;
	// (the first tile may be shorter — SW_FIRST1: the sooner the tile ends at which at most SW_UMAX measurements are still
	// unexplored, the sooner the Explored density leaves the pair loop)
	constexpr int SW_FIRST = (ZB == 1 && SW_FIRST1 < SW_TILE) ? SW_FIRST1 : SW_TILE;
	for (int c0 = 0, tl = SW_FIRST; c0 < n; c0 += tl, tl = SW_TILE) {
		{   // the measurement-space record of a component (h(m), S^-1, PD: ~650 dependent FP64 instructions, :857-870) is built by
			// one thread of the last SW_TILE / 64 waves; its map-space record (P^-1: ~110) by the first waves, SW_MAPEACH
			// components per thread — with 192-component tiles three waves share the dear part and one takes the cheap one
			if (tid < SW_NMAP) {
#pragma unroll
				for (int q = 0; q < SW_MAPEACH; q++) {
					const int cl = tid + q * SW_NMAP, c = c0 + cl;
					if (c < n && cl < tl) {
						double P[6], m[3], Pi[6], det, w;
						load_comp(vin.rec + (sb + c) * MIX_REC, w, m, P);
						inv_sym3(P, Pi, det);
						gauss_record(w, m, Pi, PHD_INV_2PI / sqrt(fabs(det)), tile + cl * SW_REC + 10);
					}
				}
			}
			else {   // (wave-uniform) also the misdetection copies that survive MinWeight
				const int cl = tid - SW_NMAP, c = c0 + cl;
				bool mis = false;
				double wm = 0;
				if (c < n && cl < tl) {
					double* tt = tile + cl * SW_REC;
					double P[6], m[3], w;
					load_comp(vin.rec + (sb + c) * MIX_REC, w, m, P);
					CompMeas q;
					double rq[9];
					rotation(rq);
					comp_measure(prm, pose, rq, m, P, q);
					tt[0] = q.zh[0]; tt[1] = q.zh[1]; tt[2] = q.zh[2];
					tt[3] = -0.5 * q.Sinv[0];
					tt[4] = -0.5 * (q.Sinv[1] + q.Sinv[3]);
					tt[5] = -0.5 * (q.Sinv[2] + q.Sinv[6]);
					tt[6] = -0.5 * q.Sinv[4];
					tt[7] = -0.5 * (q.Sinv[5] + q.Sinv[7]);
					tt[8] = -0.5 * q.Sinv[8];
					tt[9] = log(q.pd * w * q.qmult);
					wm  = (1 - q.pd) * w;        // component.Reweight((1 - PD) w), :838-839
					mis = !(wm < prm.minw);
				}
				unsigned long long bal = ballot64(mis);
				if (bal) {
					int base = 0, first = __ffsll((long long) bal) - 1;
					if (lane == first) base = atomicAdd(&s_nmis, __popcll(bal));
					base = __shfl(base, first, 64);
					if (mis) {
						int slot = base + __popcll(bal & lanemask_lt());
						if (slot < a.ecap) {
							// (weight and index only: the copy's mean and covariance are the predicted component's, which
							// k_prune_merge reads where they are — 12 bytes instead of 84 per copy)
							size_t e = (size_t) p * a.ecap + slot;
							a.emit_w[e]   = wm;
							a.emit_idx[e] = c;
						}
					}
				}
			}
		}
		__syncthreads();
		load_z();
		const int cend = min(tl, n - c0);
		if (compact) {
			const int nu = s_nu;
			for (int u = wv; u < nu; u += 4) {   // (wave-uniform) this wave alone adds to s_du[u]
				const int k = s_ulist[u];
				const double ux = zmap[k * 3], uy = zmap[k * 3 + 1], uz = zmap[k * 3 + 2];
				double acc = 0;
				for (int cc = lane; cc < cend; cc += 64) {
					const double* tt = tile + cc * SW_REC;
					const double e0 = ux - tt[10], e1 = uy - tt[11], e2 = uz - tt[12];
					const double vm = exp_neg(gauss_logw(tt + 10, e0, e1, e2), etab);
					if (e0 * e0 + e1 * e1 + e2 * e2 <= g2e) acc += vm;
				}
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
				if (lane == 0) s_du[u] += acc;
			}
		}
		auto visit = [&](int cc) {
			const int ccl = HALF ? min(cc + hoff, cend - 1) : cc;   // (HALF: this lane's own component of the visit)
			const bool cval = !HALF || cc + hoff < cend;
			const double* tt = tile + ccl * SW_REC;
#pragma unroll
			for (int b = 0; b < ZB; b++) {
				double e0 = wx[b] - tt[10], e1 = wy[b] - tt[11], e2 = wz[b] - tt[12];
				double sq = e0 * e0 + e1 * e1 + e2 * e2;
				if (!compact && (allpairs || ballot64(sq <= g2e))) {
					// w * N(x; m, P) of the component at MeasureToMap(z), inside the radius gate (Map.cs:214-217)
					double vm = exp_neg(gauss_logw(tt + 10, e0, e1, e2), etab);
					if (zv[b] && sq <= g2e && cval) dens[b] += vm;
				}
				weigh(tt, b, sq, c0 + ccl, cval);
			}
		};
		int cc = wv;
#ifdef PHD_STAMPS
		const long long st_pc = clock64(), st_pr = wall_clock64();
#endif
		if (HALF) {
			for (; cc < cend; cc += 8) visit(cc);   // components cc and cc + 4 at once
		}
		else {
			for (; cc + 4 < cend; cc += 8) { visit(cc); visit(cc + 4); }
			if (cc < cend) visit(cc);
		}
#ifdef PHD_STAMPS
		st_pairc += clock64() - st_pc; st_pairr += wall_clock64() - st_pr;
#endif
		if (!compact) {
#pragma unroll
			for (int b = 0; b < ZB; b++) part2[wv * MP + b * 64 + lane] = dens[b];
		}
		__syncthreads();
		if (!compact) {   // every wave adds up the same partial sums
			int nopen = 0;
			unsigned long long open[ZB];
#pragma unroll
			for (int b = 0; b < ZB; b++) {
				const int k = b * 64 + klane;
				const double d = sumk(part2, k);
				open[b] = ballot64(zv[b] && owner && !(d >= thr));
				nopen += __popcll(open[b]);
			}
			if (nopen <= SW_UMAX) {
				compact = true;
				if (wv == 0) {   // read after the next tile's staging barrier (or the one before the births)
					int base = 0;
#pragma unroll
					for (int b = 0; b < ZB; b++) {
						if ((open[b] >> lane) & 1ull) {
							const int u = base + __popcll(open[b] & lanemask_lt());
							s_ulist[u] = b * 64 + lane;
							s_du[u] = 0.0;
						}
						base += __popcll(open[b]);
					}
					if (lane == 0) s_nu = nopen;
				}
			}
		}
	}

	// ---- births (:806-818): measurements whose density stays below the threshold, in measurement order
#pragma unroll
	for (int b = 0; b < ZB; b++) part2[wv * MP + b * 64 + lane] = dens[b];
	__syncthreads();
	for (int k = tid; k < MP; k += 256) {
		double d = (k < M) ? sumk(part2, k) : 0.0;
		born[k] = (k < M) && !(d >= thr);   // !Explored (:808, :958)
	}
	__syncthreads();
	if (tid < s_nu) {   // the measurements finished component-per-lane: their remaining terms are in s_du
		const int k = s_ulist[tid];
		const double d = sumk(part2, k) + s_du[tid];
		born[k] = !(d >= thr);
	}
	__syncthreads();
	if (tid == 0) {
		int nb = 0;
		for (int k = 0; k < M; k++) {
			if (born[k]) {
				a.born_k[(size_t) p * a.Mcap + nb] = k;
				a.born_mean[((size_t) p * a.Mcap + nb) * 3]     = zmap[k * 3];
				a.born_mean[((size_t) p * a.Mcap + nb) * 3 + 1] = zmap[k * 3 + 1];
				a.born_mean[((size_t) p * a.Mcap + nb) * 3 + 2] = zmap[k * 3 + 2];
				born[nb] = k;   // nb <= k: entries below k are already consumed
				nb++;
			}
		}
		a.born_count[p] = nb;
		s_nb = nb;
	}
	__syncthreads();
	const int nb = s_nb;
	// the births as components n .. n + nb - 1 of the predicted mixture: measured by one thread each (records in LDS),
	// their misdetection copies emitted behind those of the prior components
	const int nmis0 = s_nmis;   // copies of the prior components (every thread reads it before the births add theirs)
	__syncthreads();
	if (tid == 0) s_nmis = 0;
	__syncthreads();
	{
		for (int b0 = 0; b0 < nb; b0 += 256) {
			const int bi = b0 + tid;
			bool mis = false;
			double wm = 0, m[3] = {0, 0, 0};
			if (bi < nb) {
				const int k = born[bi];
				m[0] = zmap[k * 3]; m[1] = zmap[k * 3 + 1]; m[2] = zmap[k * 3 + 2];
				CompMeas q;
				double rq[9];
				rotation(rq);
				comp_measure(prm, pose, rq, m, prm.birthP, q);
				double* tt = tile + bi * 13;
				tt[0] = q.zh[0]; tt[1] = q.zh[1]; tt[2] = q.zh[2];
				tt[3] = -0.5 * q.Sinv[0];
				tt[4] = -0.5 * (q.Sinv[1] + q.Sinv[3]);
				tt[5] = -0.5 * (q.Sinv[2] + q.Sinv[6]);
				tt[6] = -0.5 * q.Sinv[4];
				tt[7] = -0.5 * (q.Sinv[5] + q.Sinv[7]);
				tt[8] = -0.5 * q.Sinv[8];
				tt[9] = log(q.pd * prm.birthw * q.qmult);
				tt[10] = m[0]; tt[11] = m[1]; tt[12] = m[2];
				wm  = (1 - q.pd) * prm.birthw;
				mis = !(wm < prm.minw);
			}
			unsigned long long bal = ballot64(mis);
			if (bal) {
				int base = 0, first = __ffsll((long long) bal) - 1;
				if (lane == first) base = atomicAdd(&s_nmis, __popcll(bal));
				base = __shfl(base, first, 64);
				if (mis) {
					int slot = nmis0 + base + __popcll(bal & lanemask_lt());
					if (slot < a.ecap) {
						size_t e = (size_t) p * a.ecap + slot;
						a.emit_w[e]   = wm;
						a.emit_idx[e] = n + bi;
					}
				}
			}
		}
	}
	__syncthreads();
	load_z();
	// the births' share of the weight sums (they were born from this frame's measurements: :804, :886-890)
	for (int bi = (wv - n) & 3; bi < nb; bi += HALF ? 8 : 4) {   // component n + bi belongs to wave (n + bi) mod 4, as the prior ones do
		const int bil = HALF ? min(bi + hoff, nb - 1) : bi;
		const bool cval = !HALF || bi + hoff < nb;
		const double* tt = tile + bil * 13;
#pragma unroll
		for (int b = 0; b < ZB; b++) {
			double e0 = wx[b] - tt[10], e1 = wy[b] - tt[11], e2 = wz[b] - tt[12];
			weigh(tt, b, e0 * e0 + e1 * e1 + e2 * e2, n + bil, cval);
		}
	}
#pragma unroll
	for (int b = 0; b < ZB; b++) part[wv * MP + b * 64 + lane] = wsum[b];
	__syncthreads();
	for (int k = tid; k < M; k += 256) {
		a.denom[(size_t) p * a.Mcap + k] = prm.kappa + sumk(part, k);
	}
	if (lane == 0) a.cand_count[(size_t) p * 4 + wv] = ncand_w;   // above segcap: the segment overflowed
	if (tid == 0) {
		int ne = nmis0 + s_nmis;
		if (ne > a.ecap) { atomicOr(a.flags, PHD_FLAG_EMIT_OVERFLOW); ne = a.ecap; }
		a.emit_count[p] = ne;
	}
#ifdef PHD_STAMPS
	if (tid == 0 && a.stamps && a.stamp_kernel == 6) {
		double* o = a.stamps + (size_t) p * 16;
		o[0] = 0; o[1] = (double) (clock64() - st_c0); o[2] = (double) (wall_clock64() - st_r0); o[3] = (double) st_pairc; o[4] = (double) st_pairr;
	}
#endif
}

#ifndef PHD_SWEEP_WAVES
#define PHD_SWEEP_WAVES 4
#endif
// (four measurement blocks per lane need 220 registers: two waves per SIMD is what that kernel gets, and what it asks for)
template <int ZB, bool HALF = false>
#ifndef PHD_SWEEP2_WAVES
#define PHD_SWEEP2_WAVES PHD_SWEEP_WAVES   // ... of the two-block kernel (65 - 128 measurements)
#endif
__global__ __launch_bounds__(256, (ZB == 4 ? 2 : (ZB == 2 ? PHD_SWEEP2_WAVES : PHD_SWEEP_WAVES))) void k_sweep(const DevParams prm, const StepBufs a)
{
	__shared__ __align__(16) double pool[SweepLds<ZB>::doubles];
	PHD_TL_BEGIN;
	PHD_SET_PRIO(PHD_DENSE_PRIO);
	sweep_body<ZB, HALF>(prm, a, pool);
	PHD_TL_END(6);
}
