// phd_kernels.h — hand-written HIP kernels (gfx950, wave64) of the RB-PHD-SLAM inner loop.
//
// One workgroup per particle everywhere (particles are independent through predict / correct /
// prune / reweight: PHDNavigator.cs:326-339). Mixtures live in HBM as struct-of-arrays planes
// (w, mean x/y/z, covariance xx/xy/xz/yy/yz/zz), each [particle][slot], so a wavefront reads 64
// consecutive components of one particle as one 512-B line per plane.
//
//   k_predict_correct : PredictConditional + CorrectConditional (+ the MinWeight cut of PruneModel)
//   k_prune_merge     : PruneModel (sort by weight, MaxQuantity cap, greedy merge)
//   k_weight_alpha    : WeightAlpha = BestMapEstimate + mixture densities + SetLogLikelihood
//   k_normalise_resample, k_gather_particles : particle weights, BestParticle, systematic resampling
#pragma once
#include "phd_device.h"

// status bits written by the kernels into StepBufs::flags
#define PHD_FLAG_EMIT_OVERFLOW   1   // corrected components did not fit emit_capacity
#define PHD_FLAG_J_OVERFLOW      2   // map estimate larger than the landmark scratch
#define PHD_FLAG_BIG_CLUSTER     4   // association cluster beyond the on-device solver's cap

struct MixView {
	double* w;
	double* m[3];
	double* P[6];
	int*    count;
};

// one of the three state banks: mixture slabs (10 planes of [Pcap][cap]), counts, poses, weights
struct Bank {
	double* mix;
	int*    count;
	double* poses;    // [Pcap][7]
	double* weights;  // [Pcap]
};

#define SEL_IN  0   // bank a step reads
#define SEL_OUT 1   // bank a step writes
#define SEL_TMP 2   // bank a resampling copy goes to

struct StepBufs {
	int P;          // particles in this launch
	int cap;        // slots per particle in a mixture slab
	int M;          // measurements
	int Mcap;       // stride of per-measurement scratch
	int ecap;       // emit scratch slots per particle
	int Jcap;       // landmark scratch per particle
	size_t plane;   // doubles per plane = Pcap * cap
	Bank bank[3];
	const int* sel; // [3] device-resident roles of the banks for this step (no host round trip to rotate them)
	const double* z;         // [M][3]
	// corrected-but-unpruned components (weight >= MinWeight), unsorted
	double* emit_w;      // [P][ecap]
	int*    emit_idx;    // [P][ecap] canonical position in the reference's `corrected` list
	double* emit_rec;    // [P][ecap][9]  mean, covariance upper triangle
	int*    emit_count;  // [P]
	// births of the predict step
	int*    born_count;  // [P]
	int*    born_k;      // [P][Mcap]
	double* born_mean;   // [P][Mcap][3]
	// reweight outputs
	double* alpha;       // [P]
	double* setll;       // [P]
	int*    flags;       // [1]
	struct MurtyNodes* murty;   // [P] workspace of the big-cluster solver
	double* jscratch;    // [P] landmark-indexed arrays of k_weight_alpha when the map estimate outgrows LDS
};

__device__ __forceinline__ MixView bank_view(const StepBufs& a, int role)
{
	const Bank& b = a.bank[a.sel[role]];
	MixView v;
	v.w = b.mix;
#pragma unroll
	for (int t = 0; t < 3; t++) v.m[t] = b.mix + (size_t) (1 + t) * a.plane;
#pragma unroll
	for (int t = 0; t < 6; t++) v.P[t] = b.mix + (size_t) (4 + t) * a.plane;
	v.count = b.count;
	return v;
}

#define TILE 256   // components staged per LDS tile

// =================================================================================================
// k_predict_correct
//
// LDS: the measurement block (raw + mapped into world space), per-wave partial sums, and one tile
// of per-component quantities that the "measurement-in-lanes" loops read as broadcasts.
// Mapping: per-component work (Jacobian, innovation covariance and its inverse, detection
// probability) is done component-per-lane; every (component, measurement) pair is then visited
// measurement-per-lane with the component broadcast from LDS, so the per-measurement sums
// (explored density, PHD weight sum) are private to a lane and need no cross-lane reduction inside
// the loop; the four waves split the components of a tile and are combined once, in wave order.
// =================================================================================================
template <int ZB>
__global__ __launch_bounds__(256) void k_predict_correct(const DevParams prm, const StepBufs a)
{
	constexpr int MP = ZB * 64;
	extern __shared__ __align__(16) double smem[];
	double* zs    = smem;              // [MP][3]
	double* zmap  = zs + 3 * MP;       // [MP][3]
	double* part  = zmap + 3 * MP;     // [4][MP]
	double* denom = part + 4 * MP;     // [MP]
	double* tile  = denom + MP;        // [18][TILE]
	int*    born  = (int*) (tile + 18 * TILE);   // [MP] flags, then compacted list
	int*    cnt   = born + MP;         // [0] births, [1] emitted

	const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int M = a.M, cap = a.cap;
	const MixView vin = bank_view(a, SEL_IN);
	const Bank& bin  = a.bank[a.sel[SEL_IN]];
	const Bank& bout = a.bank[a.sel[SEL_OUT]];
	const int n = vin.count[p];
	const size_t sb = (size_t) p * cap;
	const PoseD pose = load_pose(bin.poses + (size_t) p * 7);
	double rq[9];
	conj_matrix(pose, rq);
	// the particle keeps its pose and (until the reweight kernel runs) its weight in the output bank
	if (tid < 7) bout.poses[(size_t) p * 7 + tid] = bin.poses[(size_t) p * 7 + tid];
	if (tid == 7) bout.weights[p] = bin.weights[p];

	for (int k = tid; k < MP; k += 256) {
		double z[3] = {0, 0, 1}, x[3] = {0, 0, 0};
		if (k < M) {
			z[0] = a.z[k * 3]; z[1] = a.z[k * 3 + 1]; z[2] = a.z[k * 3 + 2];
			measure_to_map(prm, pose, z, x);
		}
		zs[k * 3] = z[0]; zs[k * 3 + 1] = z[1]; zs[k * 3 + 2] = z[2];
		zmap[k * 3] = x[0]; zmap[k * 3 + 1] = x[1]; zmap[k * 3 + 2] = x[2];
	}
	if (tid < 2) cnt[tid] = 0;
	__syncthreads();

	// the measurements this lane owns
	double zx[ZB], zy[ZB], zr[ZB], wx[ZB], wy[ZB], wz[ZB];
	bool   zv[ZB];
#pragma unroll
	for (int b = 0; b < ZB; b++) {
		int k = b * 64 + lane;
		zv[b] = k < M;
		zx[b] = zs[k * 3]; zy[b] = zs[k * 3 + 1]; zr[b] = zs[k * 3 + 2];
		wx[b] = zmap[k * 3]; wy[b] = zmap[k * 3 + 1]; wz[b] = zmap[k * 3 + 2];
	}

	// ---- PredictConditional: Explored(model, MeasureToMap(z)) on the PRIOR map (PHDNavigator.cs:806-811)
	{
		double acc[ZB];
#pragma unroll
		for (int b = 0; b < ZB; b++) acc[b] = 0;
		bool wavedone = false;
		for (int c0 = 0; c0 < n; c0 += TILE) {
			int c = c0 + tid;
			if (c < n) {
				double P[6], Pi[6], det;
#pragma unroll
				for (int t = 0; t < 6; t++) P[t] = vin.P[t][sb + c];
				inv_sym3(P, Pi, det);
				tile[0 * TILE + tid] = vin.m[0][sb + c];
				tile[1 * TILE + tid] = vin.m[1][sb + c];
				tile[2 * TILE + tid] = vin.m[2][sb + c];
#pragma unroll
				for (int t = 0; t < 6; t++) tile[(3 + t) * TILE + tid] = Pi[t];
				tile[9 * TILE + tid]  = vin.w[sb + c];
				tile[10 * TILE + tid] = PHD_INV_2PI / sqrt(fabs(det));
			}
			__syncthreads();
			int cend = min(TILE, n - c0);
			for (int cc = wv; cc < cend && !wavedone; cc += 4) {
				double m0 = tile[cc], m1 = tile[TILE + cc], m2 = tile[2 * TILE + cc];
				double Pi[6];
#pragma unroll
				for (int t = 0; t < 6; t++) Pi[t] = tile[(3 + t) * TILE + cc];
				double w = tile[9 * TILE + cc], mult = tile[10 * TILE + cc];
				bool open = false;
#pragma unroll
				for (int b = 0; b < ZB; b++) {
					double d0 = wx[b] - m0, d1 = wy[b] - m1, d2 = wz[b] - m2;
					double sq = d0 * d0 + d1 * d1 + d2 * d2;
					if (zv[b] && gate_near(prm.gate_metric, sq, prm.r_explore)) {
						acc[b] += w * (mult * exp(-0.5 * quad_sym(Pi, d0, d1, d2)));   // Map.cs:216
					}
					open |= zv[b] && !(acc[b] >= prm.expl_thr);
				}
				// every term is >= 0: once this wave's partial sum of a measurement reaches the threshold the
				// full sum does too, so a wave whose measurements are all explored can stop (NaNs keep it going)
				wavedone = __ballot(open) == 0;
			}
			if (__syncthreads_and(wavedone)) break;
		}
#pragma unroll
		for (int b = 0; b < ZB; b++) part[wv * MP + b * 64 + lane] = acc[b];
		__syncthreads();
		for (int k = tid; k < MP; k += 256) {
			double dens = part[k] + part[MP + k] + part[2 * MP + k] + part[3 * MP + k];
			born[k] = (k < M) && !(dens >= prm.expl_thr);
		}
		__syncthreads();
		if (tid == 0) {   // births keep measurement order (PHDNavigator.cs:814-816)
			int nb = 0;
			for (int k = 0; k < M; k++) {
				if (born[k]) {
					born[nb] = k;   // nb <= k: in-place compaction
					a.born_k[(size_t) p * a.Mcap + nb] = k;
					a.born_mean[((size_t) p * a.Mcap + nb) * 3]     = zmap[k * 3];
					a.born_mean[((size_t) p * a.Mcap + nb) * 3 + 1] = zmap[k * 3 + 1];
					a.born_mean[((size_t) p * a.Mcap + nb) * 3 + 2] = zmap[k * 3 + 2];
					nb++;
				}
			}
			cnt[0] = nb;
			a.born_count[p] = nb;
		}
		__syncthreads();
	}
	const int np = n + cnt[0];   // predicted = prior + births

	// component c of the predicted mixture
	auto load_comp = [&](int c, double& w, double m[3], double P[6]) {
		if (c < n) {
			w = vin.w[sb + c];
#pragma unroll
			for (int t = 0; t < 3; t++) m[t] = vin.m[t][sb + c];
#pragma unroll
			for (int t = 0; t < 6; t++) P[t] = vin.P[t][sb + c];
		}
		else {
			int k = born[c - n];
			w = prm.birthw;
			m[0] = zmap[k * 3]; m[1] = zmap[k * 3 + 1]; m[2] = zmap[k * 3 + 2];
#pragma unroll
			for (int t = 0; t < 6; t++) P[t] = prm.birthP[t];
		}
	};

	auto emit = [&](bool flag, double w, int idx, const double m[3], const double P[6]) {
		unsigned long long bal = __ballot(flag);
		if (bal == 0) return;
		int base = 0;
		int first = __ffsll((long long) bal) - 1;
		if (lane == first) base = atomicAdd(&cnt[1], __popcll(bal));
		base = __shfl(base, first, 64);
		if (flag) {
			int slot = base + __popcll(bal & lanemask_lt());
			if (slot < a.ecap) {
				size_t e = (size_t) p * a.ecap + slot;
				a.emit_w[e]   = w;
				a.emit_idx[e] = idx;
				double* r = a.emit_rec + e * 9;
				r[0] = m[0]; r[1] = m[1]; r[2] = m[2];
#pragma unroll
				for (int t = 0; t < 6; t++) r[3 + t] = P[t];
			}
		}
	};

	// ---- CorrectConditional, two sweeps over the predicted mixture
	//   sweep 0: misdetection copies (:837-840) and weightsum[z] = sum over near components of PD w q(z) (:886-890)
	//   sweep 1: emission of w' = PD w q / (kappa + weightsum) with m', P' (:892-902), for w' >= MinWeight
	double wsum[ZB];
#pragma unroll
	for (int b = 0; b < ZB; b++) wsum[b] = 0;

	for (int sweep = 0; sweep < 2; sweep++) {
		for (int c0 = 0; c0 < np; c0 += TILE) {
			int  c = c0 + tid;
			bool valid = c < np;
			bool mis = false;
			double w = 0, m[3] = {0, 0, 0}, P[6] = {1, 0, 0, 1, 0, 1}, wm = 0;
			if (valid) {
				load_comp(c, w, m, P);
				CompMeas cm;
				comp_measure(prm, pose, rq, m, P, cm);
#pragma unroll
				for (int t = 0; t < 3; t++) tile[t * TILE + tid] = cm.zh[t];
#pragma unroll
				for (int t = 0; t < 9; t++) tile[(3 + t) * TILE + tid] = cm.Sinv[t];
				double pdw = cm.pd * w;
				tile[12 * TILE + tid] = cm.qmult;
				tile[13 * TILE + tid] = pdw;
#pragma unroll
				for (int t = 0; t < 3; t++) tile[(14 + t) * TILE + tid] = m[t];
				// no pair can reach MinWeight unless  PD w mult exp(-d2/2) >= MinWeight * kappa
				double dc = 2.0 * (log(pdw * cm.qmult) - prm.emit_log_floor) + 1.0;
				tile[17 * TILE + tid] = isinf(prm.emit_log_floor) ? INFINITY : dc;
				wm  = (1 - cm.pd) * w;
				mis = !(wm < prm.minw);
			}
			if (sweep == 0) emit(mis, wm, c, m, P);
			__syncthreads();
			int cend = min(TILE, np - c0);
			for (int cc = wv; cc < cend; cc += 4) {
				double zh0 = tile[cc], zh1 = tile[TILE + cc], zh2 = tile[2 * TILE + cc];
				double Si[9];
#pragma unroll
				for (int t = 0; t < 9; t++) Si[t] = tile[(3 + t) * TILE + cc];
				double qmult = tile[12 * TILE + cc], pdw = tile[13 * TILE + cc];
				double m0 = tile[14 * TILE + cc], m1 = tile[15 * TILE + cc], m2 = tile[16 * TILE + cc];
				double dc = tile[17 * TILE + cc];
				if (sweep == 0) {
#pragma unroll
					for (int b = 0; b < ZB; b++) {
						double e0 = wx[b] - m0, e1 = wy[b] - m1, e2 = wz[b] - m2;
						double sq = e0 * e0 + e1 * e1 + e2 * e2;
						double d2 = quad_gen(Si, zx[b] - zh0, zy[b] - zh1, zr[b] - zh2);
						double q  = qmult * exp(-0.5 * d2);
						if (zv[b] && gate_near(prm.gate_metric, sq, prm.r_correct)) {
							wsum[b] += pdw * q;
						}
					}
				}
				else {
#pragma unroll
					for (int b = 0; b < ZB; b++) {
						double e0 = wx[b] - m0, e1 = wy[b] - m1, e2 = wz[b] - m2;
						double sq = e0 * e0 + e1 * e1 + e2 * e2;
						double n0 = zx[b] - zh0, n1 = zy[b] - zh1, n2 = zr[b] - zh2;
						double d2 = quad_gen(Si, n0, n1, n2);
						bool cand = zv[b] && gate_near(prm.gate_metric, sq, prm.r_correct) && (d2 <= dc);
						if (__ballot(cand)) {
							double q   = qmult * exp(-0.5 * d2);
							double wgt = pdw * q / denom[b * 64 + lane];   // PHDNavigator.cs:899
							bool   em  = cand && !(wgt < prm.minw);
							if (__ballot(em)) {
								// the component is wave-uniform: every lane rebuilds its gain and posterior covariance
								int cg = c0 + cc;
								double cw, cmn[3], cP[6];
								load_comp(cg, cw, cmn, cP);
								CompMeas cm;
								comp_measure(prm, pose, rq, cmn, cP, cm);
								double K[9], Pn[6], mn[3];
								kalman_gain(cm, K);
								kalman_cov(cm, K, cP, Pn);
#pragma unroll
								for (int t = 0; t < 3; t++) {
									mn[t] = cmn[t] + (K[t * 3] * n0 + K[t * 3 + 1] * n1 + K[t * 3 + 2] * n2);
								}
								int k = b * 64 + lane;
								emit(em, wgt, np + k * np + cg, mn, Pn);
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
	if (tid == 0) {
		int ne = cnt[1];
		if (ne > a.ecap) {
			atomicOr(a.flags, PHD_FLAG_EMIT_OVERFLOW);
			ne = a.ecap;
		}
		a.emit_count[p] = ne;
	}
}

#include "phd_prune.h"

#include "phd_alpha.h"

#include "phd_resample.h"

// Deep copy of the resampled particles (PHDNavigator.cs:740-741) and rotation of the bank roles for the
// next step, decided on the device from the resampling flag so the host never waits inside a step.
//   not resampled: the new state is the OUT bank      -> next roles (IN, OUT, TMP) = (OUT, TMP, IN)
//   resampled    : particle i <- OUT[src[first + i] - first] written to TMP -> next roles = (TMP, IN, OUT)
//   frozen       : roles stay (benchmark steady state)
__global__ __launch_bounds__(256) void k_gather_rotate(const StepBufs a, const int* src, const int* info, int first,
                                                       int* sel_next, int frozen)
{
	const int i = blockIdx.x, tid = threadIdx.x;
	const int resampled = info[1];
	const int I = a.sel[SEL_IN], O = a.sel[SEL_OUT], T = a.sel[SEL_TMP];
	if (i == 0 && tid == 0) {
		if (frozen)         { sel_next[0] = I; sel_next[1] = O; sel_next[2] = T; }
		else if (resampled) { sel_next[0] = T; sel_next[1] = I; sel_next[2] = O; }
		else                { sel_next[0] = O; sel_next[1] = T; sel_next[2] = I; }
		sel_next[3] = resampled ? T : O;   // where the result of this step lives
	}
	if (!resampled) return;
	const MixView from = bank_view(a, SEL_OUT), dst = bank_view(a, SEL_TMP);
	const int s = src[first + i] - first;
	const int n = from.count[s];
	const size_t db = (size_t) i * a.cap, fb = (size_t) s * a.cap;
	for (int c = tid; c < n; c += 256) {
		dst.w[db + c] = from.w[fb + c];
#pragma unroll
		for (int t = 0; t < 3; t++) dst.m[t][db + c] = from.m[t][fb + c];
#pragma unroll
		for (int t = 0; t < 6; t++) dst.P[t][db + c] = from.P[t][fb + c];
	}
	if (tid == 0) {
		dst.count[i] = n;
		a.bank[T].weights[i] = a.bank[O].weights[i];
	}
	if (tid < 7) a.bank[T].poses[(size_t) i * 7 + tid] = a.bank[O].poses[(size_t) s * 7 + tid];
}

// replicate particle 0 of the IN bank over `P` particles of the OUT bank (PHDNavigator.reset, :256-263)
__global__ __launch_bounds__(256) void k_replicate(const StepBufs a, double weight)
{
	const int i = blockIdx.x, tid = threadIdx.x;
	const MixView from = bank_view(a, SEL_IN), dst = bank_view(a, SEL_OUT);
	const int n = from.count[0];
	const size_t db = (size_t) i * a.cap;
	for (int c = tid; c < n; c += 256) {
		dst.w[db + c] = from.w[c];
#pragma unroll
		for (int t = 0; t < 3; t++) dst.m[t][db + c] = from.m[t][c];
#pragma unroll
		for (int t = 0; t < 6; t++) dst.P[t][db + c] = from.P[t][c];
	}
	const Bank& bi = a.bank[a.sel[SEL_IN]];
	const Bank& bo = a.bank[a.sel[SEL_OUT]];
	if (tid == 0) {
		dst.count[i]  = n;
		bo.weights[i] = weight;
	}
	if (tid < 7) bo.poses[(size_t) i * 7 + tid] = bi.poses[tid];
}

// =================================================================================================
// multi-GPU resampling: particles are sharded contiguously over ranks; after the global resample a
// slot may need a particle that lives on another rank. A migrating particle travels as one
// fixed-size record: [count, pose(7), planes(10 x cap)] doubles.
// =================================================================================================
__global__ __launch_bounds__(256) void k_scatter_weights(const StepBufs a, const double* gw, int first)
{
	int i = blockIdx.x * 256 + threadIdx.x;
	if (i < a.P) a.bank[a.sel[SEL_OUT]].weights[i] = gw[first + i];
}

__global__ __launch_bounds__(256) void k_pack_particles(const StepBufs a, const int* sendlist, double* sendbuf)
{
	const int r = blockIdx.x, tid = threadIdx.x;
	const int s = sendlist[r];
	const MixView from = bank_view(a, SEL_OUT);
	const Bank& bo = a.bank[a.sel[SEL_OUT]];
	const size_t rec = (size_t) 8 + (size_t) 10 * a.cap;
	double* o = sendbuf + (size_t) r * rec;
	const int n = from.count[s];
	if (tid == 0) o[0] = (double) n;
	if (tid < 7) o[1 + tid] = bo.poses[(size_t) s * 7 + tid];
	const size_t fb = (size_t) s * a.cap;
	for (int c = tid; c < n; c += 256) {
		o[8 + c] = from.w[fb + c];
#pragma unroll
		for (int t = 0; t < 3; t++) o[8 + (size_t) (1 + t) * a.cap + c] = from.m[t][fb + c];
#pragma unroll
		for (int t = 0; t < 6; t++) o[8 + (size_t) (4 + t) * a.cap + c] = from.P[t][fb + c];
	}
}

// dstsrc[i] >= 0: local source slot in the OUT bank; < 0: record -(dstsrc[i] + 1) of the receive buffer
__global__ __launch_bounds__(256) void k_unpack_gather(const StepBufs a, const int* dstsrc, const double* recvbuf,
                                                       double weight, int* sel_next, int frozen)
{
	const int i = blockIdx.x, tid = threadIdx.x;
	const int I = a.sel[SEL_IN], O = a.sel[SEL_OUT], T = a.sel[SEL_TMP];
	if (i == 0 && tid == 0) {
		if (frozen) { sel_next[0] = I; sel_next[1] = O; sel_next[2] = T; }
		else        { sel_next[0] = T; sel_next[1] = I; sel_next[2] = O; }
		sel_next[3] = T;
	}
	const MixView from = bank_view(a, SEL_OUT), dst = bank_view(a, SEL_TMP);
	const size_t db = (size_t) i * a.cap;
	const int code = dstsrc[i];
	if (code >= 0) {
		const int n = from.count[code];
		const size_t fb = (size_t) code * a.cap;
		for (int c = tid; c < n; c += 256) {
			dst.w[db + c] = from.w[fb + c];
#pragma unroll
			for (int t = 0; t < 3; t++) dst.m[t][db + c] = from.m[t][fb + c];
#pragma unroll
			for (int t = 0; t < 6; t++) dst.P[t][db + c] = from.P[t][fb + c];
		}
		if (tid == 0) dst.count[i] = n;
		if (tid < 7) a.bank[T].poses[(size_t) i * 7 + tid] = a.bank[O].poses[(size_t) code * 7 + tid];
	}
	else {
		const size_t rec = (size_t) 8 + (size_t) 10 * a.cap;
		const double* r = recvbuf + (size_t) (-(code + 1)) * rec;
		const int n = (int) r[0];
		for (int c = tid; c < n; c += 256) {
			dst.w[db + c] = r[8 + c];
#pragma unroll
			for (int t = 0; t < 3; t++) dst.m[t][db + c] = r[8 + (size_t) (1 + t) * a.cap + c];
#pragma unroll
			for (int t = 0; t < 6; t++) dst.P[t][db + c] = r[8 + (size_t) (4 + t) * a.cap + c];
		}
		if (tid == 0) dst.count[i] = n;
		if (tid < 7) a.bank[T].poses[(size_t) i * 7 + tid] = r[1 + tid];
	}
	if (tid == 0) a.bank[T].weights[i] = weight;
}
