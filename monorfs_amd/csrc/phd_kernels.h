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

// =================================================================================================
// k_weight_alpha — WeightAlpha (PHDNavigator.cs:373-393)
// =================================================================================================

// next permutation of LexicographicalPairing (GraphCombinatorics.cs:306-332) on n <= 5 entries
__device__ __forceinline__ bool lex_last(const int* perm, int n)
{
	for (int i = 1; i < n; i++) {
		if (perm[i - 1] < perm[i]) return false;
	}
	return true;
}

__device__ __forceinline__ void lex_reverse(int* perm, int from, int to)   // [from, to)
{
	for (int i = from, j = to - 1; i < j; i++, j--) {
		int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
	}
}

__device__ __forceinline__ void lex_next(int* perm, int n, int measurestart)
{
	int x, y;
	for (x = n - 2; x > 0; x--) {
		if (perm[x] < perm[x + 1]) break;
	}
	for (y = n - 1; y > x; y--) {
		if (perm[x] < perm[y]) break;
	}
	int t = perm[x]; perm[x] = perm[y]; perm[y] = t;
	lex_reverse(perm, x + 1, n);
	lex_reverse(perm, measurestart, n);
}

// log-sum-exp over every pairing of a cluster with n <= 5 rows, enumerated exactly like
// LexicographicalPairing(component, map.Count) (`modelsize` is compared with COMPACTED row indices,
// PHDNavigator.cs:493 / GraphCombinatorics.cs:293-299). mat: n x n, row stride 5, stride `ms` between entries.
__device__ double cluster_enumerate(const double* mat, int ms, int n, int modelsize, double* rec = nullptr)
{
	int perm[5], first[5];
	int measurestart = n;
	for (int i = 0; i < n; i++) {
		if (i >= modelsize) { measurestart = i; break; }
	}
	for (int i = 0; i < n; i++) first[i] = i;
	lex_reverse(first, measurestart, n);

	double mx = -INFINITY, value = 0;
	for (int pass = 0; pass < 2; pass++) {
		for (int i = 0; i < n; i++) perm[i] = first[i];
		int m = 0;
		for (;;) {
			double v = 0;
			for (int i = 0; i < n; i++) v += mat[(i * 5 + perm[i]) * ms];   // AssignmentValue
			if (pass == 0) {
				mx = fmax(mx, v);
				if (rec) rec[m] = v;   // logcomp[m] = assignment.Item2 (PHDNavigator.cs:507)
			}
			else value += exp(v - mx);
			m++;
			if (lex_last(perm, n)) break;
			lex_next(perm, n, measurestart);
		}
		if (pass == 0 && isinf(mx) && mx < 0) return -INFINITY;   // LogSumExp, MatrixExtensions.cs:379-381
	}
	return mx + log(value);
}

// ---- clusters with more than 5 rows: MurtyPairing (GraphCombinatorics.cs:241-272) run by one wave ----
#define MURTY_NMAX 32     // rows of the largest cluster solved on the device
#define MURTY_OUT  200    // logcomp.Length (PHDNavigator.cs:469)
#define MURTY_POOL 208    // frontier (<= 201 live entries) + the node being expanded + its child

struct MurtyNodes {       // per-particle workspace in HBM, touched only when such a cluster exists
	unsigned char asg[MURTY_POOL][MURTY_NMAX];    // assignment (row -> column)
	unsigned int  elim[MURTY_POOL][MURTY_NMAX];   // eliminated columns of every row, as a bit mask
	unsigned int  forced[MURTY_POOL];             // forced rows; a forced edge is (row, asg[row])
};

__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
	return v;
}

__device__ __forceinline__ unsigned int wave_or(unsigned int v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v |= (unsigned int) __shfl_xor((int) v, o, 64);
	return v;
}

__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }

// Hungarian (GraphCombinatorics.cs:64-175) with lane i owning row i (labelx, matchx, visitx) and
// column i (labely, matchy, visity, slack, parent). Every arithmetic step is the serial algorithm's,
// the argmin keeps its first-minimum tie-break, so the assignment is the reference's.
__device__ bool wave_hungarian(const double* mat, int n, int lane, int& matchx_out)
{
	const bool active = lane < n;
	double labelx = 0, labely = 0, slack = INFINITY;
	int matchx = -1, matchy = -1, parent = 0;
	if (active) {
		double f = 0;   // FoldRows(Math.Max, 0)
		for (int k = 0; k < n; k++) f = fmax(f, mat[lane * n + k]);
		labelx = f;
	}
	for (;;) {
		unsigned long long um = __ballot(active && matchx == -1);
		if (!um) break;
		const int root = __ffsll((long long) um) - 1;
		const double lxr = __shfl(labelx, root, 64);
		parent = root;
		slack = active ? (lxr + labely - mat[root * n + lane]) : INFINITY;
		bool visitx = lane == root, visity = false;
		int imin = 0;
		for (;;) {
			double val = (active && !visity) ? slack : INFINITY;
			double delta = wave_min(val);
			if (isinf(delta) && delta > 0) return false;   // no solution
			imin = __ffsll((long long) __ballot(active && !visity && val == delta)) - 1;
			if (visitx) labelx -= delta;
			if (active) {
				if (visity) labely += delta;
				else slack -= delta;
			}
			if (lane == imin) visity = true;
			const int my = __shfl(matchy, imin, 64);
			if (my == -1) break;
			if (lane == my) visitx = true;
			const double lxm = __shfl(labelx, my, 64);
			if (active && !visity) {
				double md = lxm + labely - mat[my * n + lane];
				if (md < slack) { slack = md; parent = my; }
			}
		}
		int py = imin, px = __shfl(parent, py, 64);
		while (px != root) {
			int ty = __shfl(matchx, px, 64);
			if (lane == px) matchx = py;
			if (lane == py) matchy = px;
			py = ty;
			px = __shfl(parent, py, 64);
		}
		if (lane == px) matchx = py;
		if (lane == py) matchy = px;
	}
	matchx_out = matchx;
	return true;
}

// AssignmentValue (GraphCombinatorics.cs:183-197), summed in row order
__device__ __forceinline__ double wave_assignment_value(const double* profit, int n, int matchx)
{
	double total = 0;
	for (int i = 0; i < n; i++) total += profit[i * n + __shfl(matchx, i, 64)];
	return total;
}

// LDS scratch of the Murty path
struct MurtyLds {
	double* profit;    // [NMAX*NMAX]
	double* reduced;   // [NMAX*NMAX]
	double* logcomp;   // [MURTY_OUT]
	double* fkey;      // [MURTY_POOL] frontier priorities, ascending
	int*    fnode;     // [MURTY_POOL] frontier node slots
	int*    freelist;  // [MURTY_POOL]
	int*    L;         // [NMAX]
	int*    Z;         // [NMAX]
};
#define MURTY_LDS_DOUBLES (2 * MURTY_NMAX * MURTY_NMAX + MURTY_OUT + MURTY_POOL + (2 * MURTY_POOL + 2 * MURTY_NMAX + 1) / 2)

// Enumerate the pairings of one cluster best-first and record their values into logcomp exactly as
// the loop of SetLogLikelihood does (PHDNavigator.cs:501-509), including its read of the stale
// logcomp[m] left by earlier clusters. Returns the number of values written. Wave-uniform.
__device__ int wave_murty(const MurtyLds& ws, MurtyNodes* nodes, int n, int lane)
{
	const double* profit = ws.profit;
	int nfree = 0;
	if (lane == 0) {
		for (int i = 0; i < MURTY_POOL; i++) ws.freelist[i] = MURTY_POOL - 1 - i;
	}
	nfree = MURTY_POOL;
	lds_fence();
	int fsize = 0, m = 0;

	// frontier.Add(priority, node): keep ascending order, a new entry goes after its equals
	// (PriorityQueue.Add re-sorts the list, GraphCombinatorics.cs:638-642; canonical stable order);
	// only the best (MURTY_OUT + 1 - m) entries can ever be popped, the rest is dropped.
	auto frontier_add = [&](double key, int slot) {
		int cap = MURTY_OUT + 1 - m;
		int pos = 0;
		for (int b = 0; b < MURTY_POOL; b += 64) {
			int idx = b + lane;
			pos += __popcll(__ballot(idx < fsize && ws.fkey[idx] <= key));
		}
		if (fsize >= cap && pos == 0) {   // would be the worst of a full frontier
			if (lane == 0) ws.freelist[nfree] = slot;
			nfree++;
			lds_fence();
			return;
		}
		// shift [pos, fsize) up by one
		double kreg[4]; int nreg[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			int idx = u * 64 + lane;
			kreg[u] = (idx < fsize) ? ws.fkey[idx] : 0.0;
			nreg[u] = (idx < fsize) ? ws.fnode[idx] : 0;
		}
		lds_fence();
#pragma unroll
		for (int u = 0; u < 4; u++) {
			int idx = u * 64 + lane;
			if (idx >= pos && idx < fsize) { ws.fkey[idx + 1] = kreg[u]; ws.fnode[idx + 1] = nreg[u]; }
		}
		if (lane == 0) { ws.fkey[pos] = key; ws.fnode[pos] = slot; }
		fsize++;
		lds_fence();
		if (fsize > cap) {   // drop the worst (front)
			int dropped = ws.fnode[0];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				int idx = u * 64 + lane;
				kreg[u] = (idx < fsize) ? ws.fkey[idx] : 0.0;
				nreg[u] = (idx < fsize) ? ws.fnode[idx] : 0;
			}
			lds_fence();
#pragma unroll
			for (int u = 0; u < 4; u++) {
				int idx = u * 64 + lane;
				if (idx >= 1 && idx < fsize) { ws.fkey[idx - 1] = kreg[u]; ws.fnode[idx - 1] = nreg[u]; }
			}
			fsize--;
			if (lane == 0) ws.freelist[nfree] = dropped;
			nfree++;
			lds_fence();
		}
	};
	auto alloc = [&]() {
		nfree--;
		return ws.freelist[nfree];
	};

	// first node: no forced, no eliminated edges
	{
		int slot = alloc();
		int mx;
		bool solved = wave_hungarian(profit, n, lane, mx);
		if (lane < n) {
			nodes->asg[slot][lane]  = (unsigned char) (solved ? mx : 0);
			nodes->elim[slot][lane] = 0;
		}
		if (lane == 0) nodes->forced[slot] = 0;
		__threadfence_block();
		double value = solved ? wave_assignment_value(profit, n, mx) : -INFINITY;
		// an unsolved first node is yielded with value -inf and has no children (GraphCombinatorics.cs:245-249,473)
		frontier_add(value, solved ? slot : (slot | 0x10000));
	}

	while (fsize > 0) {
		// best = frontier.Pop(out value)
		const double value = ws.fkey[fsize - 1];
		const int    code  = ws.fnode[fsize - 1];
		fsize--;
		// foreach body of SetLogLikelihood (PHDNavigator.cs:502-509)
		if (m >= MURTY_OUT || (ws.logcomp[m] - ws.logcomp[0] < -10)) break;
		if (lane == 0) ws.logcomp[m] = value;
		m++;
		lds_fence();
		if (code & 0x10000) continue;   // unsolved: no children
		const int slot = code;
		// children (MurtyNode.Children, GraphCombinatorics.cs:469-509)
		const int pa = (lane < n) ? nodes->asg[slot][lane] : 0;
		const unsigned int pelim = (lane < n) ? nodes->elim[slot][lane] : 0u;
		const unsigned int pforced = nodes->forced[slot];
		const unsigned int rowsmask = (n >= 32) ? 0xffffffffu : ((1u << n) - 1u);
		unsigned int remaining = ~pforced & rowsmask;
		const int R = __popc(remaining);
		unsigned int extra = 0;   // rows forced on top of the parent's: remaining[0 .. c-1]
		for (int c = 0; c < R - 1; c++) {
			const int er = __ffs((int) remaining) - 1;   // remaining[c]
			remaining &= remaining - 1;
			const int ec = __shfl(pa, er, 64);
			const unsigned int cforced = pforced | extra;
			// reduceprofit (GraphCombinatorics.cs:206-234)
			unsigned int fcols = wave_or((lane < n && ((cforced >> lane) & 1u)) ? (1u << pa) : 0u);
			for (int e0 = 0; e0 < n * n; e0 += 64) {   // uniform trip count: every lane takes part in the shuffle
				const int  e  = e0 + lane;
				const bool in = e < n * n;
				const int  i  = in ? e / n : 0, k = e - i * n;
				const int  pai = __shfl(pa, i, 64);
				if (in) {
					double v = profit[e];
					if ((cforced >> i) & 1u) v = (k == pai) ? 1.0 : -INFINITY;
					else if ((fcols >> k) & 1u) v = -INFINITY;
					ws.reduced[e] = v;
				}
			}
			lds_fence();
			// eliminated edges: the parent's and (er, ec)
			if (lane < n) {
				unsigned int em = pelim | ((lane == er) ? (1u << ec) : 0u);
				while (em) {
					int k = __ffs((int) em) - 1;
					em &= em - 1;
					ws.reduced[lane * n + k] = -INFINITY;
				}
			}
			lds_fence();
			int mx;
			bool solved = wave_hungarian(ws.reduced, n, lane, mx);
			if (solved) {
				int cs = alloc();
				if (lane < n) {
					nodes->asg[cs][lane]  = (unsigned char) mx;
					nodes->elim[cs][lane] = pelim | ((lane == er) ? (1u << ec) : 0u);
				}
				if (lane == 0) nodes->forced[cs] = cforced;
				__threadfence_block();
				frontier_add(wave_assignment_value(profit, n, mx), cs);
			}
			extra |= 1u << er;
		}
		if (lane == 0) ws.freelist[nfree] = slot;
		nfree++;
		lds_fence();
	}
	return m;
}

// LDS layout of k_weight_alpha, shared with the host so the launch sizes it identically.
struct AlphaLds {
	int zs, lm, red, pick, scr;   // offsets in doubles
	int bytes;
};

__host__ __device__ inline AlphaLds alpha_lds(int MP, int JP, int ncap)
{
	AlphaLds l;
	l.zs   = 0;
	l.lm   = l.zs + 3 * MP;
	l.red  = l.lm + 3 * JP;
	l.pick = l.red + 256;
	l.scr  = l.pick + (JP + 1) / 2;
	int MW = MP / 64;
	int ph1 = 2 * ncap + JP + (ncap + JP + 1) / 2;                       // keyw, sortw, dw | sortsrc, dsrc
	int ph2 = 11 * TILE + 256;                                           // tile | part
	int ph3 = 5 * JP + JP * MW + 25 * 64 + (2 * JP + MP + 1) / 2          // zh, pdj, res | adj | mats | labl, roots, labz
	          + MURTY_LDS_DOUBLES + 2;                                   // big-cluster solver
	int mx = ph1 > ph2 ? ph1 : ph2;
	mx = mx > ph3 ? mx : ph3;
	l.bytes = (l.scr + mx) * 8;
	return l;
}

template <int ZB>
__global__ __launch_bounds__(256) void k_weight_alpha(const DevParams prm, const StepBufs a, int ncap)
{
	constexpr int MP = ZB * 64;
	constexpr int MW = ZB;   // 64-bit adjacency words per landmark
	const int JP = a.Jcap;   // multiple of 64
	extern __shared__ __align__(16) double smem[];
	const AlphaLds lay = alpha_lds(MP, JP, ncap);
	double* zs   = smem + lay.zs;          // [MP][3] measurements
	double* lm   = smem + lay.lm;          // [3][JP] landmark means of the map estimate
	double* red  = smem + lay.red;         // [256] reduction scratch
	int*    pick = (int*) (smem + lay.pick);   // [JP] component picked for landmark j
	double* scr  = smem + lay.scr;         // per-phase scratch
	__shared__ int s_J, s_changed, s_nroots, s_big;
	__shared__ double s_ccount, s_total;

	const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int M = a.M, cap = a.cap;
	const MixView vin = bank_view(a, SEL_IN), vout = bank_view(a, SEL_OUT);
	const Bank& bin  = a.bank[a.sel[SEL_IN]];
	const Bank& bout = a.bank[a.sel[SEL_OUT]];
	const int n = vin.count[p], nb = a.born_count[p], no = vout.count[p];
	const int np = n + nb;
	const size_t sbi = (size_t) p * cap, sbo = (size_t) p * cap;
	const PoseD pose = load_pose(bin.poses + (size_t) p * 7);

	for (int k = tid; k < MP * 3; k += 256) zs[k] = (k < M * 3) ? a.z[k] : 0.0;

	// ---- phase 1: BestMapEstimate (Map.cs:119-142)
	{
		double* keyw    = scr;               // [ncap] weights in map order
		double* sortw   = keyw + ncap;       // [ncap] weights, stable descending
		double* dw      = sortw + ncap;      // [JP]   derived (w - 1) entries, FIFO
		int*    sortsrc = (int*) (dw + JP);  // [ncap]
		int*    dsrc    = sortsrc + ncap;    // [JP]
		for (int c = tid; c < no; c += 256) keyw[c] = vout.w[sbo + c];
		__syncthreads();
		if (tid == 0) {   // ExpectedSize, summed in map order (Map.cs:61-71); size = (int) ExpectedSize (:126)
			double e = 0;
			for (int c = 0; c < no; c++) e += keyw[c];
			s_ccount = e;
			int J = (int) e;
			if (J < 0) J = 0;
			if (J > JP) {
				atomicOr(a.flags, PHD_FLAG_J_OVERFLOW);
				J = JP;
			}
			s_J = J;
		}
		// stable descending order by counting rank (mlist.Sort, :129)
		for (int c = tid; c < no; c += 256) {
			double wc = keyw[c];
			int rank = 0;
			for (int j = 0; j < no; j++) {
				double wj = keyw[j];
				rank += (wj > wc) || (wj == wc && j < c);
			}
			sortw[rank]   = wc;
			sortsrc[rank] = c;
		}
		__syncthreads();
		if (tid == 0) {
			// "take the i-th entry, append a copy with w - 1, sort again" (:131-138) is a two-way merge:
			// every appended weight is <= the one it came from, so the appended entries are produced in
			// non-increasing order and form a FIFO merged with the original sorted list; on a tie the
			// original entry stays first (stable sort of an appended element).
			const int J = s_J;
			int ia = 0, id = 0, nd = 0;
			for (int j = 0; j < J; j++) {
				bool takeorig;
				if (ia < no && id < nd) takeorig = !(dw[id] > sortw[ia]);
				else takeorig = ia < no;
				double wpick;
				int    src;
				if (takeorig) { wpick = sortw[ia]; src = sortsrc[ia]; ia++; }
				else          { wpick = dw[id];    src = dsrc[id];    id++; }
				dw[nd]   = wpick - 1;
				dsrc[nd] = src;
				nd++;
				pick[j] = src;
			}
		}
		__syncthreads();
	}
	const int J = s_J;
	for (int j = tid; j < J; j += 256) {
		int c = pick[j];
		lm[j] = vout.m[0][sbo + c]; lm[JP + j] = vout.m[1][sbo + c]; lm[2 * JP + j] = vout.m[2][sbo + c];
	}
	__syncthreads();

	// ---- phase 2: sum_j log v_pred(m_j), sum_j log v_corr(m_j) with v = full ungated mixture density (Map.cs:192-202)
	double plog_part = 0, clog_part = 0, pcount_part = 0;
	{
		double* tile = scr;                 // [11][TILE]
		double* part = tile + 11 * TILE;    // [4][64]
		for (int c = tid; c < n; c += 256) pcount_part += vin.w[sbi + c];
		for (int jb = 0; jb * 64 < J; jb++) {
			int  j  = jb * 64 + lane;
			bool jv = j < J;
			double x0 = jv ? lm[j] : 0, x1 = jv ? lm[JP + j] : 0, x2 = jv ? lm[2 * JP + j] : 0;
			for (int src = 0; src < 2; src++) {
				const int total = (src == 0) ? np : no;
				double acc = 0;
				for (int c0 = 0; c0 < total; c0 += TILE) {
					int c = c0 + tid;
					if (c < total) {
						double w, m[3], P[6], Pi[6], det;
						if (src == 1) {
							w = vout.w[sbo + c];
#pragma unroll
							for (int t = 0; t < 3; t++) m[t] = vout.m[t][sbo + c];
#pragma unroll
							for (int t = 0; t < 6; t++) P[t] = vout.P[t][sbo + c];
						}
						else if (c < n) {
							w = vin.w[sbi + c];
#pragma unroll
							for (int t = 0; t < 3; t++) m[t] = vin.m[t][sbi + c];
#pragma unroll
							for (int t = 0; t < 6; t++) P[t] = vin.P[t][sbi + c];
						}
						else {
							const double* bm = a.born_mean + ((size_t) p * a.Mcap + (c - n)) * 3;
							w = prm.birthw;
							m[0] = bm[0]; m[1] = bm[1]; m[2] = bm[2];
#pragma unroll
							for (int t = 0; t < 6; t++) P[t] = prm.birthP[t];
						}
						inv_sym3(P, Pi, det);
#pragma unroll
						for (int t = 0; t < 3; t++) tile[t * TILE + tid] = m[t];
#pragma unroll
						for (int t = 0; t < 6; t++) tile[(3 + t) * TILE + tid] = Pi[t];
						tile[9 * TILE + tid]  = w;
						tile[10 * TILE + tid] = PHD_INV_2PI / sqrt(fabs(det));
					}
					__syncthreads();
					int cend = min(TILE, total - c0);
					for (int cc = wv; cc < cend; cc += 4) {
						double d0 = x0 - tile[cc], d1 = x1 - tile[TILE + cc], d2 = x2 - tile[2 * TILE + cc];
						double Pi[6];
#pragma unroll
						for (int t = 0; t < 6; t++) Pi[t] = tile[(3 + t) * TILE + cc];
						acc += tile[9 * TILE + cc] * (tile[10 * TILE + cc] * exp(-0.5 * quad_sym(Pi, d0, d1, d2)));
					}
					__syncthreads();
				}
				part[wv * 64 + lane] = acc;
				__syncthreads();
				if (wv == 0 && jv) {
					double v = part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane];
					if (src == 0) plog_part += log(v);
					else          clog_part += log(v);
				}
				__syncthreads();
			}
		}
	}
	// block reductions (fixed order)
	auto block_sum = [&](double v) {
		red[tid] = v;
		__syncthreads();
		for (int s = 128; s > 0; s >>= 1) {
			if (tid < s) red[tid] += red[tid + s];
			__syncthreads();
		}
		double r = red[0];
		__syncthreads();
		return r;
	};
	const double plog = block_sum(plog_part);
	const double clog = block_sum(clog_part);
	const double pcount = block_sum(pcount_part) + nb * prm.birthw;
	const double ccount = s_ccount;

	// ---- phase 3: SetLogLikelihood (PHDNavigator.cs:462-515) on the matrix of SetLogLikeMatrix (:415-453)
	{
		double* zh   = scr;                 // [3][JP] h(m_j)
		double* pdj  = zh + 3 * JP;         // [JP] detection probability of landmark j
		double* res  = pdj + JP;            // [JP] per-cluster log-sum-exp, in cluster order
		unsigned long long* adj = (unsigned long long*) (res + JP);   // [JP][MW] gated measurements of landmark j
		double* mats = (double*) (adj + (size_t) JP * MW);             // [25][64] one 5x5 matrix per lane
		int*    labl = (int*) (mats + 25 * 64);   // [JP]
		int*    roots = labl + JP;                // [JP]
		int*    labz = roots + JP;                // [MP]

		for (int j = tid; j < J; j += 256) {
			double m[3] = {lm[j], lm[JP + j], lm[2 * JP + j]}, z[3], l[3];
			measure_perfect(prm, pose, m, z, l);
			zh[j] = z[0]; zh[JP + j] = z[1]; zh[2 * JP + j] = z[2];
			pdj[j] = detection_probability_m(prm, z);
			unsigned long long bits[MW];
#pragma unroll
			for (int b = 0; b < MW; b++) bits[b] = 0;
			for (int k = 0; k < M; k++) {
				double dist = sqrt(quad_gen(prm.Rinv, z[0] - zs[k * 3], z[1] - zs[k * 3 + 1], z[2] - zs[k * 3 + 2]));
				if (dist < 5) bits[k >> 6] |= 1ull << (k & 63);   // :436
			}
#pragma unroll
			for (int b = 0; b < MW; b++) adj[(size_t) j * MW + b] = bits[b];
			labl[j] = j;
		}
		for (int k = tid; k < M; k += 256) labz[k] = J + k;
		if (tid == 0) { s_nroots = 0; s_big = 0; }
		__syncthreads();

		// connected components of the bipartite (landmark, measurement) graph by min-label propagation;
		// a cluster's label ends as its smallest landmark index, which is also its position in the
		// reference's component list (rows with detection entries are inserted first, ascending).
		for (int it = 0; it < J + M + 1; it++) {
			if (tid == 0) s_changed = 0;
			__syncthreads();
			for (int j = tid; j < J; j += 256) {
				int l = labl[j];
#pragma unroll
				for (int b = 0; b < MW; b++) {
					unsigned long long bits = adj[(size_t) j * MW + b];
					while (bits) {
						int k = b * 64 + __ffsll((long long) bits) - 1;
						bits &= bits - 1;
						l = min(l, labz[k]);
					}
				}
				if (l < labl[j]) { labl[j] = l; s_changed = 1; }
			}
			__syncthreads();
			for (int k = tid; k < M; k += 256) {
				int l = labz[k];
				for (int j = 0; j < J; j++) {
					if ((adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull) l = min(l, labl[j]);
				}
				if (l < labz[k]) { labz[k] = l; s_changed = 1; }
			}
			__syncthreads();
			if (!s_changed) break;
			__syncthreads();
		}
		if (tid == 0) {
			int nr = 0;
			for (int j = 0; j < J; j++) {
				bool has = false;
				for (int b = 0; b < MW; b++) has |= adj[(size_t) j * MW + b] != 0;
				if (labl[j] == j && has) roots[nr++] = j;
			}
			s_nroots = nr;
		}
		__syncthreads();
		const int nroots = s_nroots;
		const double logmult = prm.logRmult;

		// clusters with n <= 5 rows: every pairing (PHDNavigator.cs:492-494), one lane per cluster
		if (wv == 0) {
			for (int r0 = 0; r0 < nroots; r0 += 64) {
				int ri = r0 + lane;
				if (ri < nroots) {
					int root = roots[ri];
					int L[5], Z[5], nl = 0, nz = 0, nrow = 0;
					for (int j = root; j < J; j++) {
						if (labl[j] == root) { if (nl < 5) L[nl] = j; nl++; }
					}
					for (int k = 0; k < M; k++) {
						if (labz[k] == root) { if (nz < 5) Z[nz] = k; nz++; }
					}
					nrow = nl + nz;
					if (nrow > 5) {
						res[ri] = NAN;   // solved below by the Murty path
						s_big = 1;
					}
					else {
						double* mat = mats + lane;   // entry e at mat[e * 64]
						for (int e = 0; e < 25; e++) mat[e * 64] = -INFINITY;
						for (int x = 0; x < nl; x++) {
							int j = L[x];
							for (int y = 0; y < nz; y++) {
								int k = Z[y];
								if ((adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull) {
									double dist = sqrt(quad_gen(prm.Rinv, zh[j] - zs[k * 3], zh[JP + j] - zs[k * 3 + 1],
									                            zh[2 * JP + j] - zs[k * 3 + 2]));
									mat[(x * 5 + y) * 64] = log(pdj[j]) + logmult - 0.5 * dist * dist;   // :439
								}
							}
							mat[(x * 5 + nz + x) * 64] = log(1 - pdj[j]);   // :445
						}
						for (int y = 0; y < nz; y++) {
							mat[((nl + y) * 5 + y) * 64] = prm.logkappa;   // :449
							for (int x = 0; x < nl; x++) mat[((nl + y) * 5 + nz + x) * 64] = 0;   // :480-488
						}
						res[ri] = cluster_enumerate(mat, 64, nrow, J);
					}
				}
			}
		}
		__syncthreads();
		if (s_big) {
			// Some cluster has more than 5 rows: it is enumerated best-first (MurtyPairing) under the
			// early-exit test of PHDNavigator.cs:503, which reads logcomp[m] as left behind by the clusters
			// before it. So wave 0 replays the clusters in order up to the last such cluster, keeping the
			// shared logcomp array; the other waves wait.
			if (wv == 0) {
				MurtyLds ws;
				double* big = (double*) (labz + MP + (MP & 1));
				big = (double*) (((size_t) big + 7) & ~(size_t) 7);
				ws.profit  = big;
				ws.reduced = ws.profit + MURTY_NMAX * MURTY_NMAX;
				ws.logcomp = ws.reduced + MURTY_NMAX * MURTY_NMAX;
				ws.fkey    = ws.logcomp + MURTY_OUT;
				ws.fnode   = (int*) (ws.fkey + MURTY_POOL);
				ws.freelist = ws.fnode + MURTY_POOL;
				ws.L = ws.freelist + MURTY_POOL;
				ws.Z = ws.L + MURTY_NMAX;
				for (int i = lane; i < MURTY_OUT; i += 64) ws.logcomp[i] = 0;   // new double[200], :469
				lds_fence();
				int lastbig = -1;
				for (int r = 0; r < nroots; r++) {
					if (isnan(res[r])) lastbig = r;
				}
				for (int ri = 0; ri <= lastbig; ri++) {
					const int root = roots[ri];
					int nl = 0, nz = 0;
					for (int j0 = root; j0 < J; j0 += 64) {
						int j = j0 + lane;
						bool in = j < J && labl[j] == root;
						unsigned long long bal = __ballot(in);
						int pos = nl + __popcll(bal & lanemask_lt());
						if (in && pos < MURTY_NMAX) ws.L[pos] = j;
						nl += __popcll(bal);
					}
					for (int k0 = 0; k0 < M; k0 += 64) {
						int k = k0 + lane;
						bool in = k < M && labz[k] == root;
						unsigned long long bal = __ballot(in);
						int pos = nz + __popcll(bal & lanemask_lt());
						if (in && pos < MURTY_NMAX) ws.Z[pos] = k;
						nz += __popcll(bal);
					}
					lds_fence();
					const int nrow = nl + nz;
					if (nrow > MURTY_NMAX) {
						if (lane == 0) { atomicOr(a.flags, PHD_FLAG_BIG_CLUSTER); res[ri] = 0; }
						continue;
					}
					// the cluster's square matrix: rows = landmarks then clutter rows, columns = measurements
					// then misdetection columns (Compact, SparseMatrix.cs:592-628; zero quadrant :480-488)
					const int stride = (nrow <= 5) ? 5 : nrow;
					double* mat = (nrow <= 5) ? ws.reduced : ws.profit;
					for (int e = lane; e < nrow * nrow; e += 64) {
						int x = e / nrow, y = e - x * nrow;
						double v = -INFINITY;
						if (x < nl) {
							int j = ws.L[x];
							if (y < nz) {
								int k = ws.Z[y];
								if ((adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull) {
									double dist = sqrt(quad_gen(prm.Rinv, zh[j] - zs[k * 3], zh[JP + j] - zs[k * 3 + 1],
									                            zh[2 * JP + j] - zs[k * 3 + 2]));
									v = log(pdj[j]) + logmult - 0.5 * dist * dist;
								}
							}
							else if (y - nz == x) v = log(1 - pdj[j]);
						}
						else {
							if (y < nz) { if (y == x - nl) v = prm.logkappa; }
							else v = 0;
						}
						mat[x * stride + y] = v;
					}
					lds_fence();
					if (nrow <= 5) {
						// only its logcomp entries matter here (res[ri] is already known)
						if (lane == 0) cluster_enumerate(mat, 1, nrow, J, ws.logcomp);
						lds_fence();
					}
					else {
						int mcount = wave_murty(ws, a.murty + p, nrow, lane);
						// LogSumExp(logcomp, 0, m), MatrixExtensions.cs:361-389
						double mx = -INFINITY, value = 0;
						for (int i = 0; i < mcount; i++) mx = fmax(mx, ws.logcomp[i]);
						double lse;
						if (isinf(mx) && mx < 0) lse = -INFINITY;
						else {
							for (int i = 0; i < mcount; i++) value += exp(ws.logcomp[i] - mx);
							lse = mx + log(value);
						}
						if (lane == 0) res[ri] = lse;
						lds_fence();
					}
				}
			}
			__syncthreads();
		}
		if (tid == 0) {
			// total in the reference's component order: clusters holding detections (ascending first
			// landmark), then the lone landmarks (misdetection only), then the lone measurements (clutter)
			double total = 0;
			for (int r = 0; r < nroots; r++) total += res[r];
			for (int j = 0; j < J; j++) {
				bool has = false;
				for (int b = 0; b < MW; b++) has |= adj[(size_t) j * MW + b] != 0;
				if (!has) total += log(1 - pdj[j]);
			}
			for (int k = 0; k < M; k++) {
				if (labz[k] == J + k) total += prm.logkappa;
			}
			s_total = total;
		}
		__syncthreads();
	}
	if (tid == 0) {
		double setll = s_total;
		double ratio = (plog - pcount) - (clog - ccount);   // :390
		double alpha = exp(setll + ratio);                  // :392
		a.setll[p] = setll;
		a.alpha[p] = alpha;
		bout.weights[p] = bin.weights[p] * alpha;         // :335
	}
}

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
