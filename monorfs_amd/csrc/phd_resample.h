// phd_resample.h — k_normalise_resample: PHDNavigator.cs:343-358 (normalise, BestParticle),
// :768-777 (ParticleDepleted) and :724-760 (ResampleParticles) on the whole weight vector (all
// ranks' particles when sharded). One workgroup of 1024 threads.
//
// Systematic resampling in the reference is a sequential floating-point recurrence
//     random = u / P;  for i: { while (random > 0 && k < P) random -= w[k++];  src[i] = k - 1;  random += 1 / P; }
// whose indices must come out bit-exact. A GPU lane runs such a chain at ~100 cycles per step, so the
// kernel gets the same indices in parallel by VERIFIED SPECULATION:
//   * in exact arithmetic the recurrence holds  random = T_i - S_k  with  T_i = fl(u/P) + i * fl(1/P)  and
//     S_k = w_0 + ... + w_(k-1);  it stops iteration i at the first k with S_k >= T_i;
//   * S (prefix sums) and T are evaluated in double-double arithmetic (error ~1e-30), in parallel;
//   * the floating-point recurrence makes at most 2P roundings of numbers no larger than
//     max(w) + fl(u/P) + fl(1/P), so its `random` differs from T_i - S_k by less than
//     B = 8 P 2^-53 (max(w) + fl(u/P) + fl(1/P)); every comparison `random > 0` it performs therefore has
//     the exact outcome whenever |T_i - S_k| > B at the two prefix sums around the crossing;
//   * if every slot passes that margin test (and no weight is negative or NaN) the parallel indices ARE
//     the recurrence's; otherwise one wave replays the recurrence literally (the fallback).
// The result is bit-identical to the sequential code in both cases.
//
// Sum of the weights and of their squares: fixed-shape parallel tree (chunk per thread, then waves, then
// the 16 wave totals), independent of the number of GPUs. (The reference sums sequentially; the last-bit
// difference is far below the tolerance on particle weights, 1e-6.)
//   w (gw or the OUT bank): in: un-normalised weights, out: normalised weights, or 1/P after resampling
//   src : [P] source slot of each particle (identity when not resampled);  info: [0] BestParticle, [1] resampled
#pragma once
#include "phd_device.h"

struct dd { double hi, lo; };

__device__ __forceinline__ dd dd_two_sum(double a, double b)
{
#pragma clang fp contract(off)

	double s = a + b, bb = s - a;
	double e = (a - (s - bb)) + (b - bb);
	return dd{s, e};
}

__device__ __forceinline__ dd dd_quick(double a, double b)   // |a| >= |b|
{
#pragma clang fp contract(off)

	double s = a + b;
	return dd{s, b - (s - a)};
}

__device__ __forceinline__ dd dd_add(dd x, dd y)
{
#pragma clang fp contract(off)

	dd s = dd_two_sum(x.hi, y.hi);
	dd t = dd_two_sum(x.lo, y.lo);
	s.lo += t.hi;
	s = dd_quick(s.hi, s.lo);
	s.lo += t.lo;
	return dd_quick(s.hi, s.lo);
}

__device__ __forceinline__ dd dd_add_d(dd x, double d)
{
#pragma clang fp contract(off)

	dd s = dd_two_sum(x.hi, d);
	s.lo += x.lo;
	return dd_quick(s.hi, s.lo);
}

__device__ __forceinline__ bool dd_ge(dd a, dd b) { return a.hi > b.hi || (a.hi == b.hi && a.lo >= b.lo); }
__device__ __forceinline__ bool dd_lt(dd a, dd b) { return !dd_ge(a, b); }
__device__ __forceinline__ double dd_diff(dd a, dd b) {
#pragma clang fp contract(off)
 return (a.hi - b.hi) + (a.lo - b.lo); }

__device__ __forceinline__ double shfl_up_d(double v, int o) { return __shfl_up(v, o, 64); }

// block-wide sums of a double, fixed order: thread partials -> wave butterflies -> wave 0 adds the 16 totals
__device__ __forceinline__ double block_sum_1024(double v, double* scratch16, int tid)
{
	const int lane = tid & 63, wv = tid >> 6, nw = (int) (blockDim.x >> 6);
	v = wave_sum(v);
	__syncthreads();
	if (lane == 0) scratch16[wv] = v;
	__syncthreads();
	double t = 0;
	for (int q = 0; q < nw; q++) t += scratch16[q];
	return t;
}

// End of a single-handle step (sel_next != NULL), folded into the same launch: the resampled particles
// (PHDNavigator.cs:740-741) and the bank roles of the next step, decided here from the resampling flag so the host never
// waits inside a step. No mixture is copied:
//   not resampled: the new state is the OUT bank                      -> (IN, OUT, TMP, INMIX) = (O, I, T, O), slots identity
//   resampled    : small arrays of particle i <- OUT[src[i]] into TMP -> (IN, OUT, TMP, INMIX) = (T, I, O, O), slots src
//   frozen       : roles and slots stay (benchmark steady state); RES / RESMIX say where the result is (slots: src)
// Called by all 1024 threads after the source vector and the final weights are in global memory.
__device__ __forceinline__ void rotate_roles(const StepBufs& a, const int* src, int resampled, int* sel_next, int frozen,
                                             int* inslot, int tid)
{
	const int I = a.sel[SEL_IN], O = a.sel[SEL_OUT], T = a.sel[SEL_TMP], X = a.sel[SEL_INMIX];
	if (tid == 0) {
		if (frozen)         { sel_next[SEL_IN] = I; sel_next[SEL_OUT] = O; sel_next[SEL_TMP] = T; sel_next[SEL_INMIX] = X; }
		else if (resampled) { sel_next[SEL_IN] = T; sel_next[SEL_OUT] = I; sel_next[SEL_TMP] = O; sel_next[SEL_INMIX] = O; }
		else                { sel_next[SEL_IN] = O; sel_next[SEL_OUT] = I; sel_next[SEL_TMP] = T; sel_next[SEL_INMIX] = O; }
		sel_next[SEL_RES]    = resampled ? T : O;
		sel_next[SEL_RESMIX] = O;
	}
	const Bank bo = bank_of(a, SEL_OUT), bt = bank_of(a, SEL_TMP);
	// two particles per thread and trip, their loads issued together: this tail is two dependent round trips to memory
	// per trip in a single workgroup, nothing else
	const int nt_ = (int) blockDim.x;
	for (int i0 = tid; i0 < a.P; i0 += 2 * nt_) {
		const int i1 = i0 + nt_;
		const bool v1 = i1 < a.P;
		const int s0 = resampled ? src[i0] : i0, s1 = (resampled && v1) ? src[i1] : i1;
		if (resampled) {
			const int c0 = bo.count[s0], c1 = v1 ? bo.count[s1] : 0;
			const double w0 = bo.weights[i0], w1 = v1 ? bo.weights[i1] : 0.0;
			double q0[7], q1[7];
#pragma unroll
			for (int t = 0; t < 7; t++) { q0[t] = bo.poses[(size_t) s0 * 7 + t]; q1[t] = v1 ? bo.poses[(size_t) s1 * 7 + t] : 0.0; }
			bt.count[i0] = c0; bt.weights[i0] = w0;
#pragma unroll
			for (int t = 0; t < 7; t++) bt.poses[(size_t) i0 * 7 + t] = q0[t];
			if (v1) {
				bt.count[i1] = c1; bt.weights[i1] = w1;
#pragma unroll
				for (int t = 0; t < 7; t++) bt.poses[(size_t) i1 * 7 + t] = q1[t];
			}
		}
		if (!frozen) { inslot[i0] = s0; if (v1) inslot[i1] = s1; }
	}
}

__global__ __launch_bounds__(1024) void k_normalise_resample(const StepBufs a, double* gw, int P, double min_eff, double u,
                                                             int force_resample, int skip_normalise, int use_lds,
                                                             int* src, int* info, int* sel_next, int frozen, int* inslot)
{
#pragma clang fp contract(off)   // the error-free transformations below must not be fused
	extern __shared__ __align__(16) double lw[];   // [P] when use_lds
	__shared__ double s_pchi[1024], s_pclo[1024];  // double-double prefix sum at the start of every thread's chunk
	__shared__ double s16[16], s16b[16];
	__shared__ int    s_i16[16];
	__shared__ int    s_res, s_ok, s_best;
	// (launched with 1024 threads, or 256 for short weight vectors: the shape of the sums depends on the vector length only)
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nt = (int) blockDim.x, nw = nt >> 6;
	if (tid == 0 && a.bigws_used && *a.bigws_used) *a.bigws_used = 0;   // the association slab is free again (every k_alpha_assoc of the step is over)
	// A kernel of this step raised a flag (emit capacity, landmark scratch): what it wrote into the OUT bank is not a
	// valid state. The step is dropped as a whole — the roles stay, nothing of the current state was touched — and the host
	// finds the flag at its next phd_sync. (Every thread reads the same word, written by earlier launches.)
	if (sel_next && *a.flags != 0) {
		if (tid < SEL_STRIDE) sel_next[tid] = a.sel[tid];
		return;
	}
	double* gwp = gw ? gw : bank_of(a, SEL_OUT).weights;
	double* w = use_lds ? lw : gwp;
	if (use_lds) {
		for (int i = tid; i < P; i += nt) lw[i] = gwp[i];
	}
	__syncthreads();
	const int CH = (P + nt - 1) / nt;              // contiguous chunk of every thread
	const int c0 = min(P, tid * CH), c1 = min(P, c0 + CH);

	// ---- normalise (:343-345)
	if (!skip_normalise) {
		double part = 0;
		for (int k = c0; k < c1; k++) part += w[k];
		double sum = block_sum_1024(part, s16, tid);
		sum = (sum == 0) ? 1 : sum;
		for (int k = c0; k < c1; k++) w[k] = w[k] / sum;
	}
	__syncthreads();

	// ---- BestParticle (first strict maximum, :347-354), sum of squares (:772-774), and the chunk sums for S
	double part2 = 0, wmax = -INFINITY;
	int    imax = 0;
	bool   bad = false;
	dd     chunk = {0, 0};
	for (int k = c0; k < c1; k++) {
		double wk = w[k];
		part2 += wk * wk;
		if (wk > wmax) { wmax = wk; imax = k; }
		bad |= !(wk >= 0);
		chunk = dd_add_d(chunk, wk);
	}
	const double cum = block_sum_1024(part2, s16, tid);
	// arg-max with the smallest index among equals
	{
		double m = wmax;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
		unsigned long long bal = ballot64(wmax == m && c0 < c1);
		int firstl = bal ? __ffsll((long long) bal) - 1 : 0;
		int idx = __shfl(imax, firstl, 64);
		__syncthreads();
		if (lane == 0) { s16b[wv] = bal ? m : -INFINITY; s_i16[wv] = idx; }
		__syncthreads();
	}
	double gmax = -INFINITY;
	int    gbest = 0;
	for (int q = 0; q < nw; q++) {
		if (s16b[q] > gmax) { gmax = s16b[q]; gbest = s_i16[q]; }
	}
	if (!(gmax > 0)) gbest = 0;   // maxweight starts at 0 and the comparison is strict (:347-353)
	bool depleted = (1.0 / cum < min_eff * P);   // :776
	if (force_resample > 0) depleted = true;
	if (force_resample < 0) depleted = false;

	if (!depleted) {   // uniform: every thread computed the same `cum`
		if (tid == 0) { info[0] = gbest; info[1] = 0; }
		for (int k = c0; k < c1; k++) {
			src[k] = k;
			if (use_lds && !skip_normalise) gwp[k] = lw[k];
		}
		if (sel_next) rotate_roles(a, src, 0, sel_next, frozen, inslot, tid);   // no thread reads another's writes here
		return;
	}

	// ---- ResampleParticles (:724-760) by verified speculation
	// exclusive double-double scan of the chunk sums: wave scan by shuffles, then the 16 wave totals
	dd incl = chunk;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		dd y = {shfl_up_d(incl.hi, o), shfl_up_d(incl.lo, o)};
		if (lane >= o) incl = dd_add(y, incl);
	}
	__syncthreads();
	if (lane == 63) { s16[wv] = incl.hi; s16b[wv] = incl.lo; }
	__syncthreads();
	dd woff = {0, 0};
	for (int q = 0; q < wv; q++) woff = dd_add(woff, dd{s16[q], s16b[q]});
	dd excl = dd_add(woff, incl);
	excl = dd_add(excl, dd{-chunk.hi, -chunk.lo});   // prefix at the start of this thread's chunk
	if (c0 == 0) excl = dd{0, 0};
	s_pchi[tid] = excl.hi;
	s_pclo[tid] = excl.lo;
	if (tid == 0) { s_ok = 1; s_best = 0x7fffffff; }
	bad = __syncthreads_or(bad);

	const double R0 = u / P, invP = 1.0 / P;
	const double B = 8.0 * P * 1.1102230246251565e-16 * (fmax(gmax, 0.0) + fabs(R0) + invP);
	const int nchunks = (P + CH - 1) / CH;
	bool   ok = !bad;
	double mybestw = -INFINITY;
	int    mybesti = 0x7fffffff;
	for (int i = tid; i < P; i += nt) {
		// T_i = R0 + i * invP, exactly, as a double-double
		double ph = (double) i * invP, pl = fma((double) i, invP, -ph);
		dd T = dd_add_d(dd{ph, pl}, R0);
		int kstar;
		double mlow = INFINITY, mhigh = INFINITY;
		if (!(T.hi > 0 || (T.hi == 0 && T.lo > 0))) {
			kstar = 0;   // random <= 0 before any subtraction (u == 0)
			mhigh = -dd_diff(T, dd{0, 0});
			if (mhigh == 0) mhigh = INFINITY;   // the comparison 0 > 0 is exact
		}
		else {
			// last chunk whose start prefix is still below T
			int lo = 0, hi = nchunks - 1;
			while (lo < hi) {
				int mid = (lo + hi + 1) >> 1;
				if (dd_lt(dd{s_pchi[mid], s_pclo[mid]}, T)) lo = mid;
				else hi = mid - 1;
			}
			dd s = {s_pchi[lo], s_pclo[lo]};
			int k = lo * CH, kend = min(P, k + CH);
			kstar = P;
			dd sprev = s;
			while (k < kend) {
				sprev = s;
				s = dd_add_d(s, w[k]);
				k++;
				if (dd_ge(s, T)) { kstar = k; break; }
			}
			if (kstar == P && k < P) {   // crossing exactly at the start of the next chunk cannot happen (its prefix >= T): defensive
				ok = false;
			}
			mlow = dd_diff(T, sprev);                       // last comparison that came out positive
			if (kstar < P) mhigh = dd_diff(s, T);           // the comparison that stopped the loop (none if k ran into P)
			else mlow = dd_diff(T, sprev);
		}
		ok = ok && (mlow > B) && (mhigh > B);
		int sidx = (kstar - 1 < 0) ? 0 : kstar - 1;
		src[i] = sidx;
		double ws = w[sidx];
		if (ws > mybestw) { mybestw = ws; mybesti = i; }    // i ascending per thread: first maximum kept
	}
	if (!ok) s_ok = 0;
	__syncthreads();
	if (s_ok) {
		// BestParticle = first slot whose source has the largest weight (:745-748)
		double m = mybestw;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
		if (lane == 0) s16[wv] = m;
		__syncthreads();
		double gm = -INFINITY;
		for (int q = 0; q < nw; q++) gm = fmax(gm, s16[q]);
		if (gm > 0 && mybestw == gm) atomicMin(&s_best, mybesti);
		__syncthreads();
		if (tid == 0) { info[0] = (gm > 0) ? s_best : 0; info[1] = 1; }
	}
	else if (wv == 0) {
		// fallback: the recurrence itself, wave-uniform (weights through v_readlane, scalar control flow)
		auto rl = [](double v, int l) {
			int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
			return __hiloint2double(hi, lo);
		};
		double random = u / P, maxweight = 0;
		int k = 0, best = 0, cb = 0;
		double cur = (lane < P) ? w[lane] : 0.0, prev = 0.0;
		for (int i = 0; i < P; i++) {
			while (k < P && __builtin_amdgcn_readfirstlane((int) (random > 0))) {
				if (k >= cb + 64) { prev = cur; cb += 64; cur = (cb + lane < P) ? w[cb + lane] : 0.0; }
				random -= rl(cur, k - cb);
				k++;
			}
			int s = (k - 1 < 0) ? 0 : k - 1;   // u == 0 would index -1 in the reference: clamped
			if (lane == 0) src[i] = s;
			random += invP;
			double ws = (s >= cb) ? rl(cur, s - cb) : rl(prev, s - (cb - 64));
			if (__builtin_amdgcn_readfirstlane((int) (ws > maxweight))) { maxweight = ws; best = i; }
		}
		if (lane == 0) { info[0] = best; info[1] = 1; }
	}
	__syncthreads();
	for (int k = c0; k < c1; k++) gwp[k] = 1.0 / P;   // :742
	if (sel_next) {
		__threadfence_block();
		__syncthreads();   // src and the weights of every particle are written
		rotate_roles(a, src, 1, sel_next, frozen, inslot, tid);
	}
}
