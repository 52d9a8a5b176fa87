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

// T_i = fl(u / P) + i * fl(1 / P), exactly, as a double-double: what `random` is at slot i before anything is subtracted
__device__ __forceinline__ dd slot_target(int i, double R0, double invP)
{
#pragma clang fp contract(off)
	const double ph = (double) i * invP, pl = fma((double) i, invP, -ph);
	return dd_add_d(dd{ph, pl}, R0);
}

// How many slots i in [0, P) have T_i <= x: an estimate from the quotient, settled by exact comparisons with the slots
// around it. *settled <- false if six steps did not settle it (never seen; the caller then lets the recurrence decide).
__device__ __forceinline__ int slots_upto(dd x, double R0, double invP, int P, bool* settled)
{
#pragma clang fp contract(off)
	const double est = (x.hi - R0) * (double) P;
	int n = (est >= (double) P) ? P : ((est < 0) ? 0 : (int) est + 1);
	int it = 0;
	for (; it < 6; it++) {
		const bool up = n < P && dd_ge(x, slot_target(n, R0, invP));
		const bool dn = !up && n > 0 && !dd_ge(x, slot_target(n - 1, R0, invP));
		if (up) n++;
		else if (dn) n--;
		else break;
	}
	if (it == 6) *settled = false;
	return n;
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

// The body: the kernel's one workgroup, or — for a small particle set — the last workgroup of k_particle_chain to finish
// (phd_kernels.h; blockDim.x = 256 there as in a launch of its own for up to 512 weights: the same tree of sums, the same bits).
__device__ __forceinline__ void normalise_resample_body(const StepBufs& a, double* gw, int P, double min_eff, double u,
                                                        int force_resample, int skip_normalise, int use_lds,
                                                        int* src, int* info, int* sel_next, int frozen, int* inslot,
                                                        double* lw /* LDS, [P] chunk-transposed when use_lds */)
{
#pragma clang fp contract(off)   // the error-free transformations below must not be fused
	__shared__ double s_pchi[1024], s_pclo[1024];  // double-double prefix sum at the start of every thread's chunk
	__shared__ double s16[16], s16b[16];
	__shared__ int    s_i16[16];
	__shared__ int    s_ok, s_best;
	// (launched with 1024 threads, or 256 for short weight vectors: the shape of the sums depends on the vector length only)
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nt = (int) blockDim.x, nw = nt >> 6;
	PHD_STAMP_DECL;
	PHD_STAMP(0);
	// (one workgroup alone on the device: every dependent trip to memory is the kernel's time. The flag word, the bank roles and
	// this thread's first weight — from all three banks, the role picks one — are requested together: one trip instead of three)
	double spec0 = 0, spec1 = 0, spec2 = 0;
	const bool spec = !gw && !a.defer && tid < P;
	if (spec) { spec0 = a.bank[0].weights[tid]; spec1 = a.bank[1].weights[tid]; spec2 = a.bank[2].weights[tid]; }
	const int flags_now = sel_next ? *a.flags : 0;
	const int sel_out = a.sel[SEL_OUT];
	if (tid == 0 && a.bigws_used && *a.bigws_used) *a.bigws_used = 0;   // the association slab is free again (every k_alpha_assoc of the step is over)
	if (tid < 4 && a.biglist) a.biglist[(size_t) tid * a.bigstride] = 0;   // ... and the sub-ranges' lists of deferred particles are empty again
	// A kernel of this step raised a flag (emit capacity, landmark scratch): what it wrote into the OUT bank is not a
	// valid state. The step is dropped as a whole — the roles stay, nothing of the current state was touched — and the host
	// finds the flag at its next phd_sync. (Every thread reads the same word, written by earlier launches.)
	if (sel_next && flags_now != 0) {
		if (tid < SEL_STRIDE) sel_next[tid] = a.sel[tid];
		return;
	}
	double* gwp = gw ? gw : bank_of(a, SEL_OUT).weights;
	if (a.defer && !gw) {
		// WeightAlpha's last line (PHDNavigator.cs:390-392, :335), left open by k_alpha_density while the set log-likelihoods of
		// the particles with big association clusters were still in the making: alpha = exp(L + density ratio), weight *= alpha
		const double* win = bank_of(a, SEL_IN).weights;
		for (int i = tid; i < P; i += nt) {
			const double alpha = exp(a.setll[i] + a.ratio[i]);
			a.alpha[i] = alpha;
			gwp[i] = win[i] * alpha;
		}
		__syncthreads();   // (a workgroup's own stores: visible to its loads behind the barrier)
	}
	const int CH = (P + nt - 1) / nt;              // contiguous chunk of every thread
	const int c0 = min(P, tid * CH), c1 = min(P, c0 + CH), cn = c1 - c0;
	// The vector is staged in LDS chunk-transposed: element j of chunk t at lw[j * LS + t], LS = nt + 1. Every loop below has
	// the lanes of a wave on the same element of neighbouring chunks — consecutive LDS words — where the plain layout put
	// them CH doubles apart (at 16 384 weights: 16 lanes on one bank, every access 16 times its cost); the odd row stride
	// spreads the accesses that run along a chunk (staging the vector, storing the sources) over the banks as well.
	const int LS = nt + 1;
	auto W = [&](int chunk, int j) -> double { return use_lds ? lw[j * LS + chunk] : gwp[chunk * CH + j]; };
	const float rCH = 1.0f / (float) CH;
	const int chshift = (CH & (CH - 1)) == 0 ? __ffs(CH) - 1 : -1;   // (a power of two — every BASELINE size — needs no division)
	// where element i of the vector sits in LDS (i < 2^24: small_div's range; a vector that long is not staged anyway)
	auto lpos = [&](int i) {
		const int c = chshift >= 0 ? (i >> chshift) : small_div(i, CH, rCH);
		return (i - c * CH) * LS + c;
	};
	if (use_lds) {
		// (global reads in whole lines; the scattered side is the LDS one)
		if (spec) lw[lpos(tid)] = (sel_out == 0) ? spec0 : ((sel_out == 1) ? spec1 : spec2);
		// (four loads in flight per thread: one element per trip to memory made the staging of 16 384 weights sixteen trips)
		for (int i0 = spec ? tid + nt : tid; i0 < P; i0 += 4 * nt) {
			double v[4];
#pragma unroll
			for (int q = 0; q < 4; q++) v[q] = (i0 + q * nt < P) ? gwp[i0 + q * nt] : 0.0;
#pragma unroll
			for (int q = 0; q < 4; q++) if (i0 + q * nt < P) lw[lpos(i0 + q * nt)] = v[q];
		}
	}
	__syncthreads();
	PHD_STAMP(1);

	// ---- normalise (:343-345); BestParticle (first strict maximum, :347-354), sum of squares (:772-774) and the chunk sums
	// for S in the same pass over the thread's chunk
	double nsum = 1;
	if (!skip_normalise) {
		double part = 0;
		for (int j = 0; j < cn; j++) part += W(tid, j);
		double sum = block_sum_1024(part, s16, tid);
		nsum = (sum == 0) ? 1 : sum;
	}
	PHD_STAMP(2);
	double part2 = 0, wmax = -INFINITY;
	int    imax = 0;
	bool   bad = false;
	dd     chunk = {0, 0};
	for (int j = 0; j < cn; j++) {
		double wk = W(tid, j);
		if (!skip_normalise) {
			wk = wk / nsum;
			if (use_lds) lw[j * LS + tid] = wk;
			else gwp[c0 + j] = wk;
		}
		part2 += wk * wk;
		if (wk > wmax) { wmax = wk; imax = c0 + j; }
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
	PHD_STAMP(3);
	bool depleted = (1.0 / cum < min_eff * P);   // :776
	if (force_resample > 0) depleted = true;
	if (force_resample < 0) depleted = false;

	if (!depleted) {   // uniform: every thread computed the same `cum`
		if (tid == 0) { info[0] = gbest; info[1] = 0; }
		for (int i = tid; i < P; i += nt) {
			src[i] = i;
			if (use_lds && !skip_normalise) gwp[i] = lw[lpos(i)];
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
	dd woff;
	{   // exclusive scan of the wave totals: lane q holds wave q's, every wave does the same four steps; its own offset is lane wv's
		dd tot = (lane < nw) ? dd{s16[lane], s16b[lane]} : dd{0, 0};
		dd inc = tot;
#pragma unroll
		for (int o = 1; o < 16; o <<= 1) {
			dd y = {shfl_up_d(inc.hi, o), shfl_up_d(inc.lo, o)};
			if (lane >= o) inc = dd_add(y, inc);
		}
		woff = (wv == 0) ? dd{0, 0} : dd{__shfl(inc.hi, wv - 1, 64), __shfl(inc.lo, wv - 1, 64)};   // (wave-uniform index)
	}
	dd excl = dd_add(woff, incl);
	excl = dd_add(excl, dd{-chunk.hi, -chunk.lo});   // prefix at the start of this thread's chunk
	if (c0 == 0) excl = dd{0, 0};
	s_pchi[tid] = excl.hi;
	s_pclo[tid] = excl.lo;
	if (tid == 0) { s_ok = 1; s_best = 0x7fffffff; }
	bad = __syncthreads_or(bad);

	PHD_STAMP(4);
	const double R0 = u / P, invP = 1.0 / P;
	const double B = 8.0 * P * 1.1102230246251565e-16 * (fmax(gmax, 0.0) + fabs(R0) + invP);
	const int nchunks = (P + CH - 1) / CH;
	bool   ok = !bad;
	double mybestw = -INFINITY;
	int    mybesti = 0x7fffffff;
	int lo0 = 0;   // (staged path) the first slot of this thread's first particle
	if (use_lds) {
		// ---- per PARTICLE, not per slot: particle k takes the slots i with S_k < T_i <= S_(k+1), i.e. [N(S_k), N(S_(k+1))) with
		// N(x) = #{i : T_i <= x}. Every thread takes the particles of its chunk; each costs the same few exact comparisons
		// whatever the weights are — no lane waits for another lane's search (a walk over the slots made every wave pay for
		// its slowest lane at every slot: most of this kernel at 16 384 weights). A particle's cell in LDS (its weight is
		// dead once its owner has read it) takes N(S_(k+1)); the runs are then laid over the slots by a max-scan.
		//   boundaries between chunks come from the scanned chunk prefixes on both sides, so the ranges tile [0, P);
		//   the recurrence's comparisons are exact (B) iff, for every particle with slots, T at its first slot clears S_k and
		//   S_(k+1) clears T at its last one — the two prefix sums around every crossing, as before.
		int2* cells = (int2*) lw;   // cell (chunk c, element j) at [j * LS + c]: {N(S_(k+1)) | source of slot k, head written for slot k}
		dd S = {s_pchi[tid], s_pclo[tid]};
		bool settled = true;
		int lo = (c0 == 0) ? 0 : slots_upto(S, R0, invP, P, &settled);   // (slot 0 belongs to the first particle even when u == 0: the clamp of :739)
		lo0 = lo;
		for (int j = 0; j < cn; j++) {
			const int k = c0 + j;
			const double wk = lw[j * LS + tid];
			const dd Sn = dd_add_d(S, wk);
			const dd Sb = (j == cn - 1 && tid + 1 < nchunks) ? dd{s_pchi[tid + 1], s_pclo[tid + 1]} : Sn;   // S_(k+1) as the next chunk knows it
			int hi = slots_upto(Sb, R0, invP, P, &settled);
			const int hinat = hi;
			if (k == P - 1) hi = P;        // the recurrence runs into P: whatever is left goes to the last particle
			if (hi < lo) { ok = false; hi = lo; }
			if (hi > lo) {
				// the last comparison that came out positive, at the particle's first slot (none for particle 0: `random` is
				// u / P itself there, compared without any rounding)
				if (k > 0) ok = ok && dd_diff(slot_target(lo, R0, invP), S) > B;
				// the comparison that stopped the loop, at its last slot (none for the slots the last particle gets by k == P)
				if (hinat > lo) ok = ok && dd_diff(Sb, slot_target(hinat - 1, R0, invP)) > B;
				if (wk > mybestw) { mybestw = wk; mybesti = lo; }   // k ascending per thread, slots ascending with k: first maximum kept
			}
			cells[j * LS + tid] = make_int2(hi, 0);
			S = Sn; lo = hi;
		}
		ok = ok && settled;
	}
	else {
	// (the vector does not fit LDS: the slots are walked in global memory)
	// Every thread takes the slots [c0, c1) and walks the prefix sums forward with them: T_i grows with i and so does the
	// crossing, which for most slots of a depleted set is the one of the slot before (a heavy particle takes a run of
	// slots) — one comparison. A slot that moves on scans the rest of its chunk; only when the target lies beyond the next
	// chunk's start is that chunk found by bisection over the chunk prefixes.
	//   s = S_k (the first k weights, k = ck * CH + jn: jn weights of chunk ck added), sprev = S_(k-1), wlast = w[k - 1]
	dd s = {0, 0}, sprev = {0, 0};
	int k = 0, ck = 0, jn = 0;
	bool have = false;                // a chunk has been entered
	double wlast = 0;
	for (int i = c0; i < c1; i++) {
		const dd T = slot_target(i, R0, invP);
		int kstar;
		double ws, mlow = INFINITY, mhigh = INFINITY;
		if (!(T.hi > 0 || (T.hi == 0 && T.lo > 0))) {
			kstar = 0;   // random <= 0 before any subtraction (u == 0)
			mhigh = -dd_diff(T, dd{0, 0});
			if (mhigh == 0) mhigh = INFINITY;   // the comparison 0 > 0 is exact
			ws = W(0, 0);                       // (the reference would index -1: clamped to the first particle)
		}
		else {
			if ((!have || dd_lt(s, T)) && k < P) {   // no crossing yet, or the one of the slot before does not reach T: move on
				// (k == P: every weight is added and the sum stays below T — the recurrence ran into P)
				int lo = -1;
				if (!have) lo = 0;
				else if (jn == min(CH, P - ck * CH)) lo = ck + 1;                                          // this chunk is used up
				else if (ck + 1 < nchunks && dd_lt(dd{s_pchi[ck + 1], s_pclo[ck + 1]}, T)) lo = ck + 1;   // the target lies beyond it
				if (lo >= 0) {
					// the last chunk, from lo on, whose start prefix is still below T. (lo's own is: the prefix of chunk 0 is 0;
					// that of ck + 1 is the sum walked so far up to the rounding of two orders of addition — should they fall on
					// different sides of T, the literal recurrence decides)
					if (lo > 0 && !dd_lt(dd{s_pchi[lo], s_pclo[lo]}, T)) ok = false;
					int hi = nchunks - 1;
					while (lo < hi) {
						int mid = (lo + hi + 1) >> 1;
						if (dd_lt(dd{s_pchi[mid], s_pclo[mid]}, T)) lo = mid;
						else hi = mid - 1;
					}
					ck = lo; jn = 0; k = lo * CH;
					s = dd{s_pchi[lo], s_pclo[lo]};
					have = true;
				}
				const int jend = min(CH, P - ck * CH);
				while (jn < jend) {
					sprev = s;
					wlast = W(ck, jn);
					s = dd_add_d(s, wlast);
					jn++; k++;
					if (dd_ge(s, T)) break;
				}
				if (dd_lt(s, T) && k < P) ok = false;   // the chunk ends below T although the next one starts at or above it: cannot happen (defensive)
			}
			const bool reached = dd_ge(s, T);
			kstar = reached ? k : P;
			mlow = dd_diff(T, sprev);                       // last comparison that came out positive
			if (reached) mhigh = dd_diff(s, T);             // the comparison that stopped the loop (none if k ran into P)
			ws = wlast;                                     // w[kstar - 1] (k ran into P: the last particle)
		}
		ok = ok && (mlow > B) && (mhigh > B);
		src[i] = (kstar - 1 < 0) ? 0 : kstar - 1;
		if (ws > mybestw) { mybestw = ws; mybesti = i; }    // i ascending per thread: first maximum kept
	}
	}   // (walk in global memory)
	if (!ok) s_ok = 0;
	__syncthreads();
	PHD_STAMP(5);
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
		if (use_lds) {
			// the runs laid over the slots: a particle with slots writes its number (+ 1) into the cell of its first slot, a
			// running maximum over the slots fills the runs (sources never decrease); then the sources leave in whole lines
			int2* cells = (int2*) lw;
			int lo = lo0;
			for (int j = 0; j < cn; j++) {
				const int hi = cells[j * LS + tid].x;
				if (hi > lo) ((int*) lw)[2 * lpos(lo) + 1] = c0 + j + 1;
				lo = hi;
			}
			__syncthreads();
			int run = 0;
			for (int j = 0; j < cn; j++) run = max(run, cells[j * LS + tid].y);
			int incl = run;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const int y = __shfl_up(incl, o, 64);
				if (lane >= o) incl = max(incl, y);
			}
			if (lane == 63) s_i16[wv] = incl;
			__syncthreads();
			int before = __shfl_up(incl, 1, 64);
			if (lane == 0) before = 0;
			for (int q = 0; q < wv; q++) before = max(before, s_i16[q]);
			for (int j = 0; j < cn; j++) {
				before = max(before, cells[j * LS + tid].y);
				cells[j * LS + tid].x = before - 1;   // (slot 0 always carries a head: the first particle with slots starts there)
			}
			__syncthreads();
			for (int i = tid; i < P; i += nt) src[i] = cells[lpos(i)].x;
		}
	}
	else if (wv == 0) {
		// fallback: the recurrence itself, wave-uniform (weights through v_readlane, scalar control flow)
		auto rl = [](double v, int l) {
			int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
			return __hiloint2double(hi, lo);
		};
		// (the staged weights have given way to the cells: read from memory — still un-normalised there when this launch
		// normalises, so divided again, by the same sum: the same bits)
		auto wk = [&](int q) { return (use_lds && !skip_normalise) ? gwp[q] / nsum : gwp[q]; };
		double random = u / P, maxweight = 0;
		int k = 0, best = 0, cb = 0;
		double cur = (lane < P) ? wk(lane) : 0.0, prev = 0.0;
		for (int i = 0; i < P; i++) {
			while (k < P && __builtin_amdgcn_readfirstlane((int) (random > 0))) {
				if (k >= cb + 64) { prev = cur; cb += 64; cur = (cb + lane < P) ? wk(cb + lane) : 0.0; }
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
	PHD_STAMP(6);
	for (int i = tid; i < P; i += nt) gwp[i] = 1.0 / P;   // :742
	PHD_STAMP(7);
#ifdef PHD_STAMPS
	if (tid == 0 && a.stamps && a.stamp_kernel == 6) { for (int s_ = 0; s_ < 8; s_++) a.stamps[s_] = (double) (stamp_[s_] - stamp_[0]); a.stamps[8] = (double) s_ok; }
#endif
	if (sel_next) {
		__threadfence_block();
		__syncthreads();   // src and the weights of every particle are written
		rotate_roles(a, src, 1, sel_next, frozen, inslot, tid);
	}
}

__global__ __launch_bounds__(1024) void k_normalise_resample(const StepBufs a, double* gw, int P, double min_eff, double u,
                                                             int force_resample, int skip_normalise, int use_lds,
                                                             int* src, int* info, int* sel_next, int frozen, int* inslot)
{
	extern __shared__ __align__(16) double lw_nr[];   // [P] when use_lds
	if (a.wait_tickets > 0) {
		// Two sub-range streams, steps posted back to back (phd_step_async): this launch sits directly behind the k_alpha_density of
		// ITS stream; the other stream's is ordered by count — every workgroup of both took a ticket behind a device-scope release of
		// its weight. Everything waited for was submitted before this launch, on whatever queue; the wait is bounded (0.2 s of the
		// 100 MHz counter: the step is then dropped with PHD_FLAG_ORDER_TIMEOUT, and so is every step behind it until phd_sync).
		if (threadIdx.x == 0) {
			const long long t0 = wall_clock64();
			// (relaxed loads: an acquire per poll would invalidate the L2 under the kernels still running)
			// (a step-stamped target on a counter that is never reset: late tickets of a step whose wait timed out cannot count towards
			// the next step; the flag a timed-out wait raises stays up — every later step is dropped — until phd_sync has reported it)
			while ((int) (__hip_atomic_load(a.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.ticket_target) < 0) {
				if (wall_clock64() - t0 > 20000000LL) { atomicOr(a.flags, PHD_FLAG_ORDER_TIMEOUT); break; }
				__builtin_amdgcn_s_sleep(32);
			}
		}
		__syncthreads();
		__threadfence();   // acquire in every wave: what the ticket holders published is what the loads below see
	}
	PHD_TL_BEGIN;
	normalise_resample_body(a, gw, P, min_eff, u, force_resample, skip_normalise, use_lds, src, info, sel_next, frozen, inslot, lw_nr);
	if (a.done_value) {
		// ... and the other stream's next k_sweep waits for THIS launch by number (k_gate)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (threadIdx.x == 0) { __threadfence(); __hip_atomic_store(a.ticket + 1, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
	}
#ifdef PHD_STAMPS
	if (threadIdx.x == 0 && a.stamps && a.stamp_kernel == 199) { a.stamps[8] = (double) tl0_; a.stamps[9] = (double) wall_clock64(); }
#endif
}

// The other sub-range stream's side of that ordering: one wave in front of its next k_sweep, through when k_normalise_resample
// number `value` is (submitted before this launch; bounded like the wait above). The launch boundary behind it is the acquire.
__global__ __launch_bounds__(64) void k_gate(const unsigned int* done, unsigned int value, int* flags)
{
	if (threadIdx.x == 0) {
		const long long t0 = wall_clock64();
		while ((int) (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
			if (wall_clock64() - t0 > 20000000LL) { atomicOr(flags, PHD_FLAG_ORDER_TIMEOUT); break; }
			__builtin_amdgcn_s_sleep(64);
		}
	}
}

// =================================================================================================================================
// The same step over a GRID of workgroups (round 5), for weight vectors of NR_GRID_MIN entries and more — what every rank of a
// multi-GPU run resamples per step (config C8: 16 384 weights on each of 8 GPUs): one workgroup of 1024 threads spent 48 us on
// them, bound by ONE compute unit's FP64 rate while 255 others idled. Here every thread owns ONE particle, a workgroup 256
// consecutive ones, and what the whole vector must agree on meets between four launches (a dependent launch boundary costs
// 2 - 3 us; a grid-wide barrier inside one launch costs the same in fences and needs every workgroup resident):
//   k_nr_sum      the workgroups' partial sums of the un-normalised weights
//   k_nr_stats    the total (fixed order over the workgroups), the normalised weights, per workgroup: sum of squares, first
//                 maximum, a negative / NaN weight seen, the double-double sum of its weights; per particle: the inclusive
//                 double-double prefix inside its workgroup
//   k_nr_slots    depleted? (PHDNavigator.cs:768-777); if not: identity sources, BestParticle, the roles. Else per particle the
//                 slots it takes, [N(S_k), N(S_k+1)) — the verified speculation of the kernel above, one particle per thread —,
//                 the margin tests, the candidate for BestParticle
//   k_nr_sources  every slot finds its source by a sixteen-way search over the particles' upper slot bounds (non-decreasing), the weights
//                 become 1 / P, the small arrays of the resampled particles are gathered, the roles rotate; one wave replays
//                 the recurrence literally when a margin test failed (never seen outside the tests that force it)
// The shape of every sum depends on the length of the vector only, as in the one-workgroup kernel: all ranks of a sharded run
// and a single handle holding all particles get the same bits. (The two kernels' sums have different shapes — a thread's chunk
// there, one weight per thread here —: which one runs is decided by the vector's length alone, too.)
#define NR_GRID_MIN 8192
#define NR_STAT 8   // doubles per workgroup in the statistics block: sum, sum of squares, max, index of the max, bad, dd sum hi / lo, spare
#define NR_SLOT 4   // ... in the slots block: ok, best weight, best slot, spare

struct NrGrid {
	double* part;     // [G][NR_STAT]
	double* slotres;  // [G][NR_SLOT]
	double* pre;      // [P][2] inclusive double-double prefix of a particle's weight inside its workgroup
	int*    hi;       // [P] N(S_(k+1)): the first slot particle k does NOT take
	int*    state;    // [4] decision of k_nr_slots: [0] depleted
	int     G;
};

// Exclusive double-double scan over the workgroups' sums (G <= 256 values, thread q holding workgroup q's) by the 256 threads of a
// workgroup — every workgroup computes the same numbers from the same block: excl <- the prefix in front of workgroup g,
// next <- the prefix in front of workgroup g + 1 (the total behind the last). s_hi / s_lo: 8 doubles of LDS each.
__device__ __forceinline__ void nr_scan_groups(dd mine, int g, double* s_hi, double* s_lo, int tid, dd& excl, dd& next)
{
#pragma clang fp contract(off)
	const int lane = tid & 63, wv = tid >> 6;
	dd inc = mine;   // workgroup `tid`'s sum (zero beyond the last workgroup)
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		dd y = {shfl_up_d(inc.hi, o), shfl_up_d(inc.lo, o)};
		if (lane >= o) inc = dd_add(y, inc);
	}
	if (lane == 63) { s_hi[wv] = inc.hi; s_lo[wv] = inc.lo; }
	__syncthreads();
	dd incl = inc;
	if (wv > 0) {
		dd off = {s_hi[0], s_lo[0]};
		for (int w = 1; w < wv; w++) off = dd_add(off, dd{s_hi[w], s_lo[w]});
		incl = dd_add(off, inc);
	}
	if (tid == g - 1) { s_hi[4] = incl.hi; s_lo[4] = incl.lo; }
	if (tid == g)     { s_hi[5] = incl.hi; s_lo[5] = incl.lo; }
	__syncthreads();
	excl = (g == 0) ? dd{0, 0} : dd{s_hi[4], s_lo[4]};
	next = dd{s_hi[5], s_lo[5]};
}


// this thread's weight of the vector the step works on: the gathered vector, or — single handle — the OUT bank's, whose role is
// read from the device: the three banks' words are requested together with the role (one trip to memory, not two)
__device__ __forceinline__ double nr_weight(const StepBufs& a, const double* gw, int k, bool live)
{
	if (gw) return live ? gw[k] : 0.0;
	const double w0 = live ? a.bank[0].weights[k] : 0.0, w1 = live ? a.bank[1].weights[k] : 0.0, w2 = live ? a.bank[2].weights[k] : 0.0;
	const int so = a.sel[SEL_OUT];
	return (so == 0) ? w0 : ((so == 1) ? w1 : w2);
}

//   graw != NULL (per-rank host): the all-gather landed as [rank][Pl + 1] (weights | status word): this launch is also the un-gather —
//   the weights into the contiguous vector gw, the status words behind it — instead of a launch of its own in front of it
__global__ __launch_bounds__(256) void k_nr_sum(const StepBufs a, double* gw, int P, int skip_normalise, int* sel_next, NrGrid nr,
                                                const double* __restrict__ graw, int Pl, int world)
{
	__shared__ double s4[4];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = blockIdx.x, k = g * 256 + tid;
	if (g == 0 && tid == 0 && a.bigws_used && *a.bigws_used) *a.bigws_used = 0;   // the association slab is free again
	if (g == 0 && tid < 4 && a.biglist) a.biglist[(size_t) tid * a.bigstride] = 0;
	const int flags_now = sel_next ? *a.flags : 0;   // (requested with the weights below: one trip)
	double w = 0;
	if (a.defer && !gw) {   // WeightAlpha's last line, left open by k_alpha_density (PHD_DEFER_BIG)
		if (k < P) {
			const double alpha = exp(a.setll[k] + a.ratio[k]);
			a.alpha[k] = alpha;
			w = bank_of(a, SEL_IN).weights[k] * alpha;
			bank_of(a, SEL_OUT).weights[k] = w;
		}
	}
	else if (graw) {
		if (k < P) {
			const int r = k / Pl;
			w = graw[(size_t) r * (Pl + 1) + (k - r * Pl)];
			gw[k] = w;
		}
		if (k < world) gw[P + k] = graw[(size_t) k * (Pl + 1) + Pl];
	}
	else w = nr_weight(a, gw, k, k < P);
	if (flags_now != 0) {   // a kernel of this step raised a flag: the step is dropped as a whole (see the body above)
		if (g == 0 && tid < SEL_STRIDE) sel_next[tid] = a.sel[tid];
		return;
	}
	double s = skip_normalise ? 0.0 : wave_sum(w);
	if (lane == 0) s4[wv] = s;
	__syncthreads();
	if (tid == 0) nr.part[(size_t) g * NR_STAT] = ((s4[0] + s4[1]) + s4[2]) + s4[3];
}

__global__ __launch_bounds__(256) void k_nr_stats(const StepBufs a, double* gw, int P, int skip_normalise, int* sel_next, NrGrid nr)
{
#pragma clang fp contract(off)
	__shared__ double s4[4], s4b[4], s_hi[4], s_lo[4];
	__shared__ int s_i4[4], s_bad;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = blockIdx.x, k = g * 256 + tid;
	const int flags_now = sel_next ? *a.flags : 0;
	double wk = nr_weight(a, gw, k, k < P);
	// (the workgroups' partial sums: one per thread, requested with the weight above — a loop over them was a chain of dependent trips
	// to memory, one per workgroup; the tree below has the same shape in every workgroup)
	double nsum = 1;
	if (!skip_normalise) {
		const double pv = wave_sum((tid < nr.G) ? nr.part[(size_t) tid * NR_STAT] : 0.0);
		if (lane == 0) s4[wv] = pv;
		__syncthreads();
		const double sum = ((s4[0] + s4[1]) + s4[2]) + s4[3];
		nsum = (sum == 0) ? 1 : sum;
		__syncthreads();   // (s4 is used again below)
	}
	if (flags_now != 0) return;
	if (tid == 0) s_bad = 0;
	if (k < P && !skip_normalise) {
		wk = wk / nsum;   // :343-345
		(gw ? gw : bank_of(a, SEL_OUT).weights)[k] = wk;
	}
	const bool bad = k < P && !(wk >= 0);
	// sum of squares (:772-774), first maximum (:347-354)
	const double sq = wave_sum(wk * wk);
	double m = (k < P) ? wk : -INFINITY;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
	const unsigned long long bal = ballot64(k < P && wk == m);
	const int firstl = bal ? __ffsll((long long) bal) - 1 : 0;
	// inclusive double-double prefix inside the workgroup
	dd inc = {wk, 0};
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		dd y = {shfl_up_d(inc.hi, o), shfl_up_d(inc.lo, o)};
		if (lane >= o) inc = dd_add(y, inc);
	}
	__syncthreads();
	if (bad) s_bad = 1;
	if (lane == 0) { s4[wv] = sq; s4b[wv] = bal ? m : -INFINITY; s_i4[wv] = g * 256 + wv * 64 + firstl; }
	if (lane == 63) { s_hi[wv] = inc.hi; s_lo[wv] = inc.lo; }
	__syncthreads();
	dd off = {0, 0};
	for (int w = 0; w < wv; w++) off = dd_add(off, dd{s_hi[w], s_lo[w]});
	const dd incl = (wv == 0) ? inc : dd_add(off, inc);
	if (k < P) { nr.pre[(size_t) k * 2] = incl.hi; nr.pre[(size_t) k * 2 + 1] = incl.lo; }
	if (tid == 255) { nr.part[(size_t) g * NR_STAT + 5] = incl.hi; nr.part[(size_t) g * NR_STAT + 6] = incl.lo; }   // (weights beyond P are 0: the total)
	if (tid == 0) {
		double gm = -INFINITY;
		int gi = 0;
		for (int q = 0; q < 4; q++) {
			if (s4b[q] > gm) { gm = s4b[q]; gi = s_i4[q]; }
		}
		double* o = nr.part + (size_t) g * NR_STAT;
		o[1] = ((s4[0] + s4[1]) + s4[2]) + s4[3];
		o[2] = gm; o[3] = (double) gi; o[4] = (double) s_bad;
	}
}

__global__ __launch_bounds__(256) void k_nr_slots(const StepBufs a, double* gw, int P, double min_eff, double u, int force_resample,
                                                  int* src, int* info, int* sel_next, int frozen, int* inslot, NrGrid nr)
{
#pragma clang fp contract(off)
	__shared__ double s_hi[12], s_lo[12], s_bw[4];
	__shared__ int s_hilast[4], s_ok, s_best;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = blockIdx.x, k = g * 256 + tid;
	const int flags_now = sel_next ? *a.flags : 0;
	const bool live = k < P;
	const double wk = nr_weight(a, gw, k, live);
	const dd incl = live ? dd{nr.pre[(size_t) k * 2], nr.pre[(size_t) k * 2 + 1]} : dd{0, 0};
	if (flags_now != 0) return;
	// what every workgroup derives alike from the statistics block: workgroup `tid`'s entries by thread `tid`, one trip to memory
	// (a loop over the workgroups was a chain of dependent trips), then a tree of the same shape everywhere
	double cum, gmax;
	int gbest;
	bool bad;
	dd gsum;   // workgroup `tid`'s double-double sum, for the scan below
	{
		const bool has = tid < nr.G;
		const double* st = nr.part + (size_t) (has ? tid : 0) * NR_STAT;
		const double q1 = has ? st[1] : 0.0, q2 = has ? st[2] : -INFINITY, q3 = has ? st[3] : 0.0, q4 = has ? st[4] : 0.0;
		gsum = has ? dd{st[5], st[6]} : dd{0, 0};
		const double csum = wave_sum(q1);
		double m = q2;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
		const unsigned long long mb = ballot64(has && q2 == m);
		const int fl = mb ? __ffsll((long long) mb) - 1 : 0;
		const int gi = (int) __shfl(q3, fl, 64);
		const unsigned long long bb = ballot64(q4 != 0.0);
		if (lane == 0) { s_hi[wv] = csum; s_lo[wv] = mb ? m : -INFINITY; s_hilast[wv] = gi; s_bw[wv] = bb ? 1.0 : 0.0; }
		__syncthreads();
		cum = ((s_hi[0] + s_hi[1]) + s_hi[2]) + s_hi[3];
		gmax = -INFINITY; gbest = 0;
		for (int w = 0; w < 4; w++) {
			if (s_lo[w] > gmax) { gmax = s_lo[w]; gbest = s_hilast[w]; }   // first maximum: the workgroups in order
		}
		bad = (s_bw[0] + s_bw[1] + s_bw[2] + s_bw[3]) != 0.0;
		__syncthreads();   // (the arrays are used again below)
	}
	if (!(gmax > 0)) gbest = 0;   // maxweight starts at 0 and the comparison is strict (:347-353)
	bool depleted = (1.0 / cum < min_eff * P);   // :776
	if (force_resample > 0) depleted = true;
	if (force_resample < 0) depleted = false;
	if (g == 0 && tid == 0) nr.state[0] = depleted ? 1 : 0;
	if (!depleted) {
		if (g == 0 && tid == 0) { info[0] = gbest; info[1] = 0; }
		if (k < P) {
			src[k] = k;
			if (sel_next && !frozen) inslot[k] = k;
		}
		if (sel_next && g == 0 && tid == 0) {
			const int I = a.sel[SEL_IN], O = a.sel[SEL_OUT], T = a.sel[SEL_TMP], X = a.sel[SEL_INMIX];
			if (frozen) { sel_next[SEL_IN] = I; sel_next[SEL_OUT] = O; sel_next[SEL_TMP] = T; sel_next[SEL_INMIX] = X; }
			else        { sel_next[SEL_IN] = O; sel_next[SEL_OUT] = I; sel_next[SEL_TMP] = T; sel_next[SEL_INMIX] = O; }
			sel_next[SEL_RES] = O; sel_next[SEL_RESMIX] = O;
		}
		return;
	}
	// ---- ResampleParticles (:724-760) by verified speculation, one particle per thread: particle k takes the slots
	// [N(S_k), N(S_(k+1))), N(x) = #{i : T_i <= x} (see the kernel above). S_k = the workgroup's offset + the prefix of the lane
	// before; the numbers on the two sides of every boundary — between lanes, waves and workgroups — are the same numbers, so the
	// ranges tile [0, P).
	dd woff, wnext;
	nr_scan_groups(gsum, g, s_hi, s_lo, tid, woff, wnext);
	if (tid == 0) { s_ok = 1; s_best = 0x7fffffff; }
	const double R0 = u / P, invP = 1.0 / P;
	const double B = 8.0 * P * 1.1102230246251565e-16 * (fmax(gmax, 0.0) + fabs(R0) + invP);
	bool ok = !bad, settled = true;
	// S_(k+1); at the workgroup's last particle: what the NEXT workgroup starts from
	const bool lastofgroup = tid == 255 && g + 1 < nr.G;
	const dd Sb = lastofgroup ? wnext : dd_add(woff, incl);
	const int hinat = live ? slots_upto(Sb, R0, invP, P, &settled) : 0;
	int hi = (k == P - 1) ? P : hinat;   // the recurrence runs into P: whatever is left goes to the last particle
	// the inclusive prefix and the upper slot bound of the particle before: the lane before, or the last lane of the wave before
	dd pin = {__shfl_up(incl.hi, 1, 64), __shfl_up(incl.lo, 1, 64)};
	int lo = __shfl_up(hi, 1, 64);
	if (lane == 63) { s_hi[8 + wv] = incl.hi; s_lo[8 + wv] = incl.lo; s_hilast[wv] = hi; }
	__syncthreads();
	if (lane == 0 && wv > 0) { pin = dd{s_hi[8 + wv - 1], s_lo[8 + wv - 1]}; lo = s_hilast[wv - 1]; }
	const dd S = (tid == 0) ? woff : dd_add(woff, pin);   // S_k
	if (tid == 0) lo = (g == 0) ? 0 : slots_upto(woff, R0, invP, P, &settled);   // (slot 0 belongs to the first particle even when u == 0: the clamp of :739)
	double mybestw = -INFINITY;
	int mybesti = 0x7fffffff;
	if (live) {
		if (hi < lo) { ok = false; hi = lo; }
		if (hi > lo) {
			// the last comparison that came out positive, at the particle's first slot (none for particle 0: `random` is u / P itself there)
			if (k > 0) ok = ok && dd_diff(slot_target(lo, R0, invP), S) > B;
			// the comparison that stopped the loop, at its last slot (none for the slots the last particle gets by k == P)
			if (hinat > lo) ok = ok && dd_diff(Sb, slot_target(hinat - 1, R0, invP)) > B;
			mybestw = wk; mybesti = lo;
		}
		nr.hi[k] = hi;
	}
	ok = ok && settled;
	if (!ok) s_ok = 0;
	// BestParticle = first slot whose source has the largest weight (:745-748): the workgroup's candidate
	double m = mybestw;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
	if (lane == 0) s_bw[wv] = m;
	__syncthreads();
	const double gmw = fmax(fmax(s_bw[0], s_bw[1]), fmax(s_bw[2], s_bw[3]));
	if (mybesti != 0x7fffffff && mybestw == gmw) atomicMin(&s_best, mybesti);
	__syncthreads();
	if (tid == 0) {
		double* o = nr.slotres + (size_t) g * NR_SLOT;
		o[0] = (double) s_ok; o[1] = gmw; o[2] = (double) s_best;
	}
}

//   pg.cnt != NULL (sharded step: the vector holds world x Pl slots): this launch also COUNTS for the migration plan (k_plan_count's
//   work, phd_kernels.h) — every thread has its slot's source in a register here, and a launch of its own for the counting cost as
//   much as the counting; k_plan_lists follows. gflags: the gathered status words (a flag on any rank: nothing is counted).
__global__ __launch_bounds__(256) void k_nr_sources(const StepBufs a, double* gw, int P, double u, int* src, int* info, int* sel_next,
                                                    int frozen, int* inslot, NrGrid nr, PlanGrid pg, int Pl, int world, int rank,
                                                    const double* gflags)
{
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = blockIdx.x, i = g * 256 + tid;
	const int flags_now = sel_next ? *a.flags : 0, depleted = nr.state[0];
	const bool plan = pg.cnt != nullptr && !plan_dropped(a.flags, gflags, world, lane);   // (its loads with the two above: one trip)
	if (flags_now != 0 || depleted == 0) return;   // a dropped step | not depleted: k_nr_slots ended the step
	double* gwp = gw ? gw : bank_of(a, SEL_OUT).weights;
	// the workgroups' verdicts and candidates: one per thread, one trip, then the same reduction everywhere
	__shared__ double s_gm[4];
	__shared__ int s_bs[4], s_okw[4];
	bool ok;
	double gm;
	int best;
	{
		const bool has = tid < nr.G;
		const double* r = nr.slotres + (size_t) (has ? tid : 0) * NR_SLOT;
		const double r0 = has ? r[0] : 1.0, r1 = has ? r[1] : -INFINITY, r2 = has ? r[2] : 2147483647.0;
		double m = r1;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
		int bs = (has && r1 == m) ? (int) r2 : 0x7fffffff;   // the smallest slot among the candidates with the largest weight
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) bs = min(bs, __shfl_xor(bs, o, 64));
		const unsigned long long nb = ballot64(r0 == 0.0);
		if (lane == 0) { s_gm[wv] = m; s_bs[wv] = bs; s_okw[wv] = nb ? 0 : 1; }
		__syncthreads();
		gm = fmax(fmax(s_gm[0], s_gm[1]), fmax(s_gm[2], s_gm[3]));
		best = 0x7fffffff;
		for (int w = 0; w < 4; w++) if (s_gm[w] == gm) best = min(best, s_bs[w]);
		ok = (s_okw[0] & s_okw[1] & s_okw[2] & s_okw[3]) != 0;
	}
	const Bank bo = bank_of(a, SEL_OUT), bt = bank_of(a, SEL_TMP);
	auto finish_slot = [&](int s_i, int slot) {   // slot `slot` takes particle s_i: the small arrays of the resampled state (rotate_roles above)
		if (sel_next) {
			const int c0 = bo.count[s_i];
			double q0[7];
#pragma unroll
			for (int t = 0; t < 7; t++) q0[t] = bo.poses[(size_t) s_i * 7 + t];
			bt.count[slot] = c0; bt.weights[slot] = 1.0 / P;
#pragma unroll
			for (int t = 0; t < 7; t++) bt.poses[(size_t) slot * 7 + t] = q0[t];
			if (!frozen) inslot[slot] = s_i;
		}
	};
	auto roles = [&]() {
		const int I = a.sel[SEL_IN], O = a.sel[SEL_OUT], T = a.sel[SEL_TMP], X = a.sel[SEL_INMIX];
		if (frozen) { sel_next[SEL_IN] = I; sel_next[SEL_OUT] = O; sel_next[SEL_TMP] = T; sel_next[SEL_INMIX] = X; }
		else        { sel_next[SEL_IN] = T; sel_next[SEL_OUT] = I; sel_next[SEL_TMP] = O; sel_next[SEL_INMIX] = O; }
		sel_next[SEL_RES] = T; sel_next[SEL_RESMIX] = O;
	};
	if (ok) {
		if (g == 0 && tid == 0) {
			info[0] = (gm > 0) ? best : 0; info[1] = 1;
			if (sel_next) roles();
		}
		// the source of slot i: the first particle whose upper slot bound lies beyond i (the bounds never decrease; the last is P)
		// (sixteen ways per trip to memory — fifteen pivots requested together — instead of two: four trips at 16 384 weights,
		// not fourteen; a workgroup alone on its CU pays every dependent trip in full)
		// With the plan's counting in this launch a thread also wants the source of the slot BEFORE its own (is its slot the head of
		// a run?): both are searched in lockstep — the two searches' pivots requested together, the same four trips.
		int lo = 0, lo2 = 0;
		if (i < P) {
			const int i2 = plan ? max(i - 1, 0) : i;
			int cnt = P, cnt2 = P;   // the answers lie in [lo, lo + cnt), [lo2, lo2 + cnt2): hi[lo + cnt - 1] > slot
			while (cnt > 1 || cnt2 > 1) {
				const int stride = (cnt + 15) >> 4, end = lo + cnt, stride2 = (cnt2 + 15) >> 4, end2 = lo2 + cnt2;
				int below = 0, below2 = 0;
				int v[15], v2[15];
#pragma unroll
				for (int j = 0; j < 15; j++) {
					const int idx = lo + (j + 1) * stride - 1, idx2 = lo2 + (j + 1) * stride2 - 1;
					v[j] = (cnt > 1 && idx < end - 1) ? nr.hi[idx] : 0x7fffffff;
					v2[j] = (plan && cnt2 > 1 && idx2 < end2 - 1) ? nr.hi[idx2] : 0x7fffffff;
				}
#pragma unroll
				for (int j = 0; j < 15; j++) { below += (v[j] <= i) ? 1 : 0; below2 += (v2[j] <= i2) ? 1 : 0; }
				if (cnt > 1) { lo += below * stride; cnt = min(stride, end - lo); }
				if (plan) { if (cnt2 > 1) { lo2 += below2 * stride2; cnt2 = min(stride2, end2 - lo2); } }
				else { lo2 = lo; cnt2 = cnt; }
			}
			src[i] = lo;
			gwp[i] = 1.0 / P;   // :742
			finish_slot(lo, i);
		}
		if (plan) {   // (uniform over the launch; whole waves: the vector holds a multiple of 64 slots)
			int t, sr;
			bool head, bad;
			plan_flags(lo, lo2, i, P, Pl, 1.0f / (float) Pl, P >= (1 << 24), t, sr, head, bad);
			plan_count_slot(pg, i, P, Pl, world, rank, lane, g * 4 + wv, lo, t, sr, head, bad);
		}
		return;
	}
	// ---- fallback: the recurrence itself, by one wave (weights through v_readlane, scalar control flow); the weights are
	// normalised in place already
	if (g != 0) return;
	if (tid < 64) {
		auto rl = [](double v, int l) {
			int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
			return __hiloint2double(hi, lo);
		};
		const double invP = 1.0 / P;
		double random = u / P, maxweight = 0;
		int k = 0, bestslot = 0, cb = 0;
		double cur = (lane < P) ? gwp[lane] : 0.0, prev = 0.0;
		for (int s = 0; s < P; s++) {
			while (k < P && __builtin_amdgcn_readfirstlane((int) (random > 0))) {
				if (k >= cb + 64) { prev = cur; cb += 64; cur = (cb + lane < P) ? gwp[cb + lane] : 0.0; }
				random -= rl(cur, k - cb);
				k++;
			}
			const int sidx = (k - 1 < 0) ? 0 : k - 1;   // u == 0 would index -1 in the reference: clamped
			if (lane == 0) src[s] = sidx;
			random += invP;
			const double ws = (sidx >= cb) ? rl(cur, sidx - cb) : rl(prev, sidx - (cb - 64));
			if (__builtin_amdgcn_readfirstlane((int) (ws > maxweight))) { maxweight = ws; bestslot = s; }
		}
		if (lane == 0) { info[0] = bestslot; info[1] = 1; }
	}
	__threadfence_block();
	__syncthreads();
	for (int s = tid; s < P; s += 256) {
		gwp[s] = 1.0 / P;
		finish_slot(src[s], s);
	}
	if (sel_next && tid == 0) roles();
	if (plan) {   // the counting for the plan, by this one workgroup over all slots (whole waves: the vector holds a multiple of 64 slots)
		for (int base = 0; base < P; base += 256) {
			const int gs = base + tid;
			int sv, t, sr;
			bool head, bad;
			plan_look(src, gs, P, Pl, 1.0f / (float) Pl, P >= (1 << 24), lane, sv, t, sr, head, bad);
			plan_count_slot(pg, gs, P, Pl, world, rank, lane, (base >> 6) + wv, sv, t, sr, head, bad);
		}
	}
}
