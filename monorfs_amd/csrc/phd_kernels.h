// phd_kernels.h — hand-written HIP kernels (gfx950, wave64) of the RB-PHD-SLAM inner loop.
//
// One workgroup per particle everywhere (particles are independent through predict / correct /
// prune / reweight: PHDNavigator.cs:326-339). Mixtures live in HBM as ONE 80-byte record per component
// (w, mean x/y/z, covariance xx/xy/xz/yy/yz/zz), [particle][slot][10] doubles (round 4; ten planes before): a
// wavefront that reads 64 consecutive components of a particle reads 5 KB of whole lines with five 16-byte loads per
// lane, and a component picked by index (the gathers of k_prune_merge, the queued pairs of k_emit_finish, the picks of
// BestMapEstimate) costs two 64-byte sectors instead of ten.
//
//   k_sweep (phd_sweep.h), k_emit_finish (phd_correct.h) : PredictConditional + CorrectConditional
//                       (+ the MinWeight cut of PruneModel)
//   k_prune_merge     : PruneModel (sort by weight, MaxQuantity cap, greedy merge)
//   k_weight_alpha    : WeightAlpha = BestMapEstimate + mixture densities + SetLogLikelihood
//   k_normalise_resample, k_gather_particles : particle weights, BestParticle, systematic resampling
#pragma once
#include "phd_device.h"

// status bits written by the kernels into StepBufs::flags
#define PHD_FLAG_EMIT_OVERFLOW   1   // corrected components did not fit emit_capacity
#define PHD_FLAG_J_OVERFLOW      2   // map estimate larger than the landmark scratch
#define PHD_FLAG_BIG_CLUSTER     4   // association cluster beyond the on-device solver's cap
#define PHD_FLAG_ORDER_TIMEOUT   8   // a kernel that orders the sub-range streams on the device gave up waiting (see k_normalise_resample, k_gate)

#define MIX_REC 10   // doubles per component record: w, m[3], P[6] (upper triangle)

struct MixView {
	double* rec;      // component i of the bank (particle slot s, index c: i = s * cap + c) at rec + i * MIX_REC
	int*    count;
};

// a component record in and out of registers: five 16-byte accesses (records are 80 bytes apart, 16-byte aligned)
__device__ __forceinline__ void load_comp(const double* __restrict__ r, double& w, double m[3], double P[6])
{
	const double2* q = (const double2*) r;
	const double2 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
	w = a.x; m[0] = a.y; m[1] = b.x; m[2] = b.y;
	P[0] = c.x; P[1] = c.y; P[2] = d.x; P[3] = d.y; P[4] = e.x; P[5] = e.y;
}

__device__ __forceinline__ void store_comp(double* __restrict__ r, double w, const double m[3], const double P[6])
{
	double2* q = (double2*) r;
	q[0] = make_double2(w, m[0]);
	q[1] = make_double2(m[1], m[2]);
	q[2] = make_double2(P[0], P[1]);
	q[3] = make_double2(P[2], P[3]);
	q[4] = make_double2(P[4], P[5]);
}

// `n` whole records from one place to another by the threads of the workgroup (16 bytes per thread and trip: whole lines)
__device__ __forceinline__ void copy_comps(double* __restrict__ dst, const double* __restrict__ src, int n, int tid, int nthreads)
{
	double2* d = (double2*) dst;
	const double2* s = (const double2*) src;
	for (int i = tid; i < n * (MIX_REC / 2); i += nthreads) d[i] = s[i];
}

// one of the three state banks: mixture records [Pcap][cap][10], counts, poses, weights
struct Bank {
	double* mix;
	int*    count;
	double* poses;    // [Pcap][7]
	double* weights;  // [Pcap]
};

// Roles of the three banks. A bank holds the mixture records of all particles and their small arrays (count, pose,
// weight). Resampling does not copy mixtures: after it the small arrays of the new particles sit in one bank (SEL_IN)
// while their mixtures are still the ones the step wrote into another (SEL_INMIX), particle p's at slot inslot[p]
// (the deep copies of PHDNavigator.cs:740-741 are what an indirection makes of them). OUT differs from IN and INMIX;
// TMP differs from IN and OUT (it may be INMIX: only its small arrays are written).
#define SEL_IN     0   // bank whose small arrays a step reads
#define SEL_OUT    1   // bank a step writes
#define SEL_TMP    2   // bank the small arrays of a resampled state go to
#define SEL_RES    3   // bank holding the small arrays of the last step's result
#define SEL_INMIX  4   // bank whose mixtures a step reads, through inslot
#define SEL_RESMIX 5   // bank holding the mixtures of the last step's result (slots: the resampling sources)
#define SEL_STRIDE 8

struct StepBufs {
	int P;          // particles of this handle
	int p0;         // first particle of this launch (a step may be split into sub-ranges on concurrent streams)
	int cap;        // slots per particle in a mixture slab
	int M;          // measurements
	int Mcap;       // stride of per-measurement scratch
	int ecap;       // emit scratch slots per particle
	int Jcap;       // landmark scratch per particle
	Bank bank[3];
	const int* sel; // [SEL_STRIDE] device-resident roles of the banks for this step (no host round trip to rotate them)
	const int* inslot;   // [P] slot of particle p's mixture in the INMIX bank
	const double* z;         // [M][3]
	// corrected-but-unpruned components (weight >= MinWeight), unsorted
	double* emit_w;      // [P][ecap]
	int*    emit_idx;    // [P][ecap] canonical position in the reference's `corrected` list
	double* emit_rec;    // [P][ecap][10] the detection updates as component records (w, mean, covariance upper triangle); not written for the misdetection copies
	int*    emit_count;  // [P]
	// births of the predict step
	int*    born_count;  // [P]
	int*    born_k;      // [P][Mcap]
	double* born_mean;   // [P][Mcap][3]
	// reweight outputs
	double* alpha;       // [P]
	double* setll;       // [P]
	int*    flags;       // [1]
	struct MurtyNodes* murty;   // [P] workspace of the big-cluster solver (clusters of 6 .. 64 rows)
	char*   bigws;       // association slab: workspaces of the clusters beyond 64 rows, handed out by a bump counter
	unsigned long long  bigws_bytes;
	unsigned long long* bigws_used;   // reset at the end of every step (k_normalise_resample) / by the host behind a stage launch; a quasi batch brings its own
	double* jscratch;    // [P] landmark-indexed arrays of k_weight_alpha when the map estimate outgrows LDS
	// (component, measurement) pairs that may reach MinWeight, queued by k_sweep for k_emit_finish
	int*    cand;        // [P][candcap]
	int     candcap;
	int*    cand_count;  // [P][4] entries in each wave's segment of the queue (above candcap / 4: it overflowed)
	double* denom;       // [P][Mcap] kappa + weightsum[z]
	double* srec;        // [P][cutcap][12] k_prune_merge: the kept records in sorted order (component record, canonical index, spare)
	double* outw;        // [P][cap] the weights of the pruned mixture once more, as a plane of their own (k_prune_merge -> BestMapEstimate of k_alpha_assoc, which reads nothing else of most components)
	// map estimate handed from k_alpha_assoc to k_alpha_density
	double* alm;         // [P][3][Jcap] landmark means
	int*    aJ;          // [P] landmarks
	double* account;     // [P] expected size of the corrected map
	// QuasiSetLogLikelihood batches (k_quasi_setll): candidate poses, the landmark set, its size
	const double* qposes;   // [P][7]
	const double* qlm;      // [qJ][3]
	int     qJ;
	double* qgrad;       // [P][6] pose gradients (k_quasi_setll_grad)
	int     qavg;        // TemperedAverage normalisation: 0 as the source reads, 1 weights / their sum
	double* wcopy;       // [P][cap + Mcap] weight of the surviving misdetection copy of predicted component c (0: none), k_prune_merge -> k_alpha_density
	int*    cover;       // [P][cap] 1: this pruned component is such a copy
	double* stamps;      // [P][16] phase stamps of the diagnostic build (NULL otherwise)
	int*    biglist;     // k_alpha_assoc_main -> k_alpha_big: [0] entries, [1 ..] the particles of this launch whose association needs the ordered replay
	int     bigstride;   // ints between the lists of two sub-ranges
	double* ratio;       // [P] k_alpha_density -> k_alpha_combine: the density part of log alpha
	int     defer;       // 1: the step runs k_alpha_assoc_main / k_alpha_big / k_alpha_combine (k_alpha_density leaves alpha open)
	int     all_pairs;   // 1: k_sweep evaluates every (component, measurement) pair, the radius gate only masks (SURVEY §8d's benchmark
	                     // mode: the unit count P C M is exact); 0: a visit whose 64 pairs all lie outside the gate is skipped
	// k_normalise_resample folded into the one-launch chain (small particle sets): the workgroup that takes the last ticket runs it
	int           fold_nr;      // 1: k_particle_chain ends the step itself
	unsigned int* ticket;       // [0] workgroups of the step's last per-particle launch(es) that are through (set back to 0 by whoever waited for them)
	                            // [1] the number of the last step whose k_normalise_resample is through
	int           tickets;      // 1: every workgroup of k_alpha_density publishes its weight and takes a ticket (device-side ordering of the two sub-range streams)
	int           wait_tickets; // k_normalise_resample: 1 = wait until the ticket counter has reached ticket_target (0: the stream has ordered it)
	unsigned int  ticket_target;// ... P times the number of device-ordered steps so far: the counter only ever grows, so tickets that arrive behind a
	                            // timed-out wait still count for their own step and the next one's wait is not satisfied early
	unsigned int  done_value;   // k_normalise_resample: the step number it publishes in ticket[1] when through (0: none)
	double        nr_u;         // the arguments k_normalise_resample would have got
	int           nr_force, nr_skip, nr_frozen;
	int*          nr_src;
	int*          nr_info;
	int*          nr_sel_next;
	int*          nr_inslot;
	int     stamp_kernel; // which kernel writes them (env PHD_STAMP_KERNEL): 2 prune, 3 assoc, 4 density, 1 correct, 5 the one-launch chain
	// k_particle_chain with a HELPER workgroup per particle for the densities of WeightAlpha (two workgroups per particle for that body)
	int           dsplit;   // 1: the launch has 2 P workgroups, the second P are the helpers (2 - 4: test and measuring switches, PHD_DSPLIT_LATE)
	unsigned int  dstamp;   // this launch's number (never 0, grows by one per launch of the chain): what the words below are compared with
	unsigned int* dsync;    // [P][3]: helper ready (16 dstamp + its XCD) | the main's word (2 dstamp + {0 it keeps the density sums, 1 the helper runs them}) | ticket (alpha_meet)
};

// The bank playing `role`. The roles rotate on the device (a.sel lives in device memory), so the index is not known
// at launch; choosing among the three kernel-argument entries by comparison keeps them in scalar registers — indexing
// the array dynamically makes the compiler copy the argument block to scratch memory.
__device__ __forceinline__ Bank bank_of(const StepBufs& a, int role)
{
	const int s = a.sel[role];
	Bank b;
	b.mix     = (s == 0) ? a.bank[0].mix     : ((s == 1) ? a.bank[1].mix     : a.bank[2].mix);
	b.count   = (s == 0) ? a.bank[0].count   : ((s == 1) ? a.bank[1].count   : a.bank[2].count);
	b.poses   = (s == 0) ? a.bank[0].poses   : ((s == 1) ? a.bank[1].poses   : a.bank[2].poses);
	b.weights = (s == 0) ? a.bank[0].weights : ((s == 1) ? a.bank[1].weights : a.bank[2].weights);
	return b;
}

// The mixtures of the bank playing `role` and the counts that go with them. For SEL_IN the records are those of the
// INMIX bank: particle p's components start at in_base(a, p), its count is count[p].
__device__ __forceinline__ MixView bank_view(const StepBufs& a, int role)
{
	const Bank b = bank_of(a, role);
	const double* mix = (role == SEL_IN) ? bank_of(a, SEL_INMIX).mix : b.mix;
	MixView v;
	v.rec = const_cast<double*>(mix);
	v.count = b.count;
	return v;
}

__device__ __forceinline__ size_t in_base(const StepBufs& a, int p) { return (size_t) a.inslot[p] * a.cap; }

#define TILE 256   // components staged per LDS tile

#define PLAN_GRID_MIN 8192               // (= NR_GRID_MIN: the plan's first half runs inside the grid resampling's last launch)
#define PLAN_GRID_MAXSLOTS 65536          // per-wave count arrays: 1024 global waves

struct PlanGrid {
	int* cnt;      // [n][n] records rank t takes from rank s            (this launch pair's set; NULL: no plan is counted)
	int* wcg;      // [Pg / 64] heads whose source is mine, per global wave
	int* lcg;      // [Pl / 64] heads among my own slots, per local wave
	unsigned int* used;   // [(Pl + 31) / 32] bit c: OUT slot c stays the source of a local particle
	int* bad;      // [1]
	int* cnt_next; unsigned int* used_next; int* bad_next;   // the other set: cleared by k_plan_lists
};

// the flags of slot g from its source s and the source of the slot before it (see k_plan_migration): owner rank, source rank, head
// of a run fed from another rank, malformed
__device__ __forceinline__ void plan_flags(int s, int prev, int g, int Pg, int Pl, float rPl, bool bigidx, int& t, int& sr, bool& head, bool& bad)
{
	auto rank_of = [&](int x) { return bigidx ? x / Pl : small_div(x, Pl, rPl); };
	bad = g < Pg && (s < 0 || s >= Pg || (g > 0 && prev > s));
	t = rank_of(min(g, Pg - 1));
	sr = rank_of(min(max(s, 0), Pg - 1));
	head = g < Pg && sr != t && (g == t * Pl || prev != s);
}

// ... with the sources read from the vector: this lane's, and the lane before's by a shuffle (lane 0: the word before)
__device__ __forceinline__ void plan_look(const int* __restrict__ gsrc, int g, int Pg, int Pl, float rPl, bool bigidx, int lane,
                                          int& s, int& t, int& sr, bool& head, bool& bad)
{
	s = (g < Pg) ? gsrc[g] : 0;
	int prev = __shfl_up(s, 1, 64);
	if (lane == 0) prev = (g > 0 && g < Pg) ? gsrc[g - 1] : 0;
	plan_flags(s, prev, g, Pg, Pl, rPl, bigidx, t, sr, head, bad);
}

// a flag raised on this rank, or — the gathered status words, one per lane — on any other: one trip to memory
__device__ __forceinline__ bool plan_dropped(const int* lflags, const double* gflags, int n, int lane)
{
	const int lf = *lflags;
	const double gf = (gflags && lane < n) ? gflags[lane] : 0.0;
	return lf != 0 || ballot64(gf != 0.0) != 0ull;
}

// What slot g adds to the plan's accumulators (all 64 lanes of a wave call it together; gwave: the wave's number among all slots'
// waves): the count matrix, the waves' head counts, the bitmap of OUT slots that stay a local source. Pl is a multiple of 64 — the
// host takes the one-workgroup kernel otherwise —, so a wave's slots belong to one rank.
__device__ __forceinline__ void plan_count_slot(const PlanGrid& pg, int g, int Pg, int Pl, int n, int rank, int lane, int gwave,
                                                int s, int t, int sr, bool head, bool bad)
{
	const int first = rank * Pl;
	if (bad) atomicOr(pg.bad, 1);
	if (head && !bad) atomicAdd(&pg.cnt[t * n + sr], 1);
	const unsigned long long mine = ballot64(head && !bad && sr == rank);
	if (lane == 0 && g < Pg) pg.wcg[gwave] = __popcll(mine);
	const bool myslot = g < Pg && t == rank;
	if (myslot) {
		// the heads among my own slots, per wave of them: the records' numbers (k_plan_lists)
		const unsigned long long lb = ballot64(head && !bad);
		if (lane == 0) pg.lcg[(g - first) >> 6] = __popcll(lb);
	}
	// The OUT slots that stay a local particle's source, as a bitmap. The sources never decrease, so a wave's 64 slots name a run
	// of neighbouring bits — mostly the same few: the lanes OR theirs together per word first (an atomic per slot was 2048
	// atomics on 64 words, serialised at the L2: most of the launch) and the first lane of each word's run adds it.
	const bool loc = myslot && sr == rank && !bad;
	const int word = loc ? (s - first) >> 5 : -1 - lane;   // (distinct negative numbers: no run)
	unsigned int bits = loc ? 1u << ((s - first) & 31) : 0u;
	// segmented OR over runs of equal `word` (the lanes of a run are neighbours): log steps, a lane takes what the lane `o`
	// further on holds when that lane belongs to the same word
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int wo = __shfl_down(word, o, 64);
		const unsigned int bo = __shfl_down(bits, o, 64);
		if (lane + o < 64 && wo == word) bits |= bo;
	}
	const int wp = __shfl_up(word, 1, 64);
	if (loc && (lane == 0 || wp != word)) atomicOr(&pg.used[word], bits);
}

#include "phd_correct.h"

// (PHD_ONLY_EP: a translation unit of k_emit_finish / k_prune_merge / k_emit_prune alone — scripts/kres.sh compiles it in seconds
// to read one kernel's registers and scratch while it is being worked on; never the product build)
#ifndef PHD_ONLY_EP
#include "phd_sweep.h"
#endif

#include "phd_prune.h"

#ifndef PHD_ONLY_EP
#include "phd_alpha.h"

#include "phd_resample.h"

#ifndef PHD_HELPER_PRIO
#define PHD_HELPER_PRIO 3   // s_setprio of the chain's helper workgroups while they run the density sums
#endif

// The per-particle chain of a step as ONE launch, for small particle sets (the real-time regime of the reference: 20 - 800
// particles at 30 Hz, plots/scripts/chap3/S4-particles.sh:14-15; BASELINE config A): a particle's predict / correct / prune /
// reweight touch nothing of another particle, so its workgroup runs the five kernels' bodies back to back, with a
// workgroup barrier where a launch boundary was, all of them in ONE LDS pool (the largest of their layouts,
// chain_lds_bytes: ~38 KB). With at most two workgroups per CU the kernels were latency-bound launches of 8 - 30 us each
// with a ramp and a tail; here there are no boundaries until the particle weights meet in k_normalise_resample, and
// registers to spare (245). On a full machine (2048 particles) a 128-register build of it at four workgroups per CU
// measured 0.785 ms per step against 0.71 for the separate kernels on two streams: not used there.
template <int ZB>
__host__ __device__ inline int chain_lds_bytes(int cutcap)
{
	int d = SweepLds<ZB>::doubles;
	if (EMIT_LDS_DOUBLES > d) d = EMIT_LDS_DOUBLES;
	if (DENS_LDS_DOUBLES > d) d = DENS_LDS_DOUBLES;
	int b = d * 8;
	const int pb = prune_lds(cutcap).bytes, ab = alpha_lds(ZB * 64, cutcap).bytes;
	if (pb > b) b = pb;
	if (ab > b) b = ab;
	return b;
}

template <int ZB, bool HALF = false>
__global__ __launch_bounds__(256, 1) void k_particle_chain(const DevParams prm, const StepBufs a, int cutcap, int with_alpha)
{
	extern __shared__ __align__(16) double smem[];
	// Two workgroups per particle where the chain allows it (a.dsplit: the launch has 2 nmain workgroups): WeightAlpha's density sums
	// (alpha_density_body) need the pruned map and the map estimate, not the association — so the HELPER of particle p, workgroup
	// nmain + p, runs them BESIDE the main workgroup's association instead of behind it. The helper says it is there (the launch's
	// number and its XCD) and waits for the main's word; the MAIN never waits: where its map estimate is final (alpha_assoc_body)
	// it looks whether its helper has reported from the same XCD and hands the sums over, or keeps them — the same body, the same
	// bits either way. The two meet in alpha_meet. A helper only ever waits for a workgroup that was handed out before it, and
	// gives up (step dropped: a.flags) after 2 s of the 100 MHz counter.
	const int nmain = a.dsplit ? (int) (gridDim.x >> 1) : (int) gridDim.x;
	if (a.dsplit && (int) blockIdx.x >= nmain) {
		// (whose helper: workgroups go to the XCDs in turn, workgroup b to XCD b mod 8, so helper h = b - nmain takes, within its group of
		// eight particles, the one whose main workgroup has b's residue — with a particle count that is no multiple of 8 the helpers
		// would otherwise all sit on another XCD than their mains and never be picked; the last, incomplete group stays as it is)
		const int h = (int) blockIdx.x - nmain;
		const int p = a.p0 + (((h | 7) < nmain) ? (h & ~7) + ((h + nmain) & 7) : h);
		__shared__ int s_go[1];
		if (!with_alpha || a.dsplit == 3) return;   // (3: a measuring switch — helpers that leave at once)
		// (while it waits: what the density sums can have ready from the prior mixture alone)
		const double pre_pcount = alpha_density_prestage(a, smem, p);
		if (threadIdx.x == 0) {
			unsigned int* w = a.dsync + 3 * (size_t) p;
			if (a.dsplit == 2) {   // (test switch PHD_DSPLIT_LATE=1: a helper that reports 0.5 ms late — every main keeps its sums)
				const long long t1 = wall_clock64();
				while (wall_clock64() - t1 < 50000LL) __builtin_amdgcn_s_sleep(64);
			}
			__hip_atomic_store(w, (a.dstamp << 4) | (a.dsplit == 4 ? 15u : my_xcd()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (4: a measuring switch — helpers nobody picks)
			const long long t0 = wall_clock64();
			int go = -1;
			for (;;) {
				const unsigned int v = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((v >> 1) == a.dstamp) { go = (int) (v & 1u); break; }
				if (wall_clock64() - t0 > 200000000LL) break;
				__builtin_amdgcn_s_sleep(8);
			}
			if (go < 0) { atomicOr(a.flags, PHD_FLAG_ORDER_TIMEOUT); go = 0; }   // (cannot happen with workgroups handed out in order; the step is dropped)
			s_go[0] = go;
		}
		__syncthreads();
		const int go = s_go[0];
		__syncthreads();
		if (!go) return;
		// (the helper's sums are the longer of the two paths behind the hand-over: its waves go ahead of the main's where both sit on one
		// SIMD — 256 particles 0.0862 -> 0.0855 ms, profiles/r05_chain_helpers_prio.txt)
		__builtin_amdgcn_s_setprio(PHD_HELPER_PRIO);
		// every wave: what the main wrote in front of its word is in the L2 both share; this CU's L1 and the scalar cache may hold older lines
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef PHD_STAMPS   // (slot 14: ticks from the main's hand-over to this point; slot 15: the helper's density sums, hand-over to end)
		const long long tw_ = wall_clock64();
		const bool fin_ = alpha_density_body(prm, a, smem, p, true, true, pre_pcount);
		if (threadIdx.x == 0 && a.stamps && a.stamp_kernel == 5) {
			const double t0_ = a.stamps[(size_t) p * 16 + 13];
			a.stamps[(size_t) p * 16 + 14] = (double) tw_ - t0_;
			a.stamps[(size_t) p * 16 + 15] = (double) wall_clock64() - t0_;
		}
		if (!fin_) return;
#else
		if (!alpha_density_body(prm, a, smem, p, true, true, pre_pcount)) return;
#endif
	}
	else {
	PHD_STAMP_DECL;
	PHD_STAMP(0);
	sweep_body<ZB, HALF>(prm, a, smem);
	__syncthreads();   // (workgroup scope: the global writes of the step before are visible to this workgroup's loads)
	PHD_STAMP(1);
	emit_finish_body<false, true>(prm, a, smem);
	__syncthreads();
	PHD_STAMP(2);
	prune_merge_body(prm, a, cutcap, smem);
	if (with_alpha) {
		__shared__ int s_hgo;
		if (threadIdx.x == 0) s_hgo = 0;
		__syncthreads();
		PHD_STAMP(3);
		alpha_assoc_body<ZB, false, false, 1>(prm, a, cutcap, smem, nullptr, -1, a.dsplit ? &s_hgo : nullptr);
		__syncthreads();
		PHD_STAMP(4);
		if (!s_hgo) alpha_density_body(prm, a, smem);
		else {
			// (the helper has the sums: this workgroup's number is the set log-likelihood it has just written)
			if (threadIdx.x == 0) {
				const int p = a.p0 + (int) blockIdx.x;
#ifdef PHD_STAMPS   // (slot 11: the association behind the hand-over, ticks)
				if (a.stamps && a.stamp_kernel == 5) a.stamps[(size_t) p * 16 + 11] = (double) wall_clock64() - a.stamps[(size_t) p * 16 + 13];
#endif
				const double sl = a.setll[p];
				__hip_atomic_store(a.setll + p, sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				s_hgo = alpha_meet(a, p, false, sl) ? 1 : 0;
			}
			__syncthreads();
			if (!s_hgo) return;
		}
		PHD_STAMP(5);
	}
	PHD_STAMP_FLUSH(5, 6);   // (diagnostic build, PHD_STAMP_KERNEL=5: the bodies' shares of the chain)
	}
	if (a.fold_nr) {
		// The end of the step without a launch of its own: behind the barrier every wave's stores have left the CU; ONE thread
		// publishes them (release fence, device scope: the XCD's L2 is written back — by every thread that is 1024 write-backs
		// per launch and cost more than the launch saved) and takes a ticket; the workgroup that takes the last has every
		// particle's weight, count and flag before it (acquire fence: L1 and the L2's stale lines invalidated) and runs
		// k_normalise_resample's body on the launch's LDS pool.
		__shared__ int s_last;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores are acknowledged by the L2
		__syncthreads();
		if (threadIdx.x == 0) {
			__threadfence();
			const int last = ((int) atomicAdd(a.ticket + 2, 1u) == nmain - 1) ? 1 : 0;   // (a word of its own: the device order's counter beside it never goes back)
			if (last) a.ticket[2] = 0;   // for the next launch (stream order)
			__threadfence();
			s_last = last;
		}
		__syncthreads();
		if (s_last) {   // (workgroup-uniform)
			normalise_resample_body(a, nullptr, a.P, prm.min_eff, a.nr_u, a.nr_force, a.nr_skip, 1, a.nr_src, a.nr_info, a.nr_sel_next,
			                        a.nr_frozen, a.nr_inslot, smem);
		}
	}
}

#ifdef PHD_WITH_FUSE_SEP   // (a measured-and-dropped variant, DESIGN §4: compiled only into diagnostic builds, -DPHD_WITH_FUSE_SEP; the product
                           // library carries neither its four instantiations nor the switch)
// k_sweep, k_emit_finish and k_prune_merge as ONE launch on a full machine (environment PHD_FUSE_SEP=1): the first three bodies
// of the chain above at k_sweep's own register budget (four workgroups per CU), k_alpha_assoc and k_alpha_density behind it as
// launches of their own (compiled into one kernel with the others they cost occupancy: 245 registers). The workgroups of a CU
// drift apart — some in the dense pair loops, some in the latency-bound prune — and two launch boundaries go.
template <int ZB, bool HALF = false>
__global__ __launch_bounds__(256, 4) void k_sweep_emit_prune(const DevParams prm, const StepBufs a, int cutcap)
{
	extern __shared__ __align__(16) double smem[];
	sweep_body<ZB, HALF>(prm, a, smem);
	__syncthreads();
	emit_finish_body(prm, a, smem);
	__syncthreads();
	prune_merge_body(prm, a, cutcap, smem);
}
#endif

// Test surface (phd_stage_map, PHD_STAGE_CORRECTED): the emitted list carries no mean / covariance for the misdetection
// copies (k_sweep writes their weight and index only); this fills them in from the predicted components.
__global__ __launch_bounds__(256) void k_expand_emit(const DevParams prm, const StepBufs a)
{
	const int p = a.p0 + blockIdx.x, tid = threadIdx.x;
	const MixView vin = bank_view(a, SEL_IN);
	const int n = vin.count[p], np = n + a.born_count[p], ne = a.emit_count[p];
	const size_t eb = (size_t) p * a.ecap;
	for (int e = tid; e < ne; e += 256) {
		const int cidx = a.emit_idx[eb + e];
		if (cidx < np) {
			double w, m[3], P[6];
			load_predicted(prm, a, vin, p, n, cidx, w, m, P);
			store_comp(a.emit_rec + (eb + e) * MIX_REC, a.emit_w[eb + e], m, P);
		}
	}
}

// The mixtures of the current state gathered into its own bank: particle i <- (INMIX, inslot[i]) written to (IN, i),
// after which INMIX = IN and the slots are the identity. Run before anything that addresses mixtures by particle
// number in bulk (uploads and downloads of whole states, single-map writes, the sharded step's migration).
// (the role INMIX = IN is written by the host once the launch has drained: a workgroup that did it here would redirect the
// reads of the workgroups that start after it)
__global__ __launch_bounds__(256) void k_materialise(const StepBufs a, int* inslot)
{
	const int i = blockIdx.x, tid = threadIdx.x;
	const MixView from = bank_view(a, SEL_IN);
	const Bank bi = bank_of(a, SEL_IN);
	const int n = bi.count[i];
	const size_t db = (size_t) i * a.cap, fb = in_base(a, i);
	copy_comps(bi.mix + db * MIX_REC, from.rec + fb * MIX_REC, n, tid, 256);
	__syncthreads();
	if (tid == 0) inslot[i] = i;   // only this workgroup reads inslot[i]
}

// Particle motion: TrackVehicle.UpdateNoisy (TrackVehicle.cs:89-102) = Pose3D.AddOdometry
// (Pose3D.cs:314-333) of the odometry reading, then of the particle's own noise vector (drawn by the host).
struct Quat4 { double w, x, y, z; };

__device__ __forceinline__ Quat4 quat_mul(const Quat4& a, const Quat4& b)   // Quaternion.cs:295-301
{
	return Quat4{a.w * b.w - (a.x * b.x + a.y * b.y + a.z * b.z),
	             a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
	             a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
	             a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}

__device__ inline void add_odometry(double s[7], const double* d)
{
	const Quat4 q{s[3], s[4], s[5], s[6]};
	const double l0 = 0.5 * d[3], l1 = 0.5 * d[4], l2 = 0.5 * d[5];   // FromLinear, Quaternion.cs:145-149
	const double phi = sqrt(l0 * l0 + l1 * l1 + l2 * l2);
	Quat4 dq{1, 0, 0, 0};                                              // Exp, :185-196
	if (!(phi < 1e-12)) {
		const double sn = sin(phi);
		dq = Quat4{cos(phi), sn * (l0 / phi), sn * (l1 / phi), sn * (l2 / phi)};
	}
	const Quat4 nq = quat_mul(q, dq);
	Quat4 mid{1, 0, 0, 0};                                             // Sqrt, :225-235
	if (!(fabs(dq.w - -1.0) < 1e-8)) {
		const double rw = sqrt(0.5 * (1 + dq.w)), alpha = 1 / (2 * rw);
		mid = Quat4{rw, alpha * dq.x, alpha * dq.y, alpha * dq.z};
	}
	const Quat4 mr = quat_mul(q, mid);
	const Quat4 dl = quat_mul(quat_mul(mr, Quat4{0, d[0], d[1], d[2]}), Quat4{mr.w, -mr.x, -mr.y, -mr.z});
	const double alpha = 1 / sqrt(nq.w * nq.w + nq.x * nq.x + nq.y * nq.y + nq.z * nq.z);   // Normalize, :240-245
	s[0] += dl.x; s[1] += dl.y; s[2] += dl.z;
	s[3] = alpha * nq.w; s[4] = alpha * nq.x; s[5] = alpha * nq.y; s[6] = alpha * nq.z;
}

// (the bank holding the current poses is resolved here, on the device: correct right behind an asynchronous step)
__global__ __launch_bounds__(256) void k_motion(const StepBufs a, int P, const double* odometry, const double* noise, int use_noise)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= P) return;
	double* poses = bank_of(a, SEL_IN).poses;
	double s[7], d[6];
#pragma unroll
	for (int t = 0; t < 7; t++) s[t] = poses[(size_t) i * 7 + t];
#pragma unroll
	for (int t = 0; t < 6; t++) d[t] = odometry[t];
	add_odometry(s, d);
	if (use_noise) {
#pragma unroll
		for (int t = 0; t < 6; t++) d[t] = noise[(size_t) i * 6 + t];
		add_odometry(s, d);
	}
#pragma unroll
	for (int t = 0; t < 7; t++) poses[(size_t) i * 7 + t] = s[t];
}

// phd_set_poses / phd_set_weights: staged values into the small arrays of the current state (the IN bank, whichever
// it is by now). One thread per double.
__global__ __launch_bounds__(256) void k_store_small(const StepBufs a, const double* poses, const double* weights, int P)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	const Bank bi = bank_of(a, SEL_IN);
	if (poses && i < P * 7) bi.poses[i] = poses[i];
	if (weights && i < P) bi.weights[i] = weights[i];
}

// replicate particle 0 of the IN bank over `P` particles of the OUT bank (PHDNavigator.reset, :256-263)
__global__ __launch_bounds__(256) void k_replicate(const StepBufs a, double weight)
{
	const int i = blockIdx.x, tid = threadIdx.x;
	const MixView from = bank_view(a, SEL_IN), dst = bank_view(a, SEL_OUT);
	const int n = from.count[0];
	const size_t db = (size_t) i * a.cap, fb = in_base(a, 0);
	copy_comps(dst.rec + db * MIX_REC, from.rec + fb * MIX_REC, n, tid, 256);
	const Bank bi = bank_of(a, SEL_IN);
	const Bank bo = bank_of(a, SEL_OUT);
	if (tid == 0) {
		dst.count[i]  = n;
		bo.weights[i] = weight;
	}
	if (tid < 7) bo.poses[(size_t) i * 7 + tid] = bi.poses[tid];
}

// =================================================================================================
// Sharded step: particles are sharded contiguously over ranks (one rank = one GPU: a process of its own
// with RCCL, or a shard of a phd_create_multi handle); after the global resampling a slot may need a particle that lives
// on another rank. A migrating particle travels as one fixed-size record: [count, pose(7), component records (cap x 10)] doubles.
// Everything between the global resampling kernel and the next step is decided ON THE DEVICE (k_plan_migration): the host
// never needs the source vector, only — where a collective wants split sizes (RCCL all-to-all) — 2 n counts.
// =================================================================================================

// The un-normalised weights of the local step, stored straight into the gathered weight vector of every destination
// (dst[t] + first): the shards of a multi-device handle write their slice into every peer's vector through peer-mapped
// pointers (the all-gather of SURVEY §5 / §8e as 16 KB of stores per peer, no copy engine, no host call per pair); the
// per-rank host hands in one destination, the buffer its collective reads. gflag: the step's status word goes along
// (slot `flagslot` behind the weights of every destination), so that a step dropped on one shard is dropped on all.
__global__ __launch_bounds__(256) void k_push_weights(const StepBufs a, double* const* dst, int ndst, int first, int flagslot)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	double* w = bank_of(a, SEL_OUT).weights;
	if (i < a.P) {
		if (a.defer) {   // WeightAlpha's last line, left open by k_alpha_density (see k_normalise_resample)
			const double alpha = exp(a.setll[i] + a.ratio[i]);
			a.alpha[i] = alpha;
			w[i] = bank_of(a, SEL_IN).weights[i] * alpha;
		}
		const double v = w[i];
		for (int t = 0; t < ndst; t++) dst[t][first + i] = v;
	}
	if (i == 0 && flagslot >= 0) {
		const double f = (double) *a.flags;
		for (int t = 0; t < ndst; t++) dst[t][flagslot] = f;
	}
}

// Per-rank host: the all-gather lands as [rank][Pl + 1] — a rank's un-normalised weights and, behind them, its status word.
// The weights go to the contiguous vector the global kernel takes (gw[world Pl]), the status words behind it (gw[world Pl + r]).
__global__ __launch_bounds__(256) void k_ungather(const double* __restrict__ graw, double* __restrict__ gw, int Pl, int world)
{
	const int g = blockIdx.x * 256 + threadIdx.x, Pg = Pl * world;
	if (g < Pg) {
		const int r = g / Pl, i = g - r * Pl;
		gw[g] = graw[(size_t) r * (Pl + 1) + i];
	}
	else if (g < Pg + world) {
		const int r = g - Pg;
		gw[g] = graw[(size_t) r * (Pl + 1) + Pl];
	}
}

// The migration plan of one rank, device-resident. counts: [0, n) records sent to rank t, [n, 2n) records received from
// rank s, then nsend, nrecv, status, resampled.
#define MIG_OK        0
#define MIG_DROPPED   1   // a kernel of the step raised a flag (on this or, multi-device handle, on any shard): nothing moves
#define MIG_BAD       2   // the source vector is not a resampling result (not non-decreasing, or out of range)
#define MIG_OVERFLOW  3   // more records than the send list holds
struct MigPlan {
	int* code;         // [Pl]  per local slot: >= 0 local source slot, < 0: -(k + 1) = record k of the receive buffer
	int* fslot;        // [Pl]  OUT-bank slot record k is unpacked into (one no local particle keeps as its source)
	int* sendlist;     // [sendcap] local slots to pack, grouped by destination rank (ascending), then by destination slot
	long long* senddst;// [sendcap] destination rank << 32 | record number in that rank's receive buffer
	int* counts;       // [2 n + 4]
	int  sendcap;
};

// exclusive prefix sum of one int per thread over the threads of the workgroup (wsum: 17 ints of LDS); *total <- the sum
__device__ __forceinline__ int block_excl_scan(int v, int* wsum, int tid, int* total)
{
	const int lane = tid & 63, wv = tid >> 6, nw = (int) (blockDim.x >> 6);
	int incl = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int y = __shfl_up(incl, o, 64);
		if (lane >= o) incl += y;
	}
	__syncthreads();   // (wsum may still be read from the scan before)
	if (lane == 63) wsum[wv] = incl;
	__syncthreads();
	int off = 0, tot = 0;
	for (int q = 0; q < nw; q++) { const int x = wsum[q]; off += (q < wv) ? x : 0; tot += x; }
	*total = tot;
	return off + incl - v;
}

// phd_plan_migration (phdhip.hip, the host statement of the same plan, kept as the ABI's pure function and as this kernel's
// reference in the tests) for the source vectors systematic resampling produces: those are non-decreasing (the recurrence of
// PHDNavigator.cs:731-738 only ever advances k), so "the previous slot of rank t whose source lies on rank s" is simply the
// slot before — every list is a prefix sum over flags of neighbouring slots. One workgroup of 1024 threads; every rank
// runs it on the same global vector and derives matching lists (the sender's order per destination is the receiver's order
// per source). The vector is read in rounds of 1024 consecutive slots (coalesced; the slot before comes from the
// neighbouring lane); positions in the lists are (flags before in the round's waves) + (flags before in the wave), the
// first from one scan over the per-round, per-wave counts.
//   gsrc [n Pl] global source of every slot; info[1] resampled; lflags: this rank's status word; gflags: the status words
//   of all ranks as gathered with the weights (NULL: per-rank host); hostcounts: pinned host memory the counts are
//   written to as well, followed by `seq` (the host polls that word instead of synchronising the stream); NULL: none
// LDS (ints): cnt[n][n] base[n] roff[n] wsum[20] | wc[rounds][16] send counts per round and wave | used[(Pl + 31) / 32]
//   | sg[n Pl] the source vector itself, when `staged` (it is read five times over: from LDS the rounds do not each wait
//   for a trip to memory)
#define PLAN_LDS_MAX (150 * 1024)
__host__ __device__ inline size_t plan_lds_fixed(int Pl, int n)
{
	const size_t rounds = ((size_t) Pl * n + 1023) / 1024;
	return ((size_t) n * n + 2 * n + 24 + rounds * 16 + 16 + (Pl + 31) / 32 + 4) * 4;
}
__host__ __device__ inline bool plan_staged(int Pl, int n) { return plan_lds_fixed(Pl, n) + (size_t) Pl * n * 4 <= PLAN_LDS_MAX; }
__host__ __device__ inline size_t plan_lds_bytes(int Pl, int n) { return plan_lds_fixed(Pl, n) + (plan_staged(Pl, n) ? (size_t) Pl * n * 4 : 0); }

//   gw (may be NULL): this rank's slice of the gathered (normalised, or 1 / P) weights goes back into its OUT bank here
__global__ __launch_bounds__(1024) void k_plan_migration(const int* __restrict__ gsrc, const int* __restrict__ info, const int* lflags,
                                                         const double* gflags, int Pl, int n, int rank, MigPlan pl, int* hostcounts, int seq,
                                                         const StepBufs a, const double* gw)
{
	extern __shared__ int sm[];
	const int tid = threadIdx.x, nt = (int) blockDim.x, lane = tid & 63, wv = tid >> 6;
	const int Pg = Pl * n, first = rank * Pl;
	if (gw) {
		double* wout = bank_of(a, SEL_OUT).weights;
		for (int i = tid; i < Pl; i += nt) wout[i] = gw[first + i];
	}
	const int rounds = (Pg + nt - 1) / nt, lrounds = (Pl + nt - 1) / nt;
	int* const cnt  = sm;                          // [n][n] records rank t takes from rank s
	int* const base = cnt + n * n;                 // [n] first send-list position of destination t
	int* const roff = base + n;                    // [n] record number, in destination t's receive buffer, of my first record for it
	int* const wsum = roff + n;                    // [20]
	int* const wc   = wsum + 20;                   // [rounds][16] flags per round and wave, then their exclusive prefix
	int* const used = wc + rounds * 16 + 16;       // [(Pl + 31) / 32] bit c: OUT slot c stays the source of a local particle
	int* const sg   = used + (Pl + 31) / 32 + 4;   // [Pg] the source vector (staged)
	const bool staged = plan_staged(Pl, n);
	__shared__ int s_bad;
	const float rPl = 1.0f / (float) Pl;
	bool drop = *lflags != 0;
	if (gflags) for (int t = 0; t < n; t++) drop = drop || gflags[t] != 0.0;
	const int resampled = info[1];
	int status = drop ? MIG_DROPPED : MIG_OK, nsend = 0, nrecv = 0;
	const bool bigidx = Pg >= (1 << 24);   // (the float quotient of small_div needs 24-bit operands; beyond that: the division)
	auto rank_of = [&](int s) { return bigidx ? s / Pl : small_div(s, Pl, rPl); };
	// the flags of slot g: does a new run start here, who owns the slot, who the source
	auto look = [&](int g, int& s, int& t, int& sr, bool& head, bool& bad) {
		int prev;
		if (staged) { s = sg[g]; prev = g > 0 ? sg[g - 1] : 0; }
		else {
			s = gsrc[g];
			prev = __shfl_up(s, 1, 64);
			if (lane == 0) prev = g > 0 ? gsrc[g - 1] : 0;
		}
		bad = s < 0 || s >= Pg || (g > 0 && prev > s);
		t = rank_of(g);
		sr = rank_of(min(max(s, 0), Pg - 1));
		head = sr != t && (g == t * Pl || prev != s);   // a new run of slots of rank t fed by a particle of rank sr: one record
	};
	if (resampled && !drop) {   // (uniform)
		for (int i = tid; i < n * n; i += nt) cnt[i] = 0;
		for (int i = tid; i < (Pl + 31) / 32; i += nt) used[i] = 0;
		if (tid == 0) s_bad = 0;
		if (staged) {   // eight loads in flight per thread, then their stores
			for (int b0 = tid; b0 < Pg; b0 += 8 * nt) {
				int v[8];
#pragma unroll
				for (int q = 0; q < 8; q++) v[q] = (b0 + q * nt < Pg) ? gsrc[b0 + q * nt] : 0;
#pragma unroll
				for (int q = 0; q < 8; q++) if (b0 + q * nt < Pg) sg[b0 + q * nt] = v[q];
			}
		}
		__syncthreads();
		// ---- all slots: who takes a record from whom; how many records of mine every round and wave holds
		bool anybad = false;
		for (int r = 0; r < rounds; r++) {
			const int g = r * nt + tid;
			int s = 0, t = 0, sr = 0;
			bool head = false, bad = false;
			if (g < Pg) look(g, s, t, sr, head, bad);
			anybad = anybad || bad;
			if (head && !bad) atomicAdd(&cnt[t * n + sr], 1);
			const unsigned long long mine = ballot64(head && !bad && sr == rank);
			if (lane == 0) wc[r * 16 + wv] = __popcll(mine);
		}
		if (anybad) s_bad = 1;
		__syncthreads();
		{   // exclusive prefix over the (round, wave) counts, in slot order: rounds x 16 entries, chunked over the threads
			const int tot = rounds * 16, CHW = (tot + nt - 1) / nt;
			const int e0 = min(tot, tid * CHW), e1 = min(tot, e0 + CHW);
			int mysum = 0;
			for (int e = e0; e < e1; e++) mysum += wc[e];
			int run = block_excl_scan(mysum, wsum, tid, &nsend);
			for (int e = e0; e < e1; e++) { const int x = wc[e]; wc[e] = run; run += x; }
		}
		if (tid < n) {
			int b = 0, r = 0;
			for (int t = 0; t < tid; t++) b += (t != rank) ? cnt[t * n + rank] : 0;     // destinations before `tid`
			for (int s = 0; s < rank; s++) r += (s != tid) ? cnt[tid * n + s] : 0;      // sources before me at destination `tid`
			base[tid] = b; roff[tid] = r;
		}
		__syncthreads();
		if (s_bad) status = MIG_BAD;
		else if (nsend > pl.sendcap) status = MIG_OVERFLOW;
		if (status == MIG_OK) {
			// ---- my send list: the heads among other ranks' slots whose source is mine, in slot order (= by destination, then slot)
			for (int r = 0; r < rounds; r++) {
				const int g = r * nt + tid;
				int s = 0, t = 0, sr = 0;
				bool head = false, bad = false;
				if (g < Pg) look(g, s, t, sr, head, bad);
				const bool mine = head && sr == rank;
				const unsigned long long bal = ballot64(mine);
				if (mine) {
					const int k = wc[r * 16 + wv] + __popcll(bal & lanemask_lt());
					pl.sendlist[k] = s - first;
					pl.senddst[k] = ((long long) t << 32) | (long long) (roff[t] + (k - base[t]));
				}
			}
			// ---- my slots: local source, or the record that feeds the run the slot belongs to (records numbered in slot order:
			// with non-decreasing sources that is the order "by source rank, then by slot" the sender packs them in)
			__syncthreads();   // (wc is reused)
			for (int r = 0; r < lrounds; r++) {
				const int i = r * nt + tid;
				bool head = false;
				if (i < Pl) {
					int s, t, sr; bool bad;
					look(first + i, s, t, sr, head, bad);
				}
				const unsigned long long bal = ballot64(head);
				if (lane == 0) wc[r * 16 + wv] = __popcll(bal);
			}
			__syncthreads();
			{
				const int tot = lrounds * 16, CHW = (tot + nt - 1) / nt;
				const int e0 = min(tot, tid * CHW), e1 = min(tot, e0 + CHW);
				int mysum = 0;
				for (int e = e0; e < e1; e++) mysum += wc[e];
				int run = block_excl_scan(mysum, wsum, tid, &nrecv);
				for (int e = e0; e < e1; e++) { const int x = wc[e]; wc[e] = run; run += x; }
			}
			__syncthreads();
			for (int r = 0; r < lrounds; r++) {
				const int i = r * nt + tid;
				int s = 0, t = 0, sr = 0;
				bool head = false, bad = false;
				if (i < Pl) look(first + i, s, t, sr, head, bad);
				const unsigned long long bal = ballot64(head);
				if (i < Pl) {
					if (sr == rank) {
						pl.code[i] = s - first;
						atomicOr(&used[(s - first) >> 5], 1 << ((s - first) & 31));
					}
					else {
						// heads up to and including this slot: the number of the record that feeds it (a slot inside a run
						// carries the count of its run's head: no head lies in between)
						const int slot = wc[r * 16 + wv] + __popcll(bal & (lanemask_lt() | (1ull << lane)));
						pl.code[i] = -slot;   // record slot - 1
					}
				}
			}
			__syncthreads();
			// ---- an arriving particle is unpacked into a slot of the OUT bank that no local particle keeps as its source
			// (at least nrecv slots are fed from elsewhere, so the Pl slots keep at most Pl - nrecv distinct local sources: at
			// least nrecv slots of the OUT bank are free)
			{
				const int words = (Pl + 31) / 32, CHW = (words + nt - 1) / nt;
				const int e0 = min(words, tid * CHW), e1 = min(words, e0 + CHW);
				int nfree = 0;
				for (int e = e0; e < e1; e++) {
					unsigned int fr = ~(unsigned int) used[e];
					if (e == words - 1 && (Pl & 31)) fr &= (1u << (Pl & 31)) - 1u;
					nfree += __popc(fr);
				}
				int totfree;
				int f = block_excl_scan(nfree, wsum, tid, &totfree);
				for (int e = e0; e < e1 && f < nrecv; e++) {
					unsigned int fr = ~(unsigned int) used[e];
					if (e == words - 1 && (Pl & 31)) fr &= (1u << (Pl & 31)) - 1u;
					while (fr && f < nrecv) {
						const int bit = __ffs((int) fr) - 1;
						fr &= fr - 1;
						pl.fslot[f++] = e * 32 + bit;
					}
				}
			}
		}
		else { nsend = 0; nrecv = 0; }
	}
	__syncthreads();
	const bool live = resampled && status == MIG_OK;
	for (int t = tid; t < n; t += nt) {
		pl.counts[t]     = (live && t != rank) ? cnt[t * n + rank] : 0;
		pl.counts[n + t] = (live && t != rank) ? cnt[rank * n + t] : 0;
	}
	if (tid == 0) {
		pl.counts[2 * n] = nsend; pl.counts[2 * n + 1] = nrecv; pl.counts[2 * n + 2] = status; pl.counts[2 * n + 3] = resampled;
	}
	if (hostcounts) {
		// the same words into pinned host memory, then the sequence number the host waits for (system-scope stores; the
		// fence orders the counts before it)
		for (int t = tid; t < n; t += nt) {
			__hip_atomic_store(hostcounts + t, (live && t != rank) ? cnt[t * n + rank] : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + n + t, (live && t != rank) ? cnt[rank * n + t] : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		if (tid == 0) {
			__hip_atomic_store(hostcounts + 2 * n, nsend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + 2 * n + 1, nrecv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + 2 * n + 2, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + 2 * n + 3, resampled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + 2 * n + 4, info[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + 2 * n + 5, *lflags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		__threadfence_system();
		__syncthreads();
		if (tid == 0) __hip_atomic_store(hostcounts + 2 * n + 6, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// =================================================================================================================================
// The same plan over a GRID of workgroups (round 5), for global vectors of PLAN_GRID_MIN slots and more: one workgroup of 1024
// threads took 34 us for the five passes over an 8 x 2048 vector — instruction-bound on ONE compute unit, on every rank, in
// every resampling step. Here every thread owns one slot of the global vector and two launches do the work:
//   k_plan_count   every slot: is it the head of a run that needs a record (flags of neighbouring slots, as above)? The n x n
//                  count matrix by atomics (only heads add: a few hundred), per wave the heads whose source is MINE (the send
//                  list's order) and, over this rank's own slots, the heads at all (the record numbers) and the bitmap of OUT
//                  slots that stay a local particle's source
//   k_plan_lists   positions = (counts of the waves before) + (heads before in the wave): the send list with each record's
//                  destination and number, the code of every local slot; one workgroup lays the arrivals' free slots out and
//                  writes the counts (to the host too, when it waits for them)
// Two sets of the accumulators alternate between launches: k_plan_lists clears the set the NEXT pair of launches adds to, so
// that no launch — and no memset on the stream — stands between the resampling kernel and k_plan_count.
__global__ __launch_bounds__(256) void k_plan_count(const int* __restrict__ gsrc, const int* __restrict__ info, const int* lflags,
                                                    const double* gflags, int Pl, int n, int rank, PlanGrid pg)
{
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int Pg = Pl * n, g = blockIdx.x * 256 + tid;
	int s, t, sr;
	bool head, bad;
	const bool bigidx = Pg >= (1 << 24);
	plan_look(gsrc, g, Pg, Pl, 1.0f / (float) Pl, bigidx, lane, s, t, sr, head, bad);   // (its loads are in flight while the status words arrive)
	const int resampled = info[1];
	const bool drop = plan_dropped(lflags, gflags, n, lane);
	if (drop || !resampled) return;   // dropped, or not resampled: nothing moves (k_plan_lists writes the status)
	plan_count_slot(pg, g, Pg, Pl, n, rank, lane, blockIdx.x * 4 + wv, s, t, sr, head, bad);
}

//   gw (may be NULL): this rank's slice of the gathered (normalised, or 1 / P) weights goes back into its OUT bank here
__global__ __launch_bounds__(256) void k_plan_lists(const int* __restrict__ gsrc, const int* __restrict__ info, const int* lflags,
                                                    const double* gflags, int Pl, int n, int rank, MigPlan pl, PlanGrid pg, int* hostcounts, int seq,
                                                    const StepBufs a, const double* gw)
{
	__shared__ int s_cnt[64 * 64], s_base[64], s_roff[64], s_w[20], s_ns, s_nr;   // (at most 64 ranks: PHD_MAX_DEVICES)
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int Pg = Pl * n, first = rank * Pl, g = blockIdx.x * 256 + tid;
	int s, t, sr;
	bool head, bad;
	const bool bigidx = Pg >= (1 << 24);
	plan_look(gsrc, g, Pg, Pl, 1.0f / (float) Pl, bigidx, lane, s, t, sr, head, bad);
	// Everything this launch reads of the counting's results is requested up front, together — the count matrix into LDS, the
	// waves' counts into registers — and only then looked at: every dependent trip to memory is a microsecond here.
	const int resampled = info[1], badword = pg.bad[0];
	const bool drop = plan_dropped(lflags, gflags, n, lane);
	if (gw && g >= first && g < first + Pl) bank_of(a, SEL_OUT).weights[g - first] = gw[g];   // this rank's slice of the weights
	// the accumulators of the NEXT pair of launches (the set the pair before this one added to): cleared whatever this step does —
	// the host alternates the sets with every pair, and a set left as a resampling step filled it would be met again two pairs on
	{
		const int gt = blockIdx.x * 256 + tid, gn = gridDim.x * 256;
		for (int q = gt; q < n * n; q += gn) pg.cnt_next[q] = 0;
		for (int q = gt; q < (Pl + 31) / 32; q += gn) pg.used_next[q] = 0u;
		if (gt == 0) pg.bad_next[0] = 0;
	}
	if (!resampled || drop) {
		// Nothing was counted (the counting returns at the same test) and nothing moves: the step's most common end on a frame
		// that does not deplete the particle set. One workgroup writes the status; this pair's set of accumulators is still clear.
		if (blockIdx.x == 0) {
			for (int q = tid; q < 2 * n; q += 256) pl.counts[q] = 0;
			if (tid == 0) { pl.counts[2 * n] = 0; pl.counts[2 * n + 1] = 0; pl.counts[2 * n + 2] = drop ? MIG_DROPPED : MIG_OK; pl.counts[2 * n + 3] = resampled; }
			if (hostcounts) {
				for (int q = tid; q < 2 * n + 2; q += 256) __hip_atomic_store(hostcounts + q, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				if (tid == 0) {
					__hip_atomic_store(hostcounts + 2 * n + 2, drop ? MIG_DROPPED : MIG_OK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					__hip_atomic_store(hostcounts + 2 * n + 3, resampled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					__hip_atomic_store(hostcounts + 2 * n + 4, info[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					__hip_atomic_store(hostcounts + 2 * n + 5, *lflags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
				__threadfence_system();
				__syncthreads();
				if (tid == 0) __hip_atomic_store(hostcounts + 2 * n + 6, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
		return;
	}
	int cv[16];
#pragma unroll
	for (int q = 0; q < 16; q++) cv[q] = (tid + 256 * q < n * n) ? pg.cnt[tid + 256 * q] : 0;
	const int gwv = blockIdx.x * 4 + wv;                 // this wave's number among all slots' waves (at most 1024)
	const bool myslots = g < Pg && t == rank;            // (wave-uniform: Pl is a multiple of 64)
	const int lw = myslots ? (g - first) >> 6 : 0;       // ... and among my own slots' waves
	int wq[16], lq[16];
#pragma unroll
	for (int q = 0; q < 16; q++) {
		wq[q] = (lane + 64 * q < gwv) ? pg.wcg[lane + 64 * q] : 0;
		lq[q] = (lane + 64 * q < lw) ? pg.lcg[lane + 64 * q] : 0;
	}
#pragma unroll
	for (int q = 0; q < 16; q++) if (tid + 256 * q < n * n) s_cnt[tid + 256 * q] = cv[q];
	__syncthreads();
	if (tid < n) {
		int b = 0, r = 0;
		for (int q = 0; q < tid; q++) b += (q != rank) ? s_cnt[q * n + rank] : 0;      // destinations before `tid`
		for (int q = 0; q < rank; q++) r += (q != tid) ? s_cnt[tid * n + q] : 0;       // sources before me at destination `tid`
		s_base[tid] = b; s_roff[tid] = r;
	}
	if (wv == 0) {
		int ns = (lane < n && lane != rank) ? s_cnt[lane * n + rank] : 0, nr_ = (lane < n && lane != rank) ? s_cnt[rank * n + lane] : 0;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { ns += __shfl_xor(ns, o, 64); nr_ += __shfl_xor(nr_, o, 64); }
		if (lane == 0) { s_ns = ns; s_nr = nr_; }
	}
	__syncthreads();
	int status = drop ? MIG_DROPPED : MIG_OK, nsend = 0, nrecv = 0;
	if (resampled && !drop) {
		if (badword) status = MIG_BAD;
		nsend = s_ns; nrecv = s_nr;
		if (status == MIG_OK && nsend > pl.sendcap) status = MIG_OVERFLOW;
	}
	const bool live = resampled && status == MIG_OK;
	if (live) {
		// heads of mine in the waves before this one; heads among my slots in the local waves before
		int acc = 0, lacc = 0;
#pragma unroll
		for (int q = 0; q < 16; q++) { acc += wq[q]; lacc += lq[q]; }
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o, 64); lacc += __shfl_xor(lacc, o, 64); }
		const bool mine = head && sr == rank;
		const unsigned long long bal = ballot64(mine);
		if (mine) {
			const int k = acc + __popcll(bal & lanemask_lt());
			pl.sendlist[k] = s - first;
			pl.senddst[k] = ((long long) t << 32) | (long long) (s_roff[t] + (k - s_base[t]));
		}
		if (myslots) {   // (wave-uniform)
			const int i = g - first;
			// heads up to and including this slot, among my slots: the number of the record that feeds it (a slot inside a run
			// carries the count of its run's head: no head lies in between)
			const unsigned long long hb = ballot64(head);
			if (sr == rank) pl.code[i] = s - first;
			else pl.code[i] = -(lacc + __popcll(hb & (lanemask_lt() | (1ull << lane))));   // record (that count) - 1
		}
	}
	if (blockIdx.x != 0) return;
	// ---- one workgroup: the arrivals' free slots, the counts
	if (live) {
		const int words = (Pl + 31) / 32, CHW = (words + 255) / 256;
		const int e0 = min(words, tid * CHW), e1 = min(words, e0 + CHW);
		int nfree = 0;
		for (int e = e0; e < e1; e++) {
			unsigned int fr = ~pg.used[e];
			if (e == words - 1 && (Pl & 31)) fr &= (1u << (Pl & 31)) - 1u;
			nfree += __popc(fr);
		}
		int totfree;
		int f = block_excl_scan(nfree, s_w, tid, &totfree);
		for (int e = e0; e < e1 && f < nrecv; e++) {
			unsigned int fr = ~pg.used[e];
			if (e == words - 1 && (Pl & 31)) fr &= (1u << (Pl & 31)) - 1u;
			while (fr && f < nrecv) {
				const int bit = __ffs((int) fr) - 1;
				fr &= fr - 1;
				pl.fslot[f++] = e * 32 + bit;
			}
		}
	}
	else { nsend = 0; nrecv = 0; }
	for (int q = tid; q < n; q += 256) {
		pl.counts[q]     = (live && q != rank) ? s_cnt[q * n + rank] : 0;
		pl.counts[n + q] = (live && q != rank) ? s_cnt[rank * n + q] : 0;
	}
	if (tid == 0) {
		pl.counts[2 * n] = nsend; pl.counts[2 * n + 1] = nrecv; pl.counts[2 * n + 2] = status; pl.counts[2 * n + 3] = resampled;
	}
	if (hostcounts) {
		for (int q = tid; q < n; q += 256) {
			__hip_atomic_store(hostcounts + q, (live && q != rank) ? s_cnt[q * n + rank] : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + n + q, (live && q != rank) ? s_cnt[rank * n + q] : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		if (tid == 0) {
			__hip_atomic_store(hostcounts + 2 * n, nsend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + 2 * n + 1, nrecv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + 2 * n + 2, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + 2 * n + 3, resampled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + 2 * n + 4, info[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(hostcounts + 2 * n + 5, *lflags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		__threadfence_system();
		__syncthreads();
		if (tid == 0) __hip_atomic_store(hostcounts + 2 * n + 6, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// Pack the particles other ranks take: record k of the send list = particle sendlist[k] of the OUT bank. sendbuf != NULL:
// the records go, in list order, into this rank's send buffer (the host's all-to-all moves them); NULL: each record is
// stored straight into its place in the destination's receive buffer (recvbase[t], a peer-mapped pointer: multi-device
// handle). The count comes from the device plan: a fixed grid strides over the records.
__global__ __launch_bounds__(256) void k_pack_particles(const StepBufs a, const MigPlan pl, int n, double* sendbuf, double* const* recvbase)
{
	const int tid = threadIdx.x;
	const int nsend = pl.counts[2 * n];
	if (pl.counts[2 * n + 2] != MIG_OK) return;
	const MixView from = bank_view(a, SEL_OUT);
	const Bank bo = bank_of(a, SEL_OUT);
	const size_t rec = (size_t) 8 + (size_t) 10 * a.cap;
	for (int r = blockIdx.x; r < nsend; r += gridDim.x) {
		const int s = pl.sendlist[r];
		double* o;
		if (sendbuf) o = sendbuf + (size_t) r * rec;
		else {
			const long long d = pl.senddst[r];
			o = recvbase[(int) (d >> 32)] + (size_t) (d & 0xffffffffll) * rec;
		}
		const int nc = from.count[s];
		if (tid == 0) o[0] = (double) nc;
		if (tid < 7) o[1 + tid] = bo.poses[(size_t) s * 7 + tid];
		copy_comps(o + 8, from.rec + (size_t) s * a.cap * MIX_REC, nc, tid, 256);
	}
}

// The landing flags (round 5): behind k_pack_particles on the sender's stream, one wave stores the step's number into word
// `rank` of the flag area at the end of EVERY peer's receive buffer (fine-grained memory, system-scope release: this launch
// begins when the pack kernel — its peer stores with it — has ended, and the fence orders whatever is still in flight
// before the flag). The receiver's k_finish_sharded waits for the words of the ranks it takes records from: the
// one-word all-reduce that played landing barrier until round 4 is a second collective the step does not need.
__global__ __launch_bounds__(64) void k_post_landing(double* const* recvbase, int n, int rank, size_t flagoff, unsigned long long seq)
{
	const int t = threadIdx.x;
	__threadfence_system();
	if (t < n && t != rank) {
		unsigned long long* w = (unsigned long long*) (recvbase[t] + flagoff) + rank;
		__hip_atomic_store(w, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// The receiver's wait as ONE wave in front of k_finish_sharded (the default): lane t polls the word of rank t when this step
// takes records from it; the launch boundary behind it is the acquire for everything k_finish_sharded reads. A grid that waits —
// the same loop inside k_finish_sharded, PHD_LANDING_INLINE=1: one launch boundary (~3 us) less — holds every slot of the
// device for as long as it waits: harmless when the senders run on OTHER devices, a standstill when ranks share one (seen:
// four processes with 2048-particle shards on one GPU), and thousands of waves polling the memory the peers are storing into.
__global__ __launch_bounds__(64) void k_wait_landing(const MigPlan pl, int n, const unsigned long long* landing, unsigned long long seq,
                                                     long long landing_ticks, int* flags)
{
	const int tid = threadIdx.x;
	const int nrecv = pl.counts[2 * n + 1], status = pl.counts[2 * n + 2], resampled = pl.counts[2 * n + 3];
	if (status != MIG_OK || !resampled || nrecv <= 0) return;
	if (tid < n && pl.counts[n + tid] > 0) {
		const long long t0 = wall_clock64();
		while ((long long) (__hip_atomic_load(landing + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
			if (wall_clock64() - t0 > landing_ticks) { atomicOr(flags, PHD_FLAG_ORDER_TIMEOUT); break; }
			__builtin_amdgcn_s_sleep(8);
		}
	}
	__threadfence_system();
}

// End of a sharded step, one workgroup per local particle; what it does is read from the device plan, not decided by the
// host (rotate_roles in phd_resample.h has the rules of the single-handle step, which are these):
//   dropped step (a flag was raised): the roles stay as they were, nothing is touched
//   not resampled: (IN, OUT, TMP, INMIX) = (O, I, T, O), slots identity
//   resampled: as in the single-handle step no local mixture is copied — particle i whose source is a local particle reads
//     that particle's slot of the OUT bank from now on; a particle that arrives from another rank (record j of the receive
//     buffer) is unpacked into a slot of the OUT bank that no local particle uses as a source (fslot[j]) and read from
//     there. Block b unpacks record b (if there is one) and sets up particle b: small arrays into TMP, slot into inslot.
//     (IN, OUT, TMP, INMIX) = (T, I, O, O)
//   frozen: roles and slots stay (benchmark steady state); RES / RESMIX / slots say where the result is
//   landing != NULL: the flag words of this rank's receive buffer (k_post_landing); a step that takes records from rank t waits
//   for word t to reach `seq` — bounded (landing_ticks of the 100 MHz counter; PHD_FLAG_ORDER_TIMEOUT: a peer that never posts has died)
__global__ __launch_bounds__(256) void k_finish_sharded(const StepBufs a, const MigPlan pl, int n, const double* recvbuf, double weight,
                                                        int* sel_next, int frozen, int* inslot, int* slots,
                                                        const unsigned long long* landing, unsigned long long seq, long long landing_ticks)
{
	const int i = blockIdx.x, tid = threadIdx.x;
	const int nrecv = pl.counts[2 * n + 1], status = pl.counts[2 * n + 2], resampled = pl.counts[2 * n + 3];
	const int I = a.sel[SEL_IN], O = a.sel[SEL_OUT], T = a.sel[SEL_TMP], X = a.sel[SEL_INMIX];
	// (the workgroups that read the receive buffer wait: the one that unpacks record i, and every particle fed by an arrival — the
	// others, whose source is a local particle, go ahead: fewer waves polling this rank's memory while the peers' stores come in,
	// fewer slots held by waiting workgroups)
	if (landing && status == MIG_OK && resampled && nrecv > 0 && (i < nrecv || pl.code[i] < 0)) {   // (uniform over the workgroup)
		if (tid < n && pl.counts[n + tid] > 0) {
			const long long t0 = wall_clock64();
			while ((long long) (__hip_atomic_load(landing + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
				if (wall_clock64() - t0 > landing_ticks) { atomicOr(a.flags, PHD_FLAG_ORDER_TIMEOUT); break; }
				__builtin_amdgcn_s_sleep(8);
			}
		}
		__syncthreads();
		__threadfence_system();   // acquire in every wave: the records behind the flags are what the loads below see
	}
	if (status != MIG_OK) {
		if (i == 0 && tid < SEL_STRIDE) sel_next[tid] = a.sel[tid];
		return;
	}
	if (i == 0 && tid == 0) {
		if (frozen)         { sel_next[SEL_IN] = I; sel_next[SEL_OUT] = O; sel_next[SEL_TMP] = T; sel_next[SEL_INMIX] = X; }
		else if (resampled) { sel_next[SEL_IN] = T; sel_next[SEL_OUT] = I; sel_next[SEL_TMP] = O; sel_next[SEL_INMIX] = O; }
		else                { sel_next[SEL_IN] = O; sel_next[SEL_OUT] = I; sel_next[SEL_TMP] = T; sel_next[SEL_INMIX] = O; }
		sel_next[SEL_RES]    = resampled ? T : O;
		sel_next[SEL_RESMIX] = O;
	}
	if (!resampled) {
		if (tid == 0) {
			slots[i] = i;
			if (!frozen) inslot[i] = i;
		}
		return;
	}
	const size_t rec = (size_t) 8 + (size_t) 10 * a.cap;
	if (i < nrecv) {
		const MixView dst = bank_view(a, SEL_OUT);
		const double* r = recvbuf + (size_t) i * rec;
		const int nc = min(max((int) r[0], 0), a.cap);   // (a record is what a peer packed; never trust a count with a store loop)
		const size_t db = (size_t) pl.fslot[i] * a.cap;
		copy_comps(dst.rec + db * MIX_REC, r + 8, nc, tid, 256);
	}
	const Bank bo = bank_of(a, SEL_OUT), bt = bank_of(a, SEL_TMP);
	const int code = pl.code[i];
	int slot;
	if (code >= 0) {
		slot = code;
		if (tid == 0) bt.count[i] = bo.count[code];
		if (tid < 7) bt.poses[(size_t) i * 7 + tid] = bo.poses[(size_t) code * 7 + tid];
	}
	else {
		const int j = -(code + 1);
		const double* r = recvbuf + (size_t) j * rec;
		slot = pl.fslot[j];
		if (tid == 0) bt.count[i] = min(max((int) r[0], 0), a.cap);
		if (tid < 7) bt.poses[(size_t) i * 7 + tid] = r[1 + tid];
	}
	if (tid == 0) {
		bt.weights[i] = weight;
		slots[i] = slot;
		if (!frozen) inslot[i] = slot;
	}
}
#endif   // PHD_ONLY_EP
