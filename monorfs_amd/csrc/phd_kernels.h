// phd_kernels.h — hand-written HIP kernels (gfx950, wave64) of the RB-PHD-SLAM inner loop.
//
// One workgroup per particle everywhere (particles are independent through predict / correct /
// prune / reweight: PHDNavigator.cs:326-339). Mixtures live in HBM as struct-of-arrays planes
// (w, mean x/y/z, covariance xx/xy/xz/yy/yz/zz), each [particle][slot], so a wavefront reads 64
// consecutive components of one particle as one 512-B line per plane.
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

// Roles of the three banks. A bank holds the mixture planes of all particles and their small arrays (count, pose,
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
	size_t plane;   // doubles per plane = Pcap * cap
	Bank bank[3];
	const int* sel; // [SEL_STRIDE] device-resident roles of the banks for this step (no host round trip to rotate them)
	const int* inslot;   // [P] slot of particle p's mixture in the INMIX bank
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
	struct MurtyNodes* murty;   // [P] workspace of the big-cluster solver (clusters of 6 .. 64 rows)
	char*   bigws;       // association slab: workspaces of the clusters beyond 64 rows, handed out by a bump counter
	unsigned long long  bigws_bytes;
	unsigned long long* bigws_used;   // reset at the end of every step (k_normalise_resample) / by the host before a stage or quasi launch
	double* jscratch;    // [P] landmark-indexed arrays of k_weight_alpha when the map estimate outgrows LDS
	// (component, measurement) pairs that may reach MinWeight, queued by k_sweep for k_emit_finish
	int*    cand;        // [P][candcap]
	int     candcap;
	int*    cand_count;  // [P][4] entries in each wave's segment of the queue (above candcap / 4: it overflowed)
	double* denom;       // [P][Mcap] kappa + weightsum[z]
	double* srec;        // [P][10][cutcap] k_prune_merge: the kept records in sorted order (mean, covariance, weight)
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
	int     stamp_kernel; // which kernel writes them (env PHD_STAMP_KERNEL): 2 prune, 3 assoc, 4 density, 1 correct, 5 the one-launch chain
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

// The mixtures of the bank playing `role` and the counts that go with them. For SEL_IN the planes are those of the
// INMIX bank: particle p's components start at in_base(a, p), its count is count[p].
__device__ __forceinline__ MixView bank_view(const StepBufs& a, int role)
{
	const Bank b = bank_of(a, role);
	const double* mix = (role == SEL_IN) ? bank_of(a, SEL_INMIX).mix : b.mix;
	MixView v;
	v.w = const_cast<double*>(mix);
#pragma unroll
	for (int t = 0; t < 3; t++) v.m[t] = const_cast<double*>(mix) + (size_t) (1 + t) * a.plane;
#pragma unroll
	for (int t = 0; t < 6; t++) v.P[t] = const_cast<double*>(mix) + (size_t) (4 + t) * a.plane;
	v.count = b.count;
	return v;
}

__device__ __forceinline__ size_t in_base(const StepBufs& a, int p) { return (size_t) a.inslot[p] * a.cap; }

#define TILE 256   // components staged per LDS tile

#include "phd_correct.h"

#include "phd_sweep.h"

#include "phd_prune.h"

#include "phd_alpha.h"

#include "phd_resample.h"

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
	PHD_STAMP_DECL;
	PHD_STAMP(0);
	sweep_body<ZB, HALF>(prm, a, smem);
	__syncthreads();   // (workgroup scope: the global writes of the step before are visible to this workgroup's loads)
	PHD_STAMP(1);
	emit_finish_body(prm, a, smem);
	__syncthreads();
	PHD_STAMP(2);
	prune_merge_body(prm, a, cutcap, smem);
	if (with_alpha) {
		__syncthreads();
		PHD_STAMP(3);
		alpha_assoc_body<ZB, false, false, 1>(prm, a, cutcap, smem);
		__syncthreads();
		PHD_STAMP(4);
		alpha_density_body(prm, a, smem);
		PHD_STAMP(5);
	}
	PHD_STAMP_FLUSH(5, 6);   // (diagnostic build, PHD_STAMP_KERNEL=5: the bodies' shares of the chain)
}

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
			double w, v[9];
			load_predicted(prm, a, vin, p, n, cidx, w, v, v + 3);
			double* r = a.emit_rec + (eb + e) * 9;
#pragma unroll
			for (int t = 0; t < 9; t++) r[t] = v[t];
		}
	}
}

// The sharded step's rotation when no rank resampled (the single-handle step does this inside k_normalise_resample,
// rotate_roles in phd_resample.h, where the rules are written down): roles (IN, OUT, TMP, INMIX) = (O, I, T, O), slots
// identity; frozen: nothing moves. One thread per particle.
__global__ __launch_bounds__(256) void k_gather_rotate(const StepBufs a, const int* src, const int* info, int first,
                                                       int* sel_next, int frozen, int* inslot)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	const int resampled = info[1];
	const int I = a.sel[SEL_IN], O = a.sel[SEL_OUT], T = a.sel[SEL_TMP], X = a.sel[SEL_INMIX];
	if (i == 0) {
		if (frozen)         { sel_next[SEL_IN] = I; sel_next[SEL_OUT] = O; sel_next[SEL_TMP] = T; sel_next[SEL_INMIX] = X; }
		else if (resampled) { sel_next[SEL_IN] = T; sel_next[SEL_OUT] = I; sel_next[SEL_TMP] = O; sel_next[SEL_INMIX] = O; }
		else                { sel_next[SEL_IN] = O; sel_next[SEL_OUT] = I; sel_next[SEL_TMP] = T; sel_next[SEL_INMIX] = O; }
		sel_next[SEL_RES]    = resampled ? T : O;
		sel_next[SEL_RESMIX] = O;
	}
	if (i >= a.P) return;
	const int s = resampled ? src[first + i] - first : i;
	if (resampled) {
		const Bank bo = bank_of(a, SEL_OUT), bt = bank_of(a, SEL_TMP);
		bt.count[i]   = bo.count[s];
		bt.weights[i] = bo.weights[i];
#pragma unroll
		for (int t = 0; t < 7; t++) bt.poses[(size_t) i * 7 + t] = bo.poses[(size_t) s * 7 + t];
	}
	if (!frozen) inslot[i] = s;
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
	double* w = bi.mix;
	for (int c = tid; c < n; c += 256) {
		w[db + c] = from.w[fb + c];
#pragma unroll
		for (int t = 0; t < 9; t++) w[(size_t) (1 + t) * a.plane + db + c] = from.w[(size_t) (1 + t) * a.plane + fb + c];
	}
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
	for (int c = tid; c < n; c += 256) {
		dst.w[db + c] = from.w[fb + c];
#pragma unroll
		for (int t = 0; t < 3; t++) dst.m[t][db + c] = from.m[t][fb + c];
#pragma unroll
		for (int t = 0; t < 6; t++) dst.P[t][db + c] = from.P[t][fb + c];
	}
	const Bank bi = bank_of(a, SEL_IN);
	const Bank bo = bank_of(a, SEL_OUT);
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
	if (i < a.P) bank_of(a, SEL_OUT).weights[i] = gw[first + i];
}

__global__ __launch_bounds__(256) void k_pack_particles(const StepBufs a, const int* sendlist, double* sendbuf)
{
	const int r = blockIdx.x, tid = threadIdx.x;
	const int s = sendlist[r];
	const MixView from = bank_view(a, SEL_OUT);
	const Bank bo = bank_of(a, SEL_OUT);
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

// End of a sharded step that resampled. As in the single-handle step no local mixture is copied: particle i whose
// source is a local particle reads that particle's slot of the OUT bank from now on; a particle that arrives from
// another rank (record j of the receive buffer) is unpacked into a slot of the OUT bank that no local particle uses
// as a source (fslot[j], chosen by the host plan: there are always enough) and read from there. Block b unpacks record
// b (if there is one) and sets up particle b: small arrays into TMP, slot into inslot / slots.
//   dstsrc[i] >= 0: local source slot; < 0: record -(dstsrc[i] + 1)
__global__ __launch_bounds__(256) void k_unpack_gather(const StepBufs a, const int* dstsrc, const double* recvbuf, int nrecv,
                                                       const int* fslot, double weight, int* sel_next, int frozen, int* inslot,
                                                       int* slots)
{
	const int i = blockIdx.x, tid = threadIdx.x;
	const int I = a.sel[SEL_IN], O = a.sel[SEL_OUT], T = a.sel[SEL_TMP];
	if (i == 0 && tid == 0) {
		if (frozen) { sel_next[SEL_IN] = I; sel_next[SEL_OUT] = O; sel_next[SEL_TMP] = T; sel_next[SEL_INMIX] = a.sel[SEL_INMIX]; }
		else        { sel_next[SEL_IN] = T; sel_next[SEL_OUT] = I; sel_next[SEL_TMP] = O; sel_next[SEL_INMIX] = O; }
		sel_next[SEL_RES] = T;
		sel_next[SEL_RESMIX] = O;
	}
	const size_t rec = (size_t) 8 + (size_t) 10 * a.cap;
	if (i < nrecv) {
		const MixView dst = bank_view(a, SEL_OUT);
		const double* r = recvbuf + (size_t) i * rec;
		const int n = (int) r[0];
		const size_t db = (size_t) fslot[i] * a.cap;
		for (int c = tid; c < n; c += 256) {
			dst.w[db + c] = r[8 + c];
#pragma unroll
			for (int t = 0; t < 3; t++) dst.m[t][db + c] = r[8 + (size_t) (1 + t) * a.cap + c];
#pragma unroll
			for (int t = 0; t < 6; t++) dst.P[t][db + c] = r[8 + (size_t) (4 + t) * a.cap + c];
		}
	}
	if (i >= a.P) return;
	const Bank bo = bank_of(a, SEL_OUT), bt = bank_of(a, SEL_TMP);
	const int code = dstsrc[i];
	int slot;
	if (code >= 0) {
		slot = code;
		if (tid == 0) bt.count[i] = bo.count[code];
		if (tid < 7) bt.poses[(size_t) i * 7 + tid] = bo.poses[(size_t) code * 7 + tid];
	}
	else {
		const int j = -(code + 1);
		const double* r = recvbuf + (size_t) j * rec;
		slot = fslot[j];
		if (tid == 0) bt.count[i] = (int) r[0];
		if (tid < 7) bt.poses[(size_t) i * 7 + tid] = r[1 + tid];
	}
	if (tid == 0) {
		bt.weights[i] = weight;
		slots[i] = slot;
		if (!frozen) inslot[i] = slot;
	}
}
