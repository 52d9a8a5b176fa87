// phdhip.hip — host side of libphdhip.so: the C-ABI of include/phdhip.h over the kernels of
// phd_kernels.h. Built for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC
//
// There is no CPU implementation behind this ABI: without a HIP device phd_create fails with
// PHD_ERR_NO_DEVICE and every other entry point needs a handle.
#include "../../include/phdhip.h"
#include "phd_kernels.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define DSPLIT_ROWS 1024     // particles the helpers' words are allocated for (k_particle_chain with helper workgroups)
#define PHD_MAX_DEVICES 64   // device ordinals a process may hand to phd_create / phd_create_multi

namespace {

thread_local std::string g_create_error;

struct Timer {          // one record per kernel launch since the last phd_timing_reset
	const char* name;
	hipEvent_t  t0, t1;
	int         t0_from;   // >= 0: the launch starts where record t0_from ended (back-to-back on one stream: one event, not two)
};

}  // namespace

struct MultiState;   // phd_create_multi: the shards of a multi-device handle (phd_multi.inc)

struct phd_navigator {
	MultiState* multi = nullptr;       // non-NULL: this handle only dispatches to its shards
	phd_params  prm;
	DevParams   dp;
	int         device = 0;
	hipStream_t stream = nullptr;      // the stream every kernel of this handle is launched on
	hipStream_t own_stream = nullptr;  // created by phd_create; `stream` unless the host lent its own
	static const int MAXSPLIT = 4;
	int         nsplit = 0;            // sub-ranges a step's per-particle kernels are split into (0: chosen from the particle count)
	int         dsplit_max = 256;      // ... and up to this many, with a helper workgroup per particle for the densities of WeightAlpha (two workgroups per
	                                   // particle: 2 P workgroups fit the chip's 512 slots of that kernel at once; env PHD_DSPLIT_MAX, 0: never)
	int         dsplit_late = 0;       // env PHD_DSPLIT_LATE (tests and measurements; StepBufs::dsplit = 1 + this): 1 the helpers report 0.5 ms late, 2 they leave at once,
	                                   // 3 they report and wait but are never picked — in all three every main workgroup keeps its density sums
	unsigned int dseq = 0;             // launches of the chain with helpers so far (StepBufs::dstamp)
	unsigned int* d_dsync = nullptr;   // the helpers' words (StepBufs); DSPLIT_ROWS particles
	int         chain_max = 512;       // up to this many particles a step's per-particle kernels run as one launch (k_particle_chain; env PHD_CHAIN_MAX)
	int         fold_nr = 0;           // 1 (env PHD_FOLD_NR): the chain ends the step itself — k_normalise_resample's body in its last workgroup; measured slower than the launch
	bool        chain_ok[3] = {false, false, false};   // ... where the bodies' LDS arrays fit one workgroup (per measurement-block count 1, 2, 4)
	hipStream_t aux[MAXSPLIT - 1] = {nullptr, nullptr, nullptr};   // streams of the sub-ranges after the first
	// Option (environment PHD_DEFER_BIG=1; off by default): the particles with an association cluster of more than
	// ALPHA_DEFER_ROWS rows are listed by k_alpha_assoc_main and their ordered replay runs inside the launch of the densities
	// (the first workgroups of k_alpha_density_big), k_normalise_resample / k_push_weights finish alpha. Built when the replay
	// was what config S's association kernel waited for (5 % of the particles, 15 times the median workgroup's lifetime); since
	// the clusters of up to 5 rows are no longer replayed one by one (phd_alpha.h, apply_small) the replay is short and the
	// option costs more than it hides (config S 3.80 against 3.62 ms: the fused launch needs the association's LDS, three
	// workgroups per CU for the densities instead of four). Kept for scenes whose clusters are large; bit-identical results.
	int         defer_big = 0;
	int         nbig = 256;            // workgroups of k_alpha_density_big that work the list off (environment PHD_NBIG)
	int         fuse_ep = -1;          // k_emit_finish and k_prune_merge as one launch (k_emit_prune): -1 = for frames of up to 64 measurements (measured on
	                                   // two streams: config B 0.677 -> 0.667 ms survey, 0.634 -> 0.615 steady; config S, 128 measurements, 3.67 -> 3.85:
	                                   // its Kalman path is long and pays for the 128 registers); environment PHD_FUSE_EP = 0 / 1 forces
	int         fuse_sep = 0;          // builds with -DPHD_WITH_FUSE_SEP only, environment PHD_FUSE_SEP=1: k_sweep, k_emit_finish and k_prune_merge as one launch
	int         last_defer = 0;        // the last launch_map left alpha open (k_normalise_resample / k_push_weights / k_alpha_combine finish it)
	int*        d_biglist = nullptr;   // [MAXSPLIT][Pcap + 2]
	double*     d_ratio = nullptr;     // [Pcap]
	hipEvent_t  ev_fork = nullptr, ev_join[MAXSPLIT - 1] = {nullptr, nullptr, nullptr};
	// Two sub-ranges, steps posted back to back (phd_step_async after phd_step_async): the end of the step runs on the stream
	// whose chain finishes LAST and no fork precedes the next step ("the step boundary"; env PHD_PIPELINE=0: a fork
	// before and a join behind every step, as for every other caller)
	int         pipeline = 1;
	bool        pipe_ok = false;       // nothing was enqueued on `stream` since the last such step: the aux stream is ordered behind all of it
	int         lagger = 1;            // which of the two streams (0 `stream`, 1 aux[0]) finishes the coming step last
	hipEvent_t  ev_res = nullptr;      // k_normalise_resample is through (recorded on the stream that ran it)
	int         device_order = 0;      // 1 (env PHD_DEVICE_ORDER): between such steps no event at all — k_normalise_resample counts the tickets of both
	                                   // streams' k_alpha_density workgroups, the other stream's next k_sweep waits behind k_gate. Measured 0.2 % faster
	                                   // than the events; two kernels that poll are not worth that by default
	unsigned    step_seq = 0;          // number of the last step ended that way
	unsigned    ticket_total = 0;      // tickets all such steps so far have handed out (P per step): what k_normalise_resample waits for — the counter is never reset
	bool sel_host_valid = false;       // h_sel mirrors the device-side bank roles without a round trip
	int Pcap = 0, cap = 0, Mcap = 0, ecap = 0, Jcap = 0, cutcap = 0;
	int P = 0, M = 0;
	bool frozen = false;
	bool all_pairs = false;            // phd_set_all_pairs: the benchmark mode of SURVEY §8d

	Bank   bank[3];
	int*   d_sel = nullptr;      // [2][SEL_STRIDE]: roles for the current / next step (+ where the last result is)
	int*   d_mslot = nullptr;    // [Pcap] sharded step: slot of every particle's mixture in the OUT bank
	const int* d_res_slots = nullptr;   // slots of the last step's result in RESMIX (frozen mode getters)
	int*   d_inslot = nullptr;   // [Pcap] slot of every particle's mixture in the INMIX bank (identity unless the last step resampled)
	int    parity = 0;
	int    h_sel[SEL_STRIDE] = {0, 1, 2, 0, 0, 0, 0, 0};

	double* d_z = nullptr;
	double* d_emit_w = nullptr;  int* d_emit_idx = nullptr; double* d_emit_rec = nullptr; int* d_emit_count = nullptr;
	int*    d_born_count = nullptr; int* d_born_k = nullptr; double* d_born_mean = nullptr;
	double* d_alpha = nullptr; double* d_setll = nullptr;
	int*    d_flags = nullptr; int* d_src = nullptr; int* d_info = nullptr;
	MurtyNodes* d_murty = nullptr;
	char* d_bigws = nullptr; unsigned long long bigws_bytes = 0; unsigned long long* d_bigws_used = nullptr;   // association slab (clusters beyond 64 rows)
	double* d_jscratch = nullptr;
	int cmcap = 0;
	int* d_cand_count = nullptr; double* d_denom = nullptr;
	int* d_cand = nullptr; int candcap = 0;
	double* d_alm = nullptr; int* d_aJ = nullptr; double* d_account = nullptr;
	double* d_stamps = nullptr;
	double* d_srec = nullptr;
	double* d_outw = nullptr;     // [Pcap][cap] the pruned weights as a plane (StepBufs::outw)
	double* d_wcopy = nullptr; int* d_cover = nullptr;   // k_prune_merge -> k_alpha_density (see StepBufs)
	double* d_motion = nullptr;   // odometry[6] + noise[P][6] of phd_update_motion
	double* d_quasi = nullptr;    // phd_quasi_set_loglik: poses[Pcap][7], landmarks[Jcap][3], z[256][3], out[Pcap]
	int*    h_status = nullptr;   // pinned mirror of [d_sel (two parities) | d_info | d_flags], one block on the device
	double* h_quasi = nullptr;    // ... its pinned mirror on the host (+ one word for the flags): one stream wait per call, no pageable copies
	double* d_gw = nullptr; int gwcap = 0;           // gathered weights of all ranks (+ their status words behind them: flagslot)
	double* d_stage = nullptr;                       // device staging of phd_set_poses / phd_set_weights (stored into the IN bank by k_store_small)
	// pinned host staging of the per-frame inputs (poses, weights, odometry + noise, measurements): the caller's buffers are
	// copied here and are free when the call returns; the copy to the device is asynchronous. Two buffers, each guarded by
	// an event recorded behind the copy that reads it.
	double* h_stage[2] = {nullptr, nullptr}; hipEvent_t ev_stage[2] = {nullptr, nullptr}; bool stage_used[2] = {false, false};
	int stage_i = 0; size_t stagecap = 0;
	int nr_static_lds = 0;                           // static LDS of k_normalise_resample
	// the step over a grid of workgroups (k_nr_*, phd_resample.h) for weight vectors of nr_grid_min .. 65 536 entries (environment
	// PHD_NR_GRID_MIN; 0: never): its scratch, made on first use
	int nr_grid_min = NR_GRID_MIN; int nrcap = 0; double* d_nrd = nullptr; int* d_nri = nullptr;
	// sharded step (one rank of a multi-GPU particle set: a process of its own, or a shard of a phd_create_multi handle)
	double*  d_lw = nullptr;                         // [Pcap] local weights, exported for the host's all-gather (per-rank host)
	double** d_dst_tab = nullptr; int ndst = 1;      // device table: where k_push_weights stores the local weights (own d_lw | every shard's d_gw)
	int      push_first = 0, push_flagslot = -1;     // ... at which offset, and where the status word goes (-1: nowhere)
	double** d_recv_tab = nullptr;                   // device table: the receive buffer of every shard / rank (filled by phd_create_multi, phd_migration_set_peers / _ipc_open)
	bool peers_set = false;                          // ... it is filled: migrating particles can be pushed
	bool gw_shared = false;                          // other shards hold the address of d_gw (multi-device handle): it cannot grow
	const double* d_gflags = nullptr;                // the gathered status words (multi-device handle)
	double* d_send = nullptr; double* d_recv = nullptr; int* d_plan = nullptr; int sendrecs = 0, recvrecs = 0;
	MigPlan plan = {nullptr, nullptr, nullptr, nullptr, nullptr, 0};   // device-resident migration plan (k_plan_migration)
	// the plan over a grid of workgroups (k_plan_count / k_plan_lists) for global vectors of plan_grid_min .. 65 536 slots whose
	// ranks hold a multiple of 64 particles (environment PHD_PLAN_GRID_MIN; 0: never): its accumulators (two sets, alternating)
	int plan_grid_min = PLAN_GRID_MIN; int* d_plang = nullptr; int plan_par = 0;
	int* h_counts = nullptr; int plan_seq = 0;       // pinned + mapped: the plan's counts as the kernel writes them, and the word the host polls
	bool plan_waiting = false;                       // a plan kernel with host counts is in flight
	int world = 1, rank = 0;                         // of the last global step
	int nsend = 0, nrecv = 0, last_world_particles = 1;
	bool sharded_used = false;
	bool sharded_ready = false;                      // every buffer of ensure_sharded is there (set behind the last allocation)
	bool recv_finegrained = false;                   // d_recv is fine-grained device memory (coherent for the peers that store into it)
	int  landing_flags = 0;                          // phd_migration_set_landing: 1 = push posts step-stamped flags into the peers' receive buffers, unpack waits for them
	unsigned long long landing_seq = 0;              // number of the last device-path global step (the flags' stamp)
	int       landing_inline = 0;                    // 1 (PHD_LANDING_INLINE): the wait inside k_finish_sharded instead of k_wait_landing in front of it
	long long landing_ticks = 1000000000LL;          // bound of the wait for a flag, in ticks of the 100 MHz counter: 10 s (environment PHD_LANDING_TIMEOUT_MS)
	double* d_graw = nullptr; int grawcap = 0;       // per-rank host: the all-gather's landing buffer, [world][P + 1] (weights | status word)
	std::vector<void*> ipc_opened;                   // peers' receive buffers opened with hipIpcOpenMemHandle (closed in phd_destroy)
	bool plan_on_device = false;                     // the last global step left its plan on the device only (phd_step_global_device_async)
	// host mirrors handed out by the getters
	std::vector<double> h_weights, h_poses, h_mw, h_mm, h_mc, h_alpha, h_setll, h_tmp;
	std::vector<int32_t> h_src;
	int  h_info[2] = {0, 0};
	int  h_flags = 0;
	bool stage_valid = false;

	std::vector<Timer>       timers;    // event pool; [0, ntimers) are live records
	size_t                   ntimers = 0;
	bool                     timing = true;
	int                      timing_period = 1;   // launches are timed on every timing_period-th step
	long long                timing_step = 0;
	bool                     timing_now = true;
	std::vector<const char*> tnames;
	std::vector<double>      tms;
	std::vector<int>         tcounts;
	std::string err;

	int fail(int code, const std::string& what)
	{
		err = what;
		return code;
	}
};

namespace {

#define HC(call)                                                                                    \
	do {                                                                                            \
		hipError_t e_ = (call);                                                                     \
		if (e_ != hipSuccess) {                                                                     \
			return nav->fail(PHD_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));    \
		}                                                                                           \
	} while (0)

void inv3_host(const double* a, double* inv, double* det)
{
	double c00 = a[4] * a[8] - a[5] * a[7];
	double c01 = a[3] * a[8] - a[5] * a[6];
	double c02 = a[3] * a[7] - a[4] * a[6];
	double d   = a[0] * c00 - a[1] * c01 + a[2] * c02;
	double id  = 1.0 / d;
	inv[0] = c00 * id;
	inv[1] = (a[2] * a[7] - a[1] * a[8]) * id;
	inv[2] = (a[1] * a[5] - a[2] * a[4]) * id;
	inv[3] = (a[5] * a[6] - a[3] * a[8]) * id;
	inv[4] = (a[0] * a[8] - a[2] * a[6]) * id;
	inv[5] = (a[2] * a[3] - a[0] * a[5]) * id;
	inv[6] = c02 * id;
	inv[7] = (a[1] * a[6] - a[0] * a[7]) * id;
	inv[8] = (a[0] * a[4] - a[1] * a[3]) * id;
	*det = d;
}

DevParams make_dev_params(const phd_params& p)
{
	DevParams d;
	d.focal  = p.measurer[0];
	d.rmin   = (double) (float) p.measurer[1];   // AForge.Range is float32 (PRM3DMeasurer.cs:65,110)
	d.rmax   = (double) (float) p.measurer[2];
	d.left   = (double) (int) p.measurer[3];     // XNA Rectangle is int (PRM3DMeasurer.cs:111)
	d.top    = (double) (int) p.measurer[4];
	d.right  = (double) ((int) p.measurer[3] + (int) p.measurer[5]);
	d.bottom = (double) ((int) p.measurer[4] + (int) p.measurer[6]);
	for (int i = 0; i < 3; i++) d.ramp[i] = p.visibility_ramp[i];
	for (int i = 0; i < 9; i++) d.R[i] = p.R[i];
	d.linear2d  = p.model == PHD_MODEL_LINEAR2D ? 1 : 0;
	d.lin_range = p.measurer[0];
	if (d.linear2d) {   // R is 2 x 2 row-major in the first four entries: pad it with a unit third coordinate (DevParams)
		const double r2[4] = {p.R[0], p.R[1], p.R[2], p.R[3]};
		const double r3[9] = {r2[0], r2[1], 0, r2[2], r2[3], 0, 0, 0, 1};
		for (int i = 0; i < 9; i++) d.R[i] = r3[i];
		d.ramp[2] = 1.0;
	}
	double det;
	inv3_host(d.R, d.Rinv, &det);
	d.logRmult = std::log(std::pow(2 * 3.14159265358979323846, -1.0) / std::sqrt(std::fabs(det)));
	d.pd       = p.pd;
	d.kappa    = p.clutter_density;
	d.logkappa = std::log(p.clutter_density);
	const double* B = p.birth_covariance;
	d.birthP[0] = B[0]; d.birthP[1] = B[1]; d.birthP[2] = B[2]; d.birthP[3] = B[4]; d.birthP[4] = B[5]; d.birthP[5] = B[8];
	d.birthw     = p.birth_weight;
	d.minw       = p.min_weight;
	d.expl_thr   = p.exploration_threshold;
	// Map.Near / Map.Evaluate(x, radius): Accord's KDTree compares its distance with the radius; with the
	// squared-Euclidean metric that is |x-m|^2 <= radius, with the Euclidean one |x-m| <= radius
	const double rc = p.density_distance_threshold, re = 3 * p.density_distance_threshold;
	d.g2_correct = (p.gate_metric == PHD_GATE_DISABLED) ? INFINITY : (p.gate_metric == PHD_GATE_SQUARED_EUCLIDEAN ? rc : rc * rc);
	d.g2_explore = (p.gate_metric == PHD_GATE_DISABLED) ? INFINITY : (p.gate_metric == PHD_GATE_SQUARED_EUCLIDEAN ? re : re * re);
	d.merge_thr2 = p.merge_threshold * p.merge_threshold;
	d.g2_assoc = 25.0;
	while (std::sqrt(std::nextafter(d.g2_assoc, 0.0)) >= 5.0) d.g2_assoc = std::nextafter(d.g2_assoc, 0.0);
	d.g2_quasi = 144.0;
	while (std::sqrt(std::nextafter(d.g2_quasi, 0.0)) >= 12.0) d.g2_quasi = std::nextafter(d.g2_quasi, 0.0);
	d.min_eff    = p.min_effective_particle;
	double floor = p.min_weight * p.clutter_density;
	d.emit_log_floor = (floor > 0) ? std::log(floor) : -INFINITY;
	d.gate_metric = p.gate_metric;
	d.maxq        = p.max_quantity;
	return d;
}

StepBufs make_bufs(phd_navigator* nav)
{
	StepBufs b;
	b.P = nav->P; b.p0 = 0; b.qposes = nullptr; b.qlm = nullptr; b.qJ = 0; b.cap = nav->cap; b.M = nav->M; b.Mcap = nav->Mcap; b.ecap = nav->ecap; b.Jcap = nav->Jcap;
	for (int i = 0; i < 3; i++) b.bank[i] = nav->bank[i];
	b.sel = nav->d_sel + nav->parity * SEL_STRIDE;
	b.inslot = nav->d_inslot;
	b.z = nav->d_z;
	b.emit_w = nav->d_emit_w; b.emit_idx = nav->d_emit_idx; b.emit_rec = nav->d_emit_rec; b.emit_count = nav->d_emit_count;
	b.born_count = nav->d_born_count; b.born_k = nav->d_born_k; b.born_mean = nav->d_born_mean;
	b.alpha = nav->d_alpha; b.setll = nav->d_setll; b.flags = nav->d_flags; b.murty = nav->d_murty; b.jscratch = nav->d_jscratch;
	b.bigws = nav->d_bigws; b.bigws_bytes = nav->bigws_bytes; b.bigws_used = nav->d_bigws_used;
	b.fold_nr = 0; b.ticket = (unsigned int*) (nav->d_bigws_used + 1); b.tickets = 0; b.wait_tickets = 0; b.ticket_target = 0; b.done_value = 0;
	b.nr_u = 0; b.nr_force = 0; b.nr_skip = 0; b.nr_frozen = 0; b.nr_src = nullptr; b.nr_info = nullptr; b.nr_sel_next = nullptr; b.nr_inslot = nullptr;
	b.cand_count = nav->d_cand_count; b.denom = nav->d_denom;
	b.cand = nav->d_cand; b.candcap = nav->candcap;
	b.dsplit = 0; b.dstamp = 0; b.dsync = nav->d_dsync;
	b.alm = nav->d_alm; b.aJ = nav->d_aJ; b.account = nav->d_account; b.srec = nav->d_srec; b.outw = nav->d_outw; b.wcopy = nav->d_wcopy; b.cover = nav->d_cover; b.stamps = nav->d_stamps; b.biglist = nav->d_biglist; b.bigstride = nav->Pcap + 2; b.ratio = nav->d_ratio; b.defer = 0; b.all_pairs = nav->all_pairs ? 1 : 0; b.stamp_kernel = getenv("PHD_STAMP_KERNEL") ? atoi(getenv("PHD_STAMP_KERNEL")) : 2;
	return b;
}

// Every entry point that may enqueue work on the handle's stream comes through here (phd_step_async alone does not): whatever
// it enqueues, the aux stream is not ordered behind it until the next step forks again.
static inline void enter(phd_navigator* nav)
{
	hipSetDevice(nav->device);
	nav->pipe_ok = false;
}

int zb_of(int M) { return M <= 64 ? 1 : (M <= 128 ? 2 : 4); }

// Non-finite numbers do not cross the ABI (include/phdhip.h, "Non-finite input"): the reference lets a NaN term poison a
// whole weight sum (PHDNavigator.cs:886-890) where the device's pair loops count a NaN exponent as 0 (exp_pair); with
// finite input the two never meet.
bool all_finite(const double* v, size_t n)
{
	if (!v) return true;
	unsigned long long bad = 0;
	for (size_t i = 0; i < n; i++) {
		unsigned long long b;
		std::memcpy(&b, v + i, 8);
		bad |= (unsigned long long) ((b & 0x7ff0000000000000ull) == 0x7ff0000000000000ull);
	}
	return bad == 0;
}
#define FINITE_OR_FAIL(nav, ptr, n, who) do { if (!all_finite((ptr), (size_t) (n))) return (nav)->fail(PHD_ERR_BAD_ARGUMENT, who ": non-finite input (NaN or infinity) is not accepted"); } while (0)

// A pinned staging buffer nobody reads any more (waits for the copy that last read it: normally long done).
double* stage_acquire(phd_navigator* nav)
{
	nav->stage_i ^= 1;
	const int i = nav->stage_i;
	if (nav->stage_used[i]) hipEventSynchronize(nav->ev_stage[i]);
	return nav->h_stage[i];
}

// ... the asynchronous copy out of it has been enqueued on the handle's stream
void stage_release(phd_navigator* nav)
{
	hipEventRecord(nav->ev_stage[nav->stage_i], nav->stream);
	nav->stage_used[nav->stage_i] = true;
}


// HIP events around every kernel launch, on the stream the kernel is launched on. `chained`: the launch follows the
// previous timed launch on the same stream with nothing in between, so that launch's end event is this one's start.
void timer_begin(phd_navigator* nav, const char* name, hipStream_t st = nullptr, bool chained = false)
{
	if (!st) st = nav->stream;
	if (!nav->timing || !nav->timing_now) return;
	if (nav->ntimers == nav->timers.size()) {
		if (nav->timers.size() >= 65536) { nav->timing = false; return; }
		Timer t;
		t.name = name;
		t.t0_from = -1;
		// (timing only, read after a stream synchronisation: no system-scope fence when they are recorded — the fence of a default event
		// writes the caches back in front of every timed launch and is part of what the step then costs)
		if (hipEventCreateWithFlags(&t.t0, hipEventDisableSystemFence) != hipSuccess || hipEventCreateWithFlags(&t.t1, hipEventDisableSystemFence) != hipSuccess) {
			(void) hipGetLastError();
			if (hipEventCreate(&t.t0) != hipSuccess || hipEventCreate(&t.t1) != hipSuccess) { nav->timing = false; return; }
		}
		nav->timers.push_back(t);
	}
	Timer& t = nav->timers[nav->ntimers];
	t.name = name;
	t.t0_from = (chained && nav->ntimers > 0) ? (int) nav->ntimers - 1 : -1;
	if (t.t0_from < 0) hipEventRecord(t.t0, st);
}

void timer_end(phd_navigator* nav, const char* name, hipStream_t st = nullptr)
{
	if (!st) st = nav->stream;
	if (!nav->timing || !nav->timing_now || nav->ntimers >= nav->timers.size()) return;
	hipEventRecord(nav->timers[nav->ntimers].t1, st);
	nav->ntimers++;
}

const char* T_SW = "k_sweep";
const char* T_EF = "k_emit_finish";
const char* T_PM = "k_prune_merge";
const char* T_EP = "k_emit_prune";
#ifdef PHD_WITH_FUSE_SEP
const char* T_SEP = "k_sweep_emit_prune";
#endif
const char* T_WA = "k_alpha_assoc";
const char* T_WD = "k_alpha_density";
const char* T_NR = "k_normalise_resample";
const char* T_GR = "k_finish_sharded";
const char* T_PL = "k_plan_migration";
const char* T_PK = "k_pack_particles";
const char* T_PW = "k_push_weights";
const char* T_CH = "k_particle_chain";

// The per-particle kernels of a step. With nsplit > 1 the particle range is cut into sub-ranges whose kernel
// chains run on concurrent streams (forked from and joined back into the handle's stream), so that the
// latency-bound kernels of one sub-range overlap the arithmetic-bound kernels of another.
// `pipe` >= 0 (phd_step_async, two sub-ranges): no fork and no join here — the caller has ordered the streams and ends the step on
// the stream that finishes last, `pipe` (0 `stream`, 1 aux[0]); the other one's chain is enqueued first.
template <int ZB>
int launch_map_kernels(phd_navigator* nav, const StepBufs& b0, bool with_alpha, int pipe = -1)
{
	const int P = nav->P;
	// two half-ranges measured best at 2048 particles (1.10 -> 1.02 ms); below ~4 workgroups per CU and sub-range it does not pay
	const int want = nav->nsplit > 0 ? nav->nsplit : (P >= 1024 ? 2 : 1);
	const int S = std::max(1, std::min(std::min(want, (int) phd_navigator::MAXSPLIT), P));
	const size_t lp = (size_t) prune_lds(nav->cutcap).bytes;
	const AlphaLds lay = alpha_lds(ZB * 64, nav->cutcap);   // (the dynamic LDS limits of the kernels were raised once, in phd_create)
	const int zi = ZB == 1 ? 0 : (ZB == 2 ? 1 : 2);
	if (nav->chain_ok[zi] && P <= nav->chain_max) {
		// a small particle set (up to two workgroups per CU): the whole per-particle chain as one launch
		timer_begin(nav, T_CH);
		// (a helper workgroup per particle for the densities when all 2 P workgroups can be on the chip together; the results are the
		// same bits whether a helper comes in time or not: k_particle_chain)
		StepBufs bc = b0;
		const bool helpers = with_alpha && nav->d_dsync && P <= nav->dsplit_max && P <= DSPLIT_ROWS;
		if (helpers) {
			if (++nav->dseq >= 0x07ffffffu) {   // (the numbers start again: no word of an earlier round may be taken for one of this round)
				nav->dseq = 1;
				HC(hipMemsetAsync(nav->d_dsync, 0, (size_t) std::min(nav->Pcap, (int) DSPLIT_ROWS) * 3 * 4, nav->stream));
			}
			bc.dsplit = nav->dsplit_late > 0 ? 1 + nav->dsplit_late : 1; bc.dstamp = nav->dseq;
		}
		const int G = helpers ? 2 * P : P;
		// (up to 32 measurements: the sweep with two components per visit, phd_sweep.h HALF)
		if (ZB == 1 && nav->M <= 32) hipLaunchKernelGGL((k_particle_chain<1, true>), dim3(G), dim3(256), (size_t) chain_lds_bytes<1>(nav->cutcap), nav->stream, nav->dp, bc, nav->cutcap, with_alpha ? 1 : 0);
		else hipLaunchKernelGGL(k_particle_chain<ZB>, dim3(G), dim3(256), (size_t) chain_lds_bytes<ZB>(nav->cutcap), nav->stream, nav->dp, bc, nav->cutcap, with_alpha ? 1 : 0);
		timer_end(nav, T_CH);
		HC(hipGetLastError());
		nav->last_defer = 0;
		return PHD_OK;
	}
	if (S > 1 && pipe < 0) {
		HC(hipEventRecord(nav->ev_fork, nav->stream));
		for (int s = 1; s < S; s++) HC(hipStreamWaitEvent(nav->aux[s - 1], nav->ev_fork, 0));
	}
	const bool defer = with_alpha && nav->defer_big != 0;
	nav->last_defer = defer ? 1 : 0;
	const size_t ldb = std::max((size_t) lay.bytes, (size_t) DENS_LDS_DOUBLES * 8);   // k_alpha_density_big: the larger of the two bodies' pools
	for (int si = 0; si < S; si++) {
		const int s = (pipe >= 0 && S == 2) ? (si == 0 ? 1 - pipe : pipe) : si;
		StepBufs b = b0;
		b.defer = defer ? 1 : 0;
		b.biglist = nav->d_biglist + (size_t) s * b0.bigstride;
		b.p0 = (int) ((long long) P * s / S);
		const int n = (int) ((long long) P * (s + 1) / S) - b.p0;
		if (n <= 0) continue;
		hipStream_t st = s == 0 ? nav->stream : nav->aux[s - 1];
#ifdef PHD_WITH_FUSE_SEP
		if (nav->fuse_sep) {
			const size_t ld3 = std::max(std::max(lp, (size_t) EMIT_LDS_DOUBLES * 8), (size_t) SweepLds<ZB>::doubles * 8);
			timer_begin(nav, T_SEP, st);
			if (ZB == 1 && nav->M <= 32) hipLaunchKernelGGL((k_sweep_emit_prune<1, true>), dim3(n), dim3(256), ld3, st, nav->dp, b, nav->cutcap);
			else hipLaunchKernelGGL(k_sweep_emit_prune<ZB>, dim3(n), dim3(256), ld3, st, nav->dp, b, nav->cutcap);
			timer_end(nav, T_SEP, st);
		}
		else
#endif
		{
		timer_begin(nav, T_SW, st);
		if (ZB == 1 && nav->M <= 32) hipLaunchKernelGGL((k_sweep<1, true>), dim3(n), dim3(256), 0, st, nav->dp, b);
		else hipLaunchKernelGGL(k_sweep<ZB>, dim3(n), dim3(256), 0, st, nav->dp, b);
		timer_end(nav, T_SW, st);
		if (nav->fuse_ep < 0 ? ZB == 1 : nav->fuse_ep != 0) {
			timer_begin(nav, T_EP, st, true);
			hipLaunchKernelGGL(k_emit_prune, dim3(n), dim3(256), std::max(lp, (size_t) EMIT_LDS_DOUBLES * 8), st, nav->dp, b, nav->cutcap);
			timer_end(nav, T_EP, st);
		}
		else {
			timer_begin(nav, T_EF, st, true);
			hipLaunchKernelGGL(k_emit_finish, dim3(n), dim3(256), 0, st, nav->dp, b);
			timer_end(nav, T_EF, st);
			timer_begin(nav, T_PM, st, true);
			hipLaunchKernelGGL(k_prune_merge, dim3(n), dim3(256), lp, st, nav->dp, b, nav->cutcap);
			timer_end(nav, T_PM, st);
		}
		}
		if (with_alpha && defer) {
			timer_begin(nav, T_WA, st, true);
			hipLaunchKernelGGL(k_alpha_assoc_main<ZB>, dim3(n), dim3(256), lay.bytes, st, nav->dp, b, nav->cutcap);
			timer_end(nav, T_WA, st);
			// the particles it listed are worked off by the first workgroups of the densities' launch
			const int nbig = std::max(1, std::min(nav->nbig, n));
			timer_begin(nav, T_WD, st, true);
			hipLaunchKernelGGL(k_alpha_density_big<ZB>, dim3(n + nbig), dim3(256), ldb, st, nav->dp, b, nav->cutcap, nbig);
			timer_end(nav, T_WD, st);
		}
		else if (with_alpha) {
			timer_begin(nav, T_WA, st, true);
			hipLaunchKernelGGL(k_alpha_assoc<ZB>, dim3(n), dim3(256), lay.bytes, st, nav->dp, b, nav->cutcap);
			timer_end(nav, T_WA, st);
			timer_begin(nav, T_WD, st, true);
			hipLaunchKernelGGL(k_alpha_density, dim3(n), dim3(256), 0, st, nav->dp, b);
			timer_end(nav, T_WD, st);
		}
	}
	for (int s = 1; s < S && pipe < 0; s++) {
		HC(hipEventRecord(nav->ev_join[s - 1], nav->aux[s - 1]));
		HC(hipStreamWaitEvent(nav->stream, nav->ev_join[s - 1], 0));
	}
	HC(hipGetLastError());
	return PHD_OK;
}

// PHDNavigator.QuasiSetLogLikelihood (PHDNavigator.cs:526-531) for a batch of candidate poses against one landmark
// set and one measurement set: one workgroup per pose through the association kernel's own code.
template <int ZB>
int launch_quasi(phd_navigator* nav, const StepBufs& b, int nposes, bool gradient)
{
	const AlphaLds lay = alpha_lds(ZB * 64, nav->cutcap);
	if (gradient) {
		hipLaunchKernelGGL(k_quasi_setll_grad<ZB>, dim3(nposes), dim3(256), lay.bytes, nav->stream, nav->dp, b, nav->cutcap);
	}
	else {
		hipLaunchKernelGGL(k_quasi_setll<ZB>, dim3(nposes), dim3(256), lay.bytes, nav->stream, nav->dp, b, nav->cutcap);
	}
	HC(hipGetLastError());
	return PHD_OK;
}

int launch_map(phd_navigator* nav, const StepBufs& b, bool with_alpha, int pipe = -1)
{
	switch (zb_of(nav->M)) {
	case 1:  return launch_map_kernels<1>(nav, b, with_alpha, pipe);
	case 2:  return launch_map_kernels<2>(nav, b, with_alpha, pipe);
	default: return launch_map_kernels<4>(nav, b, with_alpha, pipe);
	}
}

// graw / Pl / world (per-rank host): the weights still lie as the all-gather delivered them, [rank][Pl + 1]; the first launch un-gathers them
// pgp != NULL (sharded step whose plan runs over the grid too): the resampling's last launch also counts for the plan (*counted <- true)
int launch_normalise(phd_navigator* nav, const StepBufs& b, double* gw, int P, double u, int force, int skipnorm, int* src, int* info,
                     int* sel_next = nullptr, hipStream_t st = nullptr, const double* graw = nullptr, int Pl = 0, int world = 0,
                     const PlanGrid* pgp = nullptr, int rank = 0, const double* gflags = nullptr, bool* counted = nullptr)
{
	if (!st) st = nav->stream;
	if (nav->nr_grid_min > 0 && P >= nav->nr_grid_min && P <= 65536 && b.wait_tickets == 0 && b.done_value == 0) {
		// one particle per thread over a grid, four launches (phd_resample.h, "over a GRID"): 16 384 weights in ~15 us instead of 48
		if (P > nav->nrcap) {
			HC(hipDeviceSynchronize());   // (first use, or a longer vector than ever before: rare)
			hipFree(nav->d_nrd); hipFree(nav->d_nri);
			nav->d_nrd = nullptr; nav->d_nri = nullptr; nav->nrcap = 0;
			const int Gc = (P + 255) / 256;
			HC(hipMalloc((void**) &nav->d_nrd, ((size_t) Gc * (NR_STAT + NR_SLOT) + 2 * (size_t) P) * 8));
			HC(hipMalloc((void**) &nav->d_nri, ((size_t) P + 4) * 4));
			nav->nrcap = P;
		}
		NrGrid nr;
		nr.G = (P + 255) / 256;
		nr.part = nav->d_nrd;
		nr.slotres = nr.part + (size_t) nr.G * NR_STAT;
		nr.pre = nr.slotres + (size_t) nr.G * NR_SLOT;
		nr.hi = nav->d_nri;
		nr.state = nav->d_nri + P;
		const int fr = nav->frozen ? 1 : 0;
		hipLaunchKernelGGL(k_nr_sum, dim3(nr.G), dim3(256), 0, st, b, gw, P, skipnorm, sel_next, nr, graw, Pl, world);
		hipLaunchKernelGGL(k_nr_stats, dim3(nr.G), dim3(256), 0, st, b, gw, P, skipnorm, sel_next, nr);
		hipLaunchKernelGGL(k_nr_slots, dim3(nr.G), dim3(256), 0, st, b, gw, P, nav->dp.min_eff, u, force, src, info, sel_next, fr, nav->d_inslot, nr);
		PlanGrid pgn;
		std::memset(&pgn, 0, sizeof pgn);
		if (pgp) { pgn = *pgp; if (counted) *counted = true; }
		hipLaunchKernelGGL(k_nr_sources, dim3(nr.G), dim3(256), 0, st, b, gw, P, u, src, info, sel_next, fr, nav->d_inslot, nr, pgn, Pl > 0 ? Pl : P, world > 0 ? world : 1,
		                   rank, gflags);
		HC(hipGetLastError());
		return PHD_OK;
	}
	if (graw) hipLaunchKernelGGL(k_ungather, dim3((P + world + 255) / 256), dim3(256), 0, st, graw, gw, Pl, world);
	// one workgroup; 256 / 512 threads for shorter weight vectors (fewer waves to meet at every barrier), 1024 beyond 4096
	static const int nr_env = getenv("PHD_NR_THREADS") ? atoi(getenv("PHD_NR_THREADS")) : 0;
	const int nthreads = (nr_env == 256 || nr_env == 512 || nr_env == 1024) ? nr_env : (P <= 512 ? 256 : (P <= 4096 ? 512 : 1024));
	// the weight vector is staged in LDS (chunk-transposed: a whole chunk per thread, used or not) when it fits beside the
	// kernel's static arrays (160 KB per CU); above that the kernel works on the vector in global memory
	size_t lds = (size_t) ((P + nthreads - 1) / nthreads) * (nthreads + 1) * 8;
	int use_lds = lds + (size_t) nav->nr_static_lds + 256 <= 160 * 1024;
	if (!use_lds) lds = 0;
	hipLaunchKernelGGL(k_normalise_resample, dim3(1), dim3(nthreads), lds, st, b, gw, P, nav->dp.min_eff, u, force, skipnorm,
	                   use_lds, src, info, sel_next, nav->frozen ? 1 : 0, nav->d_inslot);
	HC(hipGetLastError());
	return PHD_OK;
}

int check_flags(phd_navigator* nav)
{
	int f = nav->h_flags;
	if (f & PHD_FLAG_EMIT_OVERFLOW) {
		return nav->fail(PHD_ERR_CAPACITY, "corrected mixture outgrew emit_capacity (" + std::to_string(nav->ecap) + " components per particle)");
	}
	if (f & PHD_FLAG_J_OVERFLOW) {
		return nav->fail(PHD_ERR_CAPACITY, "map estimate larger than the landmark scratch (" + std::to_string(nav->Jcap) + ")");
	}
	if (f & PHD_FLAG_ORDER_TIMEOUT) {
		return nav->fail(PHD_ERR_GENERIC, "a kernel gave up a bounded wait on the device: for work of the handle's other stream submitted before it (PHD_DEVICE_ORDER=1: 0.2 s; "
		                 "the step was dropped, the state is the one before it), or for the landing flag of a peer's migrating particles (phd_migration_set_landing: "
		                 "PHD_LANDING_TIMEOUT_MS; a rank has died, the state of this handle is undefined)");
	}
	if (f & PHD_FLAG_BIG_CLUSTER) {
		return nav->fail(PHD_ERR_ASSOCIATION, "a data-association cluster has more than " + std::to_string(MURTY_NBIG) + " rows, or the clusters beyond " +
		                 std::to_string(MURTY_NMAX) + " rows used up the association workspace (" + std::to_string(nav->bigws_bytes >> 20) +
		                 " MiB, phd_set_association_workspace); the step was dropped, the state is the one before it");
	}
	return PHD_OK;
}

// copy one particle's mixture into h_mw / h_mm / h_mc (row-major mean[3n], cov[9n]): its count from the bank of the small
// arrays, its components from `mixbank` at the slot dslots[particle] (device array; NULL or the same bank: its own slot)
int fetch_map(phd_navigator* nav, int bankidx, int particle, int* ncomp, int mixbank = -1, const int* dslots = nullptr)
{
	int n = 0, slot = particle;
	HC(hipMemcpyAsync(&n, nav->bank[bankidx].count + particle, sizeof(int), hipMemcpyDeviceToHost, nav->stream));
	if (mixbank >= 0 && mixbank != bankidx && dslots) {
		HC(hipMemcpyAsync(&slot, dslots + particle, sizeof(int), hipMemcpyDeviceToHost, nav->stream));
	}
	HC(hipStreamSynchronize(nav->stream));
	if (slot < 0 || slot >= nav->Pcap) return nav->fail(PHD_ERR_GENERIC, "corrupt particle slot");
	const int srcbank = (mixbank >= 0) ? mixbank : bankidx;
	if (n < 0 || n > nav->cap) return nav->fail(PHD_ERR_GENERIC, "corrupt component count");
	nav->h_tmp.resize((size_t) MIX_REC * std::max(n, 1));
	if (n > 0) {   // the particle's records are one contiguous piece of its bank
		HC(hipMemcpyAsync(nav->h_tmp.data(), nav->bank[srcbank].mix + (size_t) slot * nav->cap * MIX_REC, (size_t) n * MIX_REC * 8,
		                  hipMemcpyDeviceToHost, nav->stream));
		HC(hipStreamSynchronize(nav->stream));
	}
	nav->h_mw.resize(std::max(n, 1));
	nav->h_mm.resize((size_t) 3 * std::max(n, 1));
	nav->h_mc.resize((size_t) 9 * std::max(n, 1));
	const double* t = nav->h_tmp.data();
	for (int c = 0; c < n; c++) {
		const double* r = t + (size_t) c * MIX_REC;
		nav->h_mw[c] = r[0];
		for (int k = 0; k < 3; k++) nav->h_mm[c * 3 + k] = r[1 + k];
		double xx = r[4], xy = r[5], xz = r[6], yy = r[7], yz = r[8], zz = r[9];
		double* C = &nav->h_mc[(size_t) c * 9];
		C[0] = xx; C[1] = xy; C[2] = xz; C[3] = xy; C[4] = yy; C[5] = yz; C[6] = xz; C[7] = yz; C[8] = zz;
	}
	*ncomp = n;
	return PHD_OK;
}

// host arrays -> the component records of one particle (covariance: upper triangle of the given 3x3)
void pack_records(const double* w, const double* mean3, const double* cov9, int n, double* recs)
{
	static const int tri[6] = {0, 1, 2, 4, 5, 8};
	for (int c = 0; c < n; c++) {
		double* r = recs + (size_t) c * MIX_REC;
		r[0] = w[c];
		for (int k = 0; k < 3; k++) r[1 + k] = mean3[c * 3 + k];
		for (int k = 0; k < 6; k++) r[4 + k] = cov9[c * 9 + tri[k]];
	}
}

int upload_particle(phd_navigator* nav, int bankidx, int particle, const double* w, const double* mean3,
                    const double* cov9, int n)
{
	if (n < 0 || n > nav->cap) return nav->fail(PHD_ERR_CAPACITY, "map larger than max_components");
	if (n > 0) {
		std::vector<double> recs((size_t) MIX_REC * n);
		pack_records(w, mean3, cov9, n, recs.data());
		HC(hipMemcpy(nav->bank[bankidx].mix + (size_t) particle * nav->cap * MIX_REC, recs.data(), (size_t) n * MIX_REC * 8, hipMemcpyHostToDevice));
	}
	HC(hipMemcpy(nav->bank[bankidx].count + particle, &n, sizeof(int), hipMemcpyHostToDevice));
	return PHD_OK;
}

int sync_state(phd_navigator* nav)
{
	HC(hipMemcpyAsync(nav->h_status, nav->d_sel, (2 * SEL_STRIDE + 3) * sizeof(int), hipMemcpyDeviceToHost, nav->stream));
	HC(hipStreamSynchronize(nav->stream));
	std::memcpy(nav->h_sel, nav->h_status + nav->parity * SEL_STRIDE, SEL_STRIDE * sizeof(int));
	std::memcpy(nav->h_info, nav->h_status + 2 * SEL_STRIDE, 2 * sizeof(int));
	nav->h_flags = nav->h_status[2 * SEL_STRIDE + 2];
	nav->sel_host_valid = true;
	return PHD_OK;
}

int cur_bank(const phd_navigator* nav) { return nav->h_sel[SEL_IN]; }

// Where the state a getter reports lives: the small arrays' bank, the mixtures' bank and the slot array. In frozen mode
// the roles do not advance and the result of the last step is described by RES / RESMIX and the resampling sources.
int res_small(const phd_navigator* nav) { return nav->frozen ? nav->h_sel[SEL_RES] : nav->h_sel[SEL_IN]; }
int res_mix(const phd_navigator* nav) { return nav->frozen ? nav->h_sel[SEL_RESMIX] : nav->h_sel[SEL_INMIX]; }
const int* res_slots(const phd_navigator* nav) { return nav->frozen ? (nav->d_res_slots ? nav->d_res_slots : nav->d_src) : nav->d_inslot; }

// The current state is about to be replaced as a whole: its mixtures will be addressed by particle number again.
int reset_indirection(phd_navigator* nav)
{
	std::vector<int> id(nav->Pcap);
	for (int i = 0; i < nav->Pcap; i++) id[i] = i;
	HC(hipMemcpy(nav->d_inslot, id.data(), (size_t) nav->Pcap * 4, hipMemcpyHostToDevice));
	nav->h_sel[SEL_INMIX] = nav->h_sel[SEL_IN];
	nav->h_sel[SEL_RES] = nav->h_sel[SEL_IN];
	nav->h_sel[SEL_RESMIX] = nav->h_sel[SEL_IN];
	HC(hipMemcpy(nav->d_sel + nav->parity * SEL_STRIDE, nav->h_sel, SEL_STRIDE * sizeof(int), hipMemcpyHostToDevice));
	return PHD_OK;
}

// Gather the mixtures of the current state into its own bank (k_materialise) if the last step left them behind an
// indirection. Needs the host mirror of the roles (sync_state).
int materialise(phd_navigator* nav)
{
	if (nav->h_sel[SEL_INMIX] == nav->h_sel[SEL_IN] || nav->P < 1) return PHD_OK;
	StepBufs b = make_bufs(nav);
	hipLaunchKernelGGL(k_materialise, dim3(nav->P), dim3(256), 0, nav->stream, b, nav->d_inslot);
	HC(hipGetLastError());
	HC(hipStreamSynchronize(nav->stream));
	nav->h_sel[SEL_INMIX] = nav->h_sel[SEL_IN];   // RES / RESMIX keep describing the last (frozen) step's result, which this did not touch
	HC(hipMemcpy(nav->d_sel + nav->parity * SEL_STRIDE, nav->h_sel, SEL_STRIDE * sizeof(int), hipMemcpyHostToDevice));
	return PHD_OK;
}

}  // namespace

// ---- multi-device handle (phd_create_multi, phd_multi.inc): every entry point that means something for it dispatches here
int multi_reset(phd_navigator* nav, int nparticles, const double* pose7, const double* w, const double* mean3, const double* cov9, int ncomp);
int multi_set_small(phd_navigator* nav, const double* poses7, const double* weights, int nparticles);
int multi_update_motion(phd_navigator* nav, const double* odometry6, const double* noise6, int nparticles, uint8_t perfect_still);
int multi_set_map(phd_navigator* nav, int particle, const double* w, const double* mean3, const double* cov9, int ncomp);
int multi_upload(phd_navigator* nav, int nparticles, int stride, const double* planes, const int32_t* counts, const double* poses7, const double* weights);
int multi_download(phd_navigator* nav, int stride, double* planes, int32_t* counts, double* poses7, double* weights);
int multi_set_measurements(phd_navigator* nav, const double* z3, int nmeasurements);
int multi_step(phd_navigator* nav, uint8_t onlymapping, double u_resample);
int multi_sync(phd_navigator* nav);
int multi_forward_int(phd_navigator* nav, int what, long long value);
const double* multi_weights(phd_navigator* nav, int* length);
const double* multi_poses(phd_navigator* nav, int* length);
int multi_best_particle(phd_navigator* nav);
int multi_map(phd_navigator* nav, int particle, int* ncomp, const double** w, const double** mean3, const double** cov9);
const int32_t* multi_resample_sources(phd_navigator* nav, int* length, uint8_t* resampled);
void multi_destroy(phd_navigator* nav);
phd_navigator* multi_shard0(phd_navigator* nav);
static void multi_timing_reset(phd_navigator* nav, uint8_t enabled);
#define MULTI_UNSUPPORTED(nav, what) if ((nav) && (nav)->multi) return (nav)->fail(PHD_ERR_BAD_ARGUMENT, what ": not available on a multi-device handle (use a single-device handle)")

// =================================================================================================
extern "C" {

int phd_api_version(void) { return PHD_API_VERSION; }

const char* phd_create_error(void) { return g_create_error.c_str(); }

void phd_default_params(phd_params* p, int max_particles, int max_components, int max_measurements)
{
	std::memset(p, 0, sizeof(*p));
	p->model = PHD_MODEL_PRM3D;
	p->zdim  = 3;
	const double meas[7] = {575.8156, (double) 0.1f, (double) 2.0f, -320, -240, 640, 480};   // PRM3DMeasurer.cs:70-73
	std::memcpy(p->measurer, meas, sizeof(meas));
	p->R[0] = 2.0; p->R[4] = 2.0; p->R[8] = 1e-3;                                             // Config.cs:251-253
	for (int i = 0; i < 3; i++) p->visibility_ramp[i] = 3 * std::sqrt(p->R[i * 4]);          // Config.cs:257-259
	p->pd = 0.9;                                                                              // Config.cs:63,90
	p->clutter_density = 3e-7;                                                                // Config.cs:256,262
	p->birth_covariance[0] = p->birth_covariance[4] = p->birth_covariance[8] = 1e-2;          // Config.cs:77-79
	p->birth_weight = 0.05; p->min_weight = 1e-3; p->min_effective_particle = 0.1;            // Config.cs:80-82
	p->max_quantity = 600; p->merge_threshold = 0.3; p->exploration_threshold = 1e-5;         // Config.cs:83-85
	p->density_distance_threshold = 0.5;                                                      // Config.cs:74
	p->gate_metric = PHD_GATE_SQUARED_EUCLIDEAN;
	p->max_particles = max_particles; p->max_components = max_components; p->max_measurements = max_measurements;
	p->emit_capacity = 0;
}

phd_navigator* phd_create(const phd_params* params, int device)
{
	g_create_error.clear();
	if (!params) { g_create_error = "params is NULL"; return nullptr; }
	if (!((params->model == PHD_MODEL_PRM3D && params->zdim == 3) || (params->model == PHD_MODEL_LINEAR2D && params->zdim == 2))) {
		g_create_error = "model / zdim must be PRM3D / 3 or LINEAR2D / 2";
		return nullptr;
	}
	if (params->max_particles < 1 || params->max_components < 1 || params->max_measurements < 0 ||
	    params->max_measurements > 256 || params->max_quantity < 1 || params->max_components < params->max_quantity) {
		g_create_error = "capacities out of range (need max_components >= max_quantity >= 1, max_measurements <= 256)";
		return nullptr;
	}
	{
		// k_prune_merge and k_alpha_assoc keep their per-particle working set in LDS (160 KB per CU)
		const int zb = zb_of(params->max_measurements);
		const size_t need = std::max((size_t) prune_lds(params->max_quantity).bytes, (size_t) alpha_lds(zb * 64, params->max_quantity).bytes);
		if (need > 160 * 1024 - 512) {
			g_create_error = "max_quantity too large: pruning one particle needs " + std::to_string(need) + " bytes of LDS (160 KB per CU; about 3400 components fit)";
			return nullptr;
		}
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		g_create_error = "no HIP device: libphdhip has no CPU path";
		return nullptr;
	}
	if (device < 0 || device >= ndev || device >= PHD_MAX_DEVICES || hipSetDevice(device) != hipSuccess) {
		g_create_error = "bad device ordinal";
		return nullptr;
	}
	phd_navigator* nav = new phd_navigator();
	nav->prm = *params;
	nav->dp  = make_dev_params(*params);
	nav->device = device;
	nav->Pcap = params->max_particles;
	nav->cap  = (params->max_components + 63) & ~63;
	nav->Mcap = std::max(64, (params->max_measurements + 63) & ~63);
	nav->cutcap = params->max_quantity;
	int ecap = params->emit_capacity > 0 ? params->emit_capacity : 4 * (nav->cap + nav->Mcap);
	nav->ecap = (std::max(ecap, nav->cutcap) + 63) & ~63;
	nav->Jcap = std::min(1024, (nav->cutcap + 63) & ~63);
	nav->P = 0;

	auto dalloc = [&](void** ptr, size_t bytes) { return hipMalloc(ptr, std::max<size_t>(bytes, 16)) == hipSuccess; };
	bool ok = hipStreamCreateWithFlags(&nav->own_stream, hipStreamNonBlocking) == hipSuccess;
	nav->stream = nav->own_stream;
	// The events that order the sub-range streams among themselves (same device, never inspected by the host): recorded without
	// the system-scope fence a default event performs (5 us per step boundary; the kernels' own release / acquire at their ends
	// and starts is what the next stream's kernels see). PHD_EVENT_FLAGS=0: default events.
	unsigned evflags = hipEventDisableTiming | hipEventDisableSystemFence;
	if (const char* e = getenv("PHD_EVENT_FLAGS")) {
		const int m = atoi(e);
		if (m == 0) evflags = hipEventDisableTiming;
		if (m == 1) evflags = hipEventDisableTiming | hipEventReleaseToDevice;
	}
	// (a runtime that refuses the fence-less flag gets plain untimed events: slower boundaries, the same order)
	auto ev_create = [&](hipEvent_t* ev, unsigned flags) {
		if (hipEventCreateWithFlags(ev, flags) == hipSuccess) return true;
		(void) hipGetLastError();
		return hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess;
	};
	for (int i = 0; i < phd_navigator::MAXSPLIT - 1; i++) {
		ok = ok && hipStreamCreateWithFlags(&nav->aux[i], hipStreamNonBlocking) == hipSuccess;
		ok = ok && ev_create(&nav->ev_join[i], evflags);
	}
	// ev_fork keeps the default (system-scope) release: it is recorded exactly when something ELSE preceded the step on the
	// handle's stream — typically the host-to-device copy of phd_set_measurements — and is then the only link between that copy
	// and the aux stream's k_sweep. It is off the back-to-back path (phd_step_async behind phd_step_async records no fork), so its
	// 5 us do not touch the steady state.
	ok = ok && ev_create(&nav->ev_fork, hipEventDisableTiming);
	ok = ok && ev_create(&nav->ev_res, evflags);
	if (const char* e = getenv("PHD_PIPELINE")) nav->pipeline = atoi(e) != 0;
	if (const char* e = getenv("PHD_DEVICE_ORDER")) nav->device_order = atoi(e) != 0;
	if (const char* e = getenv("PHD_DEFER_BIG")) nav->defer_big = atoi(e) != 0;
	if (const char* e = getenv("PHD_FUSE_EP")) nav->fuse_ep = atoi(e) != 0 ? 1 : 0;
#ifdef PHD_WITH_FUSE_SEP
	if (const char* e = getenv("PHD_FUSE_SEP")) nav->fuse_sep = atoi(e) != 0 ? 1 : 0;
#endif
	if (const char* e = getenv("PHD_NBIG")) nav->nbig = std::max(1, atoi(e));
	if (const char* e = getenv("PHD_SPLIT")) nav->nsplit = std::max(0, atoi(e));
	if (const char* e = getenv("PHD_CHAIN_MAX")) nav->chain_max = std::max(0, atoi(e));
	if (const char* e = getenv("PHD_DSPLIT_MAX")) nav->dsplit_max = std::max(0, atoi(e));
	if (const char* e = getenv("PHD_DSPLIT_LATE")) nav->dsplit_late = atoi(e);
	if (const char* e = getenv("PHD_DSPLIT_SEQ0")) nav->dseq = (unsigned int) std::max(0LL, atoll(e));   // (tests: launch numbers that start again soon)
	if (const char* e = getenv("PHD_FOLD_NR")) nav->fold_nr = atoi(e) != 0;
	if (const char* e = getenv("PHD_NR_GRID_MIN")) nav->nr_grid_min = std::max(0, atoi(e));
	if (const char* e = getenv("PHD_PLAN_GRID_MIN")) nav->plan_grid_min = std::max(0, atoi(e));
	size_t plane = (size_t) nav->Pcap * nav->cap;
	for (int i = 0; i < 3 && ok; i++) {
		ok = ok && dalloc((void**) &nav->bank[i].mix, plane * MIX_REC * 8);
		ok = ok && dalloc((void**) &nav->bank[i].count, (size_t) nav->Pcap * 4);
		ok = ok && dalloc((void**) &nav->bank[i].poses, (size_t) nav->Pcap * 7 * 8);
		ok = ok && dalloc((void**) &nav->bank[i].weights, (size_t) nav->Pcap * 8);
		if (ok) {
			hipMemset(nav->bank[i].count, 0, (size_t) nav->Pcap * 4);
			hipMemset(nav->bank[i].weights, 0, (size_t) nav->Pcap * 8);
			hipMemset(nav->bank[i].poses, 0, (size_t) nav->Pcap * 7 * 8);
		}
	}
	size_t E = (size_t) nav->Pcap * nav->ecap;
	// the role tables (two parities), the step's info words and its flag word in ONE block: phd_sync reads it with one copy
	// queued on the stream it then waits for (three blocking copies before: 40 us of a 90 us step at config A)
	ok = ok && dalloc((void**) &nav->d_sel, (2 * SEL_STRIDE + 4) * 4) && dalloc((void**) &nav->d_inslot, (size_t) nav->Pcap * 4);
	if (ok) { nav->d_info = nav->d_sel + 2 * SEL_STRIDE; nav->d_flags = nav->d_info + 2; }
	ok = ok && hipHostMalloc((void**) &nav->h_status, (2 * SEL_STRIDE + 4) * 4, hipHostMallocDefault) == hipSuccess;
	ok = ok && dalloc((void**) &nav->d_z, (size_t) nav->Mcap * 3 * 8);
	ok = ok && dalloc((void**) &nav->d_emit_w, E * 8) && dalloc((void**) &nav->d_emit_idx, E * 4);
	ok = ok && dalloc((void**) &nav->d_emit_rec, E * MIX_REC * 8) && dalloc((void**) &nav->d_emit_count, (size_t) nav->Pcap * 4);
	ok = ok && dalloc((void**) &nav->d_born_count, (size_t) nav->Pcap * 4);
	ok = ok && dalloc((void**) &nav->d_born_k, (size_t) nav->Pcap * nav->Mcap * 4);
	ok = ok && dalloc((void**) &nav->d_born_mean, (size_t) nav->Pcap * nav->Mcap * 3 * 8);
	ok = ok && dalloc((void**) &nav->d_alpha, (size_t) nav->Pcap * 8) && dalloc((void**) &nav->d_setll, (size_t) nav->Pcap * 8);
	ok = ok && dalloc((void**) &nav->d_src, (size_t) nav->Pcap * 4);
	ok = ok && dalloc((void**) &nav->d_murty, (size_t) nav->Pcap * sizeof(MurtyNodes));
	nav->bigws_bytes = 128ull << 20;
	ok = ok && dalloc((void**) &nav->d_bigws, nav->bigws_bytes) && dalloc((void**) &nav->d_bigws_used, 32);   // [0] the slab's bump counter; 32-bit words behind it: the device order's ticket counter (never reset) and step number, the chain's own ticket
	if (ok) hipMemset(nav->d_bigws_used, 0, 32);
	nav->cmcap = nav->cap + nav->Mcap;
	ok = ok && dalloc((void**) &nav->d_cand_count, (size_t) nav->Pcap * 4 * 4) && dalloc((void**) &nav->d_denom, (size_t) nav->Pcap * nav->Mcap * 8);
	nav->candcap = 16 * nav->cmcap;   // a quarter of all pairs at 64 measurements (four wave segments); beyond it the full second sweep runs
	ok = ok && dalloc((void**) &nav->d_cand, (size_t) nav->Pcap * nav->candcap * 8);
#ifdef PHD_STAMPS
	ok = ok && dalloc((void**) &nav->d_stamps, (size_t) nav->Pcap * 16 * 8);
#endif
	ok = ok && dalloc((void**) &nav->d_srec, (size_t) nav->Pcap * PRUNE_ROW * nav->cutcap * 8);
	ok = ok && dalloc((void**) &nav->d_outw, plane * 8);
	ok = ok && dalloc((void**) &nav->d_biglist, (size_t) phd_navigator::MAXSPLIT * (nav->Pcap + 2) * 4) && dalloc((void**) &nav->d_ratio, (size_t) nav->Pcap * 8);
	if (ok) hipMemset(nav->d_biglist, 0, (size_t) phd_navigator::MAXSPLIT * (nav->Pcap + 2) * 4);
	ok = ok && dalloc((void**) &nav->d_wcopy, (size_t) nav->Pcap * (nav->cap + nav->Mcap) * 8) && dalloc((void**) &nav->d_cover, (size_t) nav->Pcap * nav->cap * 4);
	ok = ok && dalloc((void**) &nav->d_alm, (size_t) nav->Pcap * 3 * nav->Jcap * 8);
	ok = ok && dalloc((void**) &nav->d_aJ, (size_t) nav->Pcap * 4) && dalloc((void**) &nav->d_account, (size_t) nav->Pcap * 8);
	if (nav->dsplit_max > 0) {
		const size_t rows = (size_t) std::min(nav->Pcap, (int) DSPLIT_ROWS);
		ok = ok && dalloc((void**) &nav->d_dsync, rows * 3 * 4);
		ok = ok && hipMemset(nav->d_dsync, 0, rows * 3 * 4) == hipSuccess;
	}
	ok = ok && dalloc((void**) &nav->d_jscratch, (size_t) nav->Pcap * alpha_jscratch_doubles(nav->Jcap) * 8);
	nav->stagecap = std::max((size_t) nav->Pcap * 8 + 8, (size_t) 256 * 3);   // poses + weights | odometry + noise | measurements
	ok = ok && dalloc((void**) &nav->d_stage, nav->stagecap * 8) && dalloc((void**) &nav->d_motion, ((size_t) nav->Pcap * 6 + 6) * 8);
	for (int i = 0; i < 2; i++) {
		ok = ok && hipHostMalloc((void**) &nav->h_stage[i], nav->stagecap * 8, hipHostMallocDefault) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&nav->ev_stage[i], hipEventDisableTiming) == hipSuccess;
	}
	{
		// dynamic LDS limits, set when a handle is created: they depend on the handle's capacities only. The attribute belongs
		// to the function ON THE CURRENT DEVICE, and the same kernels serve every handle of the process there: per device
		// the limit is only ever raised.
		struct DevLimits { int prune = 0, alpha[3] = {0, 0, 0}, chain[4] = {0, 0, 0, 0}; };
		static DevLimits limits[PHD_MAX_DEVICES];
		static std::mutex limits_mu;   // (handles may be created from several host threads, one per GPU)
		std::lock_guard<std::mutex> limits_guard(limits_mu);
		DevLimits& lim = limits[device];
		const int lp = prune_lds(nav->cutcap).bytes;
		if (lp > lim.prune) {
			ok = ok && hipFuncSetAttribute((const void*) k_prune_merge, hipFuncAttributeMaxDynamicSharedMemorySize, lp) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_emit_prune, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(lp, (int) (EMIT_LDS_DOUBLES * 8))) == hipSuccess;
#ifdef PHD_WITH_FUSE_SEP
			const int l3[3] = {std::max(lp, (int) (SweepLds<1>::doubles * 8)), std::max(lp, (int) (SweepLds<2>::doubles * 8)), std::max(lp, (int) (SweepLds<4>::doubles * 8))};
			ok = ok && hipFuncSetAttribute((const void*) k_sweep_emit_prune<1>, hipFuncAttributeMaxDynamicSharedMemorySize, l3[0]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_sweep_emit_prune<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l3[0]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_sweep_emit_prune<2>, hipFuncAttributeMaxDynamicSharedMemorySize, l3[1]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_sweep_emit_prune<4>, hipFuncAttributeMaxDynamicSharedMemorySize, l3[2]) == hipSuccess;
#endif
			lim.prune = lp;
		}
		const int la[3] = {alpha_lds(64, nav->cutcap).bytes, alpha_lds(128, nav->cutcap).bytes, alpha_lds(256, nav->cutcap).bytes};
		if (la[0] > lim.alpha[0]) {
			ok = ok && hipFuncSetAttribute((const void*) k_alpha_assoc<1>, hipFuncAttributeMaxDynamicSharedMemorySize, la[0]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_alpha_assoc_main<1>, hipFuncAttributeMaxDynamicSharedMemorySize, la[0]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_alpha_density_big<1>, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(la[0], (int) (DENS_LDS_DOUBLES * 8))) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_quasi_setll<1>, hipFuncAttributeMaxDynamicSharedMemorySize, la[0]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_quasi_setll_grad<1>, hipFuncAttributeMaxDynamicSharedMemorySize, la[0]) == hipSuccess;
			lim.alpha[0] = la[0];
		}
		if (la[1] > lim.alpha[1]) {
			ok = ok && hipFuncSetAttribute((const void*) k_alpha_assoc<2>, hipFuncAttributeMaxDynamicSharedMemorySize, la[1]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_alpha_assoc_main<2>, hipFuncAttributeMaxDynamicSharedMemorySize, la[1]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_alpha_density_big<2>, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(la[1], (int) (DENS_LDS_DOUBLES * 8))) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_quasi_setll<2>, hipFuncAttributeMaxDynamicSharedMemorySize, la[1]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_quasi_setll_grad<2>, hipFuncAttributeMaxDynamicSharedMemorySize, la[1]) == hipSuccess;
			lim.alpha[1] = la[1];
		}
		if (la[2] > lim.alpha[2]) {
			ok = ok && hipFuncSetAttribute((const void*) k_alpha_assoc<4>, hipFuncAttributeMaxDynamicSharedMemorySize, la[2]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_alpha_assoc_main<4>, hipFuncAttributeMaxDynamicSharedMemorySize, la[2]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_alpha_density_big<4>, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(la[2], (int) (DENS_LDS_DOUBLES * 8))) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_quasi_setll<4>, hipFuncAttributeMaxDynamicSharedMemorySize, la[2]) == hipSuccess;
			ok = ok && hipFuncSetAttribute((const void*) k_quasi_setll_grad<4>, hipFuncAttributeMaxDynamicSharedMemorySize, la[2]) == hipSuccess;
			lim.alpha[2] = la[2];
		}
		// the one-launch chain: its bodies share one pool, the largest of their layouts, which must fit a workgroup (160 KB)
		// with the kernel's few static words; where it does not (a large MaxQuantity) the separate kernels run
		const int lc[3] = {chain_lds_bytes<1>(nav->cutcap), chain_lds_bytes<2>(nav->cutcap), chain_lds_bytes<4>(nav->cutcap)};
		const void* chainfn[4] = {(const void*) k_particle_chain<1>, (const void*) k_particle_chain<2>, (const void*) k_particle_chain<4>,
		                          (const void*) k_particle_chain<1, true>};
		bool fits[4] = {false, false, false, false};
		for (int z = 0; z < 4 && ok; z++) {
			const int zl = z == 3 ? 0 : z;   // (the HALF build of the one-block chain shares its layout)
			hipFuncAttributes fc;
			if (hipFuncGetAttributes(&fc, chainfn[z]) != hipSuccess) { ok = false; break; }
			if ((size_t) fc.sharedSizeBytes + (size_t) lc[zl] + 256 > 160 * 1024) continue;
			if (lc[zl] > lim.chain[z]) {
				if (hipFuncSetAttribute(chainfn[z], hipFuncAttributeMaxDynamicSharedMemorySize, lc[zl]) != hipSuccess) { (void) hipGetLastError(); continue; }
				lim.chain[z] = lc[zl];
			}
			fits[z] = true;
		}
		nav->chain_ok[0] = fits[0] && fits[3];
		nav->chain_ok[1] = fits[1];
		nav->chain_ok[2] = fits[2];
		ok = ok && hipFuncSetAttribute((const void*) k_plan_migration, hipFuncAttributeMaxDynamicSharedMemorySize, PLAN_LDS_MAX + 1024) == hipSuccess;
		hipFuncAttributes fa;
		if (ok && hipFuncGetAttributes(&fa, (const void*) k_normalise_resample) == hipSuccess) {
			nav->nr_static_lds = (int) fa.sharedSizeBytes;
			ok = ok && hipFuncSetAttribute((const void*) k_normalise_resample, hipFuncAttributeMaxDynamicSharedMemorySize,
			                               160 * 1024 - nav->nr_static_lds) == hipSuccess;
		}
		else ok = false;
	}
	if (!ok) {
		g_create_error = std::string("device allocation failed: ") + hipGetErrorString(hipGetLastError());
		phd_destroy(nav);
		return nullptr;
	}
	int sel[2 * SEL_STRIDE] = {0, 1, 2, 0, 0, 0, 0, 0, 0, 1, 2, 0, 0, 0, 0, 0};
	hipMemcpy(nav->d_sel, sel, sizeof(sel), hipMemcpyHostToDevice);
	{
		std::vector<int> id(nav->Pcap);
		for (int i = 0; i < nav->Pcap; i++) id[i] = i;
		hipMemcpy(nav->d_inslot, id.data(), (size_t) nav->Pcap * 4, hipMemcpyHostToDevice);
	}
	hipMemset(nav->d_flags, 0, 4);
	hipMemset(nav->d_info, 0, 8);
	hipMemset(nav->d_emit_count, 0, (size_t) nav->Pcap * 4);
	hipMemset(nav->d_born_count, 0, (size_t) nav->Pcap * 4);
	return nav;
}

void phd_destroy(phd_navigator* nav)
{
	if (!nav) return;
	if (nav->multi) { multi_destroy(nav); return; }
	enter(nav);
	if (nav->stream) hipStreamSynchronize(nav->stream);
	for (int i = 0; i < 3; i++) {
		hipFree(nav->bank[i].mix); hipFree(nav->bank[i].count); hipFree(nav->bank[i].poses); hipFree(nav->bank[i].weights);
	}
	hipFree(nav->d_sel); hipFree(nav->d_inslot); hipFree(nav->d_mslot); hipFree(nav->d_z); hipFree(nav->d_emit_w); hipFree(nav->d_emit_idx); hipFree(nav->d_emit_rec);
	hipFree(nav->d_emit_count); hipFree(nav->d_born_count); hipFree(nav->d_born_k); hipFree(nav->d_born_mean);
	hipFree(nav->d_alpha); hipFree(nav->d_setll); hipFree(nav->d_src);
	hipFree(nav->d_murty); hipFree(nav->d_bigws); hipFree(nav->d_bigws_used); hipFree(nav->d_jscratch); hipFree(nav->d_dsync); hipFree(nav->d_stamps); hipFree(nav->d_srec); hipFree(nav->d_outw); hipFree(nav->d_wcopy); hipFree(nav->d_cover); hipFree(nav->d_motion); hipFree(nav->d_quasi); hipFree(nav->d_alm); hipFree(nav->d_aJ); hipFree(nav->d_account); hipFree(nav->d_cand_count); hipFree(nav->d_denom); hipFree(nav->d_cand); hipFree(nav->d_gw); hipFree(nav->d_send); hipFree(nav->d_recv); hipFree(nav->d_plan); hipFree(nav->d_nrd); hipFree(nav->d_nri);
	hipFree(nav->d_lw); hipFree(nav->d_dst_tab); hipFree(nav->d_recv_tab); hipFree(nav->plan.code); hipFree(nav->plan.fslot); hipFree(nav->plan.sendlist); hipFree(nav->plan.senddst); hipFree(nav->plan.counts);
	if (nav->h_counts) hipHostFree(nav->h_counts);
	for (void* q : nav->ipc_opened) hipIpcCloseMemHandle(q);
	hipFree(nav->d_graw);
	if (nav->h_quasi) hipHostFree(nav->h_quasi);
	if (nav->h_status) hipHostFree(nav->h_status);
	for (Timer& t : nav->timers) { hipEventDestroy(t.t0); hipEventDestroy(t.t1); }
	for (int i = 0; i < 2; i++) {
		if (nav->h_stage[i]) hipHostFree(nav->h_stage[i]);
		if (nav->ev_stage[i]) hipEventDestroy(nav->ev_stage[i]);
	}
	hipFree(nav->d_stage);
	if (nav->own_stream) hipStreamDestroy(nav->own_stream);
	for (int i = 0; i < phd_navigator::MAXSPLIT - 1; i++) {
		if (nav->aux[i]) hipStreamDestroy(nav->aux[i]);
		if (nav->ev_join[i]) hipEventDestroy(nav->ev_join[i]);
	}
	if (nav->ev_fork) hipEventDestroy(nav->ev_fork);
	if (nav->ev_res) hipEventDestroy(nav->ev_res);
	hipFree(nav->d_biglist); hipFree(nav->d_ratio);
	delete nav;
}

const char* phd_last_error(const phd_navigator* nav) { return nav ? nav->err.c_str() : "null handle"; }

int phd_particle_count(phd_navigator* nav) { return nav ? nav->P : 0; }

// phd_reset with the particle weight given (a shard of a multi-device handle holds 1 / the TOTAL count)
static int reset_impl(phd_navigator* nav, int nparticles, const double* pose7, const double* w, const double* mean3,
                      const double* cov9, int ncomp, double weight)
{
	if (nparticles < 1 || nparticles > nav->Pcap || !pose7 || ncomp < 0) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_reset: bad particle count / pose / ncomp");
	enter(nav);
	int rc = sync_state(nav);
	if (rc) return rc;
	rc = reset_indirection(nav);   // the state is replaced as a whole
	if (rc) return rc;
	int I = nav->h_sel[SEL_IN];
	rc = upload_particle(nav, I, 0, w, mean3, cov9, ncomp);
	if (rc) return rc;
	HC(hipMemcpy(nav->bank[I].poses, pose7, 7 * 8, hipMemcpyHostToDevice));
	nav->P = nparticles;
	StepBufs b = make_bufs(nav);
	hipLaunchKernelGGL(k_replicate, dim3(nparticles), dim3(256), 0, nav->stream, b, weight);
	HC(hipGetLastError());
	// the replicated state is in the OUT bank: make it current
	const int O = nav->h_sel[SEL_OUT], T = nav->h_sel[SEL_TMP];
	int sel[SEL_STRIDE] = {O, I, T, O, O, O, 0, 0};   // IN, OUT, TMP, RES, INMIX, RESMIX
	HC(hipStreamSynchronize(nav->stream));
	HC(hipMemcpy(nav->d_sel + nav->parity * SEL_STRIDE, sel, sizeof(sel), hipMemcpyHostToDevice));
	std::memcpy(nav->h_sel, sel, sizeof(sel));
	nav->h_info[0] = 0; nav->h_info[1] = 0;   // BestParticle = 0 (:265)
	HC(hipMemcpy(nav->d_info, nav->h_info, 8, hipMemcpyHostToDevice));
	nav->stage_valid = false;
	return PHD_OK;
}

int phd_reset(phd_navigator* nav, int nparticles, const double* pose7, const double* w, const double* mean3,
              const double* cov9, int ncomp)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (ncomp > 0 && (!w || !mean3 || !cov9)) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_reset: map arrays are NULL");
	FINITE_OR_FAIL(nav, pose7, 7, "phd_reset");
	FINITE_OR_FAIL(nav, w, std::max(ncomp, 0), "phd_reset");
	FINITE_OR_FAIL(nav, mean3, (size_t) std::max(ncomp, 0) * 3, "phd_reset");
	FINITE_OR_FAIL(nav, cov9, (size_t) std::max(ncomp, 0) * 9, "phd_reset");
	if (nav->multi) return multi_reset(nav, nparticles, pose7, w, mean3, cov9, ncomp);
	return reset_impl(nav, nparticles, pose7, w, mean3, cov9, ncomp, 1.0 / std::max(nparticles, 1));
}

int phd_set_poses(phd_navigator* nav, const double* poses7, int nparticles)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nparticles != nav->P || !poses7) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_set_poses: particle count mismatch");
	FINITE_OR_FAIL(nav, poses7, (size_t) nparticles * 7, "phd_set_poses");
	if (nav->multi) return multi_set_small(nav, poses7, nullptr, nparticles);
	if (nparticles != nav->P || !poses7) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_set_poses: particle count mismatch");
	enter(nav);
	// the bank that holds the current poses is known to the device (the roles rotate there, at the end of a step): the poses
	// are staged and stored by a kernel that resolves it, so this is correct right behind phd_step_async and never waits
	double* hs = stage_acquire(nav);
	std::memcpy(hs, poses7, (size_t) nparticles * 7 * 8);
	HC(hipMemcpyAsync(nav->d_stage, hs, (size_t) nparticles * 7 * 8, hipMemcpyHostToDevice, nav->stream));
	stage_release(nav);
	StepBufs b = make_bufs(nav);
	hipLaunchKernelGGL(k_store_small, dim3((nparticles * 7 + 255) / 256), dim3(256), 0, nav->stream, b, (const double*) nav->d_stage, (const double*) nullptr, nparticles);
	HC(hipGetLastError());
	nav->stage_valid = false;
	return PHD_OK;
}

// TrackVehicle.UpdateNoisy for every particle, on the device: the host passes the odometry reading
// and, per particle, the noise vector it drew (RNG and Cholesky factor stay managed); no pose array crosses PCIe.
int phd_update_motion(phd_navigator* nav, const double* odometry6, const double* noise6, int nparticles, uint8_t perfect_still)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (!odometry6 || nparticles != nav->P) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_update_motion: particle count mismatch");
	FINITE_OR_FAIL(nav, odometry6, 6, "phd_update_motion");
	FINITE_OR_FAIL(nav, noise6, (size_t) nparticles * 6, "phd_update_motion");
	if (nav->multi) return multi_update_motion(nav, odometry6, noise6, nparticles, perfect_still);
	if (nav->prm.model != PHD_MODEL_PRM3D) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_update_motion: Pose3D odometry, the PRM3D model only");
	if (!odometry6 || nparticles != nav->P) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_update_motion: particle count mismatch");
	enter(nav);
	bool zero = true;
	for (int t = 0; t < 6; t++) zero = zero && odometry6[t] == 0;
	const int use_noise = noise6 && !(perfect_still && zero);   // "static friction makes the robot stay put", TrackVehicle.cs:93-94
	double* hs = stage_acquire(nav);
	std::memcpy(hs, odometry6, 6 * 8);
	if (use_noise) std::memcpy(hs + 6, noise6, (size_t) nparticles * 6 * 8);
	HC(hipMemcpyAsync(nav->d_motion, hs, (6 + (use_noise ? (size_t) nparticles * 6 : 0)) * 8, hipMemcpyHostToDevice, nav->stream));
	stage_release(nav);   // the caller's buffers are free again; nothing waits for the device
	StepBufs b = make_bufs(nav);
	hipLaunchKernelGGL(k_motion, dim3((nparticles + 255) / 256), dim3(256), 0, nav->stream, b, nparticles,
	                   (const double*) nav->d_motion, (const double*) (nav->d_motion + 6), use_noise);
	HC(hipGetLastError());
	nav->stage_valid = false;
	return PHD_OK;
}

static int quasi_batch(phd_navigator* nav, const double* poses7, int nposes, const double* landmarks3, int nlandmarks,
                       const double* z3, int nmeasurements, double* out, double* gradients6, int average_mode)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) nav = multi_shard0(nav);   // a batch of candidate poses against one landmark set does not touch the particle state
	if (nposes < 1 || nposes > nav->Pcap || nlandmarks < 0 || nlandmarks > nav->Jcap || nmeasurements < 0 ||
	    nmeasurements > nav->prm.max_measurements || !poses7 || !out || (nlandmarks && !landmarks3) || (nmeasurements && !z3)) {
		return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_quasi_set_loglik: sizes out of range (poses <= max_particles, landmarks <= min(1024, max_quantity), measurements <= max_measurements)");
	}
	FINITE_OR_FAIL(nav, poses7, (size_t) nposes * 7, "phd_quasi_set_loglik");
	FINITE_OR_FAIL(nav, landmarks3, (size_t) nlandmarks * 3, "phd_quasi_set_loglik");
	FINITE_OR_FAIL(nav, z3, (size_t) nmeasurements * 3, "phd_quasi_set_loglik");
	enter(nav);
	// One buffer on the device and its pinned mirror on the host, packed per call:
	//   poses[n][7] | landmarks[J][3] | z[M][3] | flag word | slab counter | out[n] | gradients[n][6]
	// so that a call is ONE copy in (inputs, with the two words zeroed), the kernel, ONE copy out (from the flag word on) and
	// one wait. The smoother calls this in a loop: six copies from and to pageable memory, two memsets and a blocking read of
	// the step's flag word cost more than the value kernel's 40 us. The batch has a flag word and a slab counter of its own:
	// nothing it raises can drop the next step, and the step's association kernel finds its counter as it left it.
	const size_t cap = (size_t) nav->Pcap * 14 + (size_t) nav->Jcap * 3 + 256 * 3 + 2;
	const size_t op = 0, ol = op + (size_t) nposes * 7, oz = ol + (size_t) nlandmarks * 3, of = oz + (size_t) nmeasurements * 3, ou = of + 1,
	             oo = ou + 1, og = oo + nposes, end = og + (gradients6 ? (size_t) nposes * 6 : 0);
	if (!nav->d_quasi) HC(hipMalloc((void**) &nav->d_quasi, cap * 8));
	if (!nav->h_quasi) HC(hipHostMalloc((void**) &nav->h_quasi, cap * 8, hipHostMallocDefault));
	double* const hq = nav->h_quasi;
	std::memcpy(hq + op, poses7, (size_t) nposes * 7 * 8);
	if (nlandmarks) std::memcpy(hq + ol, landmarks3, (size_t) nlandmarks * 3 * 8);
	if (nmeasurements) std::memcpy(hq + oz, z3, (size_t) nmeasurements * 3 * 8);
	hq[of] = 0; hq[ou] = 0;
	HC(hipMemcpyAsync(nav->d_quasi, hq, oo * 8, hipMemcpyHostToDevice, nav->stream));
	StepBufs b = make_bufs(nav);
	b.P = nposes; b.M = nmeasurements; b.z = nav->d_quasi + oz;
	b.qposes = nav->d_quasi + op; b.qlm = nav->d_quasi + ol; b.qJ = nlandmarks;
	b.setll = nav->d_quasi + oo;
	b.qgrad = nav->d_quasi + og;
	b.qavg  = average_mode;
	b.flags = (int*) (nav->d_quasi + of);
	b.bigws_used = (unsigned long long*) (nav->d_quasi + ou);
	const bool gradient = gradients6 != nullptr;
	int rc;
	switch (zb_of(nmeasurements)) {
	case 1:  rc = launch_quasi<1>(nav, b, nposes, gradient); break;
	case 2:  rc = launch_quasi<2>(nav, b, nposes, gradient); break;
	default: rc = launch_quasi<4>(nav, b, nposes, gradient); break;
	}
	if (rc) return rc;
	HC(hipMemcpyAsync(hq + of, nav->d_quasi + of, (end - of) * 8, hipMemcpyDeviceToHost, nav->stream));
	HC(hipStreamSynchronize(nav->stream));
	std::memcpy(out, hq + oo, (size_t) nposes * 8);
	if (gradient) std::memcpy(gradients6, hq + og, (size_t) nposes * 6 * 8);
	int flags = 0;
	std::memcpy(&flags, hq + of, 4);
	if (flags & PHD_FLAG_BIG_CLUSTER) {
		return nav->fail(PHD_ERR_ASSOCIATION, "phd_quasi_set_loglik: an association cluster exceeds the on-device solver");
	}
	return PHD_OK;
}

int phd_quasi_set_loglik(phd_navigator* nav, const double* poses7, int nposes, const double* landmarks3, int nlandmarks,
                         const double* z3, int nmeasurements, double* out)
{
	return quasi_batch(nav, poses7, nposes, landmarks3, nlandmarks, z3, nmeasurements, out, nullptr, 0);
}

int phd_quasi_set_loglik_grad(phd_navigator* nav, const double* poses7, int nposes, const double* landmarks3, int nlandmarks,
                              const double* z3, int nmeasurements, int average_mode, double* out, double* gradients6)
{
	if (nav && (!gradients6 || average_mode < 0 || average_mode > 1)) {
		return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_quasi_set_loglik_grad: gradients6 is NULL or average_mode is not 0 / 1");
	}
	return quasi_batch(nav, poses7, nposes, landmarks3, nlandmarks, z3, nmeasurements, out, gradients6, average_mode);
}

int phd_test_pairing(phd_navigator* nav, const double* matrix, int n, int mode, int modelsize, int maxcount,
                     int32_t* assignments, double* values, int* count)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) nav = multi_shard0(nav);
	if (!matrix || !assignments || !values || !count || n < 1 || n > MURTY_NBIG || maxcount < 1 || (mode == 1 && n > 5) || mode < 0 || mode > 1) {
		return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_test_pairing: n in 1..256 (1..5 for the lexicographic order), mode 0 or 1, buffers for maxcount pairings");
	}
	enter(nav);
	HC(hipStreamSynchronize(nav->stream));
	double* dm = nullptr; int* da = nullptr; double* dv = nullptr; int* dc = nullptr; char* dbig = nullptr;
	HC(hipMalloc((void**) &dm, (size_t) n * n * 8));
	if (n > MURTY_NMAX) HC(hipMalloc((void**) &dbig, murty_big_bytes(n)));
	HC(hipMalloc((void**) &da, (size_t) maxcount * n * 4));
	HC(hipMalloc((void**) &dv, (size_t) maxcount * 8));
	HC(hipMalloc((void**) &dc, 4));
	hipError_t e = hipMemcpy(dm, matrix, (size_t) n * n * 8, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		hipMemset(da, 0xff, (size_t) maxcount * n * 4);
		hipLaunchKernelGGL(k_test_pairing, dim3(1), dim3(64), 0, nav->stream, nav->d_murty, dbig, (const double*) dm, n, mode, modelsize, maxcount, da, dv, dc);
		e = hipStreamSynchronize(nav->stream);
	}
	int m = 0;
	if (e == hipSuccess) e = hipMemcpy(&m, dc, 4, hipMemcpyDeviceToHost);
	const int k = std::min(m, maxcount);
	if (e == hipSuccess && k > 0) e = hipMemcpy(assignments, da, (size_t) k * n * 4, hipMemcpyDeviceToHost);
	if (e == hipSuccess && k > 0) e = hipMemcpy(values, dv, (size_t) k * 8, hipMemcpyDeviceToHost);
	hipFree(dm); hipFree(da); hipFree(dv); hipFree(dc); hipFree(dbig);
	if (e != hipSuccess) return nav->fail(PHD_ERR_DEVICE, std::string("phd_test_pairing: ") + hipGetErrorString(e));
	*count = m;
	return PHD_OK;
}

int phd_set_weights(phd_navigator* nav, const double* weights, int nparticles)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nparticles != nav->P || !weights) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_set_weights: particle count mismatch");
	FINITE_OR_FAIL(nav, weights, nparticles, "phd_set_weights");
	if (nav->multi) return multi_set_small(nav, nullptr, weights, nparticles);
	if (nparticles != nav->P || !weights) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_set_weights: particle count mismatch");
	enter(nav);
	double* hs = stage_acquire(nav);
	std::memcpy(hs, weights, (size_t) nparticles * 8);
	HC(hipMemcpyAsync(nav->d_stage, hs, (size_t) nparticles * 8, hipMemcpyHostToDevice, nav->stream));
	stage_release(nav);
	StepBufs b = make_bufs(nav);
	hipLaunchKernelGGL(k_store_small, dim3((nparticles + 255) / 256), dim3(256), 0, nav->stream, b, (const double*) nullptr, (const double*) nav->d_stage, nparticles);
	HC(hipGetLastError());
	nav->stage_valid = false;
	return PHD_OK;
}

int phd_set_map(phd_navigator* nav, int particle, const double* w, const double* mean3, const double* cov9, int ncomp)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (ncomp < 0 || (ncomp > 0 && (!w || !mean3 || !cov9))) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_set_map: bad map arrays");
	FINITE_OR_FAIL(nav, w, ncomp, "phd_set_map");
	FINITE_OR_FAIL(nav, mean3, (size_t) ncomp * 3, "phd_set_map");
	FINITE_OR_FAIL(nav, cov9, (size_t) ncomp * 9, "phd_set_map");
	if (nav->multi) return multi_set_map(nav, particle, w, mean3, cov9, ncomp);
	if (particle < 0 || particle >= nav->P) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_set_map: particle out of range");
	enter(nav);
	int rc = sync_state(nav);
	if (rc) return rc;
	rc = materialise(nav);   // a particle's slot may be shared with the other copies of its resampling source
	if (rc) return rc;
	return upload_particle(nav, cur_bank(nav), particle, w, mean3, cov9, ncomp);
}

// bulk upload of the whole particle set in the device layout (bench / tests):
// planes[10][nparticles][stride] (w, mx, my, mz, xx, xy, xz, yy, yz, zz), counts[nparticles],
// poses[nparticles][7], weights[nparticles]. Sets the particle count.
// (plane_stride: doubles between two planes of the host array — a shard of a multi-device handle takes its rows of every plane)
static int upload_impl(phd_navigator* nav, int nparticles, int stride, const double* planes, size_t plane_stride, const int32_t* counts,
                       const double* poses7, const double* weights)
{
	if (nparticles < 1 || nparticles > nav->Pcap || stride < 0 || stride > nav->cap) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_upload_state_soa: sizes out of range");
	enter(nav);
	int rc = sync_state(nav);
	if (rc) return rc;
	rc = reset_indirection(nav);
	if (rc) return rc;
	int I = cur_bank(nav);
	if (stride > 0) {
		// the ABI's host layout is plane per field ([10][particles][stride], the round-1 device layout); the banks hold one
		// record per component: converted here, at the edge, in pieces of a few thousand particles
		const int chunk = std::max(1, (int) std::min<size_t>((size_t) nparticles, ((size_t) 64 << 20) / ((size_t) stride * MIX_REC * 8) + 1));
		std::vector<double> recs((size_t) chunk * stride * MIX_REC);
		for (int p0 = 0; p0 < nparticles; p0 += chunk) {
			const int np_ = std::min(chunk, nparticles - p0);
			for (int i = 0; i < np_; i++) {
				const int nc = std::min(std::max(counts[p0 + i], 0), stride);
				double* r = recs.data() + (size_t) i * stride * MIX_REC;
				for (int f = 0; f < MIX_REC; f++) {
					const double* src = planes + (size_t) f * plane_stride + (size_t) (p0 + i) * stride;
					for (int c = 0; c < nc; c++) r[(size_t) c * MIX_REC + f] = src[c];
				}
			}
			HC(hipMemcpy2D(nav->bank[I].mix + (size_t) p0 * nav->cap * MIX_REC, (size_t) nav->cap * MIX_REC * 8, recs.data(),
			               (size_t) stride * MIX_REC * 8, (size_t) stride * MIX_REC * 8, np_, hipMemcpyHostToDevice));
		}
	}
	HC(hipMemcpy(nav->bank[I].count, counts, (size_t) nparticles * 4, hipMemcpyHostToDevice));
	HC(hipMemcpy(nav->bank[I].poses, poses7, (size_t) nparticles * 7 * 8, hipMemcpyHostToDevice));
	HC(hipMemcpy(nav->bank[I].weights, weights, (size_t) nparticles * 8, hipMemcpyHostToDevice));
	nav->P = nparticles;
	nav->stage_valid = false;
	return PHD_OK;
}

int phd_upload_state_soa(phd_navigator* nav, int nparticles, int stride, const double* planes, const int32_t* counts,
                         const double* poses7, const double* weights)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nparticles < 1 || nparticles > nav->Pcap || stride < 0 || !counts || !poses7 || !weights || (stride > 0 && !planes)) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_upload_state_soa: sizes out of range");
	FINITE_OR_FAIL(nav, poses7, (size_t) nparticles * 7, "phd_upload_state_soa");
	FINITE_OR_FAIL(nav, weights, nparticles, "phd_upload_state_soa");
	for (int i = 0; i < nparticles; i++) {   // (the components a particle holds; what lies behind its count is not read)
		if (counts[i] < 0 || counts[i] > stride) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_upload_state_soa: a component count exceeds the stride");
		for (int f = 0; f < 10; f++) FINITE_OR_FAIL(nav, planes + ((size_t) f * nparticles + i) * stride, counts[i], "phd_upload_state_soa");
	}
	if (nav->multi) return multi_upload(nav, nparticles, stride, planes, counts, poses7, weights);
	return upload_impl(nav, nparticles, stride, planes, (size_t) nparticles * stride, counts, poses7, weights);
}

// bulk download in the same layout; planes must hold [10][P][stride] with stride >= the largest count
static int download_impl(phd_navigator* nav, int stride, double* planes, size_t plane_stride, int32_t* counts, double* poses7, double* weights)
{
	enter(nav);
	int rc = sync_state(nav);
	if (rc) return rc;
	if (stride < 0 || stride > nav->cap) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_download_state_soa: stride out of range");
	rc = materialise(nav);
	if (rc) return rc;
	int I = cur_bank(nav);
	HC(hipMemcpy(counts, nav->bank[I].count, (size_t) nav->P * 4, hipMemcpyDeviceToHost));
	if (stride > 0) {   // records -> the ABI's planes (see upload_impl); slots behind a particle's count are left as they are
		const int chunk = std::max(1, (int) std::min<size_t>((size_t) nav->P, ((size_t) 64 << 20) / ((size_t) stride * MIX_REC * 8) + 1));
		std::vector<double> recs((size_t) chunk * stride * MIX_REC);
		for (int p0 = 0; p0 < nav->P; p0 += chunk) {
			const int np_ = std::min(chunk, nav->P - p0);
			HC(hipMemcpy2D(recs.data(), (size_t) stride * MIX_REC * 8, nav->bank[I].mix + (size_t) p0 * nav->cap * MIX_REC,
			               (size_t) nav->cap * MIX_REC * 8, (size_t) stride * MIX_REC * 8, np_, hipMemcpyDeviceToHost));
			for (int i = 0; i < np_; i++) {
				const int nc = std::min(std::max(counts[p0 + i], 0), stride);
				const double* r = recs.data() + (size_t) i * stride * MIX_REC;
				for (int f = 0; f < MIX_REC; f++) {
					double* dst = planes + (size_t) f * plane_stride + (size_t) (p0 + i) * stride;
					for (int c = 0; c < nc; c++) dst[c] = r[(size_t) c * MIX_REC + f];
				}
			}
		}
	}
	if (poses7) HC(hipMemcpy(poses7, nav->bank[I].poses, (size_t) nav->P * 7 * 8, hipMemcpyDeviceToHost));
	if (weights) HC(hipMemcpy(weights, nav->bank[I].weights, (size_t) nav->P * 8, hipMemcpyDeviceToHost));
	return PHD_OK;
}

int phd_download_state_soa(phd_navigator* nav, int stride, double* planes, int32_t* counts, double* poses7, double* weights)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) return multi_download(nav, stride, planes, counts, poses7, weights);
	return download_impl(nav, stride, planes, (size_t) nav->P * stride, counts, poses7, weights);
}

int phd_set_measurements(phd_navigator* nav, const double* z3, int nmeasurements)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nmeasurements < 0 || nmeasurements > nav->prm.max_measurements || (nmeasurements > 0 && !z3)) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_set_measurements: count out of range");
	FINITE_OR_FAIL(nav, z3, (size_t) nmeasurements * 3, "phd_set_measurements");
	if (nav->multi) return multi_set_measurements(nav, z3, nmeasurements);
	if (nmeasurements < 0 || nmeasurements > nav->prm.max_measurements || (nmeasurements > 0 && !z3)) {
		return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_set_measurements: count out of range");
	}
	enter(nav);
	if (nmeasurements > 0) {
		double* hs = stage_acquire(nav);
		std::memcpy(hs, z3, (size_t) nmeasurements * 3 * 8);
		HC(hipMemcpyAsync(nav->d_z, hs, (size_t) nmeasurements * 3 * 8, hipMemcpyHostToDevice, nav->stream));
		stage_release(nav);
	}
	nav->M = nmeasurements;
	return PHD_OK;
}

int phd_set_split(phd_navigator* nav, int nsplit)
{
	if (!nav || nsplit < 0 || nsplit > phd_navigator::MAXSPLIT) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) return multi_forward_int(nav, 0, nsplit);
	nav->nsplit = nsplit;
	return PHD_OK;
}

int phd_set_association_workspace(phd_navigator* nav, int64_t bytes)
{
	if (!nav || bytes < 0) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) return multi_forward_int(nav, 1, bytes);
	enter(nav);
	HC(hipStreamSynchronize(nav->stream));
	hipFree(nav->d_bigws);
	nav->d_bigws = nullptr;
	nav->bigws_bytes = 0;
	if (bytes > 0) {
		HC(hipMalloc((void**) &nav->d_bigws, (size_t) bytes));
		nav->bigws_bytes = (unsigned long long) bytes;
	}
	HC(hipMemset(nav->d_bigws_used, 0, 8));
	return PHD_OK;
}

int phd_set_frozen(phd_navigator* nav, uint8_t frozen)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) return multi_forward_int(nav, 2, frozen);
	nav->frozen = frozen != 0;
	return PHD_OK;
}

int phd_set_all_pairs(phd_navigator* nav, uint8_t all_pairs)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) return multi_forward_int(nav, 3, all_pairs);
	nav->all_pairs = all_pairs != 0;
	return PHD_OK;
}

// Does a step of this handle run as the one-launch chain (launch_map_kernels), and may that launch end the step itself?
static bool chain_folds_normalise(const phd_navigator* nav)
{
	const int zb = zb_of(nav->M), zi = zb == 1 ? 0 : (zb == 2 ? 1 : 2);
	if (!nav->fold_nr || !nav->chain_ok[zi] || nav->P > nav->chain_max) return false;
	const size_t lw = (size_t) ((nav->P + 255) / 256) * 257 * 8;   // the weight vector, chunk-transposed, in the chain's pool
	const size_t pool = zb == 1 ? chain_lds_bytes<1>(nav->cutcap) : (zb == 2 ? chain_lds_bytes<2>(nav->cutcap) : chain_lds_bytes<4>(nav->cutcap));
	return lw <= pool;
}

int phd_step_async(phd_navigator* nav, uint8_t onlymapping, double u_resample)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) return multi_step(nav, onlymapping, u_resample);
	if (nav->P < 1) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_step: no particles (call phd_reset first)");
	hipSetDevice(nav->device);   // (not enter(): a step behind a step leaves the streams ordered)
	nav->timing_now = (nav->timing_step++ % nav->timing_period) == 0;
	StepBufs b = make_bufs(nav);
	nav->d_res_slots = nav->d_src;
	const bool folded = chain_folds_normalise(nav);
	if (folded) {
		// a small particle set: the chain's last workgroup normalises, resamples and rotates the roles (no launch of its own)
		b.fold_nr = 1; b.nr_u = u_resample; b.nr_force = onlymapping ? -1 : 0; b.nr_skip = onlymapping ? 1 : 0; b.nr_frozen = nav->frozen ? 1 : 0;
		b.nr_src = nav->d_src; b.nr_info = nav->d_info; b.nr_sel_next = nav->d_sel + (nav->parity ^ 1) * SEL_STRIDE; b.nr_inslot = nav->d_inslot;
	}
	// The step boundary with two sub-range streams. A fork before the step and a join behind it put two markers and a barrier
	// between the last k_alpha_density and k_normalise_resample, and one more between that and the next k_sweep: 22 + 21 us in
	// which the device runs nothing (scripts/timeline_step.py). Posted back to back, the steps need neither: the stream whose
	// chain starts second finishes last (`lagger`), k_normalise_resample goes behind ITS k_alpha_density (the other stream's
	// event is long recorded), its next k_sweep behind that — and so it leads the next step, the other stream, which waits for
	// the event, lags and takes the end of that one.
	const int zi_ = zb_of(nav->M) == 1 ? 0 : (zb_of(nav->M) == 2 ? 1 : 2);
	const bool chain = nav->chain_ok[zi_] && nav->P <= nav->chain_max;
	const int want = nav->nsplit > 0 ? nav->nsplit : (nav->P >= 1024 ? 2 : 1);
	const bool pipe = nav->pipeline && !chain && want == 2 && nav->P >= 2 && nav->stream == nav->own_stream;
	int rc;
	if (pipe) {
		if (!nav->pipe_ok) {
			// something else was enqueued on the stream since the last step (or this is the first): fork, `stream` leads
			HC(hipEventRecord(nav->ev_fork, nav->stream));
			HC(hipStreamWaitEvent(nav->aux[0], nav->ev_fork, 0));
			nav->lagger = 1;
		}
		hipStream_t L = nav->lagger ? nav->aux[0] : nav->stream, X = nav->lagger ? nav->stream : nav->aux[0];
		// (on the device: a localising step whose densities run as k_alpha_density — that kernel's workgroups take the tickets)
		const bool dev = nav->device_order && !onlymapping && !nav->defer_big;
		b.tickets = dev ? 1 : 0;
		rc = launch_map(nav, b, !onlymapping, nav->lagger);
		if (rc) { nav->pipe_ok = false; return rc; }
		b.defer = nav->last_defer;
		if (dev) {
			nav->step_seq++;
			if (nav->step_seq == 0) nav->step_seq = 1;
			nav->ticket_total += (unsigned) nav->P;
			b.wait_tickets = 1;
			b.ticket_target = nav->ticket_total;
			b.done_value = nav->step_seq;
		}
		else {
			HC(hipEventRecord(nav->ev_join[0], X));
			HC(hipStreamWaitEvent(L, nav->ev_join[0], 0));
		}
		timer_begin(nav, T_NR, L);
		rc = launch_normalise(nav, b, nullptr, nav->P, u_resample, onlymapping ? -1 : 0, onlymapping ? 1 : 0, nav->d_src, nav->d_info,
		                      nav->d_sel + (nav->parity ^ 1) * SEL_STRIDE, L);
		timer_end(nav, T_NR, L);
		if (rc) { nav->pipe_ok = false; return rc; }
		if (dev) {
			hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, X, b.ticket + 1, nav->step_seq, nav->d_flags);
			HC(hipGetLastError());
		}
		else {
			HC(hipEventRecord(nav->ev_res, L));
			HC(hipStreamWaitEvent(X, nav->ev_res, 0));
		}
		nav->lagger ^= 1;
		nav->pipe_ok = true;
	}
	else {
		nav->pipe_ok = false;
		rc = launch_map(nav, b, !onlymapping);
		if (rc) return rc;
		if (!folded) {
			b.defer = nav->last_defer;
			timer_begin(nav, T_NR);
			// the same launch hands the resampled particles their small arrays and rotates the bank roles (rotate_roles)
			rc = launch_normalise(nav, b, nullptr, nav->P, u_resample, onlymapping ? -1 : 0, onlymapping ? 1 : 0, nav->d_src, nav->d_info,
			                      nav->d_sel + (nav->parity ^ 1) * SEL_STRIDE);
			timer_end(nav, T_NR);
			if (rc) return rc;
		}
	}
	nav->parity ^= 1;
	nav->stage_valid = false;
	nav->sel_host_valid = false;   // the rotation depends on the resampling flag, known only on the device
	return PHD_OK;
}

int phd_sync(phd_navigator* nav)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) return multi_sync(nav);
	enter(nav);
	int rc = sync_state(nav);
	if (rc) return rc;
	rc = check_flags(nav);
	if (nav->h_flags) {
		hipMemset(nav->d_flags, 0, 4);
	}
	if (!rc && nav->plan_on_device && nav->sharded_used) {
		// the host never saw the plan of the last sharded step: what it said (a flag on another rank drops the step here too)
		int st[2] = {MIG_OK, 0};
		HC(hipMemcpy(st, nav->plan.counts + 2 * nav->world + 2, 8, hipMemcpyDeviceToHost));
		nav->h_info[1] = st[1];
		if (st[0] == MIG_DROPPED) rc = nav->fail(PHD_ERR_GENERIC, "the step was dropped: another rank raised a flag (its phd_sync says which); the state is the one before the step");
		else if (st[0] == MIG_BAD) rc = nav->fail(PHD_ERR_GENERIC, "the gathered source vector was not a resampling result (were the weights of all ranks gathered?)");
		else if (st[0] == MIG_OVERFLOW) rc = nav->fail(PHD_ERR_CAPACITY, "more migrating particles than the send list holds");
	}
	return rc;
}

int phd_slam_update(phd_navigator* nav, const double* z3, int nmeasurements, uint8_t onlymapping, double u_resample)
{
	int rc = phd_set_measurements(nav, z3, nmeasurements);
	if (rc) return rc;
	rc = phd_step_async(nav, onlymapping, u_resample);
	if (rc) return rc;
	return phd_sync(nav);
}

const double* phd_weights(phd_navigator* nav, int* length)
{
	if (!nav) return nullptr;
	if (nav->multi) return multi_weights(nav, length);
	enter(nav);
	if (sync_state(nav)) return nullptr;
	int bidx = res_small(nav);
	nav->h_weights.resize(std::max(nav->P, 1));
	if (hipMemcpy(nav->h_weights.data(), nav->bank[bidx].weights, (size_t) nav->P * 8, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
	if (length) *length = nav->P;
	return nav->h_weights.data();
}

int phd_best_particle(phd_navigator* nav)
{
	if (!nav) return -1;
	if (nav->multi) return multi_best_particle(nav);
	enter(nav);
	if (sync_state(nav)) return -1;
	return nav->h_info[0];
}

const double* phd_poses(phd_navigator* nav, int* length)
{
	if (!nav) return nullptr;
	if (nav->multi) return multi_poses(nav, length);
	enter(nav);
	if (sync_state(nav)) return nullptr;
	int bidx = res_small(nav);
	nav->h_poses.resize((size_t) std::max(nav->P, 1) * 7);
	if (hipMemcpy(nav->h_poses.data(), nav->bank[bidx].poses, (size_t) nav->P * 7 * 8, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
	if (length) *length = nav->P * 7;
	return nav->h_poses.data();
}

int phd_map(phd_navigator* nav, int particle, int* ncomp, const double** w, const double** mean3, const double** cov9)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) return multi_map(nav, particle, ncomp, w, mean3, cov9);
	if (particle < 0 || particle >= nav->P || !ncomp) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_map: particle out of range");
	enter(nav);
	int rc = sync_state(nav);
	if (rc) return rc;
	rc = fetch_map(nav, res_small(nav), particle, ncomp, res_mix(nav), res_slots(nav));
	if (rc) return rc;
	if (w) *w = nav->h_mw.data();
	if (mean3) *mean3 = nav->h_mm.data();
	if (cov9) *cov9 = nav->h_mc.data();
	return PHD_OK;
}

const int32_t* phd_resample_sources(phd_navigator* nav, int* length, uint8_t* resampled)
{
	if (!nav) return nullptr;
	if (nav->multi) return multi_resample_sources(nav, length, resampled);
	enter(nav);
	if (sync_state(nav)) return nullptr;
	nav->h_src.resize(std::max(nav->P, 1));
	if (hipMemcpy(nav->h_src.data(), nav->d_src, (size_t) nav->P * 4, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
	if (length) *length = nav->P;
	if (resampled) *resampled = nav->h_info[1] ? 1 : 0;
	return nav->h_src.data();
}

// ---- stage-level entry points ------------------------------------------------------------------
int phd_stage_run(phd_navigator* nav, const double* z3, int nmeasurements, uint8_t with_alpha)
{
	MULTI_UNSUPPORTED(nav, "phd_stage_run");
	int rc = phd_set_measurements(nav, z3, nmeasurements);
	if (rc) return rc;
	if (nav->P < 1) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_stage_run: no particles");
	rc = sync_state(nav);
	if (rc) return rc;
	StepBufs b = make_bufs(nav);
	HC(hipMemsetAsync(nav->d_bigws_used, 0, 8, nav->stream));
	rc = launch_map(nav, b, with_alpha != 0);
	if (rc) return rc;
	if (nav->last_defer) {   // (no k_normalise_resample follows a stage run: alpha is finished here)
		b.defer = 1;
		hipLaunchKernelGGL(k_alpha_combine, dim3((nav->P + 255) / 256), dim3(256), 0, nav->stream, b);
	}
	hipLaunchKernelGGL(k_expand_emit, dim3(nav->P), dim3(256), 0, nav->stream, nav->dp, b);   // PHD_STAGE_CORRECTED reads whole records
	HC(hipGetLastError());
	HC(hipMemsetAsync(nav->d_bigws_used, 0, 8, nav->stream));   // (as behind a quasi batch: no k_normalise_resample follows a stage run)
	for (int s_ = 0; s_ < phd_navigator::MAXSPLIT; s_++) HC(hipMemsetAsync(nav->d_biglist + (size_t) s_ * (nav->Pcap + 2), 0, 4, nav->stream));
	rc = sync_state(nav);
	if (rc) return rc;
	nav->stage_valid = true;
	rc = check_flags(nav);
	if (nav->h_flags) hipMemset(nav->d_flags, 0, 4);
	return rc;
}

int phd_stage_map(phd_navigator* nav, int stage, int particle, int* ncomp, const double** w, const double** mean3,
                  const double** cov9)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_stage_map");
	if (!nav->stage_valid) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_stage_map: call phd_stage_run first");
	if (particle < 0 || particle >= nav->P || !ncomp) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_stage_map: particle out of range");
	enter(nav);
	int rc = PHD_OK;
	if (stage == PHD_STAGE_PRUNED) {
		rc = fetch_map(nav, nav->h_sel[SEL_OUT], particle, ncomp);
	}
	else if (stage == PHD_STAGE_PREDICTED) {
		int n = 0;
		rc = fetch_map(nav, nav->h_sel[SEL_IN], particle, &n, nav->h_sel[SEL_INMIX], nav->d_inslot);
		if (rc) return rc;
		int nb = 0;
		HC(hipMemcpy(&nb, nav->d_born_count + particle, 4, hipMemcpyDeviceToHost));
		std::vector<double> bm((size_t) std::max(nb, 1) * 3);
		if (nb > 0) HC(hipMemcpy(bm.data(), nav->d_born_mean + (size_t) particle * nav->Mcap * 3, (size_t) nb * 3 * 8, hipMemcpyDeviceToHost));
		nav->h_mw.resize(n + nb); nav->h_mm.resize((size_t) (n + nb) * 3); nav->h_mc.resize((size_t) (n + nb) * 9);
		for (int i = 0; i < nb; i++) {
			nav->h_mw[n + i] = nav->prm.birth_weight;
			for (int k = 0; k < 3; k++) nav->h_mm[(size_t) (n + i) * 3 + k] = bm[i * 3 + k];
			for (int k = 0; k < 9; k++) nav->h_mc[(size_t) (n + i) * 9 + k] = nav->prm.birth_covariance[k];
		}
		*ncomp = n + nb;
	}
	else if (stage == PHD_STAGE_CORRECTED) {
		int ne = 0;
		HC(hipMemcpy(&ne, nav->d_emit_count + particle, 4, hipMemcpyDeviceToHost));
		size_t eb = (size_t) particle * nav->ecap;
		std::vector<double> rec((size_t) std::max(ne, 1) * MIX_REC);
		nav->h_mw.resize(std::max(ne, 1)); nav->h_mm.resize((size_t) std::max(ne, 1) * 3); nav->h_mc.resize((size_t) std::max(ne, 1) * 9);
		if (ne > 0) {
			HC(hipMemcpy(nav->h_mw.data(), nav->d_emit_w + eb, (size_t) ne * 8, hipMemcpyDeviceToHost));
			HC(hipMemcpy(rec.data(), nav->d_emit_rec + eb * MIX_REC, (size_t) ne * MIX_REC * 8, hipMemcpyDeviceToHost));
		}
		for (int c = 0; c < ne; c++) {
			const double* r = &rec[(size_t) c * MIX_REC + 1];   // (mean and covariance behind the record's weight)
			for (int k = 0; k < 3; k++) nav->h_mm[(size_t) c * 3 + k] = r[k];
			double* C = &nav->h_mc[(size_t) c * 9];
			C[0] = r[3]; C[1] = r[4]; C[2] = r[5]; C[3] = r[4]; C[4] = r[6]; C[5] = r[7]; C[6] = r[5]; C[7] = r[7]; C[8] = r[8];
		}
		*ncomp = ne;
	}
	else {
		return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_stage_map: unknown stage");
	}
	if (rc) return rc;
	if (w) *w = nav->h_mw.data();
	if (mean3) *mean3 = nav->h_mm.data();
	if (cov9) *cov9 = nav->h_mc.data();
	return PHD_OK;
}

const double* phd_stage_alpha(phd_navigator* nav, int* length)
{
	if (!nav || !nav->stage_valid) return nullptr;
	nav->h_alpha.resize(std::max(nav->P, 1));
	if (hipMemcpy(nav->h_alpha.data(), nav->d_alpha, (size_t) nav->P * 8, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
	if (length) *length = nav->P;
	return nav->h_alpha.data();
}

const double* phd_stage_setloglik(phd_navigator* nav, int* length)
{
	if (!nav || !nav->stage_valid) return nullptr;
	nav->h_setll.resize(std::max(nav->P, 1));
	if (hipMemcpy(nav->h_setll.data(), nav->d_setll, (size_t) nav->P * 8, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
	if (length) *length = nav->P;
	return nav->h_setll.data();
}

// the gathered weight vector of all ranks (+ room for their status words behind it) and the global source vector
static int ensure_gw(phd_navigator* nav, int n, bool shared = false)
{
	if (n <= nav->gwcap) return PHD_OK;
	if (nav->gw_shared) return nav->fail(PHD_ERR_GENERIC, "the gathered-weight vector of a shard cannot grow (other shards hold its address)");
	HC(hipStreamSynchronize(nav->stream));
	hipFree(nav->d_gw);
	nav->d_gw = nullptr;
	hipFree(nav->d_plan);
	nav->d_plan = nullptr;
	// (a multi-device handle's shards store their weights into each other's vectors: fine-grained, as the receive buffers)
	if (!shared || getenv("PHD_COARSE_RECV") || hipExtMallocWithFlags((void**) &nav->d_gw, ((size_t) n + PHD_MAX_DEVICES) * 8, hipDeviceMallocFinegrained) != hipSuccess) {
		(void) hipGetLastError();
		nav->d_gw = nullptr;
		HC(hipMalloc((void**) &nav->d_gw, ((size_t) n + PHD_MAX_DEVICES) * 8));
	}
	HC(hipMalloc((void**) &nav->d_plan, (size_t) n * 4));
	nav->gwcap = n;
	return PHD_OK;
}

int phd_resample(phd_navigator* nav, const double* weights, int nparticles, double u_resample, int32_t* sources,
                 int32_t* best_particle)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) nav = multi_shard0(nav);
	if (nparticles < 1 || !weights || !sources) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_resample: bad arguments");
	enter(nav);
	HC(hipStreamSynchronize(nav->stream));
	// (buffers of its own: the gathered-weight vector of a sharded handle is known to other shards by address)
	double* d_w = nullptr;
	int* d_src2 = nullptr;
	HC(hipMalloc((void**) &d_w, (size_t) nparticles * 8));
	if (hipMalloc((void**) &d_src2, (size_t) nparticles * 4 + 8) != hipSuccess) { hipFree(d_w); return nav->fail(PHD_ERR_DEVICE, "phd_resample: out of device memory"); }
	hipError_t e = hipMemcpy(d_w, weights, (size_t) nparticles * 8, hipMemcpyHostToDevice);
	StepBufs b = make_bufs(nav);
	int rc = (e == hipSuccess) ? launch_normalise(nav, b, d_w, nparticles, u_resample, 1, 1, d_src2 + 2, d_src2) : PHD_OK;
	if (e == hipSuccess) e = hipStreamSynchronize(nav->stream);
	int info[2] = {0, 0};
	if (e == hipSuccess) e = hipMemcpy(sources, d_src2 + 2, (size_t) nparticles * 4, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(info, d_src2, 8, hipMemcpyDeviceToHost);
	hipFree(d_src2); hipFree(d_w);
	if (rc) return rc;
	if (e != hipSuccess) return nav->fail(PHD_ERR_DEVICE, hipGetErrorString(e));
	if (best_particle) *best_particle = info[0];
	return PHD_OK;
}

int phd_particle_depleted(phd_navigator* nav, const double* weights, int nparticles, uint8_t* depleted)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) nav = multi_shard0(nav);
	if (nparticles < 1 || !weights || !depleted) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_particle_depleted: bad arguments");
	enter(nav);
	HC(hipStreamSynchronize(nav->stream));
	double* d_w = nullptr;
	int* d_tmp = nullptr;
	HC(hipMalloc((void**) &d_w, (size_t) nparticles * 8));
	if (hipMalloc((void**) &d_tmp, (size_t) nparticles * 4 + 8) != hipSuccess) { hipFree(d_w); return nav->fail(PHD_ERR_DEVICE, "phd_particle_depleted: out of device memory"); }
	hipError_t e = hipMemcpy(d_w, weights, (size_t) nparticles * 8, hipMemcpyHostToDevice);
	StepBufs b = make_bufs(nav);
	int rc = (e == hipSuccess) ? launch_normalise(nav, b, d_w, nparticles, 0.5, 0, 1, d_tmp + 2, d_tmp) : PHD_OK;
	if (e == hipSuccess) e = hipStreamSynchronize(nav->stream);
	int info[2] = {0, 0};
	if (e == hipSuccess) e = hipMemcpy(info, d_tmp, 8, hipMemcpyDeviceToHost);
	hipFree(d_tmp); hipFree(d_w);
	if (rc) return rc;
	if (e != hipSuccess) return nav->fail(PHD_ERR_DEVICE, hipGetErrorString(e));
	*depleted = info[1] ? 1 : 0;
	return PHD_OK;
}

#ifdef PHD_STAMPS
// diagnostic build only: per-workgroup phase stamps of the last stamped kernel, [P][16] doubles
int phd_debug_stamps(phd_navigator* nav, double* out)
{
	hipStreamSynchronize(nav->stream);
	return hipMemcpy(out, nav->d_stamps, (size_t) nav->P * 16 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

void* phd_stream(phd_navigator* nav) { return (nav && !nav->multi) ? (void*) nav->stream : nullptr; }

int phd_timing_reset(phd_navigator* nav, uint8_t enabled)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) {   // the kernels of the first shard are the ones timed; the phases of its steps too (phd_multi_report)
		multi_timing_reset(nav, enabled);
		nav = multi_shard0(nav);
	}
	enter(nav);
	hipStreamSynchronize(nav->stream);
	nav->ntimers = 0;
	nav->timing = enabled != 0;
	nav->timing_period = enabled ? enabled : 1;   // 1: every step; n: every n-th step (sampling keeps the events' own cost small)
	nav->timing_step = 0;
	nav->timing_now = true;
	return PHD_OK;
}

int phd_last_timings(phd_navigator* nav, const char*** names, const double** ms)
{
	if (!nav) return 0;
	if (nav->multi) nav = multi_shard0(nav);
	enter(nav);
	hipStreamSynchronize(nav->stream);
	nav->tnames.clear();
	nav->tms.clear();
	nav->tcounts.clear();
	for (size_t r = 0; r < nav->ntimers; r++) {
		Timer& t = nav->timers[r];
		float f = 0;
		if (hipEventElapsedTime(&f, t.t0_from >= 0 ? nav->timers[t.t0_from].t1 : t.t0, t.t1) != hipSuccess) continue;
		size_t k = 0;
		for (; k < nav->tnames.size(); k++) {
			if (nav->tnames[k] == t.name) break;
		}
		if (k == nav->tnames.size()) { nav->tnames.push_back(t.name); nav->tms.push_back(0); nav->tcounts.push_back(0); }
		nav->tms[k] += (double) f;
		nav->tcounts[k]++;
	}
	for (size_t k = 0; k < nav->tms.size(); k++) nav->tms[k] /= std::max(nav->tcounts[k], 1);
	if (names) *names = nav->tnames.data();
	if (ms) *ms = nav->tms.data();
	return (int) nav->tnames.size();
}

int phd_last_timing_counts(phd_navigator* nav, const int** counts)
{
	if (!nav) return 0;
	if (nav->multi) nav = multi_shard0(nav);
	if (counts) *counts = nav->tcounts.data();
	return (int) nav->tcounts.size();
}

// ---- multi-GPU ----------------------------------------------------------------------------------
// Particles are sharded contiguously: rank r owns global slots [r * P, (r + 1) * P). One step is
//   phd_step_local_async                      predict / correct / prune / reweight of the shard; the weights are exported
//   <all-gather of phd_device_local_weights into phd_device_global_weights, RCCL, by the host>
//   phd_step_global_async                     normalise + BestParticle + depletion test + systematic resampling over ALL
//                                             particles, identical on every rank; then the migration plan, ON THE DEVICE
//   phd_migration_plan                        the host learns the 2 n split sizes of the exchange (nothing else crosses)
//   phd_migration_pack_async, <all-to-all of the send/recv buffers>, phd_migration_unpack_async
// A phd_create_multi handle plays the same kernels from one worker thread per shard, with peer stores in place of the
// collectives and no host wait anywhere (phd_multi.inc).

// Buffers of the sharded step, made once (their addresses go into other shards' tables): export buffer, plan arrays, pinned
// counts, send / receive buffers. With non-decreasing sources a rank receives at most one record per slot and sends at
// most Pcap + world - 1 (a source particle goes to every rank its run of slots meets; runs of successive sources share at
// most one rank).
static void free_sharded(phd_navigator* nav)
{
	hipFree(nav->d_lw); nav->d_lw = nullptr;
	hipFree(nav->d_dst_tab); nav->d_dst_tab = nullptr;
	hipFree(nav->plan.code); hipFree(nav->plan.fslot); hipFree(nav->plan.sendlist); hipFree(nav->plan.senddst); hipFree(nav->plan.counts);
	nav->plan = MigPlan{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
	if (nav->h_counts) hipHostFree(nav->h_counts);
	nav->h_counts = nullptr;
	hipFree(nav->d_mslot); nav->d_mslot = nullptr;
	hipFree(nav->d_plang); nav->d_plang = nullptr;
	hipFree(nav->d_send); nav->d_send = nullptr;
	hipFree(nav->d_recv); nav->d_recv = nullptr;
	hipFree(nav->d_recv_tab); nav->d_recv_tab = nullptr;
	nav->sharded_ready = false;
}

static int ensure_sharded(phd_navigator* nav, bool need_send = true)
{
	if (nav->sharded_ready) {
		if (need_send && !nav->d_send) {   // (a handle first used without a send buffer: push-only hosts never need one)
			const size_t rec = (size_t) 8 + (size_t) MIX_REC * nav->cap;
			enter(nav);
			HC(hipMalloc((void**) &nav->d_send, (size_t) nav->sendrecs * rec * 8));
		}
		return PHD_OK;
	}
	enter(nav);
	const size_t rec = (size_t) 8 + (size_t) MIX_REC * nav->cap;
	nav->plan.sendcap = nav->Pcap + PHD_MAX_DEVICES;
	nav->recvrecs = nav->Pcap;
	nav->sendrecs = nav->plan.sendcap;
	// (all or nothing: a failure frees what was made, so that the call can be repeated and nothing downstream ever sees a
	// half-made set — the flag below is what every user of these buffers asks)
	hipError_t e = hipSuccess;
	auto want = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
	want(hipMalloc((void**) &nav->d_lw, (size_t) (nav->Pcap + 1) * 8));
	want(hipMalloc((void**) &nav->d_dst_tab, PHD_MAX_DEVICES * sizeof(double*)));
	if (e == hipSuccess) want(hipMemcpy(nav->d_dst_tab, &nav->d_lw, sizeof(double*), hipMemcpyHostToDevice));
	want(hipMalloc((void**) &nav->d_recv_tab, PHD_MAX_DEVICES * sizeof(double*)));
	want(hipMalloc((void**) &nav->plan.code, (size_t) nav->Pcap * 4));
	want(hipMalloc((void**) &nav->plan.fslot, (size_t) nav->Pcap * 4));
	want(hipMalloc((void**) &nav->plan.sendlist, (size_t) nav->plan.sendcap * 4));
	want(hipMalloc((void**) &nav->plan.senddst, (size_t) nav->plan.sendcap * 8));
	want(hipMalloc((void**) &nav->plan.counts, (2 * PHD_MAX_DEVICES + 8) * 4));
	if (e == hipSuccess) want(hipMemset(nav->plan.counts, 0, (2 * PHD_MAX_DEVICES + 8) * 4));
	// (coherent: the plan kernel's system-scope stores must reach the polling host while the kernel runs, whatever the
	// runtime's default for mapped host memory is)
	want(hipHostMalloc((void**) &nav->h_counts, (2 * PHD_MAX_DEVICES + 8) * 4, hipHostMallocMapped | hipHostMallocCoherent));
	if (e == hipSuccess) std::memset(nav->h_counts, 0, (2 * PHD_MAX_DEVICES + 8) * 4);
	want(hipMalloc((void**) &nav->d_mslot, (size_t) nav->Pcap * 4));
	{   // k_plan_count / k_plan_lists: [2 sets][cnt 64 x 64 | used Pcap / 32 + 1 | bad 1] | wcg [1024] | lcg [Pcap / 64 + 1]
		const size_t set = (size_t) PHD_MAX_DEVICES * PHD_MAX_DEVICES + (size_t) (nav->Pcap + 31) / 32 + 2;
		const size_t words = 2 * set + 1024 + (size_t) nav->Pcap / 64 + 2;
		want(hipMalloc((void**) &nav->d_plang, words * 4));
		if (e == hipSuccess) want(hipMemset(nav->d_plang, 0, words * 4));
	}
	if (need_send) want(hipMalloc((void**) &nav->d_send, (size_t) nav->sendrecs * rec * 8));   // (a shard of a multi-device handle packs straight into its peers' receive buffers)
	// The receive buffer is written by OTHER devices (peer stores of a multi-device handle's shards, or of other ranks'
	// processes through IPC) and read here: fine-grained device memory, coherent between agents without cache maintenance —
	// what RCCL allocates for its own peer-to-peer buffers. (See the head of phd_multi.inc for the visibility argument.)
	if (e == hipSuccess) {
		// (+ PHD_MAX_DEVICES words behind the records: the landing flags, one per sending rank — k_post_landing)
		const size_t recv_bytes = ((size_t) nav->recvrecs * rec + PHD_MAX_DEVICES) * 8;
		nav->recv_finegrained = getenv("PHD_COARSE_RECV") == nullptr &&
		                        hipExtMallocWithFlags((void**) &nav->d_recv, recv_bytes, hipDeviceMallocFinegrained) == hipSuccess;
		if (!nav->recv_finegrained) {
			(void) hipGetLastError();
			nav->d_recv = nullptr;
			want(hipMalloc((void**) &nav->d_recv, recv_bytes));
		}
		if (e == hipSuccess) want(hipMemset(nav->d_recv + (size_t) nav->recvrecs * rec, 0, PHD_MAX_DEVICES * 8));
	}
	if (e != hipSuccess) {
		(void) hipGetLastError();
		free_sharded(nav);
		return nav->fail(PHD_ERR_DEVICE, std::string("buffers of the sharded step: ") + hipGetErrorString(e));
	}
	nav->sharded_ready = true;
	return PHD_OK;
}

// the local part of a sharded step: the per-particle kernels, then the un-normalised weights stored where the exchange
// reads them (the host's all-gather buffer, or every shard's gathered vector)
static int step_local(phd_navigator* nav, uint8_t onlymapping)
{
	if (nav->P < 1) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_step_local: no particles");
	enter(nav);
	int rc = ensure_sharded(nav);
	if (rc) return rc;
	nav->sharded_used = true;
	nav->timing_now = (nav->timing_step++ % nav->timing_period) == 0;
	StepBufs b = make_bufs(nav);
	rc = launch_map(nav, b, !onlymapping);
	if (rc) return rc;
	b.defer = nav->last_defer;
	timer_begin(nav, T_PW);
	// (per-rank host: the export buffer holds P + 1 doubles, the step's status word behind the weights)
	hipLaunchKernelGGL(k_push_weights, dim3((nav->P + 255) / 256), dim3(256), 0, nav->stream, b, (double* const*) nav->d_dst_tab, nav->ndst,
	                   nav->push_first, nav->gw_shared ? nav->push_flagslot : nav->P);
	timer_end(nav, T_PW);
	HC(hipGetLastError());
	return PHD_OK;
}

int phd_step_local_async(phd_navigator* nav, uint8_t onlymapping)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_step_local_async");
	return step_local(nav, onlymapping);
}

void* phd_device_local_weights(phd_navigator* nav)
{
	if (!nav || nav->multi) return nullptr;
	if (ensure_sharded(nav)) return nullptr;
	return nav->d_lw;   // filled by phd_step_local_async, on the handle's stream
}

void* phd_device_global_weights(phd_navigator* nav, int world_particles)
{
	if (!nav || nav->multi || world_particles < 1) return nullptr;
	enter(nav);
	if (ensure_gw(nav, world_particles)) return nullptr;
	return nav->d_gw;
}

// The migration plan from the global source vector: over a grid of workgroups where the vector is long enough to pay for them — its
// counting inside the grid resampling's last launch (`counted`), or as k_plan_count (phd_test_migration_plan: a vector from the
// host), then k_plan_lists —, by one workgroup otherwise (k_plan_migration).
static bool plan_on_grid(const phd_navigator* nav, int Pl, int n)
{
	const long long Pg = (long long) Pl * n;
	return nav->plan_grid_min > 0 && Pg >= nav->plan_grid_min && Pg <= PLAN_GRID_MAXSLOTS && (Pl & 63) == 0 && nav->d_plang != nullptr;
}

// the accumulators of this launch pair (and the set its second launch clears for the next one)
static PlanGrid plan_grid_next(phd_navigator* nav)
{
	const size_t set = (size_t) PHD_MAX_DEVICES * PHD_MAX_DEVICES + (size_t) (nav->Pcap + 31) / 32 + 2;
	int* cur = nav->d_plang + (size_t) nav->plan_par * set;
	int* nxt = nav->d_plang + (size_t) (nav->plan_par ^ 1) * set;
	nav->plan_par ^= 1;
	PlanGrid pg;
	pg.cnt = cur; pg.used = (unsigned int*) (cur + PHD_MAX_DEVICES * PHD_MAX_DEVICES); pg.bad = cur + set - 1;
	pg.cnt_next = nxt; pg.used_next = (unsigned int*) (nxt + PHD_MAX_DEVICES * PHD_MAX_DEVICES); pg.bad_next = nxt + set - 1;
	pg.wcg = nav->d_plang + 2 * set;
	pg.lcg = pg.wcg + 1024;
	return pg;
}

static int launch_plan(phd_navigator* nav, const StepBufs& b, const int* gsrc, const int* info, const int* lflags, const double* gflags,
                       int Pl, int n, int rank, int* hostcounts, int seq, const double* gw, const PlanGrid* pgp = nullptr, bool counted = false)
{
	if (plan_on_grid(nav, Pl, n)) {
		const PlanGrid pg = pgp ? *pgp : plan_grid_next(nav);
		const int G = (int) (((long long) Pl * n + 255) / 256);
		if (!counted) hipLaunchKernelGGL(k_plan_count, dim3(G), dim3(256), 0, nav->stream, gsrc, info, lflags, gflags, Pl, n, rank, pg);
		hipLaunchKernelGGL(k_plan_lists, dim3(G), dim3(256), 0, nav->stream, gsrc, info, lflags, gflags, Pl, n, rank, nav->plan, pg, hostcounts, seq, b, gw);
	}
	else {
		hipLaunchKernelGGL(k_plan_migration, dim3(1), dim3(1024), plan_lds_bytes(Pl, n), nav->stream, gsrc, info, lflags, gflags, Pl, n, rank, nav->plan,
		                   hostcounts, seq, b, gw);
	}
	HC(hipGetLastError());
	return PHD_OK;
}

// the global part: the resampling kernel on the gathered vector, this rank's weights back into its bank, the plan.
// onlymapping: OnlyMapping keeps the weights and never resamples (PHDNavigator.cs:330-336) — the same kernel with the
// weights taken as they are and resampling off. hostcounts: the plan also writes its counts to pinned host memory.
static int step_global(phd_navigator* nav, int rank, int world_size, double u, uint8_t onlymapping, bool hostcounts, bool from_graw = false)
{
	enter(nav);
	const int Pg = nav->P * world_size;
	nav->last_world_particles = Pg;
	nav->world = world_size; nav->rank = rank;
	int rc = ensure_sharded(nav, hostcounts);
	if (!rc) rc = ensure_gw(nav, Pg);
	if (rc) return rc;
	nav->plan_on_device = !hostcounts;
	StepBufs b = make_bufs(nav);
	// (from_graw: the all-gather landed as [rank][P + 1] (weights | status word): the weights go into the contiguous vector the global
	// kernel takes, the status words behind it — a flag raised on ANY rank then drops the step on every rank alike; done by the
	// resampling's own first launch, or by k_ungather in front of the one-workgroup kernel)
	if (from_graw) nav->d_gflags = nav->d_gw + Pg;
	else if (!nav->gw_shared) nav->d_gflags = nullptr;   // (the host-plan path gathers the weights only: every rank answers for its own flags)
	// (the plan over the grid: its accumulators are chosen here, its counting rides in the grid resampling's last launch when that runs)
	const bool pgrid = plan_on_grid(nav, nav->P, world_size);
	PlanGrid pg;
	std::memset(&pg, 0, sizeof pg);
	if (pgrid) pg = plan_grid_next(nav);
	bool counted = false;
	timer_begin(nav, T_NR);
	rc = launch_normalise(nav, b, nav->d_gw, Pg, u, onlymapping ? -1 : 0, onlymapping ? 1 : 0, nav->d_plan, nav->d_info, nullptr, nullptr,
	                      from_graw ? (const double*) nav->d_graw : nullptr, nav->P, world_size, pgrid ? &pg : nullptr, rank, nav->d_gflags, &counted);
	timer_end(nav, T_NR);
	if (rc) return rc;
	int* hc = nullptr;
	if (hostcounts) {
		HC(hipHostGetDevicePointer((void**) &hc, nav->h_counts, 0));
		nav->plan_seq++;
		nav->plan_waiting = true;
	}
	timer_begin(nav, T_PL);
	rc = launch_plan(nav, b, (const int*) nav->d_plan, (const int*) nav->d_info, (const int*) nav->d_flags, nav->d_gflags, nav->P, world_size, rank, hc,
	                 nav->plan_seq, (const double*) nav->d_gw, pgrid ? &pg : nullptr, counted);
	timer_end(nav, T_PL);
	return rc;
}

int phd_step_global_async(phd_navigator* nav, int rank, int world_size, double u_resample)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_step_global_async");
	if (world_size < 1 || world_size > PHD_MAX_DEVICES || rank < 0 || rank >= world_size) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_step_global: bad rank/world (at most 64 ranks)");
	return step_global(nav, rank, world_size, u_resample, 0, true);
}

// ---- the same step with NOTHING for the host to wait for (round 4): the migrating particles are pushed by the senders,
// straight into their places in the receivers' buffers, from the plan the device made. The host of rank r
//   once:      exchanges the ranks' phd_migration_ipc_export handles and opens them (phd_migration_ipc_open), or — shards in
//              one process — hands in the raw pointers (phd_migration_set_peers)
//   per step:  phd_step_local_async            local kernels; weights | status word into phd_device_local_weights [P + 1]
//              <all-gather of those P + 1 doubles into phd_device_gather_buffer, RCCL>
//              phd_step_global_device_async    un-gather, global resampling, plan — all on the device
//              phd_migration_push_async        k_pack_particles with the peers' buffers as destinations
//              <a one-word all-reduce on the stream: every rank's records have landed>
//              phd_migration_unpack_async      k_finish_sharded
// and never learns whether the step resampled, how many particles moved, or whether a flag dropped it, before phd_sync.
void* phd_device_gather_buffer(phd_navigator* nav, int world_size)
{
	if (!nav || nav->multi || world_size < 1 || world_size > PHD_MAX_DEVICES) return nullptr;
	enter(nav);
	const int need = world_size * (nav->Pcap + 1);
	if (need > nav->grawcap) {
		if (hipStreamSynchronize(nav->stream) != hipSuccess) return nullptr;
		hipFree(nav->d_graw);
		nav->d_graw = nullptr; nav->grawcap = 0;
		if (hipMalloc((void**) &nav->d_graw, (size_t) need * 8) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
		hipMemset(nav->d_graw, 0, (size_t) need * 8);
		nav->grawcap = need;
	}
	return nav->d_graw;
}

int phd_step_global_device_async(phd_navigator* nav, int rank, int world_size, double u_resample, uint8_t onlymapping)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_step_global_device_async");
	if (world_size < 1 || world_size > PHD_MAX_DEVICES || rank < 0 || rank >= world_size) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_step_global_device: bad rank/world (at most 64 ranks)");
	if (!nav->d_graw || nav->grawcap < world_size * (nav->P + 1)) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_step_global_device: no gather buffer (phd_device_gather_buffer(world_size) first)");
	if (!nav->peers_set && world_size > 1) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_step_global_device: the peers' receive buffers are not known (phd_migration_ipc_open / phd_migration_set_peers first)");
	nav->landing_seq++;
	return step_global(nav, rank, world_size, u_resample, onlymapping, false, true);
}

// How the receiver of a device-path step learns that the migrating particles have landed (phd_migration_push_async ->
// phd_migration_unpack_async):
//   0 (default)  the caller orders the two calls itself — a collective on the stream between them (the one-word all-reduce of
//                rounds 3 - 4), or one stream for all the handles of a process;
//   1            flags: the push ends with a store of the step's number into every peer's receive buffer, the unpack waits (on
//                the device, bounded) for the flags of the ranks it takes records from. No second collective per step. Needs
//                fine-grained receive buffers (phd_migration_recv_is_finegrained): with ordinary device memory a peer's store is
//                only visible at a kernel boundary, and the call is refused.
int phd_migration_set_landing(phd_navigator* nav, int flags)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_migration_set_landing");
	int rc = ensure_sharded(nav, false);
	if (rc) return rc;
	if (flags && !nav->recv_finegrained) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_migration_set_landing: the receive buffer is not fine-grained memory: order push and unpack with a collective instead");
	enter(nav);
	HC(hipStreamSynchronize(nav->stream));
	nav->landing_flags = flags ? 1 : 0;
	if (const char* e = getenv("PHD_LANDING_TIMEOUT_MS")) nav->landing_ticks = std::max(1LL, atoll(e)) * 100000LL;
	if (const char* e = getenv("PHD_LANDING_INLINE")) nav->landing_inline = atoi(e) != 0;
	return PHD_OK;
}

// the 64-byte hipIpcMemHandle_t of this rank's receive buffer, for the other ranks' processes to open
int phd_migration_ipc_export(phd_navigator* nav, void* handle64, int64_t* buffer_bytes)
{
	if (!nav || !handle64) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_migration_ipc_export");
	int rc = ensure_sharded(nav, false);
	if (rc) return rc;
	static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI hands out 64 bytes");
	hipIpcMemHandle_t h;
	HC(hipIpcGetMemHandle(&h, nav->d_recv));
	std::memcpy(handle64, &h, 64);
	if (buffer_bytes) *buffer_bytes = (int64_t) (((size_t) nav->recvrecs * ((size_t) 8 + (size_t) MIX_REC * nav->cap) + PHD_MAX_DEVICES) * 8);
	return PHD_OK;
}

// recv_buffers[world_size]: where every rank's receive buffer is, as THIS device addresses it (entry `rank` may be NULL:
// the handle's own). Shards in one process pass each other's phd_migration_recv_buffer (with peer access enabled).
int phd_migration_set_peers(phd_navigator* nav, void* const* recv_buffers, int rank, int world_size)
{
	if (!nav || !recv_buffers) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_migration_set_peers");
	if (world_size < 1 || world_size > PHD_MAX_DEVICES || rank < 0 || rank >= world_size) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_migration_set_peers: bad rank/world (at most 64 ranks)");
	int rc = ensure_sharded(nav, false);
	if (rc) return rc;
	std::vector<double*> tab(world_size);
	for (int t = 0; t < world_size; t++) {
		tab[t] = (t == rank) ? nav->d_recv : (double*) recv_buffers[t];
		if (!tab[t]) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_migration_set_peers: the receive buffer of rank " + std::to_string(t) + " is NULL");
	}
	enter(nav);
	HC(hipStreamSynchronize(nav->stream));
	HC(hipMemcpy(nav->d_recv_tab, tab.data(), world_size * sizeof(double*), hipMemcpyHostToDevice));
	nav->peers_set = true;
	nav->world = world_size; nav->rank = rank;
	return PHD_OK;
}

// handles[world_size][64]: every rank's phd_migration_ipc_export, in rank order (the entry of `rank` itself is not opened)
int phd_migration_ipc_open(phd_navigator* nav, const void* handles, int rank, int world_size)
{
	if (!nav || !handles) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_migration_ipc_open");
	if (world_size < 1 || world_size > PHD_MAX_DEVICES || rank < 0 || rank >= world_size) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_migration_ipc_open: bad rank/world (at most 64 ranks)");
	enter(nav);
	// Work in flight on the stream may still store into the mappings of an earlier call: it ends first. From here until
	// phd_migration_set_peers has filled the table again the handle has NO peers — a failure below leaves it that way
	// (phd_step_global_device_async / phd_migration_push_async then refuse), never with a table of closed mappings.
	HC(hipStreamSynchronize(nav->stream));
	nav->peers_set = false;
	for (void* q : nav->ipc_opened) hipIpcCloseMemHandle(q);
	nav->ipc_opened.clear();
	std::vector<void*> ptrs(world_size, nullptr);
	std::vector<void*> opened;
	for (int t = 0; t < world_size; t++) {
		if (t == rank) continue;
		hipIpcMemHandle_t h;
		std::memcpy(&h, (const char*) handles + (size_t) t * 64, 64);
		void* q = nullptr;
		const hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
		if (e != hipSuccess) {
			(void) hipGetLastError();
			for (void* o : opened) hipIpcCloseMemHandle(o);   // what this attempt opened
			return nav->fail(PHD_ERR_DEVICE, "phd_migration_ipc_open: the receive buffer of rank " + std::to_string(t) + " cannot be opened: " + hipGetErrorString(e) +
			                 " (the ranks' GPUs must reach each other peer to peer; HSA_ENABLE_IPC_MODE_LEGACY=0 on this pool)");
		}
		opened.push_back(q);
		ptrs[t] = q;
	}
	const int rc = phd_migration_set_peers(nav, ptrs.data(), rank, world_size);
	if (rc) {
		for (void* o : opened) hipIpcCloseMemHandle(o);
		return rc;
	}
	nav->ipc_opened = opened;
	return PHD_OK;
}

// 1: the receive buffer is fine-grained device memory (coherent for the peers storing into it); 0: an ordinary allocation
int phd_migration_recv_is_finegrained(phd_navigator* nav)
{
	if (!nav || nav->multi) return -1;
	if (ensure_sharded(nav, false)) return -1;
	return nav->recv_finegrained ? 1 : 0;
}

// pack, with every record stored straight into its place in its destination's receive buffer (the count and the places are
// the device plan's: a fixed grid strides over the records; nothing runs when the step did not resample or was dropped)
int phd_migration_push_async(phd_navigator* nav)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_migration_push_async");
	if (!nav->sharded_ready || !nav->peers_set) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_migration_push_async: no peers (phd_migration_ipc_open / phd_migration_set_peers, then phd_step_global_device_async)");
	enter(nav);
	if (nav->world <= 1) return PHD_OK;   // (one rank: nothing ever leaves it)
	StepBufs b = make_bufs(nav);
	timer_begin(nav, T_PK);
	hipLaunchKernelGGL(k_pack_particles, dim3(std::min(nav->plan.sendcap, 256)), dim3(256), 0, nav->stream, b, nav->plan, nav->world, (double*) nullptr,
	                   (double* const*) nav->d_recv_tab);
	if (nav->landing_flags && nav->world > 1) {
		const size_t rec = (size_t) 8 + (size_t) MIX_REC * nav->cap;
		hipLaunchKernelGGL(k_post_landing, dim3(1), dim3(64), 0, nav->stream, (double* const*) nav->d_recv_tab, nav->world, nav->rank,
		                   (size_t) nav->recvrecs * rec, nav->landing_seq);
	}
	timer_end(nav, T_PK);
	HC(hipGetLastError());
	return PHD_OK;
}

// Pure host logic (no handle, no device): from the global source vector of a resampling step, which of
// this rank's particles go where and where each of its slots comes from. Both sides derive their lists
// from the same vector, so the sender's order per destination equals the receiver's order per source.
//   send_list : local indices to pack, grouped by destination rank (ascending), then by destination slot
//   dst_code  : per local slot: >= 0 local source index, < 0 -(k + 1) = record k of the receive buffer
// Returns the number of records received.
int phd_plan_migration(const int32_t* gsrc, int Pl, int world_size, int rank, int32_t* send_counts, int32_t* recv_counts,
                       int32_t* send_list, int32_t* dst_code)
{
	// A source particle travels to a rank ONCE, however many of that rank's slots take it: systematic resampling hands a
	// heavy particle to a run of consecutive slots, and the copies of a particle share its map anyway (slot indirection).
	// Sender and receiver both skip a slot whose source equals that of the slot before it (among the slots the two ranks
	// have in common), so their lists agree; a depleted particle set, which is when resampling happens, migrates a small
	// fraction of the records a copy per slot would.
	const int first = rank * Pl;
	for (int r = 0; r < world_size; r++) send_counts[r] = recv_counts[r] = 0;
	int ns = 0;
	for (int r = 0; r < world_size; r++) {
		if (r == rank) continue;
		int prev = -1;
		for (int g = r * Pl; g < (r + 1) * Pl; g++) {
			int s = gsrc[g];
			if (s >= first && s < first + Pl) {
				if (s != prev) {
					send_list[ns++] = s - first;
					send_counts[r]++;
				}
				prev = s;
			}
		}
	}
	for (int i = 0; i < Pl; i++) {
		int s = gsrc[first + i];
		if (s >= first && s < first + Pl) dst_code[i] = s - first;
	}
	int slot = 0;
	for (int r = 0; r < world_size; r++) {
		if (r == rank) continue;
		int prev = -1;
		for (int i = 0; i < Pl; i++) {
			int s = gsrc[first + i];
			if (s >= r * Pl && s < (r + 1) * Pl) {
				if (s != prev) {
					slot++;
					recv_counts[r]++;
				}
				dst_code[i] = -slot;   // record slot - 1 of the receive buffer
				prev = s;
			}
		}
	}
	return slot;
}

// Test surface of the device plan: k_plan_migration on a caller-supplied global source vector (the host statement above is
// its reference in the tests). Arrays as phd_plan_migration's, plus fslot[<= Pl] (the OUT-bank slots of the arriving
// records) and send_dst[<= Pl + 64][2] (destination rank, record number in its receive buffer). *status <- MIG_*;
// returns PHD_OK when the kernel ran.
int phd_test_migration_plan(phd_navigator* nav, const int32_t* gsrc, int particles_per_rank, int world_size, int rank, int resampled,
                            int32_t* send_counts, int32_t* recv_counts, int32_t* send_list, int32_t* dst_code, int32_t* fslot,
                            int32_t* send_dst, int32_t* nsend, int32_t* nrecv, int32_t* status)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	if (nav->multi) nav = multi_shard0(nav);
	const int Pl = particles_per_rank, n = world_size;
	if (!gsrc || Pl < 1 || Pl > nav->Pcap || n < 1 || n > PHD_MAX_DEVICES || rank < 0 || rank >= n || !send_counts || !recv_counts || !send_list || !dst_code ||
	    !fslot || !send_dst || !nsend || !nrecv || !status) {
		return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_test_migration_plan: sizes out of range (particles_per_rank <= max_particles, world_size <= 64)");
	}
	enter(nav);
	int rc = ensure_sharded(nav);
	if (rc) return rc;
	HC(hipStreamSynchronize(nav->stream));
	int* d_g = nullptr;
	HC(hipMalloc((void**) &d_g, ((size_t) Pl * n + 4) * 4));
	int* d_i = d_g + (size_t) Pl * n;   // info[2], a clear status word
	const int info[3] = {0, resampled ? 1 : 0, 0};
	hipError_t e = hipMemcpy(d_g, gsrc, (size_t) Pl * n * 4, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(d_i, info, 12, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		rc = launch_plan(nav, make_bufs(nav), (const int*) d_g, (const int*) d_i, (const int*) (d_i + 2), (const double*) nullptr, Pl, n, rank, (int*) nullptr, 0,
		                 (const double*) nullptr);
		if (rc) { hipFree(d_g); return rc; }
		e = hipStreamSynchronize(nav->stream);
	}
	std::vector<int> c(2 * n + 4);
	if (e == hipSuccess) e = hipMemcpy(c.data(), nav->plan.counts, c.size() * 4, hipMemcpyDeviceToHost);
	hipFree(d_g);
	if (e != hipSuccess) return nav->fail(PHD_ERR_DEVICE, std::string("phd_test_migration_plan: ") + hipGetErrorString(e));
	for (int r = 0; r < n; r++) { send_counts[r] = c[r]; recv_counts[r] = c[n + r]; }
	*nsend = c[2 * n]; *nrecv = c[2 * n + 1]; *status = c[2 * n + 2];
	if (*status == MIG_OK && resampled) {
		std::vector<long long> sd((size_t) std::max(*nsend, 1));
		HC(hipMemcpy(dst_code, nav->plan.code, (size_t) Pl * 4, hipMemcpyDeviceToHost));
		if (*nrecv > 0) HC(hipMemcpy(fslot, nav->plan.fslot, (size_t) *nrecv * 4, hipMemcpyDeviceToHost));
		if (*nsend > 0) {
			HC(hipMemcpy(send_list, nav->plan.sendlist, (size_t) *nsend * 4, hipMemcpyDeviceToHost));
			HC(hipMemcpy(sd.data(), nav->plan.senddst, (size_t) *nsend * 8, hipMemcpyDeviceToHost));
			for (int k = 0; k < *nsend; k++) { send_dst[2 * k] = (int32_t) (sd[k] >> 32); send_dst[2 * k + 1] = (int32_t) (sd[k] & 0xffffffffll); }
		}
	}
	return PHD_OK;
}

// The split sizes of the exchange. The plan itself was made on the device behind the resampling kernel
// (k_plan_migration); its counts arrive in pinned host memory, and the host waits for the word written behind them — no
// stream synchronisation, no copy command, and nothing of the size of the particle set crosses.
int phd_migration_plan(phd_navigator* nav, int rank, int world_size, int32_t* send_counts, int32_t* recv_counts)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_migration_plan");
	if (world_size < 1 || rank < 0 || rank >= world_size || !send_counts || !recv_counts) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_migration_plan: bad arguments");
	if (!nav->plan_waiting || world_size != nav->world || rank != nav->rank) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_migration_plan: call phd_step_global_async(rank, world_size) first");
	enter(nav);
	const int n = world_size;
	volatile int* hc = nav->h_counts;
	const auto t0 = std::chrono::steady_clock::now();
	for (long spins = 0; hc[2 * n + 6] != nav->plan_seq; spins++) {
		__builtin_ia32_pause();
		if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
			// (a device that never gets there: report what the stream says rather than spin for ever)
			HC(hipStreamSynchronize(nav->stream));
			if (hc[2 * n + 6] != nav->plan_seq) return nav->fail(PHD_ERR_DEVICE, "phd_migration_plan: the plan kernel did not report");
		}
	}
	__atomic_thread_fence(__ATOMIC_ACQUIRE);
	nav->plan_waiting = false;
	const int status = hc[2 * n + 2];
	for (int r = 0; r < n; r++) { send_counts[r] = hc[r]; recv_counts[r] = hc[n + r]; }
	nav->nsend = hc[2 * n]; nav->nrecv = hc[2 * n + 1];
	if (status == MIG_DROPPED) {
		// a kernel of the local step raised a flag: the step is dropped before anything rotates (the caller must not go on to
		// pack / unpack — if it does, those kernels find the same status and leave the state alone; the state is the one
		// before phd_step_local_async)
		nav->h_flags = hc[2 * n + 5];
		hipMemsetAsync(nav->d_flags, 0, 4, nav->stream);
		return check_flags(nav);
	}
	if (status == MIG_BAD) return nav->fail(PHD_ERR_GENERIC, "phd_migration_plan: the gathered source vector is not a resampling result (were the weights of all ranks gathered?)");
	if (status == MIG_OVERFLOW) return nav->fail(PHD_ERR_CAPACITY, "phd_migration_plan: more migrating particles than the send list holds");
	nav->h_info[0] = hc[2 * n + 4]; nav->h_info[1] = hc[2 * n + 3];
	return PHD_OK;
}

// Did the last global step resample? As the host knows it from phd_migration_plan (per-rank host) — identical on every
// rank, so all of them may skip the exchange together when it did not. -1: not known (no plan waited for yet).
int phd_last_resampled(phd_navigator* nav)
{
	if (!nav || nav->multi || !nav->sharded_used || nav->plan_waiting) return -1;
	return nav->h_info[1] ? 1 : 0;
}

void* phd_migration_send_buffer(phd_navigator* nav, int64_t* bytes_per_particle)
{
	if (!nav || nav->multi || ensure_sharded(nav)) return nullptr;
	if (bytes_per_particle) *bytes_per_particle = (int64_t) ((size_t) 8 + (size_t) 10 * nav->cap) * 8;
	return nav->d_send;   // (a fixed address for the life of the handle)
}

void* phd_migration_recv_buffer(phd_navigator* nav) { return (nav && !nav->multi && !ensure_sharded(nav)) ? nav->d_recv : nullptr; }

// grid of k_pack_particles / its like: the records are counted on the device, a fixed grid strides over them
static int pack_grid(const phd_navigator* nav) { return std::min(nav->plan.sendcap, 1024); }

int phd_migration_pack_async(phd_navigator* nav)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_migration_pack_async");
	enter(nav);
	if (!nav->sharded_ready || nav->plan_on_device) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_migration_pack_async: no plan known to the host (phd_step_global_async, phd_migration_plan first)");
	if (nav->nsend == 0) return PHD_OK;   // (the per-rank host knows the counts)
	StepBufs b = make_bufs(nav);
	timer_begin(nav, T_PK);
	hipLaunchKernelGGL(k_pack_particles, dim3(std::min(nav->nsend, pack_grid(nav))), dim3(256), 0, nav->stream, b, nav->plan, nav->world, nav->d_send,
	                   (double* const*) nullptr);
	timer_end(nav, T_PK);
	HC(hipGetLastError());
	return PHD_OK;
}

// the end of a sharded step: arrivals unpacked, small arrays gathered, roles rotated — all read from the device plan
static int step_finish(phd_navigator* nav)
{
	enter(nav);
	StepBufs b = make_bufs(nav);
	int* sel_next = nav->d_sel + (nav->parity ^ 1) * SEL_STRIDE;
	nav->d_res_slots = nav->d_mslot;
	timer_begin(nav, T_GR);
	const unsigned long long* landing = nullptr;
	if (nav->landing_flags && nav->plan_on_device && nav->world > 1) {
		const size_t rec = (size_t) 8 + (size_t) MIX_REC * nav->cap;
		landing = (const unsigned long long*) (nav->d_recv + (size_t) nav->recvrecs * rec);
		if (!nav->landing_inline) {   // one wave waits, in front of the launch that reads (k_wait_landing)
			hipLaunchKernelGGL(k_wait_landing, dim3(1), dim3(64), 0, nav->stream, nav->plan, nav->world, landing, nav->landing_seq, nav->landing_ticks, nav->d_flags);
			landing = nullptr;
		}
	}
	hipLaunchKernelGGL(k_finish_sharded, dim3(nav->P), dim3(256), 0, nav->stream, b, nav->plan, nav->world, (const double*) nav->d_recv,
	                   1.0 / (double) nav->last_world_particles, sel_next, nav->frozen ? 1 : 0, nav->d_inslot, nav->d_mslot, landing, nav->landing_seq, nav->landing_ticks);
	timer_end(nav, T_GR);
	HC(hipGetLastError());
	nav->parity ^= 1;
	nav->stage_valid = false;
	nav->sel_host_valid = false;   // the rotation depends on the plan's status and the resampling flag, known on the device
	return PHD_OK;
}

int phd_migration_unpack_async(phd_navigator* nav)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_migration_unpack_async");
	if (!nav->sharded_ready) return nav->fail(PHD_ERR_BAD_ARGUMENT, "phd_migration_unpack_async: no plan (phd_step_global_async first)");
	return step_finish(nav);
}

// Lend the handle a stream of the host (e.g. the framework's current stream) so that the library's kernels and
// the host's collectives are ordered by the stream itself (NULL = the default stream); lend = 0 returns to the handle's own.
int phd_set_stream(phd_navigator* nav, void* stream, uint8_t lend)
{
	if (!nav) return PHD_ERR_BAD_ARGUMENT;
	MULTI_UNSUPPORTED(nav, "phd_set_stream");
	enter(nav);
	HC(hipStreamSynchronize(nav->stream));
	nav->stream = lend ? (hipStream_t) stream : nav->own_stream;   // a NULL lent stream is the legacy default stream
	return PHD_OK;
}

}  // extern "C"

#include "phd_multi.inc"
