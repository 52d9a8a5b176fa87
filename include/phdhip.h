/*
 * phdhip.h — C-ABI of libphdhip.so, the MI355X (gfx950) implementation of monorfs's
 * Rao-Blackwellized PHD-SLAM inner loop.
 *
 * The library sits behind the reference's solver interface
 *     abstract class Navigator<MeasurerT, PoseT, MeasurementT>
 *     (mono-rfs-lib/SLAM/Navigators/Navigator.cs:47-396)
 * and replaces the managed body of
 *     PHDNavigator.SlamUpdate (mono-rfs-lib/SLAM/Navigators/PHDNavigator.cs:323-362)
 * for the PRM3D model (Pose3D + PixelRangeMeasurement). The calling convention mirrors the
 * reference's only native solver binding, ISAM2Lib (ISAM2Navigator.cs:600-622 <-> isam2/isam2.cpp:46-365):
 *   - an opaque handle made by a `create` call and released by a `destroy` call;
 *   - inputs are caller-owned flat `double*` / `int*`, read only for the duration of the call;
 *   - outputs are library-owned host buffers returned as pointer + length, valid until the next
 *     call on the same handle (the C# side copies them out, ISAM2Navigator.cs:507-593);
 *   - booleans cross as one byte (UnmanagedType.U1, ISAM2Navigator.cs:609);
 *   - every call returns an int status: 0 ok, >0 a specific condition, -1 generic failure
 *     (isam2.cpp:317-335); no C++ exception ever crosses the boundary.
 *
 * Threading: one caller thread per handle, no re-entrancy (Simulation.Update drives the
 * navigator from one thread, Simulation.cs:636-673). The library uses HIP streams internally.
 *
 * Non-finite input: measurements, poses, weights, odometry, noise and map components that are NaN or infinite are rejected
 * with PHD_ERR_BAD_ARGUMENT by every call that takes them (the state is untouched). The reference would carry a NaN through
 * its weight sums (PHDNavigator.cs:886-890) and end with NaN particle weights; the device's pair loops count a NaN exponent
 * as exp(-800) = 0. With finite input the two agree, so the boundary keeps the other case out.
 *
 * Matrix conventions: row-major; a 3-D Gaussian component is (weight, mean[3], cov[9]);
 * a pose is (x, y, z, qw, qx, qy, qz) as in Pose3D (Pose3D.cs:142-162); a measurement is
 * (px, py, range) as in PixelRangeMeasurement.ToLinear (PixelRangeMeasurement.cs:96-99).
 */
#ifndef PHDHIP_H
#define PHDHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHD_API_VERSION 4   /* 2: phd_create_multi, phd_set_association_workspace; phd_migration_local_async gone
                               3: phd_multi_report; phd_device_local_weights is an export buffer; the migration plan is made on the device
                               4: the sharded step without a host wait (phd_step_global_device_async, phd_migration_push_async, the IPC
                                  calls); phd_device_local_weights holds P + 1 doubles; the device keeps one 80-byte record per component */

/* status codes */
#define PHD_OK                    0
#define PHD_ERR_GENERIC          -1
#define PHD_ERR_BAD_ARGUMENT      1   /* a size/pointer/flag is out of range                        */
#define PHD_ERR_CAPACITY          2   /* a mixture outgrew its slab (raise max_components/emit_cap); the step is dropped,
                                         the state is the one before it                              */
#define PHD_ERR_ASSOCIATION       3   /* a data-association cluster of more than 256 rows, or the clusters beyond 64 rows
                                         used up the association workspace; surfaces as Data["module"]="association"
                                         (Simulation.cs:666). The step is dropped, the state is the one before it.   */
#define PHD_ERR_DEVICE            4   /* HIP runtime error, see phd_last_error                       */
#define PHD_ERR_NO_DEVICE         5   /* no gfx950 device / HIP runtime unavailable                  */

/* dynamics model. PRM3D is the product. LINEAR2D is the toy model of the reference's unit tests (PHDNavigatorTest.cs):
 * it runs through the same kernels — a pose is (x, y, 0, 1, 0, 0, 0), a measurement (x, y, 0), still three doubles, R is
 * 2 x 2 row-major in the first four entries of `R` — so that those tests' vectors can be put to the device itself.
 * phd_update_motion is PRM3D only; the gradient of phd_quasi_set_loglik_grad has its first two entries for (x, y).  */
#define PHD_MODEL_LINEAR2D 0
#define PHD_MODEL_PRM3D    1

/* radius-gate metric of Map.Near / Map.Evaluate(x, r) (Map.cs:170-184, 210-220): the reference
 * delegates to Accord 3.0.2 KDTree.Nearest(point, radius), whose metric is not in the tree.   */
#define PHD_GATE_SQUARED_EUCLIDEAN 1  /* |x-m|^2 <= radius   (Accord 3.0.x default distance) */
#define PHD_GATE_EUCLIDEAN         0  /* |x-m|   <= radius                                   */
#define PHD_GATE_DISABLED          2  /* every component is "near" (PHDNavigatorTest.Correct) */

/*
 * Every configuration value the path reads. Field <- reference origin:
 */
typedef struct phd_params {
	int32_t model;                 /* Config.Model (Config.cs:47)                                      */
	int32_t zdim;                  /* measurement dimension: 3 for PRM3D, 2 for Linear2D               */
	double  measurer[7];           /* PRM3DMeasurer.ToLinear(): focal, rangemin, rangemax, filmX, filmY,
	                                  filmW, filmH (PRM3DMeasurer.cs:92-96); rangemin/max are float32
	                                  values widened to double (PRM3DMeasurer.cs:65,73).
	                                  Linear2D: measurer[0] = Range (Linear2DMeasurer.cs:56-59)        */
	double  R[9];                  /* MeasurementCovariance * MeasurementCovarianceMultiplier, zdim x zdim
	                                  row-major in the leading entries (SimulatedVehicle.cs:159-168)  */
	double  visibility_ramp[3];    /* Config.VisibilityRamp (Config.cs:257-259)                        */
	double  pd;                    /* Config.NavigatorPD (Config.cs:90)                                */
	double  clutter_density;       /* Config.NavigatorClutterDensity (Config.cs:91)                    */
	double  birth_covariance[9];   /* Config.BirthCovariance (Config.cs:77-79)                         */
	double  birth_weight;          /* Config.BirthWeight (Config.cs:80)                                */
	double  min_weight;            /* Config.MinWeight (Config.cs:81)                                  */
	double  min_effective_particle;/* Config.MinEffectiveParticle (Config.cs:82)                       */
	int32_t max_quantity;          /* Config.MaxQuantity (Config.cs:83)                                */
	int32_t gate_metric;           /* PHD_GATE_*                                                       */
	double  merge_threshold;       /* Config.MergeThreshold (Config.cs:84)                             */
	double  exploration_threshold; /* Config.ExplorationThreshold (Config.cs:85)                       */
	double  density_distance_threshold; /* Config.DensityDistanceThreshold (Config.cs:74)              */
	/* capacities of the device slabs (not reference values) */
	int32_t max_particles;         /* particles this handle (this GPU's shard) can hold                */
	int32_t max_components;        /* components per particle slab, >= max_quantity                    */
	int32_t max_measurements;      /* measurements per frame                                           */
	int32_t emit_capacity;         /* per-particle scratch for corrected-but-unpruned components;
	                                  0 = library default                                              */
} phd_params;

typedef struct phd_navigator phd_navigator;

/* Fill `p` with the PRM3D defaults of Config.SetPRM3DDefaults (Config.cs:238-263) and the PHD
 * constants of Config.cs:74-91; capacities are set from the three arguments.                   */
void phd_default_params(phd_params* p, int max_particles, int max_components, int max_measurements);

/* ≙ `new PHDNavigator(vehicle, particlecount, onlymapping)` (PHDNavigator.cs:192-208) and
 * ISAM2Lib.newnavigator (ISAM2Navigator.cs:603). `device` is the HIP ordinal. NULL on failure;
 * phd_create_error() then holds the reason.                                                     */
phd_navigator* phd_create(const phd_params* params, int device);
/* The same over several devices of one node (SURVEY §8b "device list", §8e): ONE handle, ONE caller thread — what a C# host
 * can drive (Simulation.Update runs the solver from one thread, Simulation.cs:636-673; it has no RCCL). The particles are
 * sharded contiguously, ndevices equal shards: params->max_particles and every particle count are TOTALS and multiples of
 * ndevices. Inside, one worker thread per shard issues that shard's launches; the shards' kernels run concurrently, the
 * un-normalised weights (16 KB per pair at 2048 particles per GPU) and the migrating particles are stored by the kernels
 * straight into the other shards' memory over xGMI (peer stores into fine-grained buffers: no copy engine, no host call per
 * pair), the normalisation / BestParticle / systematic resampling and the migration plan are computed identically on every
 * device: results are bit-identical to a single-device handle holding all particles, for any ndevices. Every pair of
 * distinct devices must have direct peer access (the call fails, naming the pair, otherwise). A device may be listed more
 * than once (tests; the distinct-device path has not been run on hardware by the builder: one-GPU boxes).
 * Available on such a handle: phd_reset, phd_set_poses / _weights / _map, phd_update_motion, phd_slam_update,
 * phd_set_measurements / phd_step_async / phd_sync — phd_step_async only POSTS the step to the workers and returns (a ring
 * of 256 posted steps; nothing of a step waits for the host, what it finds surfaces at phd_sync) —, the getters, phd_upload /
 * download_state_soa, phd_set_frozen / _split / _all_pairs / _association_workspace, phd_quasi_set_loglik[_grad], phd_resample /
 * phd_particle_depleted, the timing calls (first shard); the stage-level KAT entry points and the per-rank sharding primitives
 * below return PHD_ERR_BAD_ARGUMENT.                                                                                     */
phd_navigator* phd_create_multi(const phd_params* params, const int* devices, int ndevices);
/* Diagnostics of a multi-device handle (bench.py --single-process). out8 (NINE doubles): [0..4] mean device time (ms) of the phases of the
 * sampled steps on the first shard's stream — local step | waiting for the other shards' weights | global resampling + plan
 * | pack (peer stores of the migrating particles) | waiting for the other shards' records + unpack; [5] mean time the caller
 * spent inside phd_step_async (us); [6] mean time the slowest shard's worker spent issuing one step (us); [7] sampled steps
 * (phd_timing_reset(nav, n) samples every n-th step); [8] as [6] without the waits for the other workers' event records. p2p (may be NULL): [nshards][nshards] bytes, 1 where shard s stores
 * into shard t's memory directly (peer access or the same device; phd_create_multi fails when a pair cannot).        */
int            phd_multi_report(phd_navigator* nav, double* out8, uint8_t* p2p, int* nshards);
const char*    phd_create_error(void);
/* ≙ Navigator.Dispose (Navigator.cs:395) / ISAM2Lib.deletenavigator.                             */
void           phd_destroy(phd_navigator* nav);
const char*    phd_last_error(const phd_navigator* nav);
int            phd_api_version(void);

/* ≙ PHDNavigator.reset (PHDNavigator.cs:245-266): `nparticles` equal particles at `pose`, each
 * with a deep copy of the `ncomp`-component map and weight 1/nparticles; BestParticle = 0.
 * Also serves CollapseParticles (:233-236) and ResetMapModel (:271-276, ncomp = 0).              */
int phd_reset(phd_navigator* nav, int nparticles, const double* pose7,
              const double* w, const double* mean3, const double* cov9, int ncomp);

/* The host keeps the motion model and its RNG (TrackVehicle.UpdateNoisy, TrackVehicle.cs:89-102):
 * after Navigator.Update (PHDNavigator.cs:295-314) it hands the particle poses over.
 * Like phd_set_weights, phd_update_motion and phd_set_measurements this does NOT wait for the device: the caller's array is
 * copied into pinned staging (free again when the call returns), the copy to the device and the store into the current
 * state are enqueued on the handle's stream — correct right behind phd_step_async, because the bank holding the current
 * state is looked up on the device.                                                              */
int phd_set_poses(phd_navigator* nav, const double* poses7, int nparticles);
/* SURVEY row f1 (next to the path): the particle motion step itself on the device. TrackVehicle.UpdateNoisy
 * (TrackVehicle.cs:89-102) = Pose3D.AddOdometry (Pose3D.cs:314-333) of the reading (dx dy dz dpitch dyaw droll),
 * then of the particle's own noise vector noise6[i*6..] = dt * chol(MotionCovariance) * N(0, I) drawn by the host
 * (Util.cs:173-202; NULL: none). perfect_still: a zero reading skips the noise (SimulatedVehicle.cs:190-202).
 * Replaces the per-frame phd_set_poses upload.                                                   */
int phd_update_motion(phd_navigator* nav, const double* odometry6, const double* noise6, int nparticles,
                      uint8_t perfect_still);
/* Test/bench access to the public arrays VehicleWeights / MapModels (PHDNavigator.cs:128,134). A resampling step
 * leaves the copies of a particle sharing their source's map (the deep copies of :740-741 are an index on the device):
 * phd_set_map therefore first gathers every map into its own place, one pass over the state (phd_map does not).   */
int phd_set_weights(phd_navigator* nav, const double* weights, int nparticles);
int phd_set_map(phd_navigator* nav, int particle, const double* w, const double* mean3,
                const double* cov9, int ncomp);

/* ≙ PHDNavigator.SlamUpdate (PHDNavigator.cs:323-362): predict, correct, prune and (unless
 * `onlymapping`) reweight every particle, then normalise, pick BestParticle and resample if
 * depleted. `u_resample` replaces `(double) Util.Uniform.Next()` of ResampleParticles
 * (PHDNavigator.cs:727): the RNG stays on the host. Blocking.                                    */
int phd_slam_update(phd_navigator* nav, const double* z3, int nmeasurements,
                    uint8_t onlymapping, double u_resample);

/* The same step split so that measurements can be resident before a timed region, or so that a 30 Hz host never
 * waits inside a frame: upload (asynchronous), enqueue (asynchronous on the handle's stream), wait + collect status.
 * Steps may be queued back to back; a failed one is dropped as a whole together with those queued behind it, and
 * phd_sync reports it. Up to 512 particles (environment PHD_CHAIN_MAX) the per-particle part of a step is one kernel
 * launch (up to 256 — PHD_DSPLIT_MAX, 0: never — with a second workgroup per particle that runs WeightAlpha's density sums beside
 * the association: the same results bit for bit). From 1024 particles on a step runs on two streams of the handle's own; phd_step_async directly behind
 * phd_step_async (no other call on the handle between them) is the fast path: the streams are not forked and joined around
 * each step, the end of a step runs on the stream that finishes last (results are bit-identical either way; any other call
 * on the handle, and a stream lent with phd_set_stream, take the fork / join order — INTEGRATION.md, "Posting steps back to
 * back").                                                                                         */
int phd_set_measurements(phd_navigator* nav, const double* z3, int nmeasurements);
int phd_step_async(phd_navigator* nav, uint8_t onlymapping, double u_resample);
int phd_sync(phd_navigator* nav);
/* Workspace of the set log-likelihood for data-association clusters of 65 .. 256 rows (MurtyPairing, GraphCombinatorics.cs:
 * 241-272, has no size limit; clusters of up to 64 rows work in a per-particle block and never fail): one slab per handle,
 * 128 MiB by default, about 0.4 MB per cluster of 128 rows and 1.3 MB per cluster of 256 in one step. When a step runs out
 * of it (PHD_ERR_ASSOCIATION) the state is kept: raise it here and run the step again.                                */
int phd_set_association_workspace(phd_navigator* nav, int64_t bytes);
/* Benchmark aid: when frozen, a step reads the current state but does not replace it, so every
 * step sees identical input sizes (SURVEY §8d "steady state").                                   */
int phd_set_frozen(phd_navigator* nav, uint8_t frozen);
/* Benchmark mode of SURVEY §8d: every (component, measurement) pair of the correct step is evaluated and the radius gate
 * of Map.Near (PHDNavigator.cs:882) only masks, so that the unit count P x C x M is exact. Off (the default) a group of 64
 * pairs that all lie outside the gate is skipped, as the reference never evaluates them; the results are bit-identical.  */
int phd_set_all_pairs(phd_navigator* nav, uint8_t all_pairs);
/* Scheduling knob (1..4; 0 = chosen from the particle count, the default; environment PHD_SPLIT at phd_create): the per-particle kernels of a
 * step are launched as `nsplit` particle sub-ranges on concurrent streams, forked from and joined into the
 * handle's stream. Results do not depend on it.                                                  */
int phd_set_split(phd_navigator* nav, int nsplit);

/* Getters. Buffers are library-owned and valid until the next call on the handle.               */
const double* phd_weights(phd_navigator* nav, int* length);               /* VehicleWeights       */
int           phd_best_particle(phd_navigator* nav);                       /* BestParticle         */
const double* phd_poses(phd_navigator* nav, int* length);                  /* 7 per particle       */
int           phd_particle_count(phd_navigator* nav);
/* ≙ MapModels[particle] (PHDNavigator.cs:134) / BestMapModel (:155-161): n components,
 * w[n], mean[3n], cov[9n] row-major like isam2.cpp:95-119.                                       */
int phd_map(phd_navigator* nav, int particle, int* ncomp,
            const double** w, const double** mean3, const double** cov9);
/* Source slot of every particle in the last step (identity when no resampling happened); the
 * host applies it to its own per-particle objects (trajectories). `resampled` <- 0/1.           */
const int32_t* phd_resample_sources(phd_navigator* nav, int* length, uint8_t* resampled);

/* Stage-level entry points for the unit KATs (the stages behind PHDNavigator's public methods).
 *   phd_stage_run(z, with_alpha)  runs predict -> correct -> prune (-> alpha) on EVERY particle of the handle's state, without
 *                                 the weight normalisation / resampling, and leaves each stage's result on the device:
 *   phd_stage_map(stage, particle) reads one particle's mixture after that stage (phd_map's conventions):
 *       PHD_STAGE_PREDICTED  ≙ PredictConditional (PHDNavigator.cs:793-819): state map -> predicted (prior + births)
 *       PHD_STAGE_CORRECTED  ≙ CorrectConditional (:829-906) fused with the MinWeight cut of PruneModel, unsorted
 *       PHD_STAGE_PRUNED     ≙ PruneModel (:913-948)
 *   phd_stage_alpha()        ≙ WeightAlpha (:373-393), one value per particle
 *   phd_stage_setloglik()    ≙ SetLogLikelihood (:462-515), one value per particle                                        */
#define PHD_STAGE_PREDICTED 0
#define PHD_STAGE_CORRECTED 1   /* corrected components with weight >= MinWeight, unsorted */
#define PHD_STAGE_PRUNED    2
int phd_stage_run(phd_navigator* nav, const double* z3, int nmeasurements, uint8_t with_alpha);
int phd_stage_map(phd_navigator* nav, int stage, int particle, int* ncomp,
                  const double** w, const double** mean3, const double** cov9);
const double* phd_stage_alpha(phd_navigator* nav, int* length);      /* WeightAlpha per particle  */
const double* phd_stage_setloglik(phd_navigator* nav, int* length);  /* SetLogLikelihood          */

/* ≙ ResampleParticles (:724-760), ParticleDepleted (:768-777) on caller-supplied weights.        */
int phd_resample(phd_navigator* nav, const double* weights, int nparticles, double u_resample,
                 int32_t* sources, int32_t* best_particle);
int phd_particle_depleted(phd_navigator* nav, const double* weights, int nparticles, uint8_t* depleted);

/* ---- multi-GPU (one handle per GPU, particles sharded contiguously) ---------------
 * The step is split around the one exchange the path has (PHDNavigator.cs:343-358):
 *   phd_step_local_async : predict/correct/prune/reweight of the local shard;
 *   the host all-gathers the un-normalised local weights (RCCL over xGMI);
 *   phd_step_global_async: normalise, BestParticle, depletion test and systematic resampling on
 *                          the gathered vector (identical on every rank), local part of the copy;
 * Device pointers are exposed so the collective runs on device memory without staging.          */
int   phd_step_local_async(phd_navigator* nav, uint8_t onlymapping);
void* phd_device_local_weights(phd_navigator* nav);                 /* double[local particles + 1]: an export buffer at a fixed
                                                                       address, filled by phd_step_local_async on the handle's stream —
                                                                       the un-normalised weights and, behind them, the step's status word */
void* phd_device_global_weights(phd_navigator* nav, int world_particles); /* double[world]         */
int   phd_step_global_async(phd_navigator* nav, int rank, int world_size, double u_resample);
/* Particle migration after a global resample: packs the particles other ranks need into a
 * contiguous device buffer (send), and unpacks what arrived (recv). Counts are in particles.
 * The plan is made on the device by phd_step_global_async (at most 64 ranks); phd_migration_plan only waits for its 2 n
 * split sizes, which the plan kernel writes to pinned host memory (no stream synchronisation, no vector crosses).    */
int   phd_migration_plan(phd_navigator* nav, int rank, int world_size,
                         int32_t* send_counts, int32_t* recv_counts);
/* The plan itself, as a pure host function (needs no handle and no GPU): `gsrc[world * Pl]` is the
 * global source vector; send_list (<= Pl * (world - 1) entries) = local indices to pack, grouped by
 * destination rank; dst_code[Pl] = local source index, or -(k + 1) for record k of the receive
 * buffer. Returns the number of received records. A source particle is sent to a rank once for every run of that rank's
 * consecutive slots that take it (its copies share the record, as the copies of a local particle share its map).  */
int   phd_plan_migration(const int32_t* gsrc, int particles_per_rank, int world_size, int rank,
                         int32_t* send_counts, int32_t* recv_counts, int32_t* send_list, int32_t* dst_code);
/* Test surface: the plan as the DEVICE makes it inside a step (k_plan_migration; phd_plan_migration is its host statement and
 * its reference in the tests), on a caller-supplied non-decreasing global source vector. Arrays as phd_plan_migration's, plus
 * fslot[<= particles_per_rank] (the slot each arriving record is unpacked into) and send_dst[<= particles_per_rank + 64][2]
 * (destination rank, record number in that rank's receive buffer). *status: 0 ok, 1 dropped, 2 not a resampling result
 * (decreasing / out of range), 3 send list overflow.                                                                   */
int   phd_test_migration_plan(phd_navigator* nav, const int32_t* gsrc, int particles_per_rank, int world_size, int rank, int resampled,
                              int32_t* send_counts, int32_t* recv_counts, int32_t* send_list, int32_t* dst_code, int32_t* fslot,
                              int32_t* send_dst, int32_t* nsend, int32_t* nrecv, int32_t* status);
/* 1 / 0: the last global step did / did not resample, as phd_migration_plan learnt it (the same on every rank: when it did
 * not, all ranks may skip pack and the all-to-all together; phd_migration_unpack_async still ends the step); -1: not known. */
int   phd_last_resampled(phd_navigator* nav);
void* phd_migration_send_buffer(phd_navigator* nav, int64_t* bytes_per_particle);
void* phd_migration_recv_buffer(phd_navigator* nav);
int   phd_migration_pack_async(phd_navigator* nav);
int   phd_migration_unpack_async(phd_navigator* nav);
/* The sharded step with NOTHING for the host to wait for (round 4; what `bench.py --gpus N` runs). The senders store each
 * migrating particle straight into its place in the receiver's buffer, from the plan the device made; RCCL carries exactly
 * what the north star names, the particle weights:
 *   once      every rank: phd_migration_ipc_export -> 64 bytes (hipIpcMemHandle_t of its receive buffer, fine-grained device
 *             memory); the hosts exchange them; phd_migration_ipc_open(all of them, in rank order). Shards that live in ONE
 *             process hand each other's phd_migration_recv_buffer to phd_migration_set_peers instead.
 *   per step  phd_step_local_async
 *             all-gather of the P + 1 doubles of phd_device_local_weights into phd_device_gather_buffer(world) [world][P + 1]
 *             phd_step_global_device_async   the status words of ALL ranks are read (a flag anywhere drops the step everywhere),
 *                                            global normalise / BestParticle / resampling, the migration plan — on the device
 *             phd_migration_push_async       peer stores of the migrating records
 *             "every rank's records have landed": with phd_migration_set_landing(1) (round 5) the push ends with a store of
 *                                            the step's number into a flag word of every peer's receive buffer and the unpack
 *                                            waits, on the device, for the words of the ranks it takes records from — ONE
 *                                            collective per step, the weights; with the default (0) the caller puts a
 *                                            collective on the same stream here (a one-word all-reduce, rounds 3 - 4)
 *             phd_migration_unpack_async
 * The host never learns whether the step resampled, how many particles moved, or whether a flag dropped it, before phd_sync
 * (which reports a step dropped here because ANOTHER rank raised a flag as PHD_ERR_GENERIC; that rank's own phd_sync names
 * the flag). Results are bit-identical to the host-plan sequence above and to a single handle holding all particles.   */
void* phd_device_gather_buffer(phd_navigator* nav, int world_size);   /* double[world_size][max_particles + 1]; rank r's P + 1 doubles at r * (P + 1) */
int   phd_step_global_device_async(phd_navigator* nav, int rank, int world_size, double u_resample, uint8_t onlymapping);
int   phd_migration_ipc_export(phd_navigator* nav, void* handle64, int64_t* buffer_bytes);
int   phd_migration_ipc_open(phd_navigator* nav, const void* handles /* [world_size][64] */, int rank, int world_size);
int   phd_migration_set_peers(phd_navigator* nav, void* const* recv_buffers /* [world_size], entry `rank` ignored */, int rank, int world_size);
int   phd_migration_recv_is_finegrained(phd_navigator* nav);         /* 1 / 0; -1: no buffer                                         */
int   phd_migration_push_async(phd_navigator* nav);
/* 1: landing flags (needs fine-grained receive buffers: refused with PHD_ERR_BAD_ARGUMENT otherwise); 0: the caller's collective.
 * A flag that does not arrive within 10 s (environment PHD_LANDING_TIMEOUT_MS, read by this call) ends the wait; the next phd_sync
 * reports PHD_ERR_GENERIC (a peer has died; the handle's state is then undefined: phd_reset / upload). The wait is one wave in a
 * launch of its own in front of the unpack kernel (environment PHD_LANDING_INLINE=1 at phd_create: inside that kernel's
 * workgroups instead — one launch less, but a waiting grid holds every slot of its device: never with ranks sharing a GPU).        */
int   phd_migration_set_landing(phd_navigator* nav, int flags);
void* phd_stream(phd_navigator* nav);                               /* hipStream_t of the handle   */
/* Lend the handle a host stream (hipStream_t, NULL = the default stream): kernels and the host's
 * collectives are then ordered by that stream and need no synchronisation in between;
 * lend = 0 returns to the handle's own stream.                                                   */
int   phd_set_stream(phd_navigator* nav, void* stream, uint8_t lend);

/* SURVEY row f4 (next to the path): PHDNavigator.QuasiSetLogLikelihood (PHDNavigator.cs:526-531) — the set
 * log-likelihood with everything fully visible (constant PD, gate 12) — for a BATCH of candidate poses against one
 * landmark set and one measurement set: the shape of the smoother's pose searches (LoopyPHDNavigator.cs:777-909).
 * poses7[nposes][7], landmarks3[nlandmarks][3], z3[nmeasurements][3]; out[nposes]. nposes <= max_particles,
 * nlandmarks <= min(1024, max_quantity). Synchronous; does not touch the particle state.                         */
int phd_quasi_set_loglik(phd_navigator* nav, const double* poses7, int nposes, const double* landmarks3, int nlandmarks,
                         const double* z3, int nmeasurements, double* out);

/* The same with the pose gradient: QuasiSetLogLikelihood(measurements, map, pose, out gradient) (PHDNavigator.cs:543-548,
 * calcgradient branches of :556-713) — what LoopyPHDNavigator.LogLikeGradientAscent (:916-965) and LogLikeFitCovariance
 * (:976-1021) call. gradients6[nposes][6]: d/d(translation, rotation) as MeasurementJacobianP (PRM3DMeasurer.cs:185-211)
 * defines them. Every component is enumerated in the reference's order because TemperedAverage (MatrixExtensions.cs:
 * 400-440) rewrites the shared logcomp array in place; in this mode the value can differ from phd_quasi_set_loglik's
 * where a component of more than 5 rows meets those rewritten entries in the cut of :672 — as in the reference.
 * average_mode 0: TemperedAverage as its source reads (weights.Normalize() = division by the Euclidean norm of the whole
 * 200-entry array, Accord.Math 3.0.2, not in the reference tree); 1: weights divided by their sum (the only one of the
 * two under which the reference's own LoopyPHDNavigatorTest.LogLike2D assertion holds; tests/test_oracle_kat.py).     */
int phd_quasi_set_loglik_grad(phd_navigator* nav, const double* poses7, int nposes, const double* landmarks3, int nlandmarks,
                              const double* z3, int nmeasurements, int average_mode, double* out, double* gradients6);

/* Test surface for the assignment enumerators the set log-likelihood runs on the device: MurtyPairing (mode 0,
 * GraphCombinatorics.cs:241-272; n <= 256) or LexicographicalPairing(matrix, modelsize) (mode 1, :280-334; n <= 5) on a
 * dense n x n profit matrix (row-major, -inf = no entry). assignments[k][n] (row -> column; -1: an unsolved first node),
 * values[k] for the first min(count, maxcount) pairings in the order they are produced; *count = how many there are
 * (at most 200, the length of the reference's logcomp). The vectors of GraphCombinatoricsTest.cs go through this.  */
int phd_test_pairing(phd_navigator* nav, const double* matrix, int n, int mode, int modelsize, int maxcount,
                     int32_t* assignments, double* values, int* count);

/* Per-kernel device time in milliseconds, from HIP events recorded around every launch on the
 * handle's stream: the mean over the launches since the last phd_timing_reset; names[i] -> ms[i];
 * returns the number of entries. phd_timing_reset(nav, 0) switches the events off; (nav, n) times every n-th
 * step only (an event costs the device a few microseconds).                                      */
int phd_timing_reset(phd_navigator* nav, uint8_t enabled);
int phd_last_timings(phd_navigator* nav, const char*** names, const double** ms);
/* Launches behind each mean of the last phd_last_timings call, same order (a split step launches a kernel
 * once per particle sub-range).                                                                  */
int phd_last_timing_counts(phd_navigator* nav, const int** counts);

/* Bulk upload / download of the whole particle set (benchmark and tests), host arrays plane per field:
 * planes[10][nparticles][stride] = w, mean x y z, covariance xx xy xz yy yz zz; counts[nparticles];
 * poses7[nparticles][7]; weights[nparticles]. (The device keeps ONE 80-byte record per component — w, mean, covariance
 * upper triangle —, [particle][slot][10]: the conversion happens here, at the edge; only the first counts[i] slots of a
 * particle are read / written.) The upload sets the particle count; the download gathers the maps of a resampled state
 * into place first (see phd_set_map).                                                            */
int phd_upload_state_soa(phd_navigator* nav, int nparticles, int stride, const double* planes,
                         const int32_t* counts, const double* poses7, const double* weights);
int phd_download_state_soa(phd_navigator* nav, int stride, double* planes, int32_t* counts,
                           double* poses7, double* weights);

#ifdef __cplusplus
}
#endif
#endif /* PHDHIP_H */
