// HipPHDNavigator.cs — the reference-side binding a monorfs maintainer would add next to
// mono-rfs-lib/SLAM/Navigators/PHDNavigator.cs to run the PHD inner loop on an MI355X.
//
// NOT COMPILED HERE: this image has no mono/mcs/dotnet (SURVEY.md §8c). It follows, member by member,
// the one native binding the reference already ships (ISAM2Lib, ISAM2Navigator.cs:600-622):
// DllImport on a non-generic static class, HandleRef for the opaque navigator, `fixed` pinning of the
// caller-owned arrays for the duration of a call, library-owned result buffers copied out with
// Marshal.Copy, bool as UnmanagedType.U1, and a non-zero status turned into an
// InvalidOperationException carrying Data["module"] (ISAM2Navigator.cs:239-262), which
// Simulation.Update already catches (Simulation.cs:655-670).
//
// Wiring: add `case NavigationAlgorithm.HipPHD: navigator = new HipPHDNavigator(explorer, particlecount, onlymapping);`
// to the switch in Simulation.FromFiles (Simulation.cs:352-379) and the option value to Program.cs:119.
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;

using Microsoft.Xna.Framework;

namespace monorfs
{
[StructLayout(LayoutKind.Sequential)]
public unsafe struct PhdParams          // include/phdhip.h: struct phd_params
{
	public int model, zdim;
	public fixed double measurer[7];
	public fixed double R[9];
	public fixed double visibility_ramp[3];
	public double pd, clutter_density;
	public fixed double birth_covariance[9];
	public double birth_weight, min_weight, min_effective_particle;
	public int max_quantity, gate_metric;
	public double merge_threshold, exploration_threshold, density_distance_threshold;
	public int max_particles, max_components, max_measurements, emit_capacity;
}

public class PhdHipLib
{
	const string Lib = "libphdhip.so";
	[DllImport(Lib)] public extern static IntPtr phd_create(ref PhdParams p, int device);
	[DllImport(Lib)] public extern static IntPtr phd_create_multi(ref PhdParams p, int[] devices, int ndevices);
	[DllImport(Lib)] public extern static int    phd_multi_report(HandleRef nav, double[] out9, byte[] p2p, out int nshards);
	[DllImport(Lib)] public extern static IntPtr phd_create_error();
	[DllImport(Lib)] public extern static void   phd_destroy(HandleRef nav);
	[DllImport(Lib)] public extern static IntPtr phd_last_error(HandleRef nav);
	[DllImport(Lib)] public extern static int    phd_reset(HandleRef nav, int nparticles, IntPtr pose7, IntPtr w, IntPtr mean3, IntPtr cov9, int ncomp);
	[DllImport(Lib)] public extern static int    phd_set_poses(HandleRef nav, IntPtr poses7, int nparticles);
	[DllImport(Lib)] public extern static int    phd_update_motion(HandleRef nav, IntPtr odometry6, IntPtr noise6, int nparticles, [MarshalAs(UnmanagedType.U1)] bool perfectstill);
	[DllImport(Lib)] public extern static int    phd_quasi_set_loglik_grad(HandleRef nav, IntPtr poses7, int nposes, IntPtr landmarks3, int nlandmarks, IntPtr z3, int nmeasurements, int averagemode, IntPtr result, IntPtr gradients6);
	[DllImport(Lib)] public extern static int    phd_quasi_set_loglik(HandleRef nav, IntPtr poses7, int nposes, IntPtr landmarks3, int nlandmarks, IntPtr z3, int nmeasurements, IntPtr result);
	[DllImport(Lib)] public extern static int    phd_slam_update(HandleRef nav, IntPtr z3, int nmeasurements, [MarshalAs(UnmanagedType.U1)] bool onlymapping, double uresample);
	[DllImport(Lib)] public extern static IntPtr phd_weights(HandleRef nav, out int length);
	[DllImport(Lib)] public extern static int    phd_best_particle(HandleRef nav);
	[DllImport(Lib)] public extern static IntPtr phd_poses(HandleRef nav, out int length);
	[DllImport(Lib)] public extern static int    phd_map(HandleRef nav, int particle, out int ncomp, out IntPtr w, out IntPtr mean3, out IntPtr cov9);
	[DllImport(Lib)] public extern static IntPtr phd_resample_sources(HandleRef nav, out int length, [MarshalAs(UnmanagedType.U1)] out bool resampled);
}

/// <summary>
/// PHD SLAM solver running PHDNavigator.SlamUpdate on the GPU. The motion model, its random
/// generators and every per-particle object (TrackVehicle, trajectories) stay in managed code.
/// </summary>
public unsafe class HipPHDNavigator : Navigator<PRM3DMeasurer, Pose3D, PixelRangeMeasurement>
{
	HandleRef nav;
	public int ParticleCount { get; set; }
	public TrackVehicle<PRM3DMeasurer, Pose3D, PixelRangeMeasurement>[] VehicleParticles { get; private set; }
	public double[] VehicleWeights { get; private set; }
	public int BestParticle { get; private set; }

	/// <summary>gpus: how many GPUs of the node the particles are sharded over (devices 0 .. gpus - 1, one handle, this
	/// thread: phd_create_multi); particlecount must then be a multiple of it. 1, or a mapping-only navigator (one particle):
	/// a single-device handle.</summary>
	public HipPHDNavigator(Vehicle<PRM3DMeasurer, Pose3D, PixelRangeMeasurement> vehicle, int particlecount, bool onlymapping = false, int gpus = 1)
		: base(vehicle, onlymapping)
	{
		ParticleCount = particlecount;
		PhdParams p = new PhdParams();
		p.model = 1; p.zdim = 3;
		double[] m = vehicle.Measurer.ToLinear();                         // PRM3DMeasurer.cs:92-96
		for (int i = 0; i < 7; i++) p.measurer[i] = m[i];
		double[][] R = Config.MeasurementCovarianceMultiplier.Multiply(vehicle.MeasurementCovariance);
		for (int i = 0; i < 9; i++) { p.R[i] = R[i / 3][i % 3]; p.birth_covariance[i] = Config.BirthCovariance[i / 3][i % 3]; }
		for (int i = 0; i < 3; i++) p.visibility_ramp[i] = Config.VisibilityRamp[i];
		p.pd = Config.NavigatorPD; p.clutter_density = Config.NavigatorClutterDensity;
		p.birth_weight = Config.BirthWeight; p.min_weight = Config.MinWeight;
		p.min_effective_particle = Config.MinEffectiveParticle; p.max_quantity = Config.MaxQuantity;
		p.gate_metric = 1;                                                // Accord 3.0.x KDTree: squared Euclidean
		p.merge_threshold = Config.MergeThreshold; p.exploration_threshold = Config.ExplorationThreshold;
		p.density_distance_threshold = Config.DensityDistanceThreshold;
		p.max_particles = particlecount; p.max_components = Math.Max(Config.MaxQuantity, 640); p.max_measurements = 256;
		int[] devices = new int[Math.Max(gpus, 1)];
		for (int i = 0; i < devices.Length; i++) devices[i] = i;
		IntPtr h = (gpus > 1 && !onlymapping) ? PhdHipLib.phd_create_multi(ref p, devices, devices.Length) : PhdHipLib.phd_create(ref p, 0);
		if (h == IntPtr.Zero) { throw Fail(Marshal.PtrToStringAnsi(PhdHipLib.phd_create_error()), -1); }
		nav = new HandleRef(this, h);
		reset(RefVehicle, new double[0], new double[0], new double[0], onlymapping ? 1 : particlecount);
	}

	Exception Fail(string message, int status)
	{
		var e = new InvalidOperationException(message);
		e.Data["module"] = (status == 3) ? "association" : "phdhip";     // Simulation.cs:662-670
		return e;
	}

	void Check(int status)
	{
		if (status != 0) { throw Fail(Marshal.PtrToStringAnsi(PhdHipLib.phd_last_error(nav)), status); }
	}

	void reset(Vehicle<PRM3DMeasurer, Pose3D, PixelRangeMeasurement> vehicle, double[] w, double[] mean, double[] cov, int particlecount)
	{
		double[] pose = vehicle.Pose.State;
		fixed (double* pp = pose) fixed (double* pw = w) fixed (double* pm = mean) fixed (double* pc = cov) {
			Check(PhdHipLib.phd_reset(nav, particlecount, (IntPtr) pp, (IntPtr) pw, (IntPtr) pm, (IntPtr) pc, w.Length));
		}
		VehicleParticles = new TrackVehicle<PRM3DMeasurer, Pose3D, PixelRangeMeasurement>[particlecount];
		for (int i = 0; i < particlecount; i++) {
			VehicleParticles[i] = vehicle.TrackClone(Config.MotionCovarianceMultiplier, Config.MeasurementCovarianceMultiplier,
			                                         Config.NavigatorPD, Config.NavigatorClutterDensity, true);
		}
		VehicleWeights = new double[particlecount];
		for (int i = 0; i < particlecount; i++) { VehicleWeights[i] = 1.0 / particlecount; }
		BestParticle = 0;
	}

	public override TrackVehicle<PRM3DMeasurer, Pose3D, PixelRangeMeasurement> BestEstimate { get { return VehicleParticles[BestParticle]; } }

	public override Map BestMapModel
	{
		get {
			int n; IntPtr w, m, c;
			Check(PhdHipLib.phd_map(nav, BestParticle, out n, out w, out m, out c));
			double[] ws = new double[n], ms = new double[3 * n], cs = new double[9 * n];
			if (n > 0) { Marshal.Copy(w, ws, 0, n); Marshal.Copy(m, ms, 0, 3 * n); Marshal.Copy(c, cs, 0, 9 * n); }
			Map map = new Map(3);
			for (int i = 0; i < n; i++) {
				double[][] cov = { new double[] {cs[9*i], cs[9*i+1], cs[9*i+2]}, new double[] {cs[9*i+3], cs[9*i+4], cs[9*i+5]}, new double[] {cs[9*i+6], cs[9*i+7], cs[9*i+8]} };
				map.Add(new Gaussian(new double[] {ms[3*i], ms[3*i+1], ms[3*i+2]}, cov, ws[i]));
			}
			return map;
		}
	}

	public override void ResetMapModel()
	{
		reset(RefVehicle, new double[0], new double[0], new double[0], VehicleParticles.Length);
	}

	/// <summary>Motion update: stays managed (TrackVehicle.UpdateNoisy, PHDNavigator.cs:295-314), then the poses go to the device.</summary>
	public override void Update(GameTime time, double[] reading)
	{
		if (OnlyMapping) { VehicleParticles[0].Pose = RefVehicle.Pose.DClone(); }
		else { for (int i = 0; i < VehicleParticles.Length; i++) { VehicleParticles[i].UpdateNoisy(time, reading); } }
		double[] poses = new double[7 * VehicleParticles.Length];
		for (int i = 0; i < VehicleParticles.Length; i++) { VehicleParticles[i].Pose.State.CopyTo(poses, 7 * i); }
		fixed (double* pp = poses) { Check(PhdHipLib.phd_set_poses(nav, (IntPtr) pp, VehicleParticles.Length)); }
		UpdateTrajectory(time);
	}

	/// <summary>Alternative to Update with the motion step on the device (phd_update_motion): only the reading and the
	/// noise vectors drawn here cross the boundary; the managed particles are refreshed from phd_poses when needed.</summary>
	public void UpdateOnDevice(GameTime time, double[] reading)
	{
		int      n     = VehicleParticles.Length;
		double[] noise = new double[6 * n];
		double   dt    = time.ElapsedGameTime.TotalSeconds;
		for (int i = 0; i < n; i++) {
			double[] v = dt.Multiply(Util.RandomGaussianVector(new double[6], VehicleParticles[i].MotionCovariance));   // TrackVehicle.cs:95-97
			v.CopyTo(noise, 6 * i);
		}
		fixed (double* pr = reading) fixed (double* pn = noise) {
			Check(PhdHipLib.phd_update_motion(nav, (IntPtr) pr, OnlyMapping ? IntPtr.Zero : (IntPtr) pn, n, SimulatedVehicle<PRM3DMeasurer, Pose3D, PixelRangeMeasurement>.PerfectStill));
		}
		UpdateTrajectory(time);
	}

	/// <summary>≙ static PHDNavigator.QuasiSetLogLikelihood(measurements, map, pose) (PHDNavigator.cs:526-531) for a batch of
	/// candidate poses against one map estimate (the pose searches of LoopyPHDNavigator.cs:777-909).</summary>
	public double[] QuasiSetLogLikelihood(List<PixelRangeMeasurement> measurements, IMap map, Pose3D[] poses)
	{
		double[] z = new double[3 * measurements.Count], lm = new double[3 * map.Count], p7 = new double[7 * poses.Length];
		for (int i = 0; i < measurements.Count; i++) { measurements[i].ToLinear().CopyTo(z, 3 * i); }
		int j = 0;
		foreach (Gaussian landmark in map) { landmark.Mean.CopyTo(lm, 3 * j++); }
		for (int i = 0; i < poses.Length; i++) { poses[i].State.CopyTo(p7, 7 * i); }
		double[] result = new double[poses.Length];
		fixed (double* pz = z) fixed (double* pl = lm) fixed (double* pp = p7) fixed (double* pr = result) {
			Check(PhdHipLib.phd_quasi_set_loglik(nav, (IntPtr) pp, poses.Length, (IntPtr) pl, map.Count, (IntPtr) pz, measurements.Count, (IntPtr) pr));
		}
		return result;
	}

	/// <summary>≙ static PHDNavigator.QuasiSetLogLikelihood(measurements, map, pose, out gradient) (PHDNavigator.cs:543-548)
	/// for a batch of candidate poses: what LogLikeGradientAscent and LogLikeFitCovariance evaluate
	/// (LoopyPHDNavigator.cs:916-1021). averagemode 0 follows TemperedAverage as written, 1 divides its weights by their sum.</summary>
	public double[] QuasiSetLogLikelihood(List<PixelRangeMeasurement> measurements, IMap map, Pose3D[] poses, out double[][] gradients, int averagemode = 0)
	{
		double[] z = new double[3 * measurements.Count], lm = new double[3 * map.Count], p7 = new double[7 * poses.Length];
		for (int i = 0; i < measurements.Count; i++) { measurements[i].ToLinear().CopyTo(z, 3 * i); }
		int j = 0;
		foreach (Gaussian landmark in map) { landmark.Mean.CopyTo(lm, 3 * j++); }
		for (int i = 0; i < poses.Length; i++) { poses[i].State.CopyTo(p7, 7 * i); }
		double[] result = new double[poses.Length], g = new double[6 * poses.Length];
		fixed (double* pz = z) fixed (double* pl = lm) fixed (double* pp = p7) fixed (double* pr = result) fixed (double* pg = g) {
			Check(PhdHipLib.phd_quasi_set_loglik_grad(nav, (IntPtr) pp, poses.Length, (IntPtr) pl, map.Count, (IntPtr) pz, measurements.Count, averagemode, (IntPtr) pr, (IntPtr) pg));
		}
		gradients = new double[poses.Length][];
		for (int i = 0; i < poses.Length; i++) { gradients[i] = new double[6]; Array.Copy(g, 6 * i, gradients[i], 0, 6); }
		return result;
	}

	/// <summary>≙ PHDNavigator.SlamUpdate (PHDNavigator.cs:323-362).</summary>
	public override void SlamUpdate(GameTime time, List<PixelRangeMeasurement> measurements)
	{
		double[] z = new double[3 * measurements.Count];
		for (int i = 0; i < measurements.Count; i++) { measurements[i].ToLinear().CopyTo(z, 3 * i); }
		double u = (double) Util.Uniform.Next();                           // PHDNavigator.cs:727: the RNG stays managed
		fixed (double* pz = z) { Check(PhdHipLib.phd_slam_update(nav, (IntPtr) pz, measurements.Count, OnlyMapping, u)); }

		int n; bool resampled;
		Marshal.Copy(PhdHipLib.phd_weights(nav, out n), VehicleWeights, 0, VehicleWeights.Length);
		BestParticle = PhdHipLib.phd_best_particle(nav);
		IntPtr src = PhdHipLib.phd_resample_sources(nav, out n, out resampled);
		if (resampled) {                                                    // apply the device's choice to the managed particles
			int[] sources = new int[n];
			Marshal.Copy(src, sources, 0, n);
			var particles = new TrackVehicle<PRM3DMeasurer, Pose3D, PixelRangeMeasurement>[n];
			for (int i = 0; i < n; i++) { particles[i] = RefVehicle.TrackClone(VehicleParticles[sources[i]], true); }   // :740
			VehicleParticles = particles;
		}
		UpdateMapHistory(time);
	}

	protected override void StartSlamInternal()    { Collapse(ParticleCount); }
	protected override void StartMappingInternal() { Collapse(1); }

	void Collapse(int particlecount)                                       // CollapseParticles, PHDNavigator.cs:233-236
	{
		Map best = BestMapModel;
		List<Gaussian> l = best.ToList();
		double[] w = new double[l.Count], m = new double[3 * l.Count], c = new double[9 * l.Count];
		for (int i = 0; i < l.Count; i++) {
			w[i] = l[i].Weight; l[i].Mean.CopyTo(m, 3 * i);
			for (int k = 0; k < 9; k++) { c[9 * i + k] = l[i].Covariance[k / 3][k % 3]; }
		}
		reset(RefVehicle, w, m, c, particlecount);
	}

	public override void Dispose() { if (nav.Handle != IntPtr.Zero) { PhdHipLib.phd_destroy(nav); nav = new HandleRef(this, IntPtr.Zero); } }
}
}
