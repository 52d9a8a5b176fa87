// PHDNavigator.hpp — C++ host-side mirror of the reference's solver interface over libphdhip.so.
//
// The reference's host is C# (no .NET toolchain in this image), so the native host written here is
// C++: the class keeps the member names, argument meaning and error behaviour of
//     class PHDNavigator<PRM3DMeasurer, Pose3D, PixelRangeMeasurement> : Navigator<...>
//     (mono-rfs-lib/SLAM/Navigators/PHDNavigator.cs:52-983; Navigator.cs:47-396)
// and forwards each member to the C-ABI of include/phdhip.h. Nothing is computed here.
//
// Errors: a non-zero status becomes a PhdError whose `module` is "association" for
// PHD_ERR_ASSOCIATION — what Simulation.Update looks for in Data["module"] (Simulation.cs:655-670).
#pragma once
#include "../../include/phdhip.h"

#include <array>
#include <stdexcept>
#include <string>
#include <vector>

namespace monorfs {

struct PhdError : std::runtime_error {
	int         status;
	std::string module;
	PhdError(int status, const std::string& what)
		: std::runtime_error(what), status(status), module(status == PHD_ERR_ASSOCIATION ? "association" : "phdhip") {}
};

// one Gaussian component of a Map (BaseStructures/Gaussian.cs:49-59)
struct Gaussian {
	double weight;
	std::array<double, 3> mean;
	std::array<double, 9> covariance;   // row-major
};
typedef std::vector<Gaussian> Map;       // BaseStructures/Maps/Map.cs, canonical (insertion) order
typedef std::array<double, 7> Pose3D;    // x y z qw qx qy qz (Pose3D.cs:142-162)
typedef std::array<double, 3> PixelRangeMeasurement;   // px py range

class PHDNavigator {
public:
	// ≙ PHDNavigator(vehicle, particlecount, onlymapping) (PHDNavigator.cs:192-208)
	PHDNavigator(const phd_params& params, const Pose3D& pose, int particlecount, bool onlymapping = false, int device = 0)
		: ParticleCount(particlecount), OnlyMapping(onlymapping)
	{
		nav_ = phd_create(&params, device);
		if (!nav_) throw PhdError(PHD_ERR_NO_DEVICE, phd_create_error());
		reset(pose, Map(), onlymapping ? 1 : particlecount);   // :201-207
	}
	// the particles sharded over several GPUs of the node behind one handle (phd_create_multi): params.max_particles and
	// particlecount are totals, multiples of devices.size()
	PHDNavigator(const phd_params& params, const Pose3D& pose, int particlecount, const std::vector<int>& devices, bool onlymapping = false)
		: ParticleCount(particlecount), OnlyMapping(onlymapping)
	{
		nav_ = phd_create_multi(&params, devices.data(), (int) devices.size());
		if (!nav_) throw PhdError(PHD_ERR_NO_DEVICE, phd_create_error());
		reset(pose, Map(), particlecount);
	}
	~PHDNavigator() { Dispose(); }
	PHDNavigator(const PHDNavigator&) = delete;
	PHDNavigator& operator=(const PHDNavigator&) = delete;

	void Dispose()   // Navigator.cs:395
	{
		if (nav_) phd_destroy(nav_);
		nav_ = nullptr;
	}

	int  ParticleCount;   // :118
	bool OnlyMapping;     // Navigator.cs:129

	// ≙ Update (:295-314): the motion model (TrackVehicle.UpdateNoisy) and its RNG stay on the host
	void Update(const std::vector<Pose3D>& particleposes)
	{
		check(phd_set_poses(nav_, particleposes.empty() ? nullptr : particleposes[0].data(), (int) particleposes.size()));
	}

	// The same step with the motion model on the device (phd_update_motion): the odometry reading and
	// one noise vector per particle (dt * chol(MotionCovariance) * N(0, I), drawn by the host; empty: none)
	void UpdateOdometry(const std::array<double, 6>& reading, const std::vector<std::array<double, 6>>& noise, bool perfectstill)
	{
		check(phd_update_motion(nav_, reading.data(), noise.empty() ? nullptr : noise[0].data(), ParticleCount, perfectstill ? 1 : 0));
	}

	// ≙ static QuasiSetLogLikelihood(measurements, map, pose) (:526-531), for a batch of candidate poses (row f4)
	std::vector<double> QuasiSetLogLikelihood(const std::vector<PixelRangeMeasurement>& measurements,
	                                          const std::vector<std::array<double, 3>>& landmarks, const std::vector<Pose3D>& poses)
	{
		std::vector<double> out(poses.size());
		check(phd_quasi_set_loglik(nav_, poses.empty() ? nullptr : poses[0].data(), (int) poses.size(),
		                           landmarks.empty() ? nullptr : landmarks[0].data(), (int) landmarks.size(),
		                           measurements.empty() ? nullptr : measurements[0].data(), (int) measurements.size(), out.data()));
		return out;
	}

	// ≙ static QuasiSetLogLikelihood(measurements, map, pose, out gradient) (:543-548) for a batch of candidate poses;
	// averagemode: TemperedAverage as its source reads (0) or with weights that sum to one (1), see phdhip.h
	std::vector<double> QuasiSetLogLikelihood(const std::vector<PixelRangeMeasurement>& measurements,
	                                          const std::vector<std::array<double, 3>>& landmarks, const std::vector<Pose3D>& poses,
	                                          std::vector<std::array<double, 6>>& gradients, int averagemode = 0)
	{
		std::vector<double> out(poses.size());
		gradients.assign(poses.size(), std::array<double, 6>{});
		check(phd_quasi_set_loglik_grad(nav_, poses.empty() ? nullptr : poses[0].data(), (int) poses.size(),
		                                landmarks.empty() ? nullptr : landmarks[0].data(), (int) landmarks.size(),
		                                measurements.empty() ? nullptr : measurements[0].data(), (int) measurements.size(), averagemode,
		                                out.data(), gradients.empty() ? nullptr : gradients[0].data()));
		return out;
	}

	// ≙ SlamUpdate (:323-362); `uniform` replaces (double) Util.Uniform.Next() of ResampleParticles (:727)
	void SlamUpdate(const std::vector<PixelRangeMeasurement>& measurements, double uniform)
	{
		check(phd_slam_update(nav_, measurements.empty() ? nullptr : measurements[0].data(), (int) measurements.size(),
		                      OnlyMapping ? 1 : 0, uniform));
	}

	std::vector<double> VehicleWeights()   // :128
	{
		int n = 0;
		const double* w = phd_weights(nav_, &n);
		if (!w) throw PhdError(PHD_ERR_DEVICE, phd_last_error(nav_));
		return std::vector<double>(w, w + n);
	}

	int BestParticle() { return phd_best_particle(nav_); }   // :139

	std::vector<Pose3D> VehicleParticles()   // poses of :123
	{
		int n = 0;
		const double* p = phd_poses(nav_, &n);
		if (!p) throw PhdError(PHD_ERR_DEVICE, phd_last_error(nav_));
		std::vector<Pose3D> out(n / 7);
		for (int i = 0; i < n / 7; i++) std::copy(p + i * 7, p + i * 7 + 7, out[i].begin());
		return out;
	}

	Pose3D BestEstimate() { return VehicleParticles()[BestParticle()]; }   // :144-150

	Map MapModels(int particle)   // :134
	{
		int n = 0;
		const double *w, *m, *c;
		check(phd_map(nav_, particle, &n, &w, &m, &c));
		return tomap(n, w, m, c);
	}

	Map BestMapModel() { return MapModels(BestParticle()); }   // :155-161

	// ≙ reset (:245-266)
	void reset(const Pose3D& pose, const Map& model, int particlecount)
	{
		std::vector<double> w, m, c;
		frommap(model, w, m, c);
		check(phd_reset(nav_, particlecount, pose.data(), w.data(), m.data(), c.data(), (int) model.size()));
	}

	void CollapseParticles(int particlecount) { reset(BestEstimate(), BestMapModel(), particlecount); }   // :233-236
	void StartSlamInternal() { CollapseParticles(ParticleCount); }                                        // :214-217
	void StartMappingInternal() { CollapseParticles(1); }                                                 // :224-227

	void ResetMapModel()   // :271-276
	{
		int P = phd_particle_count(nav_);
		for (int i = 0; i < P; i++) check(phd_set_map(nav_, i, nullptr, nullptr, nullptr, 0));
	}

	// ≙ ResampleParticles (:724-760) / ParticleDepleted (:768-777) on given weights
	std::vector<int32_t> ResampleParticles(const std::vector<double>& weights, double uniform, int* best = nullptr)
	{
		std::vector<int32_t> src(weights.size());
		int32_t b = 0;
		check(phd_resample(nav_, weights.data(), (int) weights.size(), uniform, src.data(), &b));
		if (best) *best = b;
		return src;
	}

	bool ParticleDepleted(const std::vector<double>& weights)
	{
		uint8_t d = 0;
		check(phd_particle_depleted(nav_, weights.data(), (int) weights.size(), &d));
		return d != 0;
	}

	// after a SlamUpdate: the slot every particle was copied from (identity if no resampling), so the
	// host can permute what it keeps per particle (trajectories, TrackVehicle objects)
	std::vector<int32_t> ResampleSources(bool* resampled = nullptr)
	{
		int n = 0;
		uint8_t r = 0;
		const int32_t* s = phd_resample_sources(nav_, &n, &r);
		if (!s) throw PhdError(PHD_ERR_DEVICE, phd_last_error(nav_));
		if (resampled) *resampled = r != 0;
		return std::vector<int32_t>(s, s + n);
	}

	phd_navigator* handle() { return nav_; }

private:
	void check(int status)
	{
		if (status != PHD_OK) throw PhdError(status, phd_last_error(nav_));
	}

	static Map tomap(int n, const double* w, const double* m, const double* c)
	{
		Map out(n);
		for (int i = 0; i < n; i++) {
			out[i].weight = w[i];
			std::copy(m + i * 3, m + i * 3 + 3, out[i].mean.begin());
			std::copy(c + i * 9, c + i * 9 + 9, out[i].covariance.begin());
		}
		return out;
	}

	static void frommap(const Map& map, std::vector<double>& w, std::vector<double>& m, std::vector<double>& c)
	{
		w.assign(map.size() + 1, 0);
		m.assign(map.size() * 3 + 3, 0);
		c.assign(map.size() * 9 + 9, 0);
		for (size_t i = 0; i < map.size(); i++) {
			w[i] = map[i].weight;
			std::copy(map[i].mean.begin(), map[i].mean.end(), m.begin() + i * 3);
			std::copy(map[i].covariance.begin(), map[i].covariance.end(), c.begin() + i * 9);
		}
	}

	phd_navigator* nav_ = nullptr;
};

}  // namespace monorfs
