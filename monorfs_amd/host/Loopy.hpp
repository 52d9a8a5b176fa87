// Loopy.hpp — C++ host mirror of the pose searches of the reference's smoother over libphdhip.so:
//
//   LoopyPHDNavigator.Filter / FilterMissing          SLAM/Navigators/LoopyPHDNavigator.cs:713-762
//   LoopyPHDNavigator.LogLikeGradient                  :876-909   (12 evaluations, one device batch)
//   LoopyPHDNavigator.LogLikeGradientAscent            :916-965   (any number of initial estimates side by side; the 16
//                                                                  step sizes of a line search are one device batch)
//   Pose3D.Add / Subtract                              BaseStructures/Poses/Pose3D.cs:282-308
//
// The evaluations are PHDNavigator::QuasiSetLogLikelihood (phd_quasi_set_loglik / phd_quasi_set_loglik_grad); per pose the
// results are those of the reference's one-evaluation-at-a-time loops. monorfs_amd/loopy.py is the Python twin and also
// carries LogLikeFitCovariance and GuidedFitMixture (which need a 6 x 6 eigen-decomposition and pseudo-inverse).
#pragma once
#include "PHDNavigator.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <utility>

namespace monorfs {

typedef std::array<double, 6> Odometry;   // linear pose / odometry delta: dx dy dz, rotation vector

namespace detail {
struct Q { double w, x, y, z; };
inline Q qmul(const Q& a, const Q& b)   // Quaternion.cs:295-301
{
	return Q{a.w * b.w - (a.x * b.x + a.y * b.y + a.z * b.z), a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
	         a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline Q qconj(const Q& q) { return Q{q.w, -q.x, -q.y, -q.z}; }
inline Q qnormalize(const Q& q)
{
	double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
	return Q{q.w / n, q.x / n, q.y / n, q.z / n};
}
}  // namespace detail

// Pose3D.Add (Pose3D.cs:282-291): x + q dx q*, q Exp(dr / 2) normalised
inline Pose3D PoseAdd(const Pose3D& pose, const Odometry& delta)
{
	using namespace detail;
	const Q q{pose[3], pose[4], pose[5], pose[6]};
	const double lie[3] = {0.5 * delta[3], 0.5 * delta[4], 0.5 * delta[5]};
	const double phi = std::sqrt(lie[0] * lie[0] + lie[1] * lie[1] + lie[2] * lie[2]);
	Q dq{1, 0, 0, 0};
	if (!(phi < 1e-12)) {   // Quaternion.Exp, Quaternion.cs:185-196
		const double s = std::sin(phi);
		dq = Q{std::cos(phi), s * (lie[0] / phi), s * (lie[1] / phi), s * (lie[2] / phi)};
	}
	const Q nq = qnormalize(qmul(q, dq));
	const Q dl = qmul(qmul(q, Q{0, delta[0], delta[1], delta[2]}), qconj(q));
	return Pose3D{pose[0] + dl.x, pose[1] + dl.y, pose[2] + dl.z, nq.w, nq.x, nq.y, nq.z};
}

// Pose3D.Subtract (Pose3D.cs:296-308)
inline Odometry PoseSubtract(const Pose3D& pose, const Pose3D& origin)
{
	using namespace detail;
	const Q qo{origin[3], origin[4], origin[5], origin[6]};
	Q dq = qnormalize(qmul(qconj(qo), Q{pose[3], pose[4], pose[5], pose[6]}));
	const Q dx = qmul(qmul(qconj(qo), Q{0, pose[0] - origin[0], pose[1] - origin[1], pose[2] - origin[2]}), qo);
	const double phi = std::acos(std::min(1.0, std::max(-1.0, dq.w)));   // Quaternion.Log, Quaternion.cs:204-218
	const double mag = std::sqrt(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z);
	Odometry out{dx.x, dx.y, dx.z, 0, 0, 0};
	if (!(mag < 1e-12)) {
		out[3] = 2 * phi * (dq.x / mag); out[4] = 2 * phi * (dq.y / mag); out[5] = 2 * phi * (dq.z / mag);
	}
	return out;
}

// ≙ LoopyPHDNavigator.FilterMissing (:729-762): a one-particle, mapping-only filter over trajectory[i] with the
// measurement sets factors[i], frame `index` left out, frames from `to` on ignored; `filter` is reset first
// (≙ InnerFilter = new PHDNavigator(RefVehicle, 1, true)). Returns BestMapModel.
inline Map FilterMissing(PHDNavigator& filter, const std::vector<std::pair<double, Pose3D>>& trajectory,
                         const std::vector<std::pair<double, std::vector<PixelRangeMeasurement>>>& factors, int index, int to)
{
	to    = std::min((int) trajectory.size(), to);
	index = (index < 0) ? to : std::min(to, index);
	filter.OnlyMapping = true;
	filter.reset(trajectory.empty() ? Pose3D{0, 0, 0, 1, 0, 0, 0} : trajectory[0].second, Map(), 1);
	for (int i = 0; i < to; i++) {
		if (i == index) continue;
		filter.Update({trajectory[i].second});          // InnerFilter.BestEstimate.Pose = trajectory[i].Item2
		filter.SlamUpdate(factors[i].second, 0.5);       // mapping only: no resampling, the uniform is not used
	}
	return filter.BestMapModel();
}

// ≙ LoopyPHDNavigator.Filter (:718-721)
inline Map Filter(PHDNavigator& filter, const std::vector<std::pair<double, Pose3D>>& trajectory,
                  const std::vector<std::pair<double, std::vector<PixelRangeMeasurement>>>& factors)
{
	return FilterMissing(filter, trajectory, factors, (int) trajectory.size(), (int) trajectory.size());
}

// ≙ LoopyPHDNavigator.LogLikeGradient (:876-909): central differences, eps = 1e-5
inline Odometry LogLikeGradient(PHDNavigator& nav, const Odometry& pose, const std::vector<PixelRangeMeasurement>& measurements,
                                const std::vector<std::array<double, 3>>& landmarks, const Pose3D& linearpoint)
{
	const double eps = 1e-5;
	std::vector<Pose3D> cand;
	for (int i = 0; i < 6; i++) {
		for (double sign : {1.0, -1.0}) {
			Odometry d = pose;
			d[i] += sign * eps;
			cand.push_back(PoseAdd(linearpoint, d));
		}
	}
	const std::vector<double> l = nav.QuasiSetLogLikelihood(measurements, landmarks, cand);
	Odometry g;
	for (int i = 0; i < 6; i++) g[i] = (l[2 * i] - l[2 * i + 1]) / (2 * eps);
	return g;
}

// ≙ LoopyPHDNavigator.LogLikeGradientAscent (:916-965) for n initial estimates side by side; loglike[a] is the value at
// the returned poses[a]. Per estimate exactly the reference's loop: the analytic gradient at the LAST TRIED pose
// (`nextvehicle`, :933-934), clipped to GradientClip (Config.cs:95), step GradientAscentRate (Config.cs:94) halved up
// to 16 times until the value does not decrease, until an iteration gains no more than 1e-3.
// maxbatch: poses per device call (<= the handle's max_particles). averagemode: see phd_quasi_set_loglik_grad.
inline std::vector<Odometry> LogLikeGradientAscent(PHDNavigator& nav, const std::vector<Odometry>& initial,
                                                   const std::vector<PixelRangeMeasurement>& measurements,
                                                   const std::vector<std::array<double, 3>>& landmarks, const Pose3D& linearpoint,
                                                   std::vector<double>& loglike, int maxbatch, int averagemode = 0,
                                                   double gradientascentrate = 1e-2, double gradientclip = 10)
{
	const int n = (int) initial.size();
	auto values = [&](const std::vector<Pose3D>& poses, std::vector<std::array<double, 6>>* grads) {
		std::vector<double> out;
		if (grads) grads->clear();
		for (size_t s = 0; s < poses.size(); s += maxbatch) {
			std::vector<Pose3D> part(poses.begin() + s, poses.begin() + std::min(poses.size(), s + (size_t) maxbatch));
			std::vector<std::array<double, 6>> g;
			std::vector<double> v = grads ? nav.QuasiSetLogLikelihood(measurements, landmarks, part, g, averagemode)
			                              : nav.QuasiSetLogLikelihood(measurements, landmarks, part);
			out.insert(out.end(), v.begin(), v.end());
			if (grads) grads->insert(grads->end(), g.begin(), g.end());
		}
		return out;
	};
	std::vector<Odometry> pose = initial;
	std::vector<Pose3D> nextpose(n);
	for (int a = 0; a < n; a++) nextpose[a] = PoseAdd(linearpoint, pose[a]);
	std::vector<std::array<double, 6>> grad;
	loglike = values(nextpose, &grad);   // :928-929
	std::vector<double> prevvalue(n, -std::numeric_limits<double>::infinity());
	std::vector<int> active;
	for (int a = 0; a < n; a++) if (loglike[a] - prevvalue[a] > 1e-3) active.push_back(a);
	while (!active.empty()) {
		std::vector<Pose3D> at;
		for (int a : active) at.push_back(nextpose[a]);
		values(at, &grad);   // :933-934
		std::vector<Odometry> cand6(active.size() * 16);
		std::vector<Pose3D>   cand7(active.size() * 16);
		for (size_t u = 0; u < active.size(); u++) {
			Odometry g = grad[u];
			double size = 0;
			for (double x : g) size += x * x;
			size = std::sqrt(size);
			if (size > gradientclip) for (double& x : g) x = x * (gradientclip / size);
			double multiplier = gradientascentrate;
			for (int c = 0; c < 16; c++) {
				for (int t = 0; t < 6; t++) cand6[u * 16 + c][t] = pose[active[u]][t] + multiplier * g[t];
				cand7[u * 16 + c] = PoseAdd(linearpoint, cand6[u * 16 + c]);
				multiplier /= 2.0;
			}
		}
		const std::vector<double> vals = values(cand7, nullptr);
		std::vector<int> still;
		for (size_t u = 0; u < active.size(); u++) {
			const int a = active[u];
			int c = 0;
			double nextloglike;
			do {   // :943-951
				nextloglike = vals[u * 16 + c];
				c++;
			} while (nextloglike < loglike[a] && c < 16);
			nextpose[a]  = cand7[u * 16 + c - 1];
			prevvalue[a] = loglike[a];
			if (nextloglike > loglike[a]) {
				pose[a]    = cand6[u * 16 + c - 1];
				loglike[a] = nextloglike;
			}
			if (loglike[a] - prevvalue[a] > 1e-3) still.push_back(a);
		}
		active = still;
	}
	return pose;
}

}  // namespace monorfs
