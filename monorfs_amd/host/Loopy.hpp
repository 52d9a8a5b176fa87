// Loopy.hpp — C++ host mirror of the pose searches of the reference's smoother over libphdhip.so:
//
//   LoopyPHDNavigator.Filter / FilterMissing          SLAM/Navigators/LoopyPHDNavigator.cs:713-762
//   LoopyPHDNavigator.LogLikeGradient                  :876-909   (12 evaluations, one device batch)
//   LoopyPHDNavigator.LogLikeGradientAscent            :916-965   (any number of initial estimates side by side; the 16
//                                                                  step sizes of a line search are one device batch)
//   Pose3D.Add / Subtract                              BaseStructures/Poses/Pose3D.cs:282-308
//
// The evaluations are PHDNavigator::QuasiSetLogLikelihood (phd_quasi_set_loglik / phd_quasi_set_loglik_grad); per pose the
// results are those of the reference's one-evaluation-at-a-time loops. monorfs_amd/loopy.py is the Python twin.
//   LoopyPHDNavigator.LogLikeFitCovariance / FitGaussian / GuidedFitMixture     :777-869, :976-1021 (at the end of this file)
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

// ---- the covariance fit and the guided mixture fit (LoopyPHDNavigator.cs:777-852, :976-1021) ----

typedef std::array<double, 36> Matrix6;   // row-major

namespace detail {
// eigen-decomposition of a symmetric 6 x 6 matrix by cyclic Jacobi rotations: A = V diag(vals) V'
inline void SymEig6(const Matrix6& A, std::array<double, 6>& vals, Matrix6& V)
{
	Matrix6 a = A;
	V.fill(0.0);
	for (int i = 0; i < 6; i++) V[i * 6 + i] = 1.0;
	for (int sweep = 0; sweep < 60; sweep++) {
		double off = 0, diag = 0;
		for (int i = 0; i < 6; i++) for (int k = 0; k < 6; k++) (i == k ? diag : off) += a[i * 6 + k] * a[i * 6 + k];
		if (off <= 1e-30 * diag || off == 0) break;
		for (int p = 0; p < 5; p++) {
			for (int q = p + 1; q < 6; q++) {
				const double apq = a[p * 6 + q];
				if (apq == 0) continue;
				const double theta = (a[q * 6 + q] - a[p * 6 + p]) / (2 * apq);
				const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
				const double c = 1 / std::sqrt(t * t + 1), sn = t * c;
				for (int k = 0; k < 6; k++) {   // A <- A J
					const double akp = a[k * 6 + p], akq = a[k * 6 + q];
					a[k * 6 + p] = c * akp - sn * akq; a[k * 6 + q] = sn * akp + c * akq;
				}
				for (int k = 0; k < 6; k++) {   // A <- J' A
					const double apk = a[p * 6 + k], aqk = a[q * 6 + k];
					a[p * 6 + k] = c * apk - sn * aqk; a[q * 6 + k] = sn * apk + c * aqk;
				}
				for (int k = 0; k < 6; k++) {
					const double vkp = V[k * 6 + p], vkq = V[k * 6 + q];
					V[k * 6 + p] = c * vkp - sn * vkq; V[k * 6 + q] = sn * vkp + c * vkq;
				}
			}
		}
	}
	for (int i = 0; i < 6; i++) vals[i] = a[i * 6 + i];
}
}  // namespace detail

// ≙ LoopyPHDNavigator.LogLikeFitCovariance (:976-1021): the Hessian of the quasi set log-likelihood at `pose` by central
// differences (eps = 1e-5) of the analytic gradient — 12 gradient evaluations, one device batch —, NaN -> zero matrix,
// positive eigenvalues clipped to zero, covariance = pseudo-inverse of the negated result. The reference decomposes
// the not exactly symmetric finite-difference Hessian with Accord's general EigenvalueDecomposition (outside its tree);
// here, as in monorfs_amd/loopy.py, the symmetric part is decomposed (the same thing for a symmetric Hessian).
inline Matrix6 LogLikeFitCovariance(PHDNavigator& nav, const Odometry& pose, const std::vector<PixelRangeMeasurement>& measurements,
                                    const std::vector<std::array<double, 3>>& landmarks, const Pose3D& linearpoint, int averagemode = 0)
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
	std::vector<std::array<double, 6>> g;
	nav.QuasiSetLogLikelihood(measurements, landmarks, cand, g, averagemode);
	Matrix6 H;
	bool nan = false;
	for (int i = 0; i < 6; i++) {
		for (int k = 0; k < 6; k++) {
			H[i * 6 + k] = (g[2 * i][k] - g[2 * i + 1][k]) / (2 * eps);
			nan = nan || std::isnan(H[i * 6 + k]);
		}
	}
	if (nan) H.fill(0.0);
	Matrix6 S, V;
	for (int i = 0; i < 6; i++) for (int k = 0; k < 6; k++) S[i * 6 + k] = 0.5 * (H[i * 6 + k] + H[k * 6 + i]);
	std::array<double, 6> vals;
	detail::SymEig6(S, vals, V);
	double lmax = 0;
	for (double& v : vals) { v = -std::min(0.0, v); lmax = std::max(lmax, v); }   // eigenvalues of -(clipped Hessian)
	Matrix6 cov;
	cov.fill(0.0);
	for (int e = 0; e < 6; e++) {
		if (!(vals[e] > 1e-15 * lmax) || vals[e] == 0) continue;   // pseudo-inverse: directions without information stay zero
		for (int i = 0; i < 6; i++) for (int k = 0; k < 6; k++) cov[i * 6 + k] += V[i * 6 + e] * V[k * 6 + e] / vals[e];
	}
	return cov;
}

// ≙ LoopyPHDNavigator.FitGaussian (:863-869): mean and covariance of the weight-1 Gaussian fitted near pose0
inline std::pair<Odometry, Matrix6> FitGaussian(PHDNavigator& nav, const Odometry& pose0, const std::vector<PixelRangeMeasurement>& measurements,
                                                const std::vector<std::array<double, 3>>& landmarks, const Pose3D& linearpoint,
                                                int maxbatch, int averagemode = 0)
{
	std::vector<double> loglike;
	const Odometry maxpose = LogLikeGradientAscent(nav, {pose0}, measurements, landmarks, linearpoint, loglike, maxbatch, averagemode)[0];
	return {maxpose, LogLikeFitCovariance(nav, maxpose, measurements, landmarks, linearpoint, averagemode)};
}

// Map.BestMapEstimate (Map.cs:119-142): the means of (int) ExpectedSize picks from the weight-sorted list, every pick
// re-entered with its weight less one (stable order on ties, as on the device)
inline std::vector<std::array<double, 3>> BestMapEstimate(const Map& map)
{
	double sum = 0;
	for (const Gaussian& g : map) sum += g.weight;
	const int size = (int) sum;
	std::vector<std::pair<double, int>> lst;
	for (int i = 0; i < (int) map.size(); i++) lst.push_back({map[i].weight, i});
	auto bywd = [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; };
	std::stable_sort(lst.begin(), lst.end(), bywd);
	std::vector<std::array<double, 3>> best;
	for (int i = 0; i < size; i++) {
		const std::pair<double, int> e = lst[i];
		best.push_back(map[e.second].mean);
		lst.push_back({e.first - 1, e.second});
		std::stable_sort(lst.begin(), lst.end(), bywd);
	}
	return best;
}

// PRM3DMeasurer.FitToMeasurement (PRM3DMeasurer.cs:224-244): the pose near pose0 from which `landmark` is measured as given
inline Pose3D FitToMeasurement(double visionfocal, const Pose3D& pose0, const PixelRangeMeasurement& measurement, const std::array<double, 3>& landmark)
{
	using namespace detail;
	auto rotate = [](const Q& q, const double v[3], double out[3]) {   // Quaternion.ToMatrix() * v
		const Q r = qmul(qmul(q, Q{0, v[0], v[1], v[2]}), qconj(q));
		out[0] = r.x; out[1] = r.y; out[2] = r.z;
	};
	const Q q0{pose0[3], pose0[4], pose0[5], pose0[6]};
	const double diff[3] = {landmark[0] - pose0[0], landmark[1] - pose0[1], landmark[2] - pose0[2]};
	double ll[3], ml[3];
	rotate(qconj(q0), diff, ll);
	const double invf = 1.0 / visionfocal;
	ml[2] = measurement[2] / std::sqrt(1 + (measurement[0] * measurement[0] + measurement[1] * measurement[1]) * invf * invf);
	ml[0] = measurement[0] * ml[2] * invf;
	ml[1] = measurement[1] * ml[2] * invf;
	const double nl = std::sqrt(ll[0] * ll[0] + ll[1] * ll[1] + ll[2] * ll[2]), nm = std::sqrt(ml[0] * ml[0] + ml[1] * ml[1] + ml[2] * ml[2]);
	const double a[3] = {ll[0] / nl, ll[1] / nl, ll[2] / nl}, b[3] = {ml[0] / nm, ml[1] / nm, ml[2] / nm};
	// Quaternion.VectorRotator(a, b) (Quaternion.cs:281-284)
	const Q align = qnormalize(Q{1 + (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]), a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]});
	const Q rotation = qmul(qconj(align), q0);
	double moved[3];
	rotate(rotation, ml, moved);
	const Q rn = qnormalize(rotation);
	return Pose3D{landmark[0] - moved[0], landmark[1] - moved[1], landmark[2] - moved[2], rn.w, rn.x, rn.y, rn.z};
}

struct PoseComponent {   // a component of the fitted mixture over linear poses
	double   weight;
	Odometry mean;
	Matrix6  covariance;
};

// ≙ LoopyPHDNavigator.GuidedFitMixture (:777-852): see monorfs_amd/loopy.py for the walk-through; all guesses climb side
// by side, the covariance is fitted once at pose0 (the reference's `maxpose` never moves).
inline std::vector<PoseComponent> GuidedFitMixture(PHDNavigator& nav, double visionfocal, const Odometry& pose0,
                                                   const std::vector<PixelRangeMeasurement>& measurements, const Map& map,
                                                   const Pose3D& linearpoint, int maxbatch, double* emptyspace, int averagemode = 0)
{
	const Pose3D initpose = PoseAdd(linearpoint, pose0);
	const std::vector<std::array<double, 3>> jmap = BestMapEstimate(map);
	std::vector<Odometry> guesses = {pose0};
	for (const auto& landmark : jmap) {
		for (const auto& measurement : measurements) {
			const Pose3D guess = FitToMeasurement(visionfocal, initpose, measurement, landmark);
			const Odometry d = PoseSubtract(guess, initpose);
			double sq = 0;
			for (double x : d) sq += x * x;
			if (sq < 0.5 * 0.5) guesses.push_back(PoseSubtract(guess, linearpoint));
		}
	}
	std::vector<Pose3D> first = {PoseAdd(Pose3D{0, 0, 0, 1, 0, 0, 0}, Odometry{1e5, 1e5, 1e5, 1e5, 1e5, 1e5})};
	for (const Odometry& g : guesses) first.push_back(PoseAdd(linearpoint, g));
	std::vector<double> v;
	for (size_t s = 0; s < first.size(); s += maxbatch) {
		std::vector<Pose3D> part(first.begin() + s, first.begin() + std::min(first.size(), s + (size_t) maxbatch));
		std::vector<double> pv = nav.QuasiSetLogLikelihood(measurements, jmap, part);
		v.insert(v.end(), pv.begin(), pv.end());
	}
	if (emptyspace) *emptyspace = v[0];
	std::vector<Odometry> climbing;
	for (size_t g = 0; g < guesses.size(); g++) if (!(v[1 + g] - v[0] < 0)) climbing.push_back(guesses[g]);
	std::vector<PoseComponent> components;
	if (climbing.empty()) return components;
	std::vector<double> values;
	const std::vector<Odometry> poses = LogLikeGradientAscent(nav, climbing, measurements, jmap, linearpoint, values, maxbatch, averagemode);
	bool havecov = false;
	Matrix6 localcov{};
	auto pinvquad = [](const Matrix6& cov, const Odometry& d) {   // d' pinv(cov) d
		Matrix6 V;
		std::array<double, 6> vals;
		detail::SymEig6(cov, vals, V);
		double lmax = 0, q = 0;
		for (double x : vals) lmax = std::max(lmax, std::fabs(x));
		for (int e = 0; e < 6; e++) {
			if (!(std::fabs(vals[e]) > 1e-15 * lmax)) continue;
			double proj = 0;
			for (int i = 0; i < 6; i++) proj += V[i * 6 + e] * d[i];
			q += proj * proj / vals[e];
		}
		return q;
	};
	for (size_t a = 0; a < poses.size(); a++) {
		bool counted = false;
		for (const PoseComponent& c : components) {
			Odometry d;
			for (int t = 0; t < 6; t++) d[t] = c.mean[t] - poses[a][t];
			if (std::sqrt(std::max(0.0, pinvquad(c.covariance, d))) < 0.1) { counted = true; break; }
		}
		if (counted) continue;
		if (!havecov) { localcov = LogLikeFitCovariance(nav, pose0, measurements, jmap, linearpoint, averagemode); havecov = true; }
		// pseudo-determinant: the product of the non-zero eigenvalues; Math.Pow(2 pi, -6 / 2) with integer division
		Matrix6 V;
		std::array<double, 6> vals;
		detail::SymEig6(localcov, vals, V);
		double lmax = 0, pdet = 1;
		bool any = false;
		for (double x : vals) lmax = std::max(lmax, std::fabs(x));
		for (double x : vals) if (std::fabs(x) > 6 * 2.220446049250313e-16 * lmax) { pdet *= std::fabs(x); any = true; }
		if (!any) pdet = 0;
		const double logmultiplier = std::log(std::pow(2 * 3.14159265358979323846, -3.0)) - 0.5 * std::log(pdet);
		components.push_back(PoseComponent{std::exp(values[a] - logmultiplier), poses[a], localcov});
	}
	return components;
}

}  // namespace monorfs
