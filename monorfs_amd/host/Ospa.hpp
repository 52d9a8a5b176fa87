// Ospa.hpp — the map-error metric of the reference's post-analysis, host side, C++.
//
//   Plot.OSPA(a, b, out cardinalityerror)   postanalysis/Plot.cs:531-581     (LandmarkDistance :583-586)
//   Plot.MapError, Plot.VisitedMap           postanalysis/Plot.cs:478-529, :230-248   (at the end of this file)
//
// OSPA of order P with cut-off C between two landmark sets: with m = |a| <= n = |b|,
//     ( ( min over assignments of sum_i min(C, |a_i - b_pi(i)|)^P  +  C^P (n - m) ) / n )^(1/P)
// and its cardinality part C ((n - m) / n)^(1/P). The reference builds the transport problem on C^P - d^P, drops
// entries below 1e-5 and maximises with its Hungarian; the optimum VALUE does not depend on which optimal assignment
// is found, so this file solves the same problem with a shortest-augmenting-path assignment (O(n^3)) of its own.
// `monorfs::Map` / phd_map output goes in directly through the mean arrays.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <limits>
#include <vector>

namespace monorfs {

// minimum-cost assignment of the rows of an n x n cost matrix (row-major) to its columns; returns the cost
inline double AssignmentMinCost(const std::vector<double>& cost, int n, std::vector<int>* rowtocol = nullptr)
{
	const double INF = std::numeric_limits<double>::infinity();
	std::vector<double> u(n + 1, 0.0), v(n + 1, 0.0), minv(n + 1);
	std::vector<int> p(n + 1, 0), way(n + 1, 0);
	std::vector<char> used(n + 1);
	for (int i = 1; i <= n; i++) {
		p[0] = i;
		int j0 = 0;
		std::fill(minv.begin(), minv.end(), INF);
		std::fill(used.begin(), used.end(), 0);
		do {
			used[j0] = 1;
			int i0 = p[j0], j1 = 0;
			double delta = INF;
			for (int j = 1; j <= n; j++) {
				if (used[j]) continue;
				double cur = cost[(size_t) (i0 - 1) * n + (j - 1)] - u[i0] - v[j];
				if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
				if (minv[j] < delta) { delta = minv[j]; j1 = j; }
			}
			for (int j = 0; j <= n; j++) {
				if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
				else minv[j] -= delta;
			}
			j0 = j1;
		} while (p[j0] != 0);
		do {
			int j1 = way[j0];
			p[j0] = p[j1];
			j0 = j1;
		} while (j0);
	}
	double total = 0;
	if (rowtocol) rowtocol->assign(n, -1);
	for (int j = 1; j <= n; j++) {
		if (p[j] == 0) continue;
		total += cost[(size_t) (p[j] - 1) * n + (j - 1)];
		if (rowtocol) (*rowtocol)[p[j] - 1] = j - 1;
	}
	return total;
}

// Plot.OSPA: landmark sets as arrays of 3-D points
inline double OSPA(const std::vector<std::array<double, 3>>& a, const std::vector<std::array<double, 3>>& b, double C, double P,
                   double* cardinalityerror = nullptr)
{
	const std::vector<std::array<double, 3>>& s = (a.size() > b.size()) ? b : a;   // the smaller set
	const std::vector<std::array<double, 3>>& l = (a.size() > b.size()) ? a : b;
	const int m = (int) s.size(), n = (int) l.size();
	if (m == 0) {                                                                   // :539-542
		double c = (n == 0) ? 0.0 : C;
		if (cardinalityerror) *cardinalityerror = c;
		return c;
	}
	const double CP = std::pow(C, P);
	std::vector<double> cost((size_t) n * n, CP);                                   // unmatched rows pay the cut-off
	for (int i = 0; i < m; i++) {
		for (int k = 0; k < n; k++) {
			double d0 = s[i][0] - l[k][0], d1 = s[i][1] - l[k][1], d2 = s[i][2] - l[k][2];
			double dist = std::pow(std::min(C, std::sqrt(d0 * d0 + d1 * d1 + d2 * d2)), P);
			if (CP - dist > 1e-5) cost[(size_t) i * n + k] = dist;                  // :562: smaller gains are not stored
		}
	}
	if (cardinalityerror) *cardinalityerror = C * std::pow((double) (n - m) / n, 1.0 / P);
	return std::pow(AssignmentMinCost(cost, n) / n, 1.0 / P);
}

// ---- Plot.MapError (postanalysis/Plot.cs:478-529) and Plot.VisitedMap (:230-248) ----

// VisitedMap: the landmarks seen (weight > 0) in any frame so far, each once — a landmark is new when nothing already
// kept lies within 1e-5 of it (`cumulative.Near(landmark.Mean, 1e-5)`; the KD-tree's metric is outside the reference
// tree, at this radius only exact repeats matter). frames[i] = visible landmarks of frame i as (x, y, z, weight).
inline std::vector<std::array<double, 3>> VisitedMap(const std::vector<std::vector<std::array<double, 4>>>& frames)
{
	std::vector<std::array<double, 3>> cumulative;
	for (const auto& frame : frames) {
		for (const auto& l : frame) {
			if (!(l[3] > 0)) continue;
			bool near = false;
			for (const auto& c : cumulative) {
				double d0 = c[0] - l[0], d1 = c[1] - l[1], d2 = c[2] - l[2];
				if (std::sqrt(d0 * d0 + d1 * d1 + d2 * d2) <= 1e-5) { near = true; break; }
			}
			if (!near) cumulative.push_back({l[0], l[1], l[2]});
		}
	}
	return cumulative;
}

// One frame of MapError: the map estimate (means of Map.BestMapEstimate) is first moved by the difference between
// the estimated and the true pose at the reference time — delta = FromLinear(estimate.Subtract(truth)) (:502-503),
// p' = R(delta.q*) (p - c) + c - delta.x with c the estimated location (:505-507, :514-515) — then compared with the
// visited map by OSPA; the spatial part is (ospa^P - cardinality^P)^(1/P) (:521). Poses are x y z qw qx qy qz.
// hasreference = false (the estimate is shorter than RefTime, :500): no alignment.
inline double MapError(const std::vector<std::array<double, 3>>& visited, const std::vector<std::array<double, 3>>& estimate,
                       bool hasreference, const std::array<double, 7>& estimatedpose, const std::array<double, 7>& truepose,
                       double C, double P, double* spatialerror = nullptr)
{
	std::vector<std::array<double, 3>> refmap = estimate;
	if (hasreference) {
		auto mul = [](const std::array<double, 4>& a, const std::array<double, 4>& b) {   // Quaternion.cs:295-301
			return std::array<double, 4>{a[0] * b[0] - (a[1] * b[1] + a[2] * b[2] + a[3] * b[3]),
			                             a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
			                             a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
			                             a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]};
		};
		auto normalized = [](std::array<double, 4> q) {
			double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
			for (double& x : q) x /= n;
			return q;
		};
		// estimate.Subtract(truth) (Pose3D.cs:296-308)
		const std::array<double, 4> qe = normalized({estimatedpose[3], estimatedpose[4], estimatedpose[5], estimatedpose[6]});
		const std::array<double, 4> qt = normalized({truepose[3], truepose[4], truepose[5], truepose[6]});
		const std::array<double, 4> qtc = {qt[0], -qt[1], -qt[2], -qt[3]};
		const std::array<double, 4> dq = normalized(mul(qtc, qe));
		const std::array<double, 4> dx = mul(mul(qtc, {0, estimatedpose[0] - truepose[0], estimatedpose[1] - truepose[1],
		                                                estimatedpose[2] - truepose[2]}), qt);
		const double phi = std::acos(std::min(1.0, std::max(-1.0, dq[0])));   // Quaternion.Log, Quaternion.cs:204-218
		const double mag = std::sqrt(dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
		double lie[3] = {0, 0, 0};                                             // dq.ToLinear() / 2
		if (!(mag < 1e-12)) { lie[0] = phi * dq[1] / mag; lie[1] = phi * dq[2] / mag; lie[2] = phi * dq[3] / mag; }
		// FromLinear = Identity.Add (Pose3D.cs:248-251, :282-291): location = the translation part, orientation = Exp(lie)
		const double ang = std::sqrt(lie[0] * lie[0] + lie[1] * lie[1] + lie[2] * lie[2]);
		std::array<double, 4> dori = {1, 0, 0, 0};
		if (!(ang < 1e-12)) {
			const double sn = std::sin(ang);
			dori = normalized({std::cos(ang), sn * (lie[0] / ang), sn * (lie[1] / ang), sn * (lie[2] / ang)});
		}
		const double X = -dori[1], Y = -dori[2], Z = -dori[3], W = dori[0];    // Conjugate().ToMatrix(), Quaternion.cs:327-342
		const double R[9] = {1 - 2 * (Y * Y + Z * Z), 2 * (X * Y - Z * W), 2 * (X * Z + Y * W),
		                     2 * (X * Y + Z * W), 1 - 2 * (X * X + Z * Z), 2 * (Y * Z - X * W),
		                     2 * (X * Z - Y * W), 2 * (Y * Z + X * W), 1 - 2 * (X * X + Y * Y)};
		for (auto& m : refmap) {
			const double d[3] = {m[0] - estimatedpose[0], m[1] - estimatedpose[1], m[2] - estimatedpose[2]};
			for (int i = 0; i < 3; i++) {
				m[i] = (R[i * 3] * d[0] + R[i * 3 + 1] * d[1] + R[i * 3 + 2] * d[2]) + estimatedpose[i] + (-1.0) * dx[i + 1];
			}
		}
	}
	double card = 0;
	const double ospa = OSPA(visited, refmap, C, P, &card);
	if (spatialerror) *spatialerror = std::pow(std::pow(ospa, P) - std::pow(card, P), 1.0 / P);
	return ospa;
}

}  // namespace monorfs
