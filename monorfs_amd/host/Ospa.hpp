// Ospa.hpp — the map-error metric of the reference's post-analysis, host side, C++.
//
//   Plot.OSPA(a, b, out cardinalityerror)   postanalysis/Plot.cs:531-581     (LandmarkDistance :583-586)
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

}  // namespace monorfs
