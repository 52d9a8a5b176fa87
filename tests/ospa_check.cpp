// Reads pairs of landmark sets from stdin (text: C P na nb, then na + nb points), prints OSPA and its cardinality part
// per case: monorfs_amd/host/Ospa.hpp against the oracle's restatement of Plot.OSPA (tests/test_ospa_host.py).
#include "../monorfs_amd/host/Ospa.hpp"

#include <cstdio>

int main()
{
	double C, P;
	int na, nb;
	while (std::scanf("%lf %lf %d %d", &C, &P, &na, &nb) == 4) {
		std::vector<std::array<double, 3>> a(na), b(nb);
		for (auto& x : a) if (std::scanf("%lf %lf %lf", &x[0], &x[1], &x[2]) != 3) return 1;
		for (auto& x : b) if (std::scanf("%lf %lf %lf", &x[0], &x[1], &x[2]) != 3) return 1;
		double card = 0;
		double d = monorfs::OSPA(a, b, C, P, &card);
		std::printf("%.17g %.17g\n", d, card);
	}
	return 0;
}
