// Reads pairs of landmark sets from stdin (text: C P na nb, then na + nb points), prints OSPA and its cardinality part
// per case: monorfs_amd/host/Ospa.hpp against the oracle's restatement of Plot.OSPA (tests/test_ospa_host.py).
#include "../monorfs_amd/host/Ospa.hpp"

#include <cstdio>
#include <string>

int main(int argc, char** argv)
{
	double C, P;
	int na, nb;
	if (argc > 1 && std::string(argv[1]) == "maperror") {
		// C P nv ne hasreference, estimated pose (7), true pose (7), nv visited + ne estimate points -> MapError and its spatial part
		int hasref;
		while (std::scanf("%lf %lf %d %d %d", &C, &P, &na, &nb, &hasref) == 5) {
			std::array<double, 7> est, tru;
			for (double& x : est) if (std::scanf("%lf", &x) != 1) return 1;
			for (double& x : tru) if (std::scanf("%lf", &x) != 1) return 1;
			std::vector<std::array<double, 3>> a(na), b(nb);
			for (auto& x : a) if (std::scanf("%lf %lf %lf", &x[0], &x[1], &x[2]) != 3) return 1;
			for (auto& x : b) if (std::scanf("%lf %lf %lf", &x[0], &x[1], &x[2]) != 3) return 1;
			double spatial = 0;
			double d = monorfs::MapError(a, b, hasref != 0, est, tru, C, P, &spatial);
			std::printf("%.17g %.17g\n", d, spatial);
		}
		return 0;
	}
	if (argc > 1 && std::string(argv[1]) == "visited") {
		// nframes, then per frame: count and count x (x y z weight) -> the visited map
		int nf;
		if (std::scanf("%d", &nf) != 1) return 1;
		std::vector<std::vector<std::array<double, 4>>> frames(nf);
		for (auto& f : frames) {
			int n;
			if (std::scanf("%d", &n) != 1) return 1;
			f.resize(n);
			for (auto& x : f) if (std::scanf("%lf %lf %lf %lf", &x[0], &x[1], &x[2], &x[3]) != 4) return 1;
		}
		for (auto& x : monorfs::VisitedMap(frames)) std::printf("%.17g %.17g %.17g\n", x[0], x[1], x[2]);
		return 0;
	}
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
