// Runs the C++ host mirror of the smoother's pose searches (monorfs_amd/host/Loopy.hpp) on a scene read from a text
// file and prints the results with 17 digits; tests/test_gpu_loopy.py compares them with monorfs_amd/loopy.py.
// scene file: "J M n T" then J landmarks (3), M measurements (3), linearpoint (7), n initial estimates (6),
// T frames of (time, pose 7, count, count measurements (3)).
#include "../monorfs_amd/host/Loopy.hpp"

#include <cstdio>
#include <cstdlib>

static double rd(FILE* f)
{
	double v;
	if (std::fscanf(f, "%lf", &v) != 1) { std::fprintf(stderr, "scene file too short\n"); std::exit(2); }
	return v;
}

int main(int argc, char** argv)
{
	if (argc < 3) return 2;
	FILE* f = std::fopen(argv[1], "r");
	if (!f) return 2;
	const int mode = std::atoi(argv[2]);
	const int J = (int) rd(f), M = (int) rd(f), n = (int) rd(f), T = (int) rd(f);
	std::vector<std::array<double, 3>> lm(J);
	std::vector<monorfs::PixelRangeMeasurement> z(M);
	for (auto& l : lm) for (double& x : l) x = rd(f);
	for (auto& m : z) for (double& x : m) x = rd(f);
	monorfs::Pose3D lin;
	for (double& x : lin) x = rd(f);
	std::vector<monorfs::Odometry> starts(n);
	for (auto& s : starts) for (double& x : s) x = rd(f);
	std::vector<std::pair<double, monorfs::Pose3D>> trajectory(T);
	std::vector<std::pair<double, std::vector<monorfs::PixelRangeMeasurement>>> factors(T);
	for (int t = 0; t < T; t++) {
		trajectory[t].first = factors[t].first = rd(f);
		for (double& x : trajectory[t].second) x = rd(f);
		factors[t].second.resize((int) rd(f));
		for (auto& m : factors[t].second) for (double& x : m) x = rd(f);
	}
	std::fclose(f);

	phd_params prm;
	phd_default_params(&prm, 256, 600, 16);
	try {
		monorfs::PHDNavigator nav(prm, lin, 1);
		std::vector<double> loglike;
		std::vector<monorfs::Odometry> poses = monorfs::LogLikeGradientAscent(nav, starts, z, lm, lin, loglike, 256, mode);
		for (int a = 0; a < n; a++) {
			std::printf("ascent %.17g", loglike[a]);
			for (double x : poses[a]) std::printf(" %.17g", x);
			std::printf("\n");
		}
		monorfs::Odometry g = monorfs::LogLikeGradient(nav, starts[0], z, lm, lin);
		std::printf("gradient");
		for (double x : g) std::printf(" %.17g", x);
		std::printf("\n");
		monorfs::Odometry d = monorfs::PoseSubtract(monorfs::PoseAdd(lin, starts[0]), lin);
		std::printf("roundtrip");
		for (double x : d) std::printf(" %.17g", x);
		std::printf("\n");
		monorfs::Matrix6 cov = monorfs::LogLikeFitCovariance(nav, starts[0], z, lm, lin, 1);
		std::printf("covariance");
		for (double x : cov) std::printf(" %.17g", x);
		std::printf("\n");
		monorfs::Map model;
		for (auto& l : lm) model.push_back(monorfs::Gaussian{1.0001, l, {1e-4, 0, 0, 0, 1e-4, 0, 0, 0, 1e-4}});
		double emptyspace = 0;
		std::vector<monorfs::PoseComponent> mix = monorfs::GuidedFitMixture(nav, prm.measurer[0], starts[0], z, model, lin, 256, &emptyspace, 1);
		std::printf("emptyspace %.17g\n", emptyspace);
		for (auto& c : mix) {
			std::printf("mixture %.17g", c.weight);
			for (double x : c.mean) std::printf(" %.17g", x);
			std::printf("\n");
		}
		monorfs::Map map = monorfs::FilterMissing(nav, trajectory, factors, 1, T);
		for (auto& c : map) {
			std::printf("component %.17g", c.weight);
			for (double x : c.mean) std::printf(" %.17g", x);
			for (double x : c.covariance) std::printf(" %.17g", x);
			std::printf("\n");
		}
		return 0;
	}
	catch (const monorfs::PhdError& e) {
		std::printf("PhdError status=%d module=%s: %s\n", e.status, e.module.c_str(), e.what());
		return 3;
	}
}
