// Exercises the C++ host mirror (monorfs_amd/host/PHDNavigator.hpp) over the C-ABI.
// Without a HIP device it must fail loudly (exit 3); with one it runs two SLAM updates (exit 0).
#include "../monorfs_amd/host/PHDNavigator.hpp"

#include <cmath>
#include <cstdio>

int main()
{
	phd_params prm;
	phd_default_params(&prm, 8, 600, 16);
	monorfs::Pose3D pose = {0, 0, 0, 1, 0, 0, 0};
	try {
		monorfs::PHDNavigator nav(prm, pose, 8);
		std::vector<monorfs::PixelRangeMeasurement> z = {{10, 20, 1.0}, {-50, 30, 0.8}, {100, -60, 1.4}};
		nav.SlamUpdate(z, 0.5);   // three births per particle, corrected by their own measurements in the same frame
		monorfs::Map m0 = nav.BestMapModel();
		if (m0.size() < 3 || m0.size() > 6) { std::printf("expected 3..6 components after the first frame, got %zu\n", m0.size()); return 1; }
		// the motion step on the device: a small forward move, one noise vector per particle
		std::vector<std::array<double, 6>> noise(8, std::array<double, 6>{1e-4, 0, -1e-4, 0, 1e-5, 0});
		nav.UpdateOdometry({0.01, 0, 0, 0, 0, 0.002}, noise, false);
		nav.SlamUpdate(z, 0.5);   // now detected
		monorfs::Map m1 = nav.BestMapModel();
		double sumw = 0;
		for (auto& g : m1) sumw += g.weight;
		std::vector<double> w = nav.VehicleWeights();
		double s = 0;
		for (double x : w) s += x;
		if (std::fabs(s - 1.0) > 1e-9 && s != 0) { std::printf("weights do not sum to one: %g\n", s); return 1; }
		std::printf("host smoke ok: %zu components, expected size %.3f, best particle %d\n", m1.size(), sumw, nav.BestParticle());
		return 0;
	}
	catch (const monorfs::PhdError& e) {
		std::printf("PhdError status=%d module=%s: %s\n", e.status, e.module.c_str(), e.what());
		return 3;
	}
}
