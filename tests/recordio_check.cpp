// Checks monorfs_amd/host/RecordIO.hpp against the hand-written record members of tests/golden/record/:
// parse, compare with the values the files were written from, serialise back and compare with the text.
#include "../monorfs_amd/host/RecordIO.hpp"

#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>

using namespace monorfs;
using namespace monorfs::recordio;

static std::string slurp(const std::string& path)
{
	std::ifstream f(path);
	std::stringstream ss;
	ss << f.rdbuf();
	return ss.str();
}

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

template <class F> static std::string error_of(F f)
{
	try { f(); } catch (const FormatError& e) { return e.what(); }
	return "";
}

int main(int argc, char** argv)
{
	const std::string dir = argc > 1 ? argv[1] : "tests/golden/record";

	Scene scene = SceneFromDescriptor(slurp(dir + "/scene.world"));
	CHECK(scene.pose[0] == 0.1 && scene.pose[3] == 1 && scene.hasparams && scene.params[0] == 575.816 && scene.params[6] == 480);
	CHECK(scene.landmarks.size() == 3 && scene.landmarks[2][1] == -0.4);
	CHECK(SerializeScene(scene) == slurp(dir + "/scene.world"));
	CHECK(SceneFromDescriptor("pose\n\t0 0 0 1 0 0 0\nfocal\n\t500 0.1 2 -320 -240 640 480\nlandmarks\n").params[0] == 500);   // deprecated alias
	CHECK(error_of([] { SceneFromDescriptor("pose\n\t0 0 0 1 0 0 0\nlandmarks\n\t1 2\n"); }) == "Map landmarks must be 3D");
	CHECK(ParseDictionary("\tchild first\nkey\n").empty());   // can't start with a child node (Util.cs:243-246)

	const std::string odotext = slurp(dir + "/odometry.out");
	TimedArray odo = TimedArrayFromDescriptor(Split(odotext, "\n", true), 6);
	CHECK(odo.size() == 3 && odo[2].first == 0.0666667 && odo[2].second[2] == 1.5e-06 && odo[2].second[5] == -1.25e-05);
	CHECK(SerializeTimedArray(odo) == odotext);
	CHECK(error_of([&] { TimedArrayFromDescriptor(Split(odotext, "\n", true), 7); }) == "wrong state dimension");

	const std::string ztext = slurp(dir + "/measurements.out");
	TimedMeasurements z = MeasurementsFromDescriptor(ztext, 3);
	CHECK(z.size() == 3 && z[0].second.empty() && z[1].second.size() == 2 && z[1].second[0][2] == 1.125 && z[2].second[0][0] == 100.123456789012);
	CHECK(SerializeMeasurements(z) == ztext);
	CHECK(error_of([] { MeasurementsFromDescriptor("0.1 1 2 3", 3); }) == "bad measurement format: no ':' delimiter found");
	CHECK(error_of([] { MeasurementsFromDescriptor("0.1:1 2", 3); }) == "wrong measurement dimension");
	CHECK(error_of([] { MeasurementsFromDescriptor("x:1 2 3", 3); }) == "bad measurement format: missing time");
	CHECK(error_of([] { MeasurementsFromDescriptor("0.1:1 2 y", 3); }) == "bad measurement format: invalid point");

	const std::string mtext = slurp(dir + "/maps.out");
	TimedMapModel maps = MapHistoryFromDescriptor(mtext, 3);
	CHECK(maps.size() == 2 && maps[0].second.size() == 2 && maps[1].second.size() == 1);
	CHECK(maps[0].second[1].weight == 0.05 && maps[0].second[1].covariance[1] == 1e-05 && maps[1].second[0].covariance[8] == 1e12);
	CHECK(SerializeMaps(maps) == mtext);
	CHECK(error_of([] { ParseGaussianDescriptor("1;0 0 0;1 0 0 1"); }) == "covariance has the wrong size");
	CHECK(error_of([] { ParseGaussianDescriptor("1;0 0;1 0 0 1"); }) == "wrong gaussian dimension");

	const std::string etext = slurp(dir + "/estimate.out");
	TimedTrajectory est = TrajectoryHistoryFromDescriptor(etext, 7);
	CHECK(est.size() == 2 && est[1].second.size() == 3 && est[1].second[2].second[3] == 0.999998);
	CHECK(SerializeTrajectories(est) == etext);
	TimedTrajectory filt = TrajectoryHistoryFromDescriptor(etext, 7, true);
	CHECK(filt[1].second.size() == 2 && filt[1].second[1].first == 0.0666667);   // one (the newest) pose per frame

	TimedMessage tags = TimedMessageFromDescriptor(Split(slurp(dir + "/tags.out"), "\n", true));
	CHECK(tags.size() == 2 && tags[0].second == "SLAM mode on" && tags[1].first == 1.25 && tags[1].second == "Mapping mode on");

	std::vector<std::vector<double>> cmd = CommandsFromDescriptor({"0.01 0 0 0 0.002 0", "0 0 0 0 0 0 1", "0 0 0 0 0 0 -1 1 0.5 0.2 3"});
	CHECK(cmd[0].size() == 6 && cmd[1][6] == 1 && cmd[2].size() == 11);

	CHECK(G6(1234567.0) == "1.23457e+06" && G6(0.000012345678) == "1.23457e-05" && G6(0.5) == "0.5" && G6(100000.0) == "100000");
	// Config (Config.cs:155-209, 268-309) and the parameter block built from it
	{
		Config cfg;
		std::vector<std::string> skipped;
		ConfigFromDescriptor(Split(slurp(dir + "/config.cfg"), "\n", true), cfg, &skipped);
		CHECK(skipped.size() == 1 && skipped[0] == "this line has no colon and is skipped");
		CHECK(cfg.MaxQuantity == 250 && cfg.MinEffectiveParticle == 0.3 && cfg.ClutterDensity == 3e-7 && cfg.NavigatorPD == 0.85);
		CHECK(cfg.MeasurementCovariance[2][2] == 0.002 && cfg.MotionCovariance[5][5] == 0.0002 && cfg.BirthCovariance[1][1] == 1e-2);
		CHECK(cfg.GradientClip == 10 && !cfg.PerfectStill);
		bool future = false;
		for (auto& o : cfg.Others) future = future || (o.first == "SomeFutureField" && o.second == "17");
		CHECK(future);
		std::string text = SerializeConfig(cfg);
		CHECK(text.find("MeasurementCovariance: [2.5 0 0; 0 2.5 0; 0 0 0.002]") != std::string::npos);
		CHECK(text.find("ClutterDensity: 3E-07") != std::string::npos && text.find("PerfectStill: False") != std::string::npos);
		Config again;
		ConfigFromDescriptor(Split(text, "\n", true), again);
		CHECK(SerializeConfig(again) == text);
		phd_params p = PhdParamsFromConfig(cfg, 8, 600, 16);
		CHECK(p.R[0] == 5.0 && p.R[8] == 0.004 && p.pd == 0.85 && p.clutter_density == 6e-7 && p.max_quantity == 250);
		CHECK(p.merge_threshold == 0.5 && p.min_weight == 0.002 && p.birth_weight == 0.04 && p.max_components >= 600);
		CHECK(error_of([&] { Config bad; ConfigFromDescriptor({"MaxQuantity: many"}, bad); }) == "Input string was not in a correct format.");
		CHECK(SerializeTags(tags) == slurp(dir + "/tags.out"));
	}
	std::printf("recordio ok\n");
	return 0;
}
