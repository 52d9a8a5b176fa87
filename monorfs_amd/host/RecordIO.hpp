// RecordIO.hpp — the reference's record / replay wire formats, host side, C++.
//
// Readers follow mono-rfs-lib/Util/FileParser.cs, Util.ParseDictionary (Util.cs:232-264) and
// SimulatedVehicle.FromFile (SimulatedVehicle.cs:346-385); writers follow Simulation.Serialized* (Simulation.cs:
// 155-231) and Gaussian.ToString("g6") (Gaussian.cs:391-431). With them the same measurements.out / odometry.out
// stream can drive the C# solver (`-i=record`) and the HIP solver (scripts/replay.py does the latter through the
// ctypes mirror; a C++ host uses this header). The zip container of Simulation.SaveToFile (Simulation.cs:391-488) is
// left to the caller: these functions take and return the text of its members
//     scene.world  trajectory.out  odometry.out  measurements.out  estimate.out  maps.out  tags.out
//
// Errors: FormatError carries the reference's own FormatException messages.
#pragma once
#include "PHDNavigator.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace monorfs {
namespace recordio {

struct FormatError : std::runtime_error {
	explicit FormatError(const std::string& what) : std::runtime_error(what) {}
};

typedef std::vector<std::pair<double, std::vector<double>>>              TimedArray;          // time, vector
typedef std::vector<std::pair<double, std::vector<std::vector<double>>>> TimedMeasurements;   // time, points
typedef std::vector<std::pair<double, Map>>                              TimedMapModel;       // time, map
typedef std::vector<std::pair<double, TimedArray>>                       TimedTrajectory;     // time, trajectory so far
typedef std::vector<std::pair<double, std::string>>                      TimedMessage;        // time, text

// string.Split(separator) with or without StringSplitOptions.RemoveEmptyEntries
inline std::vector<std::string> Split(const std::string& s, const std::string& sep, bool removeempty)
{
	std::vector<std::string> out;
	size_t from = 0;
	for (;;) {
		size_t at = s.find(sep, from);
		std::string part = s.substr(from, at == std::string::npos ? std::string::npos : at - from);
		if (!(removeempty && part.empty())) out.push_back(part);
		if (at == std::string::npos) break;
		from = at + sep.size();
	}
	return out;
}

// double.Parse: the whole token must be a number
inline double ParseDouble(const std::string& token, const std::string& error)
{
	if (token.empty()) throw FormatError(error);
	char* end = nullptr;
	double v = std::strtod(token.c_str(), &end);
	while (end && (*end == ' ' || *end == '\t' || *end == '\r')) end++;   // double.Parse allows surrounding white space
	if (!end || *end != 0) throw FormatError(error);
	return v;
}

// FileParser.ParseDoubleList (FileParser.cs:279-294): space separated, empty entries dropped
inline std::vector<double> ParseDoubleList(const std::string& descriptor)
{
	std::vector<double> point;
	for (const std::string& v : Split(descriptor, " ", true)) {
		point.push_back(ParseDouble(v, "the double descriptor '" + descriptor + "' is malformed"));
	}
	return point;
}

// FileParser.TimedArrayFromDescriptor (:104-119): lines "t v1 ... vdim" (trajectory.out, odometry.out)
inline TimedArray TimedArrayFromDescriptor(const std::vector<std::string>& lines, int dim)
{
	TimedArray array;
	for (const std::string& line : lines) {
		std::vector<double> values = ParseDoubleList(line);
		if ((int) values.size() != dim + 1) throw FormatError("wrong state dimension");
		array.emplace_back(values[0], std::vector<double>(values.begin() + 1, values.end()));
	}
	return array;
}

// FileParser.MeasurementsFromDescriptor (:179-230): lines "t:x y r;x y r;..." (measurements.out). Like the reference,
// every line must carry the ':' (no trailing newline), points are split on single spaces.
inline TimedMeasurements MeasurementsFromDescriptor(const std::string& descriptor, int dim)
{
	TimedMeasurements history;
	for (const std::string& frame : Split(descriptor, "\n", false)) {
		std::vector<std::string> parts = Split(frame, ":", false);
		if (parts.size() != 2) throw FormatError("bad measurement format: no ':' delimiter found");
		double time = ParseDouble(parts[0], "bad measurement format: missing time");
		std::vector<std::vector<double>> measurements;
		for (const std::string& point : Split(parts[1], ";", false)) {
			if (point.empty()) continue;
			std::vector<std::string> strcomps = Split(point, " ", false);
			if ((int) strcomps.size() != dim) throw FormatError("wrong measurement dimension");
			std::vector<double> components;
			for (const std::string& c : strcomps) components.push_back(ParseDouble(c, "bad measurement format: invalid point"));
			measurements.push_back(components);
		}
		history.emplace_back(time, measurements);
	}
	return history;
}

// FileParser.ParseGaussianDescriptor (:302-339): "w;m1 m2 m3;c11 c12 ... c33"
inline Gaussian ParseGaussianDescriptor(const std::string& descriptor, int dim = 3)
{
	const std::string bad = "the double descriptor '" + descriptor + "' is malformed";
	std::vector<std::string> parts = Split(descriptor, ";", false);
	if (parts.size() < 3) throw FormatError(bad);
	Gaussian g;
	g.weight = ParseDouble(parts[0], bad);
	std::vector<std::string> meanvals = Split(parts[1], " ", false), covvals = Split(parts[2], " ", false);
	if (covvals.size() != meanvals.size() * meanvals.size()) throw FormatError("covariance has the wrong size");
	if ((int) meanvals.size() != dim) throw FormatError("wrong gaussian dimension");   // MapFromDescriptor, :164-166
	for (int i = 0; i < 3; i++) g.mean[i] = ParseDouble(meanvals[i], bad);
	for (int i = 0; i < 9; i++) g.covariance[i] = ParseDouble(covvals[i], bad);
	return g;
}

// FileParser.MapFromDescriptor (:156-170)
inline Map MapFromDescriptor(const std::vector<std::string>& lines, int dim = 3)
{
	Map map;
	for (const std::string& line : lines) map.push_back(ParseGaussianDescriptor(line, dim));
	return map;
}

// FileParser.MapHistoryFromDescriptor (:128-148): frames separated by "\n|\n", first line = time (maps.out)
inline TimedMapModel MapHistoryFromDescriptor(const std::string& descriptor, int dim = 3)
{
	TimedMapModel history;
	for (const std::string& frame : Split(descriptor, "\n|\n", true)) {
		std::vector<std::string> lines = Split(frame, "\n", true);
		if (lines.empty()) throw FormatError("bad map format: missing time");
		double time = ParseDouble(lines[0], "bad map format: missing time");
		history.emplace_back(time, MapFromDescriptor(std::vector<std::string>(lines.begin() + 1, lines.end()), dim));
	}
	return history;
}

// FileParser.TrajectoryHistoryFromDescriptor (:65-95) (estimate.out)
inline TimedTrajectory TrajectoryHistoryFromDescriptor(const std::string& descriptor, int dim, bool filterhistory = false)
{
	TimedTrajectory history;
	TimedArray filtered;
	for (const std::string& frame : Split(descriptor, "\n|\n", true)) {
		std::vector<std::string> lines = Split(frame, "\n", true);
		if (lines.empty()) throw FormatError("bad trajectory format: missing time");
		double time = ParseDouble(lines[0], "bad trajectory format: missing time");
		TimedArray trajectory = TimedArrayFromDescriptor(std::vector<std::string>(lines.begin() + 1, lines.end()), dim);
		if (filterhistory) {
			if (trajectory.empty()) throw FormatError("bad trajectory format: empty frame");
			filtered.push_back(trajectory.back());
			history.emplace_back(time, filtered);
		}
		else history.emplace_back(time, trajectory);
	}
	return history;
}

// FileParser.TimedMessageFromDescriptor (:238-256): "t message" (tags.out)
inline TimedMessage TimedMessageFromDescriptor(const std::vector<std::string>& lines)
{
	TimedMessage array;
	for (const std::string& line : lines) {
		size_t sp = line.find(' ');
		if (sp == std::string::npos) throw FormatError("the TimedMessage descriptor '" + line + "' is malformed");
		double time = ParseDouble(line.substr(0, sp), "the TimedMessage descriptor '" + line + "' is malformed");
		array.emplace_back(time, line.substr(sp + 1));
	}
	return array;
}

// FileParser.CommandsFromDescriptor (:263-274): one odometry command per line (6 values, optional mode / screenshot fields)
inline std::vector<std::vector<double>> CommandsFromDescriptor(const std::vector<std::string>& commandstr)
{
	std::vector<std::vector<double>> commands;
	for (const std::string& line : commandstr) commands.push_back(ParseDoubleList(line));
	return commands;
}

// Util.ParseDictionary (Util.cs:232-264): key lines have no leading white space, value lines start with one tab
inline std::map<std::string, std::vector<std::string>> ParseDictionary(std::string descriptor)
{
	std::map<std::string, std::vector<std::string>> dictionary;
	std::string norm;
	for (size_t i = 0; i < descriptor.size(); i++) {
		if (descriptor[i] == '\r') { norm += '\n'; if (i + 1 < descriptor.size() && descriptor[i + 1] == '\n') i++; }
		else norm += descriptor[i];
	}
	std::vector<std::string> lines = Split(norm, "\n", false);
	auto blank = [](const std::string& l) { return l.find_first_not_of(" \t\r\n\f\v") == std::string::npos; };
	if (!lines.empty() && (lines[0].empty() || lines[0][0] == ' ' || lines[0][0] == '\t')) return dictionary;   // can't start with a child
	std::string key;
	for (const std::string& line : lines) {
		if (blank(line)) continue;
		if (line[0] != '\t') { key = line; dictionary[key]; }
		else dictionary[key].push_back(line.substr(1));
	}
	return dictionary;
}

// SimulatedVehicle.FromFile (SimulatedVehicle.cs:346-385): scene.world
struct Scene {
	Pose3D pose;
	bool   hasparams;
	std::array<double, 7> params;   // PRM3DMeasurer.FromLinear: focal rangemin rangemax filmX filmY filmW filmH (:103-114)
	std::vector<std::array<double, 3>> landmarks;
};

inline Scene SceneFromDescriptor(const std::string& descriptor)
{
	std::map<std::string, std::vector<std::string>> dict = ParseDictionary(descriptor);
	if (!dict.count("pose") || dict["pose"].empty() || !dict.count("landmarks")) throw FormatError("scene: missing pose or landmarks");
	Scene s;
	std::vector<double> pose = ParseDoubleList(dict["pose"][0]);
	if (pose.size() != 7) throw FormatError("wrong state dimension");
	for (int i = 0; i < 7; i++) s.pose[i] = pose[i];
	const char* key = dict.count("focal") ? "focal" : (dict.count("params") ? "params" : "");   // "focal" is the deprecated alias
	s.hasparams = key[0] != 0 && !dict[key].empty();
	s.params.fill(0);
	if (s.hasparams) {
		std::vector<double> m = ParseDoubleList(dict[key][0]);
		if (m.size() != 7) throw FormatError("wrong measurer parameter count");
		for (int i = 0; i < 7; i++) s.params[i] = m[i];
	}
	for (const std::string& line : dict["landmarks"]) {
		std::vector<double> lm = ParseDoubleList(line);
		if (lm.size() != 3) throw FormatError("Map landmarks must be 3D");
		s.landmarks.push_back({lm[0], lm[1], lm[2]});
	}
	return s;
}

// ---- writers --------------------------------------------------------------------------------------------------
// double.ToString("g6"): six significant digits, scientific with a two-digit exponent outside [1e-5, 1e6) — what
// printf's %.6g prints
inline std::string G6(double x)
{
	char buf[64];
	std::snprintf(buf, sizeof buf, "%.6g", x);
	return buf;
}

// double.ToString(): the shortest of 15 significant digits ("G15" of the .NET Framework), used for measurements
inline std::string G15(double x)
{
	char buf[64];
	std::snprintf(buf, sizeof buf, "%.15g", x);
	return buf;
}

// Gaussian.ToString("g6") (Gaussian.cs:391-431)
inline std::string GaussianToString(const Gaussian& g)
{
	std::string s = G6(g.weight) + ";" + G6(g.mean[0]) + " " + G6(g.mean[1]) + " " + G6(g.mean[2]) + ";";
	for (int i = 0; i < 9; i++) s += (i ? " " : "") + G6(g.covariance[i]);
	return s;
}

// Simulation.SerializeWayPoints / SerializedOdometry (Simulation.cs:155-166, 225-231)
inline std::string SerializeTimedArray(const TimedArray& a)
{
	std::string s;
	for (size_t i = 0; i < a.size(); i++) {
		if (i) s += "\n";
		s += G6(a[i].first);
		for (double v : a[i].second) s += " " + G6(v);
	}
	return s;
}

// Simulation.SerializedMeasurements (:186-193)
inline std::string SerializeMeasurements(const TimedMeasurements& m)
{
	std::string s;
	for (size_t i = 0; i < m.size(); i++) {
		if (i) s += "\n";
		s += G6(m[i].first) + ":";
		for (size_t k = 0; k < m[i].second.size(); k++) {
			if (k) s += ";";
			for (size_t c = 0; c < m[i].second[k].size(); c++) s += (c ? " " : "") + G15(m[i].second[k][c]);
		}
	}
	return s;
}

// Vehicle.ToString("g6") (Vehicle.cs:513-524): the scene file (scene.world of a record, `-f=` of the command line)
inline std::string SerializeScene(const Scene& scene)
{
	std::string s = "pose\n\t";
	for (size_t i = 0; i < scene.pose.size(); i++) s += (i ? " " : "") + G6(scene.pose[i]);
	s += "\nparams\n\t";
	for (size_t i = 0; i < scene.params.size(); i++) s += (i ? " " : "") + G6(scene.params[i]);
	s += "\nlandmarks\n\t";
	for (size_t l = 0; l < scene.landmarks.size(); l++) {
		if (l) s += "\n\t";
		for (size_t i = 0; i < scene.landmarks[l].size(); i++) s += (i ? " " : "") + G6(scene.landmarks[l][i]);
	}
	return s + "\n";
}

// Simulation.SerializedMaps (:199-206)
inline std::string SerializeMaps(const TimedMapModel& maps)
{
	std::string s;
	for (size_t i = 0; i < maps.size(); i++) {
		if (i) s += "\n|\n";
		s += G6(maps[i].first);
		for (const Gaussian& g : maps[i].second) s += "\n" + GaussianToString(g);
	}
	return s;
}

// Simulation.SerializedEstimate (:172-181)
inline std::string SerializeTrajectories(const TimedTrajectory& t)
{
	std::string s;
	for (size_t i = 0; i < t.size(); i++) {
		if (i) s += "\n|\n";
		s += G6(t[i].first) + "\n" + SerializeTimedArray(t[i].second);
	}
	return s;
}

// ---- Config (mono-rfs-lib/Config.cs): `FieldName: value` lines, matrices in Octave syntax ---------------------------------
// The fields the PHD path reads are typed; every other field of the reference's Config travels as text, so that a
// configuration read from a record is written back whole.
struct Config {
	std::string Model = "PRM3D";
	std::vector<std::vector<double>> MotionCovariance = {{5e-3, 0, 0, 0, 0, 0}, {0, 5e-3, 0, 0, 0, 0}, {0, 0, 5e-3, 0, 0, 0},
	                                                     {0, 0, 0, 2e-4, 0, 0}, {0, 0, 0, 0, 2e-4, 0}, {0, 0, 0, 0, 0, 2e-4}};
	std::vector<std::vector<double>> MeasurementCovariance = {{2.0, 0, 0}, {0, 2.0, 0}, {0, 0, 1e-3}};   // SetPRM3DDefaults, :238-263
	std::vector<std::vector<double>> BirthCovariance = {{1e-2, 0, 0}, {0, 1e-2, 0}, {0, 0, 1e-2}};
	std::vector<double> VisibilityRamp = {3 * std::sqrt(2.0), 3 * std::sqrt(2.0), 3 * std::sqrt(1e-3)};
	double DetectionProbability = 0.9, ClutterDensity = 3e-7, DensityDistanceThreshold = 0.5;
	double BirthWeight = 0.05, MinWeight = 1e-3, MinEffectiveParticle = 0.1, MergeThreshold = 0.3, ExplorationThreshold = 1e-5;
	double MotionCovarianceMultiplier = 1.0, MeasurementCovarianceMultiplier = 1.0, NavigatorPD = 0.9, NavigatorClutterDensity = 3e-7;
	double GradientAscentRate = 1e-2, GradientClip = 10;
	int    MaxQuantity = 600;
	bool   PerfectStill = false;
	std::vector<std::pair<std::string, std::string>> Others;   // fields outside the path (NParallel, MapClip, ...), verbatim
};

inline std::string Trim(const std::string& s)
{
	size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
	return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}

// `[a b; c d]` (Accord's OctaveMatrixFormatProvider, the syntax Config.FromDescriptor hands to Matrix.ParseJagged)
inline std::vector<std::vector<double>> ParseOctaveMatrix(const std::string& text)
{
	std::string t = Trim(text);
	if (!t.empty() && t.front() == '[') t = t.substr(1);
	if (!t.empty() && t.back() == ']') t.pop_back();
	std::vector<std::vector<double>> rows;
	for (const std::string& row : Split(t, ";", false)) {
		if (Trim(row).empty()) continue;
		std::vector<double> r;
		std::string clean = row;
		for (char& c : clean) if (c == ',') c = ' ';
		for (const std::string& v : Split(clean, " ", true)) r.push_back(ParseDouble(Trim(v), "the matrix descriptor '" + text + "' is malformed"));
		rows.push_back(r);
	}
	return rows;
}

// Config.FromDescriptor (Config.cs:155-209): unknown fields are ignored by the reference (kept in Others here), a missing
// parameter is left as it is, a line without a colon is reported (skipped lines land in `skipped`) and skipped
inline void ConfigFromDescriptor(const std::vector<std::string>& lines, Config& c, std::vector<std::string>* skipped = nullptr)
{
	for (const std::string& line : lines) {
		size_t colon = line.find(':');
		if (colon == std::string::npos) {
			if (skipped) skipped->push_back(line);
			continue;
		}
		const std::string name = Trim(line.substr(0, colon)), value = Trim(line.substr(colon + 1));
		auto num = [&]() { return ParseDouble(value, "Input string was not in a correct format."); };
		auto boolean = [&]() {
			std::string v = value;
			for (char& ch : v) ch = (char) std::tolower((unsigned char) ch);
			if (v != "true" && v != "false") throw FormatError("String was not recognized as a valid Boolean.");
			return v == "true";
		};
		if      (name == "Model") c.Model = value;
		else if (name == "MotionCovariance") c.MotionCovariance = ParseOctaveMatrix(value);
		else if (name == "MeasurementCovariance") c.MeasurementCovariance = ParseOctaveMatrix(value);
		else if (name == "BirthCovariance") c.BirthCovariance = ParseOctaveMatrix(value);
		else if (name == "VisibilityRamp") c.VisibilityRamp = ParseOctaveMatrix(value).at(0);
		else if (name == "DetectionProbability") c.DetectionProbability = num();
		else if (name == "ClutterDensity") c.ClutterDensity = num();
		else if (name == "DensityDistanceThreshold") c.DensityDistanceThreshold = num();
		else if (name == "BirthWeight") c.BirthWeight = num();
		else if (name == "MinWeight") c.MinWeight = num();
		else if (name == "MinEffectiveParticle") c.MinEffectiveParticle = num();
		else if (name == "MergeThreshold") c.MergeThreshold = num();
		else if (name == "ExplorationThreshold") c.ExplorationThreshold = num();
		else if (name == "MotionCovarianceMultiplier") c.MotionCovarianceMultiplier = num();
		else if (name == "MeasurementCovarianceMultiplier") c.MeasurementCovarianceMultiplier = num();
		else if (name == "NavigatorPD") c.NavigatorPD = num();
		else if (name == "NavigatorClutterDensity") c.NavigatorClutterDensity = num();
		else if (name == "GradientAscentRate") c.GradientAscentRate = num();
		else if (name == "GradientClip") c.GradientClip = num();
		else if (name == "MaxQuantity") {
			char* end = nullptr;
			long v = std::strtol(value.c_str(), &end, 10);
			if (value.empty() || *end) throw FormatError("Input string was not in a correct format.");
			c.MaxQuantity = (int) v;
		}
		else if (name == "PerfectStill") c.PerfectStill = boolean();
		else {
			bool found = false;
			for (auto& o : c.Others) if (o.first == name) { o.second = value; found = true; }
			if (!found) c.Others.emplace_back(name, value);
		}
	}
}

// double.ToString(): "3E-07" where printf writes "3e-07"
inline std::string NetDouble(double x)
{
	std::string s = G15(x);
	for (char& ch : s) if (ch == 'e') ch = 'E';
	return s;
}

inline std::string OctaveMatrix(const std::vector<std::vector<double>>& m)
{
	std::string s = "[";
	for (size_t i = 0; i < m.size(); i++) {
		if (i) s += "; ";
		for (size_t k = 0; k < m[i].size(); k++) s += (k ? " " : "") + NetDouble(m[i][k]);
	}
	return s + "]";
}

// Config.ToString (Config.cs:268-309) for the typed fields, then the others as they came
inline std::string SerializeConfig(const Config& c)
{
	std::string s;
	auto line = [&](const std::string& name, const std::string& value) { s += (s.empty() ? "" : "\n") + name + ": " + value; };
	line("Model", c.Model);
	line("MotionCovariance", OctaveMatrix(c.MotionCovariance));
	line("MeasurementCovariance", OctaveMatrix(c.MeasurementCovariance));
	line("DetectionProbability", NetDouble(c.DetectionProbability));
	line("ClutterDensity", NetDouble(c.ClutterDensity));
	line("PerfectStill", c.PerfectStill ? "True" : "False");
	line("VisibilityRamp", OctaveMatrix({c.VisibilityRamp}));
	line("DensityDistanceThreshold", NetDouble(c.DensityDistanceThreshold));
	line("BirthCovariance", OctaveMatrix(c.BirthCovariance));
	line("BirthWeight", NetDouble(c.BirthWeight));
	line("MinWeight", NetDouble(c.MinWeight));
	line("MinEffectiveParticle", NetDouble(c.MinEffectiveParticle));
	line("MaxQuantity", std::to_string(c.MaxQuantity));
	line("MergeThreshold", NetDouble(c.MergeThreshold));
	line("ExplorationThreshold", NetDouble(c.ExplorationThreshold));
	line("MotionCovarianceMultiplier", NetDouble(c.MotionCovarianceMultiplier));
	line("MeasurementCovarianceMultiplier", NetDouble(c.MeasurementCovarianceMultiplier));
	line("NavigatorPD", NetDouble(c.NavigatorPD));
	line("NavigatorClutterDensity", NetDouble(c.NavigatorClutterDensity));
	line("GradientAscentRate", NetDouble(c.GradientAscentRate));
	line("GradientClip", NetDouble(c.GradientClip));
	for (const auto& o : c.Others) line(o.first, o.second);
	return s;
}

// The values PHDNavigator reads from Config as the parameter block of libphdhip: its particles are clones of the
// reference vehicle with MeasurementCovarianceMultiplier, NavigatorPD and NavigatorClutterDensity (PHDNavigator.cs:257-259)
inline phd_params PhdParamsFromConfig(const Config& c, int maxparticles, int maxcomponents, int maxmeasurements)
{
	if (c.Model != "PRM3D") throw FormatError("libphdhip implements the PRM3D model");
	if (c.MeasurementCovariance.size() != 3 || c.BirthCovariance.size() != 3 || c.VisibilityRamp.size() < 3) throw FormatError("the PRM3D model needs 3 x 3 covariances and a visibility ramp of 3");
	phd_params p;
	phd_default_params(&p, maxparticles, std::max(maxcomponents, c.MaxQuantity), maxmeasurements);
	for (int i = 0; i < 3; i++) {
		for (int k = 0; k < 3; k++) {
			p.R[i * 3 + k] = c.MeasurementCovarianceMultiplier * c.MeasurementCovariance[i].at(k);
			p.birth_covariance[i * 3 + k] = c.BirthCovariance[i].at(k);
		}
		p.visibility_ramp[i] = c.VisibilityRamp[i];
	}
	p.pd = c.NavigatorPD; p.clutter_density = c.NavigatorClutterDensity;
	p.birth_weight = c.BirthWeight; p.min_weight = c.MinWeight; p.min_effective_particle = c.MinEffectiveParticle;
	p.max_quantity = c.MaxQuantity; p.merge_threshold = c.MergeThreshold; p.exploration_threshold = c.ExplorationThreshold;
	p.density_distance_threshold = c.DensityDistanceThreshold;
	return p;
}

// Manipulator.SerializedTags (Manipulator.cs:294-304)
inline std::string SerializeTags(const TimedMessage& tags)
{
	std::string s;
	for (size_t i = 0; i < tags.size(); i++) s += (i ? "\n" : "") + G6(tags[i].first) + " " + tags[i].second;
	return s;
}

}  // namespace recordio
}  // namespace monorfs
