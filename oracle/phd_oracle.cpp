/*
 * phd_oracle.cpp — CPU restatement (IEEE double, C++17) of monorfs's RB-PHD-SLAM inner loop.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it. The product (libphdhip.so) never links, loads or
 * calls anything in this directory.
 *
 * Why a restatement: the reference is C#/.NET 4.5 (Mono); this image has no mono/mcs/dotnet and
 * the NuGet dependencies (Accord 3.0.2, AForge 2.2.5) are not vendored, so the reference cannot
 * be built or run here. Each function below cites the reference lines it follows.
 *
 * Parity status ("pinned" = checked against golden vectors of the reference's own NUnit tests,
 * the .json files of tests/golden, by tests/test_oracle_kat.py):
 *   pinned   : PredictConditional, CorrectConditional (ungated formula), PruneModel/Merge
 *              (PHDNavigatorTest.cs:85-265, Linear2D model); Hungarian, Murty order, lexicographic
 *              order, MurtyNode children, connected components, AssignmentValue
 *              (GraphCombinatoricsTest.cs:66-404); systematic resampling (SimulationTest.cs:225-270).
 *   UNPINNED : WeightAlpha, BestMapEstimate, SetLogLikelihood values, the PRM3D measurement model
 *              inside CorrectConditional and FuzzyVisibleM have no numeric test in the reference;
 *              for those this file's reading of the source is the definition ("parity unpinned").
 *
 * Third-party arithmetic that is not under /root/reference and is replaced by a stated
 * canonical form (all "parity unpinned" at that boundary):
 *   - Accord.Math 3.0.2 PseudoInverse / PseudoDeterminant (SVD; Gaussian.cs:152-153): here the
 *     closed-form inverse and |det| of the full (un-symmetrised) 2x2 / 3x3 matrix, which is what
 *     the SVD forms equal for a full-rank matrix.
 *   - Accord.MachineLearning 3.0.2 KDTree<T> enumeration order and Nearest(point, radius)
 *     (Map.cs:46,93,173,196,214): canonical order = insertion order; the radius test is exact
 *     (no tree pruning) with the metric chosen by phd_params.gate_metric, `<=`.
 *   - .NET List.Sort (unstable introsort; PHDNavigator.cs:920, Map.cs:129,137,
 *     GraphCombinatorics.cs:641): canonical = stable sort (ties keep list order).
 *   - .NET Dictionary enumeration order (SparseMatrix.cs:48, `Any` :150-174): canonical =
 *     insertion order, which is what the CLR does when nothing is re-inserted after a removal.
 *
 * Quirks of the reference that are reproduced on purpose (SURVEY.md "Notes for the oracle author"):
 *   integer division in the Gaussian multiplier exponent, float32 range clip, stale `logcomp`
 *   in SetLogLikelihood, `modelsize` compared against compacted indices, non-Joseph un-symmetrised
 *   covariance update, raw-second-moment Merge.
 */
#include "../include/phdhip.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

const double INF = std::numeric_limits<double>::infinity();
const double PI  = 3.14159265358979323846;  // Math.PI

// ---------------------------------------------------------------------------------------------
// small dense algebra (Accord jagged-array semantics, same summation order: k ascending)
// ---------------------------------------------------------------------------------------------

// inverse and determinant of a full dim x dim matrix (dim = 2 or 3), row-major.
// Stands in for Accord PseudoInverse / PseudoDeterminant (Gaussian.cs:152-153).
void invdet(const double* a, int dim, double* inv, double* det)
{
	if (dim == 3) {
		double c00 = a[4] * a[8] - a[5] * a[7];
		double c01 = a[3] * a[8] - a[5] * a[6];
		double c02 = a[3] * a[7] - a[4] * a[6];
		double d   = a[0] * c00 - a[1] * c01 + a[2] * c02;
		double id  = 1.0 / d;
		inv[0] = c00 * id;
		inv[1] = (a[2] * a[7] - a[1] * a[8]) * id;
		inv[2] = (a[1] * a[5] - a[2] * a[4]) * id;
		inv[3] = (a[5] * a[6] - a[3] * a[8]) * id;
		inv[4] = (a[0] * a[8] - a[2] * a[6]) * id;
		inv[5] = (a[2] * a[3] - a[0] * a[5]) * id;
		inv[6] = c02 * id;
		inv[7] = (a[1] * a[6] - a[0] * a[7]) * id;
		inv[8] = (a[0] * a[4] - a[1] * a[3]) * id;
		*det = d;
	}
	else if (dim == 2) {
		double d  = a[0] * a[3] - a[1] * a[2];
		double id = 1.0 / d;
		inv[0] =  a[3] * id;
		inv[1] = -a[1] * id;
		inv[2] = -a[2] * id;
		inv[3] =  a[0] * id;
		*det = d;
	}
	else {
		double d = a[0];
		inv[0] = 1.0 / d;
		*det = d;
	}
}

// Gaussian.cs:155 — Math.Pow(2 * Math.PI, -mean.Length / 2) / Math.Sqrt(det); `-mean.Length / 2`
// is an integer division: -1 for dim 3 and dim 2, 0 for dim 1. The pseudo-determinant of a
// full-rank matrix is the product of its singular values = |det|.
double multiplier(int dim, double det)
{
	int e = -dim / 2;   // C++ integer division truncates toward zero exactly like C#
	return std::pow(2 * PI, (double) e) / std::sqrt(std::fabs(det));
}

// Gaussian.Evaluate (Gaussian.cs:199-204) without the weight: mult * exp(-0.5 d^T Pinv d),
// d = x - mean; the quadratic form is d . (Pinv d) with rows accumulated k ascending.
double quadform(const double* pinv, const double* d, int dim)
{
	double q = 0;
	for (int i = 0; i < dim; i++) {
		double r = 0;
		for (int k = 0; k < dim; k++) {
			r += pinv[i * dim + k] * d[k];
		}
		q += d[i] * r;
	}
	return q;
}

// a component with the cached members of Gaussian (Gaussian.cs:49-90)
struct Comp {
	double w;
	double m[3];
	double P[9];
	double Pinv[9];
	double mult;
};

Comp make_comp(const double* m, const double* P, double w)
{
	Comp c;
	c.w = std::isnan(w) ? 0.0 : w;          // Gaussian.cs:154
	std::memcpy(c.m, m, sizeof(c.m));
	std::memcpy(c.P, P, sizeof(c.P));
	double det;
	invdet(P, 3, c.Pinv, &det);
	c.mult = multiplier(3, det);
	return c;
}

double comp_eval(const Comp& c, const double* x)   // Gaussian.Evaluate
{
	double d[3] = {x[0] - c.m[0], x[1] - c.m[1], x[2] - c.m[2]};
	return c.mult * std::exp(-0.5 * quadform(c.Pinv, d, 3));
}

double comp_sqmahal(const Comp& c, const double* x)   // Gaussian.SquareMahalanobis (Gaussian.cs:365-369)
{
	double d[3] = {c.m[0] - x[0], c.m[1] - x[1], c.m[2] - x[2]};
	return quadform(c.Pinv, d, 3);
}

typedef std::vector<Comp> Mixture;   // a Map in canonical (insertion) order

// ---------------------------------------------------------------------------------------------
// measurement models
// ---------------------------------------------------------------------------------------------

struct Quat { double w, x, y, z; };

Quat qmul(const Quat& a, const Quat& b)   // Quaternion.cs:295-301
{
	return Quat{a.w * b.w - (a.x * b.x + a.y * b.y + a.z * b.z),
	            a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
	            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
	            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}

Quat qconj(const Quat& q) { return Quat{q.w, -q.x, -q.y, -q.z}; }   // Quaternion.cs:155-158

void qmatrix(const Quat& q, double* r)   // Quaternion.ToMatrix, Quaternion.cs:327-342
{
	double xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
	double xy = q.x * q.y, xz = q.x * q.z, xw = q.x * q.w;
	double yz = q.y * q.z, yw = q.y * q.w, zw = q.z * q.w;
	r[0] = 1 - 2 * (yy + zz); r[1] = 2 * (xy - zw);     r[2] = 2 * (xz + yw);
	r[3] = 2 * (xy + zw);     r[4] = 1 - 2 * (xx + zz); r[5] = 2 * (yz - xw);
	r[6] = 2 * (xz - yw);     r[7] = 2 * (yz + xw);     r[8] = 1 - 2 * (xx + yy);
}

struct Pose {   // Pose3D state (Pose3D.cs:142-162): the orientation is normalised on construction
	double t[3];
	Quat   q;
};

Pose make_pose(const double* s)
{
	Pose p;
	p.t[0] = s[0]; p.t[1] = s[1]; p.t[2] = s[2];
	double a = 1.0 / std::sqrt(s[3] * s[3] + s[4] * s[4] + s[5] * s[5] + s[6] * s[6]);
	p.q = Quat{a * s[3], a * s[4], a * s[5], a * s[6]};
	return p;
}

struct Model {
	const phd_params* p;
	int    zdim;
	double Rinv[9];
	double Rmult;   // multiplier of a Gaussian with covariance R (SetLogLikeMatrix, PHDNavigator.cs:429)
};

Model make_model(const phd_params* p)
{
	Model m;
	m.p    = p;
	m.zdim = p->zdim;
	double det;
	invdet(p->R, p->zdim, m.Rinv, &det);
	m.Rmult = multiplier(p->zdim, det);
	return m;
}

// MeasurePerfect: PRM3DMeasurer.cs:138-149 / Linear2DMeasurer.cs:110-113
void measure_perfect(const Model& md, const Pose& pose, const double* lm, double* z)
{
	if (md.p->model == PHD_MODEL_LINEAR2D) {
		z[0] = lm[0] - pose.t[0];
		z[1] = lm[1] - pose.t[1];
		return;
	}
	double f = md.p->measurer[0];
	double diff[3] = {lm[0] - pose.t[0], lm[1] - pose.t[1], lm[2] - pose.t[2]};
	Quat local = qmul(qmul(qconj(pose.q), Quat{0, diff[0], diff[1], diff[2]}), pose.q);
	double euclid = std::sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
	int    sgn    = (local.z > 0) - (local.z < 0);   // Math.Sign
	z[2] = sgn * euclid;
	z[0] = f * local.x / local.z;
	z[1] = f * local.y / local.z;
}

// MeasurementJacobianL: PRM3DMeasurer.cs:157-177 / Linear2DMeasurer.cs:115-119. H is zdim x 3.
void jacobian_l(const Model& md, const Pose& pose, const double* lm, double* H)
{
	if (md.p->model == PHD_MODEL_LINEAR2D) {
		H[0] = 1; H[1] = 0; H[2] = 0;
		H[3] = 0; H[4] = 1; H[5] = 0;
		return;
	}
	double f = md.p->measurer[0];
	double diff[3] = {lm[0] - pose.t[0], lm[1] - pose.t[1], lm[2] - pose.t[2]};
	Quat l = qmul(qmul(qconj(pose.q), Quat{0, diff[0], diff[1], diff[2]}), pose.q);
	double mag = ((l.z > 0) ? 1 : -1) * std::sqrt(l.x * l.x + l.y * l.y + l.z * l.z);
	double jp[9] = {f / l.z, 0,       -f * l.x / (l.z * l.z),
	                0,       f / l.z, -f * l.y / (l.z * l.z),
	                l.x / mag, l.y / mag, l.z / mag};
	double jr[9];
	qmatrix(qconj(pose.q), jr);
	for (int i = 0; i < 3; i++) {
		for (int j = 0; j < 3; j++) {
			double s = 0;
			for (int k = 0; k < 3; k++) {
				s += jp[i * 3 + k] * jr[k * 3 + j];
			}
			H[i * 3 + j] = s;
		}
	}
}

// MeasurementJacobianP: PRM3DMeasurer.cs:185-211 / Linear2DMeasurer.cs:133-137. Jp is zdim x odo (odo = 6 / 2).
void jacobian_p(const Model& md, const Pose& pose, const double* lm, double* Jp)
{
	if (md.p->model == PHD_MODEL_LINEAR2D) {
		Jp[0] = -1; Jp[1] = 0;
		Jp[2] = 0;  Jp[3] = -1;
		return;
	}
	double f = md.p->measurer[0];
	double diff[3] = {lm[0] - pose.t[0], lm[1] - pose.t[1], lm[2] - pose.t[2]};
	Quat l = qmul(qmul(qconj(pose.q), Quat{0, diff[0], diff[1], diff[2]}), pose.q);
	double mag = ((l.z > 0) ? 1 : -1) * std::sqrt(l.x * l.x + l.y * l.y + l.z * l.z);
	double jp[9] = {f / l.z, 0,       -f * l.x / (l.z * l.z),
	                0,       f / l.z, -f * l.y / (l.z * l.z),
	                l.x / mag, l.y / mag, l.z / mag};
	// jlocation = -R(q*) ; jrotation = jlocation [diff]_x ; jlocal = [jlocation | jrotation]   (:202-207)
	double rc[9], jloc[9], cross[9] = {0, -diff[2], diff[1],  diff[2], 0, -diff[0],  -diff[1], diff[0], 0}, jlocal[18];
	qmatrix(qconj(pose.q), rc);
	for (int i = 0; i < 9; i++) jloc[i] = -1.0 * rc[i];
	for (int i = 0; i < 3; i++) {
		for (int j = 0; j < 3; j++) {
			double s = 0;
			for (int k = 0; k < 3; k++) s += jloc[i * 3 + k] * cross[k * 3 + j];
			jlocal[i * 6 + j]     = jloc[i * 3 + j];
			jlocal[i * 6 + 3 + j] = s;
		}
	}
	for (int i = 0; i < 3; i++) {
		for (int j = 0; j < 6; j++) {
			double s = 0;
			for (int k = 0; k < 3; k++) s += jp[i * 3 + k] * jlocal[k * 6 + j];
			Jp[i * 6 + j] = s;
		}
	}
}

// MeasureToMap: PRM3DMeasurer.cs:299-312 / Linear2DMeasurer.cs:181-184
void measure_to_map(const Model& md, const Pose& pose, const double* z, double* x)
{
	if (md.p->model == PHD_MODEL_LINEAR2D) {
		x[0] = pose.t[0] + z[0];
		x[1] = pose.t[1] + z[1];
		x[2] = 0;
		return;
	}
	double f = md.p->measurer[0];
	double px = z[0], py = z[1], range = z[2];
	double alpha = range / std::sqrt(f * f + px * px + py * py);
	double diff[3] = {alpha * px, alpha * py, alpha * f};
	Quat r = qmul(qmul(pose.q, Quat{0, diff[0], diff[1], diff[2]}), qconj(pose.q));
	x[0] = pose.t[0] + r.x;
	x[1] = pose.t[1] + r.y;
	x[2] = pose.t[2] + r.z;
}

// FuzzyVisibleM: PRM3DMeasurer.cs:277-291 / Linear2DMeasurer.cs:151-162.
// FilmArea is an XNA Rectangle of ints (Left = X, Right = X + Width, Top = Y, Bottom = Y + Height,
// PRM3DMeasurer.FromLinear :111); RangeClip is an AForge.Range of float32 (:110).
double fuzzy_visible(const Model& md, const double* z)
{
	const double* ramp = md.p->visibility_ramp;
	double mind = INF;
	if (md.p->model == PHD_MODEL_LINEAR2D) {
		double range = md.p->measurer[0];
		mind = std::min(mind, (z[0] - -range) / ramp[0]);
		mind = std::min(mind, (range - z[0]) / ramp[0]);
		mind = std::min(mind, (z[1] - -range) / ramp[1]);
		mind = std::min(mind, (range - z[1]) / ramp[1]);
	}
	else {
		double rmin   = (double) (float) md.p->measurer[1];
		double rmax   = (double) (float) md.p->measurer[2];
		int    left   = (int) md.p->measurer[3];
		int    top    = (int) md.p->measurer[4];
		int    right  = left + (int) md.p->measurer[5];
		int    bottom = top + (int) md.p->measurer[6];
		mind = std::min(mind, (z[0] - left) / ramp[0]);
		mind = std::min(mind, (right - z[0]) / ramp[0]);
		mind = std::min(mind, (z[1] - top) / ramp[1]);
		mind = std::min(mind, (bottom - z[1]) / ramp[1]);
		mind = std::min(mind, (z[2] - rmin) / ramp[2]);
		mind = std::min(mind, (rmax - z[2]) / ramp[2]);
	}
	return std::max(0.0, std::min(1.0, mind));
}

// SimulatedVehicle.DetectionProbabilityM / DetectionProbability (SimulatedVehicle.cs:324-339)
double pd_m(const Model& md, const double* z) { return fuzzy_visible(md, z) * md.p->pd; }

double pd_landmark(const Model& md, const Pose& pose, const double* lm)
{
	double z[3];
	measure_perfect(md, pose, lm, z);
	return pd_m(md, z);
}

// Map.Near / Map.Evaluate(point, radius) gate (Map.cs:170-184, 210-220)
bool is_near(const Model& md, const double* x, const double* m, double radius)
{
	if (md.p->gate_metric == PHD_GATE_DISABLED) {
		return true;
	}
	double d0 = x[0] - m[0], d1 = x[1] - m[1], d2 = x[2] - m[2];
	double sq = d0 * d0 + d1 * d1 + d2 * d2;
	if (md.p->gate_metric == PHD_GATE_SQUARED_EUCLIDEAN) {
		return sq <= radius;
	}
	return std::sqrt(sq) <= radius;
}

// ---------------------------------------------------------------------------------------------
// PHD map update
// ---------------------------------------------------------------------------------------------

// Explored (PHDNavigator.cs:956-959) = Map.Evaluate(x, 3 * DensityDistanceThreshold) >= ExplorationThreshold
bool explored(const Model& md, const Mixture& model, const double* x)
{
	double radius = 3 * md.p->density_distance_threshold;
	double value  = 0;
	for (const Comp& c : model) {
		if (is_near(md, x, c.m, radius)) {
			value += c.w * comp_eval(c, x);   // Map.cs:216
		}
	}
	return value >= md.p->exploration_threshold;
}

// PredictConditional (PHDNavigator.cs:793-819). `births` receives the unexplored candidates.
Mixture predict(const Model& md, const Pose& pose, const Mixture& model, const double* z, int M,
                std::vector<int>* bornfrom = nullptr)
{
	Mixture predicted(model);   // new Map(model), :802
	for (int k = 0; k < M; k++) {
		double cand[3];
		measure_to_map(md, pose, z + k * md.zdim, cand);   // :807
		if (!explored(md, model, cand)) {                  // :808 — tested against the PRIOR map
			predicted.push_back(make_comp(cand, md.p->birth_covariance, md.p->birth_weight));   // :815
			if (bornfrom) {
				bornfrom->push_back(k);
			}
		}
	}
	return predicted;
}

// CorrectConditional (PHDNavigator.cs:829-906)
Mixture correct(const Model& md, const Pose& pose, const Mixture& model, const double* z, int M)
{
	const int zd = md.zdim;
	const int n  = (int) model.size();
	Mixture corrected;
	corrected.reserve(n);

	// misdetection copies, :837-840
	for (const Comp& c : model) {
		Comp r = c;   // Reweight = MemberwiseClone with a new weight (Gaussian.cs:186-192)
		r.w = (1 - pd_landmark(md, pose, c.m)) * c.w;
		corrected.push_back(r);
	}

	// per-component measurement-space quantities, :857-870
	std::vector<double> mp(n * 3), H(n * 9), PH(n * 9), Sinv(n * 9), qmult(n), PD(n);
	for (int i = 0; i < n; i++) {
		const Comp& c = model[i];
		measure_perfect(md, pose, c.m, &mp[i * 3]);
		jacobian_l(md, pose, c.m, &H[i * 9]);
		// PH = P.MultiplyByTranspose(H): 3 x zd
		for (int a = 0; a < 3; a++) {
			for (int b = 0; b < zd; b++) {
				double s = 0;
				for (int k = 0; k < 3; k++) {
					s += c.P[a * 3 + k] * H[i * 9 + b * 3 + k];
				}
				PH[i * 9 + a * zd + b] = s;
			}
		}
		// S = H.Multiply(PH).Add(R): zd x zd
		double S[9];
		for (int a = 0; a < zd; a++) {
			for (int b = 0; b < zd; b++) {
				double s = 0;
				for (int k = 0; k < 3; k++) {
					s += H[i * 9 + a * 3 + k] * PH[i * 9 + k * zd + b];
				}
				S[a * zd + b] = s + md.p->R[a * zd + b];
			}
		}
		double det;
		invdet(S, zd, &Sinv[i * 9], &det);   // mc[n] = new Gaussian(mp, S, w), :865
		qmult[i] = multiplier(zd, det);
		PD[i]    = pd_m(md, &mp[i * 3]);     // :866
	}

	// per measurement, :881-903
	double radius = md.p->density_distance_threshold;
	std::vector<int>    near;
	std::vector<double> q;
	for (int k = 0; k < M; k++) {
		const double* zk = z + k * zd;
		double x[3];
		measure_to_map(md, pose, zk, x);
		near.clear();
		q.clear();
		double weightsum = 0;
		for (int i = 0; i < n; i++) {
			if (!is_near(md, x, model[i].m, radius)) {
				continue;
			}
			double d[3];
			for (int a = 0; a < zd; a++) {
				d[a] = zk[a] - mp[i * 3 + a];
			}
			double qi = qmult[i] * std::exp(-0.5 * quadform(&Sinv[i * 9], d, zd));   // mc[i].Evaluate(mlinear)
			near.push_back(i);
			q.push_back(qi);
			weightsum += PD[i] * model[i].w * qi;   // :889
		}

		for (size_t h = 0; h < near.size(); h++) {
			int i = near[h];
			const Comp& c = model[i];
			// gain = PH . Sinv : 3 x zd   (:895)
			double K[9];
			for (int a = 0; a < 3; a++) {
				for (int b = 0; b < zd; b++) {
					double s = 0;
					for (int e = 0; e < zd; e++) {
						s += PH[i * 9 + a * zd + e] * Sinv[i * 9 + e * zd + b];
					}
					K[a * zd + b] = s;
				}
			}
			// mean = m + K (z - mp)   (:896)
			double nu[3], mean[3];
			for (int a = 0; a < zd; a++) {
				nu[a] = zk[a] - mp[i * 3 + a];
			}
			for (int a = 0; a < 3; a++) {
				double s = 0;
				for (int e = 0; e < zd; e++) {
					s += K[a * zd + e] * nu[e];
				}
				mean[a] = c.m[a] + s;
			}
			// covariance = (I - K H) P   (:897), I hard-coded 3x3, no symmetrisation
			double IKH[9], cov[9];
			for (int a = 0; a < 3; a++) {
				for (int b = 0; b < 3; b++) {
					double s = 0;
					for (int e = 0; e < zd; e++) {
						s += K[a * zd + e] * H[i * 9 + e * 3 + b];
					}
					IKH[a * 3 + b] = ((a == b) ? 1.0 : 0.0) - s;
				}
			}
			for (int a = 0; a < 3; a++) {
				for (int b = 0; b < 3; b++) {
					double s = 0;
					for (int e = 0; e < 3; e++) {
						s += IKH[a * 3 + e] * c.P[e * 3 + b];
					}
					cov[a * 3 + b] = s;
				}
			}
			double weight = PD[i] * c.w * q[h] / (md.p->clutter_density + weightsum);   // :899
			corrected.push_back(make_comp(mean, cov, weight));
		}
	}
	return corrected;
}

// Gaussian.Merge (Gaussian.cs:297-347)
Comp merge(const std::vector<const Comp*>& comps)
{
	double weight = 0, mean[3] = {0, 0, 0}, cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	for (const Comp* c : comps) {
		double w = c->w;
		weight += w;
		for (int a = 0; a < 3; a++) {
			mean[a] = mean[a] + w * c->m[a];
		}
		for (int a = 0; a < 3; a++) {
			for (int b = 0; b < 3; b++) {
				cov[a * 3 + b] = cov[a * 3 + b] + w * (c->P[a * 3 + b] + c->m[a] * c->m[b]);
			}
		}
	}
	if (weight < 1e-15) {   // :339-341, Util.InfiniteCovariance (Util.cs:138-148)
		double infc[9] = {1e12, 0, 0, 0, 1e12, 0, 0, 0, 1e12};
		return make_comp(comps[0]->m, infc, 0.0);
	}
	for (int a = 0; a < 3; a++) {
		mean[a] = mean[a] / weight;
	}
	for (int a = 0; a < 3; a++) {
		for (int b = 0; b < 3; b++) {
			cov[a * 3 + b] = cov[a * 3 + b] / weight - mean[a] * mean[b];
		}
	}
	return make_comp(mean, cov, weight);
}

// descending by weight, comparator Math.Sign(b.w - a.w) (PHDNavigator.cs:920, Map.cs:129);
// canonical: stable
template <class T, class W>
void sort_desc(std::vector<T>& v, W weight)
{
	std::stable_sort(v.begin(), v.end(), [&](const T& a, const T& b) { return weight(b) - weight(a) < 0; });
}

// PruneModel (PHDNavigator.cs:913-948)
Mixture prune(const Model& md, const Mixture& model)
{
	std::vector<const Comp*> lm;
	lm.reserve(model.size());
	for (const Comp& c : model) {
		lm.push_back(&c);
	}
	sort_desc(lm, [](const Comp* c) { return c->w; });

	int weightcut = 0;
	int limit = std::min(md.p->max_quantity, (int) lm.size());
	for (weightcut = 0; weightcut < limit; weightcut++) {
		if (lm[weightcut]->w < md.p->min_weight) {
			break;
		}
	}
	// only the first `weightcut` entries can ever be touched (removals shrink the window with them)
	lm.resize(weightcut);

	Mixture pruned;
	double  thr2 = md.p->merge_threshold * md.p->merge_threshold;
	std::vector<const Comp*> close;
	for (int i = 0; i < (int) lm.size(); i++) {
		close.clear();
		close.push_back(lm[i]);
		for (int k = i + 1; k < (int) lm.size(); k++) {
			if (comp_sqmahal(*lm[i], lm[k]->m) < thr2) {   // Gaussian.AreClose, Gaussian.cs:243-246
				close.push_back(lm[k]);
				lm.erase(lm.begin() + k);
				k--;
			}
		}
		pruned.push_back(merge(close));
	}
	return pruned;
}

// Map.BestMapEstimate (Map.cs:119-142): returns the picked means in pick order
std::vector<std::pair<double, const Comp*>> best_map_estimate(const Mixture& map)
{
	double expected = 0;
	for (const Comp& c : map) {
		expected += c.w;   // ExpectedSize, Map.cs:61-71
	}
	int size = (int) expected;
	std::vector<std::pair<double, const Comp*>> mlist;
	for (const Comp& c : map) {
		mlist.emplace_back(c.w, &c);
	}
	auto wt = [](const std::pair<double, const Comp*>& e) { return e.first; };
	sort_desc(mlist, wt);
	std::vector<std::pair<double, const Comp*>> best;
	for (int i = 0; i < size; i++) {
		best.push_back(mlist[i]);
		mlist.emplace_back(mlist[i].first - 1, mlist[i].second);
		sort_desc(mlist, wt);
	}
	return best;
}

// ---------------------------------------------------------------------------------------------
// GraphCombinatorics on dense matrices (missing entry == -inf, the SparseMatrix default used on
// this path: PHDNavigator.cs:420, GraphCombinatorics.cs:208)
// ---------------------------------------------------------------------------------------------

struct Dense {
	int n;
	std::vector<double> v;   // n x n
	double  at(int i, int k) const { return v[i * n + k]; }
	double& at(int i, int k) { return v[i * n + k]; }
};

// Hungarian (GraphCombinatorics.cs:64-175). Returns false when there is no solution (null).
bool hungarian(const Dense& mx, std::vector<int>& matchx)
{
	const int n = mx.n;
	std::vector<double> labelx(n, 0.0), labely(n, 0.0), slack(n);
	std::vector<int>    matchy(n, -1), parent(n);
	std::vector<char>   visitx(n), visity(n);
	matchx.assign(n, -1);

	// rowmax = FoldRows(Math.Max, 0) (:67, SparseMatrix.FoldRows): max over DEFINED entries, seeded 0
	for (int i = 0; i < n; i++) {
		double f = 0;
		for (int k = 0; k < n; k++) {
			f = std::max(f, mx.at(i, k));   // a missing (-inf) entry never wins, same as skipping it
		}
		labelx[i] = f;
	}

	for (;;) {
		int root = -1;
		for (int i = 0; i < n; i++) {
			if (matchx[i] == -1) { root = i; break; }
		}
		if (root == -1) {
			break;
		}
		for (int i = 0; i < n; i++) {
			parent[i] = root;
			slack[i]  = labelx[root] + labely[i] - mx.at(root, i);
		}
		std::fill(visitx.begin(), visitx.end(), 0);
		std::fill(visity.begin(), visity.end(), 0);
		visitx[root] = 1;

		int  iminslack = 0;
		bool found = false;
		while (!found) {
			iminslack = std::numeric_limits<int>::max();
			double delta = INF;
			for (int i = 0; i < n; i++) {
				if (!visity[i] && slack[i] < delta) {
					iminslack = i;
					delta     = slack[i];
				}
			}
			if (std::isinf(delta) && delta > 0) {
				return false;
			}
			for (int i = 0; i < n; i++) {
				if (visitx[i]) {
					labelx[i] -= delta;
				}
			}
			for (int i = 0; i < n; i++) {
				if (visity[i]) {
					labely[i] += delta;
				}
				else {
					slack[i] -= delta;
				}
			}
			visity[iminslack] = 1;
			if (matchy[iminslack] != -1) {
				int match = matchy[iminslack];
				visitx[match] = 1;
				for (int i = 0; i < n; i++) {
					if (!visity[i]) {
						double mdelta = labelx[match] + labely[i] - mx.at(match, i);
						if (mdelta < slack[i]) {
							slack[i]  = mdelta;
							parent[i] = match;
						}
					}
				}
			}
			else {
				found = true;
			}
		}
		int px, py, ty;
		for (py = iminslack, px = parent[py]; px != root; py = ty, px = parent[py]) {
			ty = matchx[px];
			matchx[px] = py;
			matchy[py] = px;
		}
		matchx[px] = py;
		matchy[py] = px;
	}
	return true;
}

// AssignmentValue (GraphCombinatorics.cs:183-197)
double assignment_value(const Dense& profit, const std::vector<int>& matches)
{
	double total = 0;
	for (int i = 0; i < (int) matches.size(); i++) {
		total += profit.at(i, matches[i]);
	}
	return total;
}

typedef std::pair<int, int> Key;   // MatrixKey (I, K)

struct MurtyNode {
	std::vector<Key> forced, eliminated;
	std::vector<int> assignment;
	bool             solved = false;   // Assignment != null
};

// MurtyNode.Children (GraphCombinatorics.cs:469-509)
std::vector<MurtyNode> murty_children(const MurtyNode& node)
{
	std::vector<MurtyNode> children;
	if (!node.solved) {
		return children;
	}
	auto isforced = [&](const Key& k) { return std::find(node.forced.begin(), node.forced.end(), k) != node.forced.end(); };
	std::vector<Key> remaining;
	for (int i = 0; i < (int) node.assignment.size(); i++) {
		Key k(i, node.assignment[i]);
		if (!isforced(k)) {
			remaining.push_back(k);
		}
	}
	for (int i = 0; i < (int) remaining.size() - 1; i++) {
		MurtyNode child;
		child.eliminated = node.eliminated;
		child.eliminated.push_back(remaining[i]);
		child.forced = node.forced;
		for (int k = 0; k < i; k++) {
			child.forced.push_back(remaining[k]);
		}
		children.push_back(child);
	}
	return children;
}

// reduceprofit (GraphCombinatorics.cs:206-234)
Dense reduce_profit(const Dense& full, const MurtyNode& node)
{
	Dense r = full;
	for (const Key& f : node.forced) {
		for (int k = 0; k < r.n; k++) {
			r.at(f.first, k) = -INF;    // RemoveRows
		}
	}
	for (const Key& f : node.forced) {
		for (int i = 0; i < r.n; i++) {
			r.at(i, f.second) = -INF;   // RemoveColumns
		}
	}
	for (const Key& f : node.forced) {
		r.at(f.first, f.second) = 1;
	}
	for (const Key& e : node.eliminated) {
		r.at(e.first, e.second) = -INF;
	}
	return r;
}

// MurtyPairing (GraphCombinatorics.cs:241-272) as a pull generator. The frontier is the reference's
// PriorityQueue (:595-707): a list re-sorted ascending after each Add, popped from the back;
// canonical stable order => among equal priorities the newest entry pops first.
class MurtyEnumerator {
public:
	explicit MurtyEnumerator(const Dense& profit) : profit_(profit)
	{
		MurtyNode first;
		first.solved = hungarian(profit_, first.assignment);
		push(first.solved ? assignment_value(profit_, first.assignment) : -INF, first);
	}

	bool next(std::vector<int>* assignment, double* value, bool* solved)
	{
		if (pending_) {
			expand();
		}
		if (frontier_.empty()) {
			return false;
		}
		best_ = frontier_.back().second;
		*value = frontier_.back().first;
		frontier_.pop_back();
		*assignment = best_.assignment;
		*solved     = best_.solved;
		pending_    = true;   // the children are generated when the consumer asks for more (yield semantics)
		return true;
	}

private:
	void push(double key, const MurtyNode& node)
	{
		auto it = std::upper_bound(frontier_.begin(), frontier_.end(), key,
		                           [](double k, const std::pair<double, MurtyNode>& e) { return k < e.first; });
		frontier_.insert(it, std::make_pair(key, node));
	}

	void expand()
	{
		pending_ = false;
		for (MurtyNode& child : murty_children(best_)) {
			Dense reduced = reduce_profit(profit_, child);
			child.solved  = hungarian(reduced, child.assignment);
			if (child.solved) {
				push(assignment_value(profit_, child.assignment), child);
			}
		}
	}

	Dense     profit_;
	MurtyNode best_;
	bool      pending_ = false;
	std::vector<std::pair<double, MurtyNode>> frontier_;
};

bool last_permutation(const std::vector<int>& p)   // GraphCombinatorics.cs:341-350
{
	for (size_t i = 1; i < p.size(); i++) {
		if (p[i - 1] < p[i]) {
			return false;
		}
	}
	return true;
}

// LexicographicalPairing (GraphCombinatorics.cs:280-334). `emit` gets each permutation and value.
template <class F>
void lexicographical_pairing(const Dense& profit, int modelsize, F emit)
{
	const int n = profit.n;
	std::vector<int> perm(n);
	for (int i = 0; i < n; i++) {
		perm[i] = i;   // row keys, sorted
	}
	int measurestart = n;
	for (int i = 0; i < n; i++) {
		if (perm[i] >= modelsize) {
			measurestart = i;
			break;
		}
	}
	std::reverse(perm.begin() + measurestart, perm.end());
	if (!emit(perm, assignment_value(profit, perm))) {
		return;
	}
	while (!last_permutation(perm)) {
		int a, b;
		for (a = n - 2; a > 0; a--) {
			if (perm[a] < perm[a + 1]) {
				break;
			}
		}
		for (b = n - 1; b > a; b--) {
			if (perm[a] < perm[b]) {
				break;
			}
		}
		std::swap(perm[a], perm[b]);
		std::reverse(perm.begin() + a + 1, perm.end());
		std::reverse(perm.begin() + measurestart, perm.end());
		if (!emit(perm, assignment_value(profit, perm))) {
			return;
		}
	}
}

// LogSumExp (MatrixExtensions.cs:361-389)
double log_sum_exp(const double* v, int begin, int end)
{
	double mx = -INF, value = 0;
	for (int i = begin; i < end; i++) {
		mx = std::max(mx, v[i]);
	}
	if (std::isinf(mx) && mx < 0) {
		return -INF;
	}
	for (int i = begin; i < end; i++) {
		value += std::exp(v[i] - mx);
	}
	return mx + std::log(value);
}

// union-find for ConnectedComponents (GraphCombinatorics.cs:358-425): the partition is traversal
// independent; the ORDER of the returned list is that of each component's first row in the
// dictionary (= insertion) order of the matrix rows.
struct DSU {
	std::vector<int> p;
	explicit DSU(int n) : p(n) { for (int i = 0; i < n; i++) p[i] = i; }
	int  find(int a) { while (p[a] != a) { p[a] = p[p[a]]; a = p[a]; } return a; }
	void join(int a, int b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};

// SetLogLikelihood (PHDNavigator.cs:462-515) over SetLogLikeMatrix (:415-453).
// `lm` = landmark means of the map estimate (J x 3).
// quasi: QuasiSetLogLikelihood (:526-713, value only) — the same sum with everything fully visible: constant PD
// (logPD / log1PD, :574-575), zprobs of weight 1 (:583) and the detection gate at 12 (:600, :615).
// gradient (quasi only; :543-548 with calcgradient): d/dpose of the value, `odo` doubles (6 for Pose3D, 2 for
// LinearPose2D): per defined detection entry dlldp[i,k] = (z_k - zhat_i)' R^-1 Jp_i (:605-608), per enumerated pairing the
// sum of its entries' vectors (:676-678), per component TemperedAverage(dlogcompdp, logcomp, 0, m) (:707). That
// function (MatrixExtensions.cs:400-440) works IN PLACE on logcomp: entries [0, m) are replaced by exp(l - max) and
// stay that way, so in gradient mode the stale values the Murty cut (:672) reads are these, not log values; and its
// `weights.Normalize()` is Accord.Math's vector Normalize over the WHOLE 200-entry array — division by the Euclidean
// norm (not by the sum), stale entries beyond m included. Accord 3.0.2 is not in the tree: that reading is unpinned;
// `average_mode` 1 switches to weights / sum over [0, m) (what the summary comment of TemperedAverage describes) for
// the comparison the KAT file makes, 0 is what the source says and what the device implements.
double set_log_likelihood(const Model& md, const Pose& pose, const double* lm, int J, const double* z, int M,
                          int* nclusters = nullptr, int* maxcluster = nullptr, bool quasi = false,
                          double* gradient = nullptr, int average_mode = 0)
{
	const int zd = md.zdim;
	const int odo = (md.p->model == PHD_MODEL_LINEAR2D) ? 2 : 6;
	if (gradient) {
		for (int t = 0; t < odo; t++) gradient[t] = 0;
	}
	const double gate = quasi ? 12 : 5;
	double logclutter = std::log(md.p->clutter_density);
	std::vector<double> zhat(J * 3), pdj(J);
	for (int i = 0; i < J; i++) {
		measure_perfect(md, pose, lm + i * 3, &zhat[i * 3]);
		pdj[i] = quasi ? md.p->pd : pd_m(md, &zhat[i * 3]);   // zprobs[i].Weight (quasi: pose.PD)
	}

	// detection block: defined iff Mahalanobis(z_k ; zhat_i, R) < 5 (:433-442)
	struct Edge { int k; double v; double g[6]; };
	std::vector<std::vector<Edge>> det(J);
	DSU dsu(J + M);   // node i < J: landmark i ; node J + k: measurement k
	for (int i = 0; i < J; i++) {
		double Jp[18];
		if (gradient) {
			jacobian_p(md, pose, lm + i * 3, Jp);   // :591
		}
		for (int k = 0; k < M; k++) {
			double d[3];
			for (int a = 0; a < zd; a++) {
				d[a] = zhat[i * 3 + a] - z[k * zd + a];   // Mahalanobis: Mean - point (Gaussian.cs:354-358)
			}
			double dist = std::sqrt(quadform(md.Rinv, d, zd));
			if (dist < gate) {
				Edge e{k, std::log(pdj[i]) + std::log(md.Rmult) - 0.5 * dist * dist, {0, 0, 0, 0, 0, 0}};
				if (gradient) {   // (m - mean)' CovarianceInverse, then times the jacobian (:605-608)
					double u[3];
					for (int b = 0; b < zd; b++) {
						double sacc = 0;
						for (int a = 0; a < zd; a++) sacc += (z[k * zd + a] - zhat[i * 3 + a]) * md.Rinv[a * zd + b];
						u[b] = sacc;
					}
					for (int t = 0; t < odo; t++) {
						double sacc = 0;
						for (int b = 0; b < zd; b++) sacc += u[b] * Jp[b * odo + t];
						e.g[t] = sacc;
					}
				}
				det[i].push_back(e);
				dsu.join(i, J + k);
			}
		}
	}

	// row insertion order of the sparse matrix: landmark rows that got a detection entry (ascending),
	// then the other landmark rows (misdetection diagonal, :444-446), then the clutter rows (:448-450)
	std::vector<int> roworder;
	for (int i = 0; i < J; i++) if (!det[i].empty()) roworder.push_back(i);
	for (int i = 0; i < J; i++) if (det[i].empty())  roworder.push_back(i);
	for (int k = 0; k < M; k++) roworder.push_back(J + k);

	// members of every component, ascending: landmarks then measurements
	std::vector<std::vector<int>> Ls(J + M), Zs(J + M);
	for (int i = 0; i < J; i++) Ls[dsu.find(i)].push_back(i);
	for (int k = 0; k < M; k++) Zs[dsu.find(J + k)].push_back(k);

	std::vector<char> done(J + M, 0);
	double logcomp[200];
	std::memset(logcomp, 0, sizeof(logcomp));   // new double[200]
	std::vector<std::array<double, 6>> dlogcomp(gradient ? 200 : 0);
	double total = 0;
	int    ncl = 0, maxcl = 0;

	for (int r : roworder) {
		int root = dsu.find(r);   // row r of the matrix is node r (landmark) or node J+k (clutter row of k)
		if (done[root]) {
			continue;
		}
		done[root] = 1;
		const std::vector<int>& L = Ls[root];
		const std::vector<int>& Z = Zs[root];
		const int nl = (int) L.size(), nz = (int) Z.size(), n = nl + nz;
		ncl++;
		maxcl = std::max(maxcl, n);

		// Compact (:475, SparseMatrix.cs:592-628): rows = L then clutter rows of Z; cols = Z then
		// misdetection columns of L; then the (clutter x misdetection) quadrant is zeroed (:480-488)
		Dense comp;
		comp.n = n;
		comp.v.assign(n * n, -INF);
		for (int a = 0; a < nl; a++) {
			for (const Edge& e : det[L[a]]) {
				int col = (int) (std::lower_bound(Z.begin(), Z.end(), e.k) - Z.begin());
				comp.at(a, col) = e.v;
			}
			comp.at(a, nz + a) = std::log(1 - pdj[L[a]]);
		}
		for (int b = 0; b < nz; b++) {
			comp.at(nl + b, b) = logclutter;
			for (int a = 0; a < nl; a++) {
				comp.at(nl + b, nz + a) = 0;
			}
		}

		// dcomp = dlldp.Submatrix(rows, cols) (:643): the vector of a defined detection entry, zeros elsewhere
		auto pairing_gradient = [&](const std::vector<int>& asg, double* out) {   // :676-678
			for (int t = 0; t < odo; t++) out[t] = 0;
			for (int a = 0; a < nl; a++) {
				if (asg[a] >= nz) continue;
				for (const Edge& e : det[L[a]]) {
					if (e.k == Z[asg[a]]) {
						for (int t = 0; t < odo; t++) out[t] += e.g[t];
					}
				}
			}
		};

		int m = 0;
		if (n <= 5) {
			lexicographical_pairing(comp, J, [&](const std::vector<int>& perm, double value) {
				if (m >= 200) {
					return false;
				}
				if (gradient) pairing_gradient(perm, dlogcomp[m].data());
				logcomp[m++] = value;
				return true;
			});
		}
		else {
			MurtyEnumerator murty(comp);
			std::vector<int> asg;
			double value;
			bool   solved;
			while (murty.next(&asg, &value, &solved)) {
				if (m >= 200 || (logcomp[m] - logcomp[0] < -10)) {   // stale read, :503 / :672
					break;
				}
				if (gradient) pairing_gradient(asg, dlogcomp[m].data());
				logcomp[m++] = value;
			}
		}
		total += log_sum_exp(logcomp, 0, m);

		if (gradient) {   // TemperedAverage(dlogcompdp, logcomp, 0, m), MatrixExtensions.cs:400-440
			double mx = -INF;
			for (int i = 0; i < m; i++) mx = std::max(mx, logcomp[i]);
			if (!(std::isinf(mx) && mx < 0)) {
				for (int i = 0; i < m; i++) logcomp[i] = std::exp(logcomp[i] - mx);   // in place (:429-431)
				double norm = 0;
				if (average_mode == 0) {
					for (int i = 0; i < 200; i++) norm += logcomp[i] * logcomp[i];     // weights.Normalize(), :433
					norm = std::sqrt(norm);
				}
				else {
					for (int i = 0; i < m; i++) norm += logcomp[i];
				}
				double value[6] = {0, 0, 0, 0, 0, 0};
				for (int i = 0; i < m; i++) {
					double wn = (norm == 0) ? logcomp[i] : logcomp[i] / norm;
					for (int t = 0; t < odo; t++) value[t] += wn * dlogcomp[i][t];
				}
				for (int t = 0; t < odo; t++) gradient[t] += value[t];
			}
		}
	}
	if (nclusters)  *nclusters  = ncl;
	if (maxcluster) *maxcluster = maxcl;
	return total;
}

double map_evaluate(const Mixture& map, const double* x)   // Map.Evaluate(point), Map.cs:192-202
{
	double value = 0;
	for (const Comp& c : map) {
		value += c.w * comp_eval(c, x);
	}
	return value;
}

double expected_size(const Mixture& map)
{
	double e = 0;
	for (const Comp& c : map) {
		e += c.w;
	}
	return e;
}

// WeightAlpha (PHDNavigator.cs:373-393)
double weight_alpha(const Model& md, const Pose& pose, const Mixture& predicted, const Mixture& corrected,
                    const double* z, int M, double* setloglik_out = nullptr)
{
	auto jmap = best_map_estimate(corrected);
	const int J = (int) jmap.size();
	std::vector<double> lm(J * 3 + 3);
	double plog = 0, clog = 0;
	for (int j = 0; j < J; j++) {
		const double* mean = jmap[j].second->m;
		std::memcpy(&lm[j * 3], mean, 3 * sizeof(double));
		plog += std::log(map_evaluate(predicted, mean));
		clog += std::log(map_evaluate(corrected, mean));
	}
	double pcount = expected_size(predicted);
	double ccount = expected_size(corrected);
	double setll  = set_log_likelihood(md, pose, lm.data(), J, z, M);
	double ratio  = (plog - pcount) - (clog - ccount);
	if (setloglik_out) {
		*setloglik_out = setll;
	}
	return std::exp(setll + ratio);
}

// ResampleParticles (PHDNavigator.cs:724-760). u = (double) Util.Uniform.Next(). Returns the best slot.
// A source index of -1 (u == 0, where the reference would throw) is clamped to 0.
int resample(const double* w, int P, double u, int32_t* src)
{
	double random = u / P;
	double maxweight = 0;
	int    best = 0;
	for (int i = 0, k = 0; i < P; i++) {
		for (; random > 0 && k < P; k++) {
			random -= w[k];
		}
		int s = (k - 1 < 0) ? 0 : k - 1;
		src[i] = s;
		random += 1.0 / P;
		if (w[s] > maxweight) {
			maxweight = w[s];
			best = i;
		}
	}
	return best;
}

bool particle_depleted(const Model& md, const double* w, int P)   // :768-777
{
	double cum = 0;
	for (int i = 0; i < P; i++) {
		cum += w[i] * w[i];
	}
	return 1.0 / cum < md.p->min_effective_particle * P;
}

// flat <-> Mixture
Mixture load_mixture(const double* w, const double* mean, const double* cov, int n)
{
	Mixture mx;
	mx.reserve(n);
	for (int i = 0; i < n; i++) {
		mx.push_back(make_comp(mean + i * 3, cov + i * 9, w[i]));
		mx.back().w = w[i];
	}
	return mx;
}

int store_mixture(const Mixture& mx, int cap, double* w, double* mean, double* cov)
{
	int n = (int) mx.size();
	for (int i = 0; i < n && i < cap; i++) {
		w[i] = mx[i].w;
		std::memcpy(mean + i * 3, mx[i].m, 3 * sizeof(double));
		std::memcpy(cov + i * 9, mx[i].P, 9 * sizeof(double));
	}
	return n;
}

}  // namespace

// =============================================================================================
// C entry points used through ctypes by tests/ and bench.py's cpu_baseline leg
// =============================================================================================
extern "C" {

int orc_predict(const phd_params* p, const double* pose7, const double* z, int M,
                const double* w, const double* mean, const double* cov, int n,
                int cap, double* ow, double* omean, double* ocov)
{
	Model md = make_model(p);
	Pose  ps = make_pose(pose7);
	return store_mixture(predict(md, ps, load_mixture(w, mean, cov, n), z, M), cap, ow, omean, ocov);
}

int orc_correct(const phd_params* p, const double* pose7, const double* z, int M,
                const double* w, const double* mean, const double* cov, int n,
                int cap, double* ow, double* omean, double* ocov)
{
	Model md = make_model(p);
	Pose  ps = make_pose(pose7);
	return store_mixture(correct(md, ps, load_mixture(w, mean, cov, n), z, M), cap, ow, omean, ocov);
}

int orc_prune(const phd_params* p, const double* w, const double* mean, const double* cov, int n,
              int cap, double* ow, double* omean, double* ocov)
{
	Model md = make_model(p);
	return store_mixture(prune(md, load_mixture(w, mean, cov, n)), cap, ow, omean, ocov);
}

// Gaussian.Merge on a list (for the Prune KAT's expected values)
void orc_merge(const double* w, const double* mean, const double* cov, int n, double* ow, double* omean, double* ocov)
{
	Mixture mx = load_mixture(w, mean, cov, n);
	std::vector<const Comp*> ptr;
	for (const Comp& c : mx) ptr.push_back(&c);
	Comp r = merge(ptr);
	*ow = r.w;
	std::memcpy(omean, r.m, sizeof(r.m));
	std::memcpy(ocov, r.P, sizeof(r.P));
}

int orc_best_map_estimate(const double* w, const double* mean, const double* cov, int n, int cap, double* omean, int* osrc)
{
	Mixture mx = load_mixture(w, mean, cov, n);
	auto best = best_map_estimate(mx);
	for (int j = 0; j < (int) best.size() && j < cap; j++) {
		std::memcpy(omean + j * 3, best[j].second->m, 3 * sizeof(double));
		if (osrc) osrc[j] = (int) (best[j].second - mx.data());
	}
	return (int) best.size();
}

double orc_set_log_likelihood(const phd_params* p, const double* pose7, const double* lm, int J,
                              const double* z, int M, int* nclusters, int* maxcluster)
{
	Model md = make_model(p);
	Pose  ps = make_pose(pose7);
	return set_log_likelihood(md, ps, lm, J, z, M, nclusters, maxcluster);
}

void orc_pose_add(const double* pose7, const double* delta6, double* out7);

// LoopyPHDNavigator.LogLikeGradient (LoopyPHDNavigator.cs:876-909): central differences of QuasiSetLogLikelihood
// at linearpoint.Add(pose +- eps e_i), eps = 1e-5
void orc_loglike_gradient(const phd_params* p, const double* pose6, const double* linearpoint7, const double* lm, int J,
                          const double* z, int M, double* gradient6)
{
	Model md = make_model(p);
	const double eps = 1e-5;
	for (int i = 0; i < 6; i++) {
		double l[2];
		for (int s = 0; s < 2; s++) {
			double d[6], q7[7];
			for (int t = 0; t < 6; t++) d[t] = pose6[t];
			d[i] += s == 0 ? eps : -eps;
			orc_pose_add(linearpoint7, d, q7);
			l[s] = set_log_likelihood(md, make_pose(q7), lm, J, z, M, nullptr, nullptr, true);
		}
		gradient6[i] = (l[0] - l[1]) / (2 * eps);
	}
}

// PHDNavigator.QuasiSetLogLikelihood(measurements, map, pose) (:526-531), SURVEY row f4
double orc_quasi_set_log_likelihood(const phd_params* p, const double* pose7, const double* lm, int J, const double* z, int M)
{
	Model md = make_model(p);
	Pose  ps = make_pose(pose7);
	return set_log_likelihood(md, ps, lm, J, z, M, nullptr, nullptr, true);
}

// PHDNavigator.QuasiSetLogLikelihood(measurements, map, pose, out gradient) (:543-548). gradient: 6 doubles (Pose3D) or
// 2 (LinearPose2D, whose pose is (x, y) of pose7). average_mode: see set_log_likelihood.
double orc_quasi_set_log_likelihood_grad(const phd_params* p, const double* pose7, const double* lm, int J, const double* z,
                                         int M, double* gradient, int average_mode)
{
	Model md = make_model(p);
	return set_log_likelihood(md, make_pose(pose7), lm, J, z, M, nullptr, nullptr, true, gradient, average_mode);
}

void orc_jacobian_p(const phd_params* p, const double* pose7, const double* lm, double* Jp)
{
	Model md = make_model(p);
	jacobian_p(md, make_pose(pose7), lm, Jp);
}

double orc_weight_alpha(const phd_params* p, const double* pose7, const double* z, int M,
                        const double* pw, const double* pmean, const double* pcov, int pn,
                        const double* cw, const double* cmean, const double* ccov, int cn, double* setloglik)
{
	Model md = make_model(p);
	Pose  ps = make_pose(pose7);
	return weight_alpha(md, ps, load_mixture(pw, pmean, pcov, pn), load_mixture(cw, cmean, ccov, cn), z, M, setloglik);
}

int orc_resample(const double* w, int P, double u, int32_t* src) { return resample(w, P, u, src); }

int orc_particle_depleted(const phd_params* p, const double* w, int P)
{
	Model md = make_model(p);
	return particle_depleted(md, w, P) ? 1 : 0;
}

// ---- measurement-model probes ---------------------------------------------------------------
void orc_measure_perfect(const phd_params* p, const double* pose7, const double* lm, double* z)
{
	Model md = make_model(p);
	measure_perfect(md, make_pose(pose7), lm, z);
}
void orc_measure_to_map(const phd_params* p, const double* pose7, const double* z, double* x)
{
	Model md = make_model(p);
	measure_to_map(md, make_pose(pose7), z, x);
}
void orc_jacobian_l(const phd_params* p, const double* pose7, const double* lm, double* H)
{
	Model md = make_model(p);
	jacobian_l(md, make_pose(pose7), lm, H);
}
double orc_detection_probability(const phd_params* p, const double* pose7, const double* lm)
{
	Model md = make_model(p);
	return pd_landmark(md, make_pose(pose7), lm);
}
void orc_quat_matrix(const double* q4, double* r9) { qmatrix(Quat{q4[0], q4[1], q4[2], q4[3]}, r9); }
void orc_quat_rotate(const double* q4, const double* v3, double* o3)
{
	Quat q{q4[0], q4[1], q4[2], q4[3]};
	Quat r = qmul(qmul(q, Quat{0, v3[0], v3[1], v3[2]}), qconj(q));
	o3[0] = r.x; o3[1] = r.y; o3[2] = r.z;
}

// ---- GraphCombinatorics probes (dense n x n, `missing` marks undefined entries) ---------------
// default value of the matrix: tests in GraphCombinatoricsTest use 0, the PHD path uses -inf.
int orc_hungarian(const double* v, int n, int* match)
{
	Dense d; d.n = n; d.v.assign(v, v + n * n);
	std::vector<int> mx;
	if (!hungarian(d, mx)) return 0;
	for (int i = 0; i < n; i++) match[i] = mx[i];
	return 1;
}

double orc_assignment_value(const double* v, int n, const int* match)
{
	Dense d; d.n = n; d.v.assign(v, v + n * n);
	return assignment_value(d, std::vector<int>(match, match + n));
}

int orc_murty(const double* v, int n, int maxcount, int* assignments, double* values)
{
	Dense d; d.n = n; d.v.assign(v, v + n * n);
	MurtyEnumerator e(d);
	std::vector<int> a; double val; bool solved; int m = 0;
	while (m < maxcount && e.next(&a, &val, &solved)) {
		for (int i = 0; i < n; i++) assignments[m * n + i] = solved ? a[i] : -1;
		values[m++] = val;
	}
	return m;
}

int orc_lexicographic(const double* v, int n, int modelsize, int maxcount, int* perms, double* values)
{
	Dense d; d.n = n; d.v.assign(v, v + n * n);
	int m = 0;
	lexicographical_pairing(d, modelsize, [&](const std::vector<int>& p, double val) {
		if (m >= maxcount) return false;
		for (int i = 0; i < n; i++) perms[m * n + i] = p[i];
		values[m++] = val;
		return true;
	});
	return m;
}

// children of a Murty node: forced/eliminated as (i,k) pairs; out arrays sized by the caller.
// Returns the number of children; child c has nforced[c]/nelim[c] keys laid out with stride `stride`.
int orc_murty_children(const int* forced, int nf, const int* elim, int ne, const int* assignment, int n,
                       int stride, int* cforced, int* cnforced, int* celim, int* cnelim)
{
	MurtyNode node;
	for (int i = 0; i < nf; i++) node.forced.emplace_back(forced[2 * i], forced[2 * i + 1]);
	for (int i = 0; i < ne; i++) node.eliminated.emplace_back(elim[2 * i], elim[2 * i + 1]);
	node.assignment.assign(assignment, assignment + n);
	node.solved = true;
	auto ch = murty_children(node);
	for (size_t c = 0; c < ch.size(); c++) {
		cnforced[c] = (int) ch[c].forced.size();
		cnelim[c]   = (int) ch[c].eliminated.size();
		for (size_t i = 0; i < ch[c].forced.size(); i++) {
			cforced[(c * stride + i) * 2]     = ch[c].forced[i].first;
			cforced[(c * stride + i) * 2 + 1] = ch[c].forced[i].second;
		}
		for (size_t i = 0; i < ch[c].eliminated.size(); i++) {
			celim[(c * stride + i) * 2]     = ch[c].eliminated[i].first;
			celim[(c * stride + i) * 2 + 1] = ch[c].eliminated[i].second;
		}
	}
	return (int) ch.size();
}

// ConnectedComponents on an h x w definedness mask: labels every defined entry's row and column;
// returns the number of components. rowlabel/collabel = -1 for rows/columns without entries.
int orc_connected_components(const uint8_t* defined, int h, int w, int* rowlabel, int* collabel)
{
	DSU dsu(h + w);
	std::vector<char> used(h + w, 0);
	for (int i = 0; i < h; i++) {
		for (int k = 0; k < w; k++) {
			if (defined[i * w + k]) {
				dsu.join(i, h + k);
				used[i] = used[h + k] = 1;
			}
		}
	}
	std::vector<int> id(h + w, -1);
	int count = 0;
	for (int a = 0; a < h + w; a++) {
		if (!used[a]) continue;
		int r = dsu.find(a);
		if (id[r] < 0) id[r] = count++;
	}
	for (int i = 0; i < h; i++) rowlabel[i] = used[i] ? id[dsu.find(i)] : -1;
	for (int k = 0; k < w; k++) collabel[k] = used[h + k] ? id[dsu.find(h + k)] : -1;
	return count;
}

double orc_log_sum_exp(const double* v, int begin, int end) { return log_sum_exp(v, begin, end); }

// ---- whole step -------------------------------------------------------------------------------
// PHDNavigator.SlamUpdate (PHDNavigator.cs:323-362) on flat state:
//   poses[P*7]; per particle slab of `cap` components: w[P*cap], mean[P*cap*3], cov[P*cap*9], n[P];
//   weights[P]. In place. `src[P]` <- resample sources (identity if none), returns BestParticle,
//   *resampled <- 0/1. `threads` ≙ Config.NParallel (Parallel.For, :326-327). Optional per-stage
//   wall-clock seconds summed over particles in stage_s[4] = predict, correct, prune, alpha.
int orc_slam_update(const phd_params* p, int P, double* poses, int cap,
                    double* w, double* mean, double* cov, int* n, double* weights,
                    const double* z, int M, int onlymapping, double u, int threads,
                    int32_t* src, int* resampled, double* alpha_out, double* stage_s)
{
	Model md = make_model(p);
	int overflow = 0;
	double st[4] = {0, 0, 0, 0};
#ifdef _OPENMP
	if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : st[:4]) reduction(| : overflow)
#endif
	for (int i = 0; i < P; i++) {
		Pose ps = make_pose(poses + i * 7);
		size_t o = (size_t) i * cap;
#ifdef _OPENMP
		double t0 = omp_get_wtime();
#endif
		Mixture model     = load_mixture(w + o, mean + o * 3, cov + o * 9, n[i]);
		Mixture predicted = predict(md, ps, model, z, M);
#ifdef _OPENMP
		double t1 = omp_get_wtime();
#endif
		Mixture corrected = correct(md, ps, predicted, z, M);
#ifdef _OPENMP
		double t2 = omp_get_wtime();
#endif
		corrected = prune(md, corrected);
#ifdef _OPENMP
		double t3 = omp_get_wtime();
#endif
		double alpha = 1.0;
		if (!onlymapping) {
			alpha = weight_alpha(md, ps, predicted, corrected, z, M);
			weights[i] *= alpha;
		}
#ifdef _OPENMP
		double t4 = omp_get_wtime();
		st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2; st[3] += t4 - t3;
#endif
		if (alpha_out) alpha_out[i] = alpha;
		if ((int) corrected.size() > cap) overflow |= 1;
		n[i] = std::min((int) corrected.size(), cap);
		store_mixture(corrected, cap, w + o, mean + o * 3, cov + o * 9);
	}
	if (stage_s) std::memcpy(stage_s, st, sizeof(st));
	if (overflow) return -2;

	int best = 0;
	*resampled = 0;
	for (int i = 0; i < P; i++) src[i] = i;
	if (!onlymapping) {
		double sum = 0;
		for (int i = 0; i < P; i++) sum += weights[i];   // Accord Sum(), sequential
		sum = (sum == 0) ? 1 : sum;
		for (int i = 0; i < P; i++) weights[i] = weights[i] / sum;
		double maxweight = 0;
		for (int i = 0; i < P; i++) {
			if (weights[i] > maxweight) { maxweight = weights[i]; best = i; }
		}
		if (particle_depleted(md, weights, P)) {
			*resampled = 1;
			best = resample(weights, P, u, src);
			// deep copies (:740-742)
			std::vector<double> nw((size_t) P * cap), nm((size_t) P * cap * 3), nc((size_t) P * cap * 9), np((size_t) P * 7);
			std::vector<int> nn(P);
			for (int i = 0; i < P; i++) {
				size_t d = (size_t) i * cap, s = (size_t) src[i] * cap;
				nn[i] = n[src[i]];
				std::memcpy(&nw[d], w + s, nn[i] * sizeof(double));
				std::memcpy(&nm[d * 3], mean + s * 3, nn[i] * 3 * sizeof(double));
				std::memcpy(&nc[d * 9], cov + s * 9, nn[i] * 9 * sizeof(double));
				std::memcpy(&np[i * 7], poses + src[i] * 7, 7 * sizeof(double));
			}
			std::memcpy(w, nw.data(), nw.size() * sizeof(double));
			std::memcpy(mean, nm.data(), nm.size() * sizeof(double));
			std::memcpy(cov, nc.data(), nc.size() * sizeof(double));
			std::memcpy(poses, np.data(), np.size() * sizeof(double));
			std::memcpy(n, nn.data(), nn.size() * sizeof(int));
			for (int i = 0; i < P; i++) weights[i] = 1.0 / P;
		}
	}
	return best;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

// ---- particle motion -------------------------------------------------------------
// Quaternion.Exp (Quaternion.cs:185-196), Sqrt (:225-235), Normalize (:240-245), Log / ToLinear (:135-139, :203-217)
static Quat qexp(const double* lie)
{
	double phi = std::sqrt(lie[0] * lie[0] + lie[1] * lie[1] + lie[2] * lie[2]);
	if (phi < 1e-12) return Quat{1, 0, 0, 0};
	double s = std::sin(phi);
	return Quat{std::cos(phi), s * (lie[0] / phi), s * (lie[1] / phi), s * (lie[2] / phi)};
}

static Quat qsqrt(const Quat& q)
{
	if (std::fabs(q.w - -1.0) < 1e-8) return Quat{1, 0, 0, 0};
	double rw = std::sqrt(0.5 * (1 + q.w)), alpha = 1 / (2 * rw);
	return Quat{rw, alpha * q.x, alpha * q.y, alpha * q.z};
}

static Quat qnormalize(const Quat& q)
{
	double alpha = 1 / std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
	return Quat{alpha * q.w, alpha * q.x, alpha * q.y, alpha * q.z};
}

// Pose3D.AddOdometry (Pose3D.cs:314-333): state in, state out (x y z qw qx qy qz). The orientation of the input
// is used as stored (the reference's Pose3D(location, orientation) constructor does not normalise, :169-176).
void orc_add_odometry(const double* pose7, const double* delta6, double* out7)
{
	const Quat q{pose7[3], pose7[4], pose7[5], pose7[6]};
	const double half[3] = {0.5 * delta6[3], 0.5 * delta6[4], 0.5 * delta6[5]};   // FromLinear, Quaternion.cs:145-149
	Quat dorientation   = qexp(half);
	Quat neworientation = qmul(q, dorientation);
	Quat middelta       = qsqrt(dorientation);
	Quat midrotation    = qmul(q, middelta);
	Quat dl = qmul(qmul(midrotation, Quat{0, delta6[0], delta6[1], delta6[2]}), qconj(midrotation));
	Quat o  = qnormalize(neworientation);
	out7[0] = pose7[0] + dl.x; out7[1] = pose7[1] + dl.y; out7[2] = pose7[2] + dl.z;
	out7[3] = o.w; out7[4] = o.x; out7[5] = o.y; out7[6] = o.z;
}

// Pose3D.Add (Pose3D.cs:282-291): a linear vector in semi-Lie space added around the pose
void orc_pose_add(const double* pose7, const double* delta6, double* out7)
{
	const Quat q{pose7[3], pose7[4], pose7[5], pose7[6]};
	const double half[3] = {0.5 * delta6[3], 0.5 * delta6[4], 0.5 * delta6[5]};   // Quaternion.Add, Quaternion.cs:165-168
	Quat nq = qnormalize(qmul(q, qexp(half)));
	Quat dl = qmul(qmul(q, Quat{0, delta6[0], delta6[1], delta6[2]}), qconj(q));
	out7[0] = pose7[0] + dl.x; out7[1] = pose7[1] + dl.y; out7[2] = pose7[2] + dl.z;
	out7[3] = nq.w; out7[4] = nq.x; out7[5] = nq.y; out7[6] = nq.z;
}

// Pose3D.DiffOdometry (Pose3D.cs:338-356): the delta that takes `origin` to `pose`
void orc_diff_odometry(const double* pose7, const double* origin7, double* delta6)
{
	const Quat qa{pose7[3], pose7[4], pose7[5], pose7[6]}, qo{origin7[3], origin7[4], origin7[5], origin7[6]};
	Quat dq = qmul(qconj(qo), qa);
	Quat mid = qmul(qo, qsqrt(dq));
	Quat dx = qmul(qmul(qconj(mid), Quat{0, pose7[0] - origin7[0], pose7[1] - origin7[1], pose7[2] - origin7[2]}), mid);
	Quat n = qnormalize(dq);                                      // Log normalises first, :205
	double phi = std::acos(n.w), mag = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
	double lie[3] = {0, 0, 0};
	if (!(mag < 1e-12)) { lie[0] = phi * (n.x / mag); lie[1] = phi * (n.y / mag); lie[2] = phi * (n.z / mag); }
	delta6[0] = dx.x; delta6[1] = dx.y; delta6[2] = dx.z;
	delta6[3] = 2 * lie[0]; delta6[4] = 2 * lie[1]; delta6[5] = 2 * lie[2];   // ToLinear = 2 Log, :135-139
}

// Quaternion.CreateFromYawPitchRoll (Quaternion.cs:254-273), for the fixtures of Pose3DTest
void orc_quaternion_ypr(double yaw, double pitch, double roll, double* q4)
{
	double y2 = 0.5 * yaw, p2 = 0.5 * pitch, r2 = 0.5 * roll;
	double sy = std::sin(y2), cy = std::cos(y2), sp = std::sin(p2), cp = std::cos(p2), sr = std::sin(r2), cr = std::cos(r2);
	q4[0] = cy * cp * cr + sy * sp * sr; q4[1] = cy * sp * cr + sy * cp * sr;
	q4[2] = sy * cp * cr - cy * sp * sr; q4[3] = cy * cp * sr - sy * sp * cr;
}

// TrackVehicle.UpdateNoisy (TrackVehicle.cs:89-102) for every particle: the odometry reading, then the particle's own
// noise vector (dt * chol(Q) * N(0, I), drawn by the host: Util.cs:173-202) unless PerfectStill holds and the reading is 0
void orc_update_motion(double* poses7, int nparticles, const double* odometry6, const double* noise6, int perfect_still)
{
	bool zero = true;
	for (int t = 0; t < 6; t++) zero = zero && odometry6[t] == 0;
	for (int i = 0; i < nparticles; i++) {
		double tmp[7];
		orc_add_odometry(poses7 + (size_t) i * 7, odometry6, tmp);
		if (noise6 && !(perfect_still && zero)) orc_add_odometry(tmp, noise6 + (size_t) i * 6, poses7 + (size_t) i * 7);
		else for (int t = 0; t < 7; t++) poses7[(size_t) i * 7 + t] = tmp[t];
	}
}

// OSPA distance between two landmark sets (postanalysis/Plot.cs:531-581; LandmarkDistance :583-586), the acceptance
// metric of SURVEY 8d (C = 1, P = 1 there). The transport problem is solved on C^P - d^P (entries below 1e-5 are left
// at the sparse matrix's default 0, :562), maximised by the same Hungarian as the association step (:573), then mapped
// back with x -> C^P - x, default included (:575; SparseMatrix.Apply, SparseMatrix.cs:526-542).
double orc_ospa(const double* a3, int na, const double* b3, int nb, double C, double P, double* cardinality)
{
	if (na > nb) { std::swap(a3, b3); std::swap(na, nb); }
	if (na == 0) {
		double c = (nb == 0) ? 0 : C;
		if (cardinality) *cardinality = c;
		return c;
	}
	const double CP = std::pow(C, P);
	Dense t; t.n = nb; t.v.assign((size_t) nb * nb, 0.0);
	for (int i = 0; i < na; i++) {
		for (int k = 0; k < nb; k++) {
			double d0 = a3[i * 3] - b3[k * 3], d1 = a3[i * 3 + 1] - b3[k * 3 + 1], d2 = a3[i * 3 + 2] - b3[k * 3 + 2];
			double dist = std::pow(std::min(C, std::sqrt(d0 * d0 + d1 * d1 + d2 * d2)), P);
			if (CP - dist > 1e-5) t.v[(size_t) i * nb + k] = CP - dist;
		}
	}
	std::vector<int> best;
	hungarian(t, best);
	for (double& x : t.v) x = CP - x;
	if (cardinality) *cardinality = C * std::pow((double) (nb - na) / nb, 1.0 / P);
	return std::pow(assignment_value(t, best) / nb, 1.0 / P);
}

// Plot.MapError, one frame (postanalysis/Plot.cs:489-524): the map estimate aligned by the pose error at the reference
// time, then OSPA against the visited map and its spatial part.
double orc_map_error(const double* visited3, int nv, const double* estimate3, int ne, int hasreference,
                     const double* estimatedpose7, const double* truepose7, double C, double P, double* spatial)
{
	std::vector<double> ref(estimate3, estimate3 + (size_t) ne * 3);
	if (hasreference) {
		// delta = dummy.FromLinear(estimate.Subtract(truth)) (:502-503); Subtract: Pose3D.cs:296-308
		const Pose e = make_pose(estimatedpose7), t = make_pose(truepose7);
		Quat dq = qnormalize(qmul(qconj(t.q), e.q));
		Quat dx = qmul(qmul(qconj(t.q), Quat{0, e.t[0] - t.t[0], e.t[1] - t.t[1], e.t[2] - t.t[2]}), t.q);
		double phi = std::acos(std::min(1.0, std::max(-1.0, dq.w)));
		double mag = std::sqrt(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z);
		double lin[6] = {dx.x, dx.y, dx.z, 0, 0, 0};   // ToLinear = 2 Log (Quaternion.cs:176-179, :204-218)
		if (!(mag < 1e-12)) { lin[3] = 2 * phi * dq.x / mag; lin[4] = 2 * phi * dq.y / mag; lin[5] = 2 * phi * dq.z / mag; }
		const double identity[7] = {0, 0, 0, 1, 0, 0, 0};
		double delta[7];
		orc_pose_add(identity, lin, delta);            // FromLinear = Identity.Add (Pose3D.cs:248-251)
		double R[9];
		qmatrix(qconj(Quat{delta[3], delta[4], delta[5], delta[6]}), R);   // drotation (:506)
		for (int j = 0; j < ne; j++) {                 // :514-515
			double d[3] = {ref[j * 3] - e.t[0], ref[j * 3 + 1] - e.t[1], ref[j * 3 + 2] - e.t[2]};
			for (int i = 0; i < 3; i++) {
				ref[j * 3 + i] = (R[i * 3] * d[0] + R[i * 3 + 1] * d[1] + R[i * 3 + 2] * d[2]) + e.t[i] + (-1.0) * delta[i];
			}
		}
	}
	double card = 0;
	double ospa = orc_ospa(visited3, nv, ref.data(), ne, C, P, &card);
	if (spatial) *spatial = std::pow(std::pow(ospa, P) - std::pow(card, P), 1.0 / P);   // :521
	return ospa;
}

}  // extern "C"
