// phd_device.h — device-side parameter block and FP64 math of the PRM3D measurement model.
// gfx950 only. Everything is IEEE double like the reference (all state is `double[]`/`double[][]`).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PHD_WAVE 64

// The reference runs on the CLR: IEEE doubles, no fused multiply-add. Everything that restates its arithmetic — the
// measurement model, S and its inverse, the Kalman update, Merge, the closeness test — is compiled without FP contraction so
// that it rounds like the reference (and like the oracle, built with -ffp-contract=off): means and covariances then leave
// CorrectConditional and PruneModel bit for bit as the oracle's, and discrete decisions (AreClose, the MinWeight cut) see
// the same numbers. Only the dense pair loops (gauss_logw + exp_neg: sums of thousands of terms, compared with a
// tolerance) keep their fused multiply-adds. First statement of a function body.
#define PHD_REF_ARITH _Pragma("clang fp contract(off)")

// In-kernel stamps for a separate DIAGNOSTIC build only (hipcc -DPHD_STAMPS -> libphdhip_stamps.so, see
// scripts/stamps.py): thread 0 of every workgroup records the shader clock at phase boundaries and leaves the
// differences in a debug slab that no kernel reads. The product build compiles them to nothing.
#ifdef PHD_STAMPS
#define PHD_STAMP_DECL long long stamp_[12]; for (int s_ = 0; s_ < 12; s_++) stamp_[s_] = 0
#define PHD_STAMP(i) stamp_[i] = clock64()
#define PHD_STAMP_FLUSH(id, n) if (threadIdx.x == 0 && a.stamps && a.stamp_kernel == (id)) { for (int s_ = 0; s_ < (n); s_++) a.stamps[(size_t) (a.p0 + blockIdx.x) * 16 + s_] = (double) (stamp_[s_] - stamp_[0]); }
// ... and the launch's TIMELINE (PHD_STAMP_KERNEL = 100 + id): every workgroup's start and end on the constant 100 MHz counter and the
// place it ran at (HW_ID: SIMD / CU / SE; XCC_ID), for scripts/timeline.py: ramp, rounds and tail of a launch.
#define PHD_TL_BEGIN const long long tl0_ = wall_clock64()
#define PHD_TL_END(id) if (threadIdx.x == 0 && a.stamps && a.stamp_kernel == 100 + (id)) { double* o_ = a.stamps + (size_t) (a.p0 + blockIdx.x) * 16; \
	o_[0] = (double) tl0_; o_[1] = (double) wall_clock64(); o_[2] = (double) __builtin_amdgcn_s_getreg((31 << 11) | 4); o_[3] = (double) __builtin_amdgcn_s_getreg((31 << 11) | 20); } \
	else if (threadIdx.x == 0 && a.stamps && a.stamp_kernel == 199) { double* o_ = a.stamps + (size_t) (a.p0 + blockIdx.x) * 16 + ((id) == 6 ? 0 : ((id) == 2 ? 2 : ((id) == 3 ? 4 : 6))); \
	o_[0] = (double) tl0_; o_[1] = (double) wall_clock64(); }   /* 199: all four kernels of a step side by side (scripts/timeline_step.py) */
#else
#define PHD_STAMP_DECL
#define PHD_STAMP(i)
#define PHD_STAMP_FLUSH(id, n)
#define PHD_TL_BEGIN
#define PHD_TL_END(id)
#endif

// Wave priority of the latency-bound kernels / the dense ones (s_setprio 0..3; experiment switches, 0 = leave it alone)
#ifndef PHD_LAT_PRIO
#define PHD_LAT_PRIO 0
#endif
#ifndef PHD_DENSE_PRIO
#define PHD_DENSE_PRIO 0
#endif
#define PHD_SET_PRIO(x) do { if ((x) > 0) __builtin_amdgcn_s_setprio(x); } while (0)

// Parameters as the kernels consume them (built once on the host from phd_params).
struct DevParams {
	// PRM3DMeasurer (PRM3DMeasurer.cs:55-73): focal, float32 range clip, integer film rectangle
	double focal;
	double rmin, rmax;
	double left, right, top, bottom;
	double ramp[3];
	double R[9];            // measurement covariance
	double Rinv[9];         // its inverse (Gaussian(mlinear, R, pd) in SetLogLikeMatrix, PHDNavigator.cs:429)
	double logRmult;        // log of that Gaussian's multiplier
	double pd, kappa, logkappa;
	double birthP[6];       // BirthCovariance, upper triangle
	double birthw;
	double minw;
	double expl_thr;
	double g2_correct;      // radius gate of Map.Near(x, DensityDistanceThreshold) (PHDNavigator.cs:882) as a bound on the
	double g2_explore;      // SQUARED distance; same for Map.Evaluate(x, 3 * DensityDistanceThreshold) (:958). Squared-
	                        // Euclidean metric: the radius itself; Euclidean: radius^2; gate disabled: +inf
	double merge_thr2;      // MergeThreshold^2                 (Gaussian.cs:245)
	double g2_quasi;        // smallest q with sqrt(q) >= 12: the gate of QuasiSetLogLikelihood (:615)
	double g2_assoc;        // smallest q with sqrt(q) >= 5: `Mahalanobis < 5` (PHDNavigator.cs:436) is q < g2_assoc exactly
	double min_eff;
	double emit_log_floor;  // log(minw * kappa): no emitted weight can come from below it
	int    gate_metric;
	int    maxq;
	int    linear2d;        // 1: the Linear2D toy model of the reference's unit tests (LinearPose2D, LinearMeasurement2D,
	                        // Linear2DMeasurer.cs) carried in the three-dimensional machinery: the third measurement
	                        // coordinate is always 0, its row of H is 0 and R gets a 1 there, so S, its determinant, K and
	                        // every quadratic form are those of the 2-D model (the multiplier exponent is -1 in both, Gaussian.cs:155)
	double lin_range;       // Linear2DMeasurer.Range: the visible square [-range, range]^2
};

#define PHD_INV_2PI 0.15915494309189535   // Math.Pow(2 * Math.PI, -3 / 2) with C# integer division (Gaussian.cs:155)

struct PoseD {
	double t[3];
	double qw, qx, qy, qz;   // normalised (Pose3D.cs:157-161)
};

__device__ __forceinline__ PoseD load_pose(const double* __restrict__ s)
{
	PHD_REF_ARITH
	PoseD p;
	p.t[0] = s[0]; p.t[1] = s[1]; p.t[2] = s[2];
	double w = s[3], x = s[4], y = s[5], z = s[6];
	double a = 1.0 / sqrt(w * w + x * x + y * y + z * z);
	p.qw = a * w; p.qx = a * x; p.qy = a * y; p.qz = a * z;
	return p;
}

// Hamilton product (Quaternion.cs:295-301)
__device__ __forceinline__ void qmul(double aw, double ax, double ay, double az,
                                     double bw, double bx, double by, double bz,
                                     double& w, double& x, double& y, double& z)
{
	PHD_REF_ARITH
	w = aw * bw - (ax * bx + ay * by + az * bz);
	x = aw * bx + ax * bw + ay * bz - az * by;
	y = aw * by + ay * bw + az * bx - ax * bz;
	z = aw * bz + az * bw + ax * by - ay * bx;
}

// local = q* (0, d) q : world -> sensor frame (PRM3DMeasurer.cs:141-142)
__device__ __forceinline__ void to_local(const PoseD& p, const double d[3], double l[3])
{
	PHD_REF_ARITH
	double w1, x1, y1, z1, w2;
	qmul(p.qw, -p.qx, -p.qy, -p.qz, 0.0, d[0], d[1], d[2], w1, x1, y1, z1);
	qmul(w1, x1, y1, z1, p.qw, p.qx, p.qy, p.qz, w2, l[0], l[1], l[2]);
}

// Quaternion.Conjugate().ToMatrix() (Quaternion.cs:327-342 on (w, -x, -y, -z))
__device__ __forceinline__ void conj_matrix(const PoseD& p, double r[9])
{
	PHD_REF_ARITH
	double X = -p.qx, Y = -p.qy, Z = -p.qz, W = p.qw;
	double xx = X * X, yy = Y * Y, zz = Z * Z;
	double xy = X * Y, xz = X * Z, xw = X * W;
	double yz = Y * Z, yw = Y * W, zw = Z * W;
	r[0] = 1 - 2 * (yy + zz); r[1] = 2 * (xy - zw);     r[2] = 2 * (xz + yw);
	r[3] = 2 * (xy + zw);     r[4] = 1 - 2 * (xx + zz); r[5] = 2 * (yz - xw);
	r[6] = 2 * (xz - yw);     r[7] = 2 * (yz + xw);     r[8] = 1 - 2 * (xx + yy);
}

// MeasureToMap (PRM3DMeasurer.cs:299-312)
__device__ __forceinline__ void measure_to_map(const DevParams& prm, const PoseD& p, const double z[3], double x[3])
{
	PHD_REF_ARITH
	if (prm.linear2d) {   // Linear2DMeasurer.MeasureToMap (Linear2DMeasurer.cs:181-184)
		x[0] = p.t[0] + z[0]; x[1] = p.t[1] + z[1]; x[2] = 0.0;
		return;
	}
	double f = prm.focal;
	double alpha = z[2] / sqrt(f * f + z[0] * z[0] + z[1] * z[1]);
	double dx = alpha * z[0], dy = alpha * z[1], dz = alpha * f;
	double w1, x1, y1, z1, w2, rx, ry, rz;
	qmul(p.qw, p.qx, p.qy, p.qz, 0.0, dx, dy, dz, w1, x1, y1, z1);
	qmul(w1, x1, y1, z1, p.qw, -p.qx, -p.qy, -p.qz, w2, rx, ry, rz);
	x[0] = p.t[0] + rx; x[1] = p.t[1] + ry; x[2] = p.t[2] + rz;
}

// MeasurePerfect (PRM3DMeasurer.cs:138-149); also returns the local vector for the Jacobian
__device__ __forceinline__ void measure_perfect(const DevParams& prm, const PoseD& p, const double m[3],
                                                double zh[3], double l[3])
{
	PHD_REF_ARITH
	if (prm.linear2d) {   // Linear2DMeasurer.MeasurePerfect (Linear2DMeasurer.cs:110-113)
		zh[0] = m[0] - p.t[0]; zh[1] = m[1] - p.t[1]; zh[2] = 0.0;
		l[0] = 0.0; l[1] = 0.0; l[2] = 1.0;
		return;
	}
	double d[3] = {m[0] - p.t[0], m[1] - p.t[1], m[2] - p.t[2]};
	to_local(p, d, l);
	double euclid = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
	double sgn = (l[2] > 0) ? 1.0 : ((l[2] < 0) ? -1.0 : 0.0);   // Math.Sign
	zh[2] = sgn * euclid;
	zh[0] = prm.focal * l[0] / l[2];
	zh[1] = prm.focal * l[1] / l[2];
}

// FuzzyVisibleM (PRM3DMeasurer.cs:277-291) * detectionProbability (SimulatedVehicle.cs:335-338)
__device__ __forceinline__ double detection_probability_m(const DevParams& prm, const double z[3])
{
	PHD_REF_ARITH
	if (prm.linear2d) {   // Linear2DMeasurer.FuzzyVisibleM (Linear2DMeasurer.cs:151-162)
		double mind = (z[0] + prm.lin_range) / prm.ramp[0];
		mind = fmin(mind, (prm.lin_range - z[0]) / prm.ramp[0]);
		mind = fmin(mind, (z[1] + prm.lin_range) / prm.ramp[1]);
		mind = fmin(mind, (prm.lin_range - z[1]) / prm.ramp[1]);
		return fmax(0.0, fmin(1.0, mind)) * prm.pd;
	}
	// (the reference divides all six distances; a correctly rounded division by a positive number is monotone, so the
	// smaller quotient of a pair is the quotient of the smaller distance, bit for bit: three divisions)
	double mind = fmin(z[0] - prm.left, prm.right - z[0]) / prm.ramp[0];
	mind = fmin(mind, fmin(z[1] - prm.top, prm.bottom - z[1]) / prm.ramp[1]);
	mind = fmin(mind, fmin(z[2] - prm.rmin, prm.rmax - z[2]) / prm.ramp[2]);
	return fmax(0.0, fmin(1.0, mind)) * prm.pd;
}

// MeasurementJacobianL (PRM3DMeasurer.cs:157-177): H = Jproj(local) * R(q*)
__device__ __forceinline__ void jacobian_l(const DevParams& prm, const double l[3], const double rq[9], double H[9])
{
	PHD_REF_ARITH
	if (prm.linear2d) {   // Linear2DMeasurer.MeasurementJacobianL (Linear2DMeasurer.cs:115-119), a zero third row
		H[0] = 1; H[1] = 0; H[2] = 0;  H[3] = 0; H[4] = 1; H[5] = 0;  H[6] = 0; H[7] = 0; H[8] = 0;
		return;
	}
	double f = prm.focal;
	double mag = ((l[2] > 0) ? 1.0 : -1.0) * sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
	// H = jp * rq with jp = {f / l2, 0, -f l0 / l2^2;  0, f / l2, -f l1 / l2^2;  l / mag} (the reference multiplies the full
	// matrices, zeros included). A term 0 * rq is +-0 (rq is finite: the ABI rejects other poses) and the running sum it would be
	// added to is either non-zero or +0 — it starts as +0, and (+0) + (-0) = +0 — so leaving the two zero terms of the first two
	// rows out changes no bit; the leading "0.0 +" stays (it turns a first product of -0 into the +0 the reference's sum holds).
	// Six multiplications and six additions less per component — and, the rotation being the same for all of a particle's
	// components, six hoisted products less for the compiler to keep (it spilled them inside k_emit_prune).
	const double j00 = f / l[2], j02 = -f * l[0] / (l[2] * l[2]), j12 = -f * l[1] / (l[2] * l[2]);
	const double j20 = l[0] / mag, j21 = l[1] / mag, j22 = l[2] / mag;
#pragma unroll
	for (int j = 0; j < 3; j++) {
		H[j]     = (0.0 + j00 * rq[j]) + j02 * rq[6 + j];
		H[3 + j] = (0.0 + j00 * rq[3 + j]) + j12 * rq[6 + j];
		H[6 + j] = ((0.0 + j20 * rq[j]) + j21 * rq[3 + j]) + j22 * rq[6 + j];
	}
}

// inverse (upper triangle) and determinant of a symmetric 3x3 given as xx,xy,xz,yy,yz,zz — the
// cofactor formulas of the general inverse with the symmetric entries substituted.
__device__ __forceinline__ void inv_sym3(const double P[6], double inv[6], double& det)
{
	PHD_REF_ARITH
	double xx = P[0], xy = P[1], xz = P[2], yy = P[3], yz = P[4], zz = P[5];
	double c00 = yy * zz - yz * yz;
	double c01 = xy * zz - yz * xz;
	double c02 = xy * yz - yy * xz;
	det = xx * c00 - xy * c01 + xz * c02;
	double id = 1.0 / det;
	inv[0] = c00 * id;
	inv[1] = (xz * yz - xy * zz) * id;
	inv[2] = (xy * yz - xz * yy) * id;
	inv[3] = (xx * zz - xz * xz) * id;
	inv[4] = (xz * xy - xx * yz) * id;
	inv[5] = (xx * yy - xy * xy) * id;
}

// inverse and determinant of a general 3x3, row-major
__device__ __forceinline__ void inv_gen3(const double a[9], double inv[9], double& det)
{
	PHD_REF_ARITH
	double c00 = a[4] * a[8] - a[5] * a[7];
	double c01 = a[3] * a[8] - a[5] * a[6];
	double c02 = a[3] * a[7] - a[4] * a[6];
	det = a[0] * c00 - a[1] * c01 + a[2] * c02;
	double id = 1.0 / det;
	inv[0] = c00 * id;
	inv[1] = (a[2] * a[7] - a[1] * a[8]) * id;
	inv[2] = (a[1] * a[5] - a[2] * a[4]) * id;
	inv[3] = (a[5] * a[6] - a[3] * a[8]) * id;
	inv[4] = (a[0] * a[8] - a[2] * a[6]) * id;
	inv[5] = (a[2] * a[3] - a[0] * a[5]) * id;
	inv[6] = c02 * id;
	inv[7] = (a[1] * a[6] - a[0] * a[7]) * id;
	inv[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

// d^T A d for symmetric A (upper triangle), accumulated row by row like Gaussian.Evaluate
__device__ __forceinline__ double quad_sym(const double A[6], double d0, double d1, double d2)
{
	PHD_REF_ARITH
	double r0 = A[0] * d0 + A[1] * d1 + A[2] * d2;
	double r1 = A[1] * d0 + A[3] * d1 + A[4] * d2;
	double r2 = A[2] * d0 + A[4] * d1 + A[5] * d2;
	return d0 * r0 + d1 * r1 + d2 * r2;
}

__device__ __forceinline__ double quad_gen(const double A[9], double d0, double d1, double d2)
{
	PHD_REF_ARITH
	double r0 = A[0] * d0 + A[1] * d1 + A[2] * d2;
	double r1 = A[3] * d0 + A[4] * d1 + A[5] * d2;
	double r2 = A[6] * d0 + A[7] * d1 + A[8] * d2;
	return d0 * r0 + d1 * r1 + d2 * r2;
}

// exp(x) for the Gaussian exponents of the pair loops (x <= 0 in exact arithmetic):
//   x = (32 k + j) ln2/32 + r,  exp(x) = 2^k * 2^(j/32) * e^r,  |r| <= ln2/64
// with 2^(j/32) from a 32-entry table in LDS and e^r by a degree-6 Taylor polynomial: a dependent chain of
// 6 fused multiply-adds instead of the ~14 of a table-free evaluation. Measured against libm on [-700, 0]:
// <= 2 ulp (3.9e-16). A NaN stays a NaN, anything below -800 gives 0 like exp does.
#define EXPTAB_N 256
__device__ __forceinline__ void exp_tab_init(double* T, int tid)
{
	if (tid < EXPTAB_N) T[tid] = exp2((double) tid / EXPTAB_N);
}

// exp(x) for x in (-inf, 700]: x = (256 n' + j) ln2 / 256 + r, exp(x) = 2^n' * T[j] * p(r) with |r| <= ln2 / 512 and
// p the degree-4 Taylor polynomial (truncation 4e-17 relative). Below -800 the result is 0; a NaN stays a NaN.
__device__ __forceinline__ double exp_neg(double x, const double* __restrict__ T)
{
	x = (x < -800.0) ? -800.0 : x;
	const double n = rint(x * 369.3299304675746);                // 256 / ln 2
	double r = fma(-n, 0.002707606166950427, x);                 // ln2/256, high part (trailing bits zero)
	r = fma(-n, 7.111859369156821e-12, r);                      //          low part
	const int ni = (int) n;
	const double t = T[ni & (EXPTAB_N - 1)];
	double p = fma(r, 0.041666666666666664, 0.16666666666666666);
	p = fma(p, r, 0.5);
	p = fma(p, r, 1.0);
	p = fma(p, r, 1.0);
	return ldexp(t * p, ni >> 8);
}

// The same for the dense pair loops (k_sweep's weight sums, k_alpha_density), four instructions leaner at the front: the
// argument is clamped by one v_max_f64 and rounded by adding 1.5 * 2^52, which leaves the integer in the low word of the
// sum (no v_rndne / v_cvt). Same table, a polynomial one degree shorter (relative 1.4e-13, below); a NaN
// argument counts as exp(-800) = 0 here (v_max_f64 returns the other operand), where exp_neg keeps it a NaN.
// keep = false: the result is 0 (the exponent handed to v_ldexp_f64 is replaced: one 32-bit select instead of two on the value)
__device__ __forceinline__ double exp_pair(double x, const double* __restrict__ T, bool keep = true)
{
	// (written out: fmax() puts a canonicalising v_max_f64 in front of the clamp, and the compiler turns the first step of
	// the polynomial into a register copy + v_fmac_f64 — two instructions each where one does)
	const double lo = -800.0, c3 = 0.16666666666666666;
	asm("v_max_f64 %0, %1, %2" : "=v"(x) : "v"(x), "s"(lo));
	const double nd = fma(x, 369.3299304675746, 6755399441055744.0);   // 256 / ln 2; 1.5 * 2^52
	const int ni = __double2loint(nd);
	const double n = nd - 6755399441055744.0;
	// (ln 2 / 256 as ONE constant: the product inside the fused multiply-add is exact, what is lost is the constant's own
	// rounding, 2.2e-19, times n = 369 |x| — a relative 8e-17 |x| of exp(x), i.e. below an ulp for the terms that carry a sum
	// (|x| of a few) and far below the sum's own rounding for the terms that are small against them. exp_neg keeps both parts.)
	const double r = fma(-n, 0.0027076061740622863, x);
	const double t = T[ni & (EXPTAB_N - 1)];
	// (degree 3 here — exp_neg keeps degree 4 —: |r| <= ln 2 / 512, so the first term left out, r^4 / 24, is a relative 1.4e-13 of
	// every term of a sum that is compared at 1e-9; one fused multiply-add less of the 36 / 28 instructions per pair)
	double p;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "s"(c3), "v"(r), "v"(0.5));
	p = fma(p, r, 1.0);
	p = fma(p, r, 1.0);
	return ldexp(t * p, keep ? (ni >> 8) : -4096);
}

// A weighted Gaussian w N(x; m, P) in the form the dense evaluation loops use: exp(g9 + d^T G d), d = x - m, with
// G = -P^-1 / 2 folded for the upper-triangle sum (off-diagonal entries doubled) and g9 = log(w mult).
__device__ __forceinline__ void gauss_record(double w, const double m[3], const double Pi[6], double mult, double* g)
{
	g[0] = m[0]; g[1] = m[1]; g[2] = m[2];
	g[3] = -0.5 * Pi[0]; g[4] = -Pi[1]; g[5] = -Pi[2];
	g[6] = -0.5 * Pi[3]; g[7] = -Pi[4];
	g[8] = -0.5 * Pi[5];
	g[9] = log(w * mult);
}

__device__ __forceinline__ double gauss_logw(const double* __restrict__ g, double d0, double d1, double d2)
{
	const double t0 = fma(g[5], d2, fma(g[4], d1, g[3] * d0));
	const double t1 = fma(g[7], d2, g[6] * d1);
	const double t2 = g[8] * d2;
	return fma(t0, d0, fma(t1, d1, fma(t2, d2, g[9])));
}

// per-component measurement-space quantities of CorrectConditional (PHDNavigator.cs:857-870)
struct CompMeas {
	double zh[3];     // h(m)
	double H[9];
	double PH[9];     // P H^T
	double Sinv[9];   // (H P H^T + R)^-1, full (S is not bitwise symmetric)
	double qmult;     // multiplier of N(.; zh, S)
	double pd;        // detection probability of the component
};

__device__ __forceinline__ void comp_measure(const DevParams& prm, const PoseD& pose, const double rq[9],
                                             const double m[3], const double P[6], CompMeas& o)
{
	PHD_REF_ARITH
	double l[3];
	measure_perfect(prm, pose, m, o.zh, l);
	jacobian_l(prm, l, rq, o.H);
	const double Pf[9] = {P[0], P[1], P[2], P[1], P[3], P[4], P[2], P[4], P[5]};
#pragma unroll
	for (int a = 0; a < 3; a++) {
#pragma unroll
		for (int b = 0; b < 3; b++) {
			double s = 0;
#pragma unroll
			for (int k = 0; k < 3; k++) {
				s += Pf[a * 3 + k] * o.H[b * 3 + k];
			}
			o.PH[a * 3 + b] = s;
		}
	}
	double S[9];
#pragma unroll
	for (int a = 0; a < 3; a++) {
#pragma unroll
		for (int b = 0; b < 3; b++) {
			double s = 0;
#pragma unroll
			for (int k = 0; k < 3; k++) {
				s += o.H[a * 3 + k] * o.PH[k * 3 + b];
			}
			S[a * 3 + b] = s + prm.R[a * 3 + b];
		}
	}
	double det;
	inv_gen3(S, o.Sinv, det);
	o.qmult = PHD_INV_2PI / sqrt(fabs(det));
	o.pd    = detection_probability_m(prm, o.zh);
}

// Kalman update of one (component, measurement) pair (PHDNavigator.cs:895-897):
// K = PH Sinv, m' = m + K nu, P' = (I - K H) P (upper triangle kept)
__device__ __forceinline__ void kalman_gain(const CompMeas& cm, double K[9])
{
	PHD_REF_ARITH
#pragma unroll
	for (int a = 0; a < 3; a++) {
#pragma unroll
		for (int b = 0; b < 3; b++) {
			double s = 0;
#pragma unroll
			for (int e = 0; e < 3; e++) {
				s += cm.PH[a * 3 + e] * cm.Sinv[e * 3 + b];
			}
			K[a * 3 + b] = s;
		}
	}
}

__device__ __forceinline__ void kalman_cov(const CompMeas& cm, const double K[9], const double P[6], double Pn[6])
{
	PHD_REF_ARITH
	const double Pf[9] = {P[0], P[1], P[2], P[1], P[3], P[4], P[2], P[4], P[5]};
	double IKH[9];
#pragma unroll
	for (int a = 0; a < 3; a++) {
#pragma unroll
		for (int b = 0; b < 3; b++) {
			double s = 0;
#pragma unroll
			for (int e = 0; e < 3; e++) {
				s += K[a * 3 + e] * cm.H[e * 3 + b];
			}
			IKH[a * 3 + b] = ((a == b) ? 1.0 : 0.0) - s;
		}
	}
	const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
	for (int t = 0; t < 6; t++) {
		double s = 0;
#pragma unroll
		for (int e = 0; e < 3; e++) {
			s += IKH[ia[t] * 3 + e] * Pf[e * 3 + ib[t]];
		}
		Pn[t] = s;
	}
}

// wave-level helpers (wave = 64)
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// The lanes for which `p` holds, as a mask. (HIP's __ballot takes an int and compares it with 0 again: two vector
// instructions per call in the pair loops; this one is the compare's own result.)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// orders a wave's LDS writes before the reads of its other lanes (with __builtin_amdgcn_wave_barrier)
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }

__device__ __forceinline__ unsigned long long lanemask_lt()
{
	return (1ull << lane_id()) - 1ull;
}

// A value that is the same in every lane (a particle's pose, its rotation matrix), moved to scalar registers: it then
// costs no vector registers for the rest of the kernel.
__device__ __forceinline__ double uniform_d(double v)
{
	const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
	return __hiloint2double(hi, lo);
}

// s / d for 0 <= s < 2^24 and 1 <= d without the integer-division sequence (rd = 1.0f / d): a float quotient is off by at
// most one
__device__ __forceinline__ int small_div(int s, int d, float rd)
{
	int q = (int) ((float) s * rd);
	q -= (q * d > s) ? 1 : 0;
	q += ((q + 1) * d <= s) ? 1 : 0;
	return q;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		v += __shfl_xor(v, o, 64);
	}
	return v;
}
