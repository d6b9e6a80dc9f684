// vgs_math.h -- the arithmetic specification of the VGS/SVGS hot path ("DevMath").
//
// Every floating-point quantity the MI355X kernels compute (3x3 eigen solve, eigen features,
// the five perceptual-grouping distances, the exp weight) is defined HERE, once, as plain
// float32 code built only from IEEE-exact primitives: + - * / sqrt fma and integer bit ops.
// No libm call, no fast-math, no implicit contraction (build with -ffp-contract=off; FMAs
// appear only as explicit vm_fma).  The same header therefore produces bit-identical results
// under hipcc for gfx950 and under g++ -mfma on the host.  That is what lets the parity
// tests demand *identical* voxel->segment label maps between the HIP path and the CPU oracle's
// DevMath mode.  The oracle's RefMath mode (oracle/refcpu.cpp) is an independent restatement
// with libm and the reference's C++ promotion rules; tests bound DevMath-vs-RefMath (P1/P2).
//
// Reference semantics restated (paths under /root/reference):
//   eigen features        voxel_segmentation.h:1147-1228   supervoxel_segmentation.h:745-847
//   measuringDistance     voxel_segmentation.h:1597-1720   supervoxel_segmentation.h:1756-1878
//   distanceWeight        voxel_segmentation.h:1722-1740   supervoxel_segmentation.h:1880-1905
//   pcl::eigen33          third party (PCL 1.8.1 common/eigen.hpp), call sites VS:1166,1403
#ifndef VGS_MATH_H_
#define VGS_MATH_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define VGS_HD __host__ __device__ __forceinline__
#else
#define VGS_HD inline
#endif

// ------------------------------------------------------------------------------------------
// IEEE-exact primitives
// ------------------------------------------------------------------------------------------
VGS_HD float vm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
VGS_HD float vm_sqrt(float x) { return __builtin_sqrtf(x); }
VGS_HD float vm_abs(float x) { return __builtin_fabsf(x); }
VGS_HD uint32_t vm_bits(float x) { union { float f; uint32_t u; } v; v.f = x; return v.u; }
VGS_HD float vm_from_bits(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }
VGS_HD float vm_nan() { return vm_from_bits(0x7fc00000u); }
VGS_HD bool vm_isnan(float x) { return x != x; }

#define VM_PI_F 3.1415926f /* the reference's PI literal (VS:1611), rounded to float */

// ------------------------------------------------------------------------------------------
// acos on [-1,1]; NaN outside (the reference does not clamp: VS:1650-1657, quirk Q3).
// Classic three-range evaluation with a (3,1) rational kernel for asin(x)/x - 1.
// ------------------------------------------------------------------------------------------
VGS_HD float vm_asin_kernel(float z) {
  const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f;
  const float qS1 = -7.0662963390e-01f;
  float p = z * (pS0 + z * (pS1 + z * pS2));
  float q = 1.0f + z * qS1;
  return p / q;
}

VGS_HD float vm_acos(float x) {
  const float pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
  const float pi_f = 3.1415925026e+00f;
  float ax = vm_abs(x);
  if (!(ax <= 1.0f)) return vm_nan();
  if (ax < 0.5f) {
    if (ax < 7.4505806e-09f) return pio2_hi + pio2_lo;
    float z = x * x;
    float r = vm_asin_kernel(z);
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (x < 0.0f) {
    float z = (1.0f + x) * 0.5f;
    float s = vm_sqrt(z);
    float w = vm_asin_kernel(z) * s - pio2_lo;
    return pi_f - 2.0f * (s + w);
  }
  float z = (1.0f - x) * 0.5f;
  float s = vm_sqrt(z);
  float df = vm_from_bits(vm_bits(s) & 0xfffff000u);
  float c = (z - df * df) / (s + df);
  float w = vm_asin_kernel(z) * s + c;
  return 2.0f * (df + w);
}

// ------------------------------------------------------------------------------------------
// exp for x <= 88; results below the normal range are defined as exactly 0
// (the reference computes exp in double and stores a float: anything that small is a dead edge).
// ------------------------------------------------------------------------------------------
VGS_HD float vm_exp(float x) {
  if (vm_isnan(x)) return x;
  if (x < -87.0f) return 0.0f;
  if (x > 88.0f) x = 88.0f;
  const float log2e = 1.4426950216e+00f;
  const float ln2_hi = 6.9314575195e-01f, ln2_lo = 1.4286067653e-06f;
  float t = x * log2e;
  float kf = (t + 12582912.0f) - 12582912.0f;  // round to nearest integer (|t| < 2^22)
  float r = vm_fma(kf, -ln2_hi, x);
  r = vm_fma(kf, -ln2_lo, r);
  float p = 1.9841270e-04f;
  p = vm_fma(p, r, 1.3888889e-03f);
  p = vm_fma(p, r, 8.3333338e-03f);
  p = vm_fma(p, r, 4.1666668e-02f);
  p = vm_fma(p, r, 1.6666667e-01f);
  p = vm_fma(p, r, 0.5f);
  p = vm_fma(p, r, 1.0f);
  p = vm_fma(p, r, 1.0f);
  int k = (int)kf;
  return p * vm_from_bits((uint32_t)(k + 127) << 23);
}

// ------------------------------------------------------------------------------------------
// natural log for x > 0 (normal or subnormal)
// ------------------------------------------------------------------------------------------
VGS_HD float vm_log(float x) {
  const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
  const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
  uint32_t ix = vm_bits(x);
  int k = 0;
  if (!(x > 0.0f)) return (x == 0.0f) ? -vm_from_bits(0x7f800000u) : vm_nan();
  if (ix >= 0x7f800000u) return x;  // +inf
  if (ix < 0x00800000u) {  // subnormal: scale up by 2^25
    x = x * 33554432.0f;
    ix = vm_bits(x);
    k = -25;
  }
  ix += 0x3f800000u - 0x3f3504f3u;
  k += (int)(ix >> 23) - 127;
  ix = (ix & 0x007fffffu) + 0x3f3504f3u;
  float m = vm_from_bits(ix);
  float f = m - 1.0f;
  float s = f / (2.0f + f);
  float z = s * s;
  float w = z * z;
  float t1 = w * (Lg2 + w * Lg4);
  float t2 = z * (Lg1 + w * Lg3);
  float R = t2 + t1;
  float hfsq = 0.5f * f * f;
  float dk = (float)k;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// ------------------------------------------------------------------------------------------
// atan for x >= 0 and atan2(y >= 0, x) (only the branch pcl::eigen33's root solver needs)
// ------------------------------------------------------------------------------------------
VGS_HD float vm_atan_pos(float x) {
  const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
  const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
  const float aT0 = 3.3333328366e-01f, aT1 = -1.9999158382e-01f, aT2 = 1.4253635705e-01f,
              aT3 = -1.0648017377e-01f, aT4 = 6.1687607318e-02f;
  int id;
  float hi = 0.0f, lo = 0.0f;
  if (x >= 6.7108864e+07f) return hi3 + lo3;  // 2^26
  if (x < 0.4375f) {
    id = -1;
  } else if (x < 1.1875f) {
    if (x < 0.6875f) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); hi = hi0; lo = lo0; }
    else { id = 1; x = (x - 1.0f) / (x + 1.0f); hi = hi1; lo = lo1; }
  } else {
    if (x < 2.4375f) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); hi = hi2; lo = lo2; }
    else { id = 3; x = -1.0f / x; hi = hi3; lo = lo3; }
  }
  float z = x * x;
  float w = z * z;
  float s1 = z * (aT0 + w * (aT2 + w * aT4));
  float s2 = w * (aT1 + w * aT3);
  if (id < 0) return x - x * (s1 + s2);
  return hi - ((x * (s1 + s2) - lo) - x);
}

VGS_HD float vm_atan2_ypos(float y, float x) {  // y >= 0
  const float pi_f = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const float pio2 = 1.5707963705e+00f;
  if (y == 0.0f) return (x >= 0.0f) ? 0.0f : pi_f;
  if (x == 0.0f) return pio2;
  float a = vm_atan_pos(y / vm_abs(x));
  if (x > 0.0f) return a;
  return pi_f - (a - pi_lo);
}

// sin / cos on [0, 1.1] (theta = atan2(...)/3 in [0, pi/3])
VGS_HD float vm_sin_k(float x) {  // |x| <= pi/4
  const float S1 = -1.6666667163e-01f, S2 = 8.3333337680e-03f, S3 = -1.9841270114e-04f, S4 = 2.7557314297e-06f;
  float z = x * x;
  float p = vm_fma(S4, z, S3);
  p = vm_fma(p, z, S2);
  p = vm_fma(p, z, S1);
  return vm_fma(x * z, p, x);
}
VGS_HD float vm_cos_k(float x) {  // |x| <= pi/4
  const float C1 = -0.5f, C2 = 4.1666667908e-02f, C3 = -1.3888889225e-03f, C4 = 2.4801587642e-05f, C5 = -2.7557314297e-07f;
  float z = x * x;
  float p = vm_fma(C5, z, C4);
  p = vm_fma(p, z, C3);
  p = vm_fma(p, z, C2);
  p = vm_fma(p, z, C1);
  return vm_fma(p, z, 1.0f);
}
VGS_HD void vm_sincos_small(float x, float* s, float* c) {  // 0 <= x <= ~1.1
  const float pio4 = 7.8539818525e-01f;
  const float pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
  if (x <= pio4) { *s = vm_sin_k(x); *c = vm_cos_k(x); return; }
  float y = (pio2_hi - x) + pio2_lo;
  *s = vm_cos_k(y);
  *c = vm_sin_k(y);
}

// ------------------------------------------------------------------------------------------
// 3x3 symmetric eigen decomposition, restating pcl::eigen33(mat, evecs, evals) for float
// (PCL 1.8.1 common/impl/eigen.hpp: computeRoots / computeRoots2 / eigen33; recalled from
// upstream -- PCL is not under /root/reference, see DESIGN.md "parity unpinned").
// m is row-major m[r*3+c]; evals ascending; evecs column k = evec[.][k] stored as ev[r*3+k].
// ------------------------------------------------------------------------------------------
#define VM_EPS_F 1.1920929e-07f
#define VM_FLT_MIN 1.17549435e-38f

VGS_HD void vm_roots2(float b, float c, float* roots) {
  roots[0] = 0.0f;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) d = 0.0f;
  float sd = vm_sqrt(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

VGS_HD void vm_roots3(const float* m, float* roots) {
  float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
  float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  float c2 = m[0] + m[4] + m[8];
  if (vm_abs(c0) < VM_EPS_F) { vm_roots2(c2, c1, roots); return; }
  const float s_inv3 = 0.33333334f;
  const float s_sqrt3 = 1.7320508f;
  float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0f) a_over_3 = 0.0f;
  float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0f) q = 0.0f;
  float rho = vm_sqrt(-a_over_3);
  float theta = vm_atan2_ypos(vm_sqrt(-q), half_b) * s_inv3;
  float sn, cs;
  vm_sincos_small(theta, &sn, &cs);
  roots[0] = c2_over_3 + 2.0f * rho * cs;
  roots[1] = c2_over_3 - rho * (cs + s_sqrt3 * sn);
  roots[2] = c2_over_3 - rho * (cs - s_sqrt3 * sn);
  float t;
  if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
  if (roots[1] >= roots[2]) {
    t = roots[1]; roots[1] = roots[2]; roots[2] = t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
  }
  if (roots[0] <= 0.0f) vm_roots2(c2, c1, roots);
}

VGS_HD void vm_cross(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
VGS_HD float vm_sqnorm(const float* a) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }

// largest of the three row cross products of (M - lambda I); returns its squared length
VGS_HD float vm_best_cross(const float* sm, float lambda, float* out) {
  float t[9];
  for (int i = 0; i < 9; ++i) t[i] = sm[i];
  t[0] -= lambda; t[4] -= lambda; t[8] -= lambda;
  float v1[3], v2[3], v3[3];
  vm_cross(&t[0], &t[3], v1);
  vm_cross(&t[0], &t[6], v2);
  vm_cross(&t[3], &t[6], v3);
  float l1 = vm_sqnorm(v1), l2 = vm_sqnorm(v2), l3 = vm_sqnorm(v3);
  // (values, not a pointer to one of the three: a selected POINTER keeps all three arrays in scratch memory on the GPU -- 40 bytes per
  // lane in k_features and in the supervoxel stage's tile set-up, a gigabyte of scratch traffic per launch of the latter; round 5)
  float vx, vy, vz, l;
  if (l1 >= l2 && l1 >= l3) { vx = v1[0]; vy = v1[1]; vz = v1[2]; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { vx = v2[0]; vy = v2[1]; vz = v2[2]; l = l2; }
  else { vx = v3[0]; vy = v3[1]; vz = v3[2]; l = l3; }
  float s = vm_sqrt(l);
  out[0] = vx / s; out[1] = vy / s; out[2] = vz / s;
  return l;
}

// Eigen's Vector3f::unitOrthogonal()
VGS_HD void vm_unit_orthogonal(const float* v, float* o) {
  // if (!isMuchSmallerThan(x, z) || !isMuchSmallerThan(y, z)) with precision 1e-5 (float dummy_precision)
  const float prec = 1e-5f;
  bool x_small = vm_abs(v[0]) <= vm_abs(v[2]) * prec;
  bool y_small = vm_abs(v[1]) <= vm_abs(v[2]) * prec;
  if (!x_small || !y_small) {
    float invnm = 1.0f / vm_sqrt(v[0] * v[0] + v[1] * v[1]);
    o[0] = -v[1] * invnm; o[1] = v[0] * invnm; o[2] = 0.0f;
  } else {
    float invnm = 1.0f / vm_sqrt(v[1] * v[1] + v[2] * v[2]);
    o[0] = 0.0f; o[1] = -v[2] * invnm; o[2] = v[1] * invnm;
  }
}

VGS_HD void vm_normalize3(float* v) {
  // Eigen normalized(): v / sqrt(squaredNorm) when squaredNorm > 0
  float n2 = vm_sqnorm(v);
  if (n2 > 0.0f) { float n = vm_sqrt(n2); v[0] /= n; v[1] /= n; v[2] /= n; }
}

VGS_HD void vm_eigen33(const float* mat, float* evecs /*[r*3+k]*/, float* evals) {
  float scale = 0.0f;
  for (int i = 0; i < 9; ++i) { float a = vm_abs(mat[i]); if (a > scale) scale = a; }
  if (scale <= VM_FLT_MIN) scale = 1.0f;
  float sm[9];
  for (int i = 0; i < 9; ++i) sm[i] = mat[i] / scale;
  vm_roots3(sm, evals);
  float c0[3], c1[3], c2[3];
  if ((evals[2] - evals[0]) <= VM_EPS_F) {
    c0[0] = 1; c0[1] = 0; c0[2] = 0; c1[0] = 0; c1[1] = 1; c1[2] = 0; c2[0] = 0; c2[1] = 0; c2[2] = 1;
  } else if ((evals[1] - evals[0]) <= VM_EPS_F) {
    vm_best_cross(sm, evals[2], c2);
    vm_unit_orthogonal(c2, c1);
    vm_cross(c1, c2, c0);
  } else if ((evals[2] - evals[1]) <= VM_EPS_F) {
    vm_best_cross(sm, evals[0], c0);
    vm_unit_orthogonal(c0, c1);
    vm_cross(c0, c1, c2);
  } else {
    // (the same selections as an array of column pointers indexed by min_el / mid_el would make)
    unsigned min_el = 2, max_el = 2;
    const float m2 = vm_best_cross(sm, evals[2], c2);
    const float m1 = vm_best_cross(sm, evals[1], c1);
    float mmin = m2, mmaxv = m2;
    if (m1 <= mmin) { min_el = 1u; mmin = m1; }
    if (m1 > mmaxv) { max_el = 1u; mmaxv = m1; }
    const float m0 = vm_best_cross(sm, evals[0], c0);
    if (m0 <= mmin) { min_el = 0u; mmin = m0; }
    if (m0 > mmaxv) { max_el = 0u; mmaxv = m0; }
    unsigned mid_el = 3u - min_el - max_el;
    // column k := normalised cross product of columns k + 1 and k + 2 (mod 3), k = min_el first, then mid_el -- spelled out per k: any
    // run-time choice among c0 / c1 / c2, pointer OR value (the compiler folds a select of loads back into a load through a selected
    // pointer), keeps all three columns in scratch memory
    float tmp[3];
    if (min_el == 0u) { vm_cross(c1, c2, tmp); vm_normalize3(tmp); c0[0] = tmp[0]; c0[1] = tmp[1]; c0[2] = tmp[2]; }
    else if (min_el == 1u) { vm_cross(c2, c0, tmp); vm_normalize3(tmp); c1[0] = tmp[0]; c1[1] = tmp[1]; c1[2] = tmp[2]; }
    else { vm_cross(c0, c1, tmp); vm_normalize3(tmp); c2[0] = tmp[0]; c2[1] = tmp[1]; c2[2] = tmp[2]; }
    if (mid_el == 0u) { vm_cross(c1, c2, tmp); vm_normalize3(tmp); c0[0] = tmp[0]; c0[1] = tmp[1]; c0[2] = tmp[2]; }
    else if (mid_el == 1u) { vm_cross(c2, c0, tmp); vm_normalize3(tmp); c1[0] = tmp[0]; c1[1] = tmp[1]; c1[2] = tmp[2]; }
    else { vm_cross(c0, c1, tmp); vm_normalize3(tmp); c2[0] = tmp[0]; c2[1] = tmp[1]; c2[2] = tmp[2]; }
  }
  for (int r = 0; r < 3; ++r) { evecs[r * 3 + 0] = c0[r]; evecs[r * 3 + 1] = c1[r]; evecs[r * 3 + 2] = c2[r]; }
  evals[0] *= scale; evals[1] *= scale; evals[2] *= scale;
}

// ------------------------------------------------------------------------------------------
// Eigen features from ascending eigenvalues (VS:1169-1219 order L,P,S,C,A,E,Sum,O;
// svgs_order != 0 gives SS:808-837 order L,P,S,A,C,E,Sum,O with A inside the e1==0 branch).
// ------------------------------------------------------------------------------------------
VGS_HD void vm_eigen_features(const float* ev, int svgs_order, float* F) {
  if (ev[0] == 0.0f && ev[1] == 0.0f && ev[2] == 0.0f) {
    for (int i = 0; i < 8; ++i) F[i] = 0.0f;
    return;
  }
  float s = vm_sqrt(ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2]);
  float e3 = ev[0] / s, e2 = ev[1] / s, e1 = ev[2] / s;
  float lin, pla, sca, ani;
  float sum = e1 + e2 + e3;
  float cur = e3 / sum;
  if (e1 == 0.0f) { lin = 0.0f; pla = 1.0f; sca = 0.0f; }
  else { lin = (e1 - e2) / e1; pla = (e2 - e3) / e1; sca = e3 / e1; }
  if (svgs_order) ani = (e1 == 0.0f) ? 0.0f : (e1 - e3) / e1;
  else ani = (e2 == 0.0f) ? 0.0f : (e1 - e3) / e1;
  float prod = e1 * e2 * e3;
  float ent, omn;
  if (prod == 0.0f) { ent = 0.0f; omn = 0.0f; }
  else {
    ent = -1.0f * (e1 * vm_log(e1) + e2 * vm_log(e2) + e3 * vm_log(e3));
    // pow(prod, 1/3): prod > 0 here (a negative product gives NaN in the reference's powf too)
    omn = (prod > 0.0f) ? vm_exp(vm_log(prod) * 0.33333334f) : vm_nan();
  }
  F[0] = lin; F[1] = pla; F[2] = sca;
  if (svgs_order) { F[3] = ani; F[4] = cur; } else { F[3] = cur; F[4] = ani; }
  F[5] = ent; F[6] = sum; F[7] = omn;
}

// ------------------------------------------------------------------------------------------
// Node record: what one graph node (voxel or supervoxel) contributes to a pair distance.
// flags bit0 = position valid (all three centroid components != 0, VS:1829), bit1 = normal
// valid (VS:1840), bit2 = eigen vector has 8 entries (node "used"; unused nodes hold {0}).
// ------------------------------------------------------------------------------------------
#define VGS_F_POS 1u
#define VGS_F_NRM 2u
#define VGS_F_EIG 4u

struct alignas(16) VgsNode {  // 64 bytes: what the local-graph kernel gathers per neighbour
  float c[3];
  float n[3];
  float f[8];
  uint32_t flags;
  uint32_t pad;
};

struct VgsWeightParams {
  float inv_sig_p, inv_sig_n, inv_sig_o, inv_sig_e, inv_sig_c;  // VGS: 1/sigma ; SVGS: 1/sigma (applied to d^2)
  float inv_sig_w2;                                            // 1/(sig_w*sig_w)
  int svgs;                                                    // 0 = VS formulas, 1 = SS formulas
};

VGS_HD float vm_dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// The five distances S,A,T,E,C (VS:1597-1720; SVGS differences SS:1756-1878).
// SPLIT: the eigen sum as two loops with constant bounds (same additions in the same order) -- for callers that hold the
// records in registers, which a loop starting at a run-time index would force into scratch memory; the kernels that read
// records through pointers keep the single loop (the split costs the one-wavefront local cut 16 bytes of spills per lane).
// convx_swapped (round 5, optional): the convexity distance of the pair TAKEN THE OTHER WAY ROUND, (v2, v1).  Of the five distances it is the
// only one that depends on the order of the arguments: swapping them negates the connecting direction u and the cross product exactly
// (float subtraction and the sign of a product are exact), so the centroid distance, the angle between the normals, the stair term
// (p and q change places under a commutative sum), the eigen term and the angle of the cross product come out bit for bit the same, while
// the two angles between a normal and u become acos(-cos2) and acos(-cos1) -- which differ from pi - acos(cos2), pi - acos(cos1) in the last
// bit.  A builder that stores both orientations of a pair (near-pair lists, pair lists) pays two more acos instead of a second evaluation.
// (BOTH is a template parameter, not a run-time null check: the plain instantiation must compile to exactly what it was -- the
// one-wavefront local cut sits at its register limit and a dead store more costs it three spilled registers)
template <bool SPLIT, bool BOTH = false>
VGS_HD void vm_pair_distances_t(const VgsNode& v1, const VgsNode& v2, int svgs, float* out, float* convx_swapped = nullptr) {
  float dist_space = 100.0f, dist_angle = 100.0f, dist_stair = 100.0f, dist_eigen = 100.0f, dist_convx = 100.0f;
  [[maybe_unused]] float dist_convx_sw = 100.0f;
  float d12 = 0.0f;
  float u[3] = {0.0f, 0.0f, 0.0f};
  float prod[3] = {0.0f, 0.0f, 0.0f};
  const bool pv = (v1.flags & VGS_F_POS) && (v2.flags & VGS_F_POS);
  const bool nv = (v1.flags & VGS_F_NRM) && (v2.flags & VGS_F_NRM);
  const bool ev = (v1.flags & VGS_F_EIG) && (v2.flags & VGS_F_EIG);
  if (pv) {
    float dx = v1.c[0] - v2.c[0], dy = v1.c[1] - v2.c[1], dz = v1.c[2] - v2.c[2];
    d12 = vm_sqrt((dx * dx + dy * dy) + dz * dz);
    dist_space = d12;
    if (d12 != 0.0f) {
      u[0] = dx / d12; u[1] = dy / d12; u[2] = dz / d12;
      vm_cross(v1.c, v2.c, prod);
    }
  }
  if (nv) {
    float a_1 = 0.0f, a_2 = 0.0f, a_1_2 = 0.0f, a_d_s1 = 0.0f, a_d_s2 = 0.0f;
    [[maybe_unused]] float a_1s = 0.0f, a_2s = 0.0f;
    // VS guards on dist_space (100 when positions are invalid -> the reference then reads empty
    // vectors, UB; here u = prod = 0).  SS guards on dist_v1_v2 and sets dist_stair = 0 otherwise.
    const bool guard = svgs ? (d12 != 0.0f) : (dist_space != 0.0f);
    if (guard) {
      float cos12 = vm_dot3(v1.n, v2.n);
      float cos1 = vm_dot3(v1.n, u);
      float cos2 = vm_dot3(v2.n, u);
      float cosds = vm_dot3(prod, u);
      a_1 = vm_acos(cos1);
      a_2 = vm_acos(cos2);
      if constexpr (BOTH) { a_1s = vm_acos(-cos2); a_2s = vm_acos(-cos1); }   // dot(v2.n, -u), dot(v1.n, -u)
      a_1_2 = vm_acos(cos12);
      a_d_s1 = vm_acos(cosds);
      a_d_s2 = VM_PI_F - a_d_s1;
      dist_angle = a_1_2;
      float dv1 = vm_dot3(v1.n, v1.c), dv2 = vm_dot3(v2.n, v2.c);
      float do1 = vm_dot3(v1.n, v2.c), do2 = vm_dot3(v2.n, v1.c);
      if (d12 != 0.0f) {
        float p = do1 - dv1, q = do2 - dv2;
        dist_stair = vm_sqrt(p * p + q * q);
      } else {
        dist_stair = 0.0f;
      }
    } else if (svgs) {
      dist_stair = 0.0f;
    }
    float thr = (VM_PI_F * 0.5f) / (1.0f + vm_exp(-0.5f * (a_1_2 - VM_PI_F / 6.0f)));
    float a_d_s = a_d_s1;
    if (a_d_s1 > a_d_s2) a_d_s = a_d_s2;
    if (a_d_s > thr) dist_convx = vm_abs(a_1 - a_2);
    else dist_convx = VM_PI_F;
    if constexpr (BOTH) {
      if (a_d_s > thr) dist_convx_sw = vm_abs(a_1s - a_2s);
      else dist_convx_sw = VM_PI_F;
    }
  }
  if constexpr (BOTH) *convx_swapped = dist_convx_sw;
  if (ev) {
    float ec = 0.0f, e1 = 0.0f, e2 = 0.0f;
    // entries 4..7 (VS) or 0..7 (SS), summed in index order
    if (SPLIT) {
      if (svgs) {
        for (int i = 0; i < 4; ++i) {
          ec = ec + v1.f[i] * v2.f[i];
          e1 = e1 + v1.f[i] * v1.f[i];
          e2 = e2 + v2.f[i] * v2.f[i];
        }
      }
      for (int i = 4; i < 8; ++i) {
        ec = ec + v1.f[i] * v2.f[i];
        e1 = e1 + v1.f[i] * v1.f[i];
        e2 = e2 + v2.f[i] * v2.f[i];
      }
    } else {
      for (int i = svgs ? 0 : 4; i < 8; ++i) {
        ec = ec + v1.f[i] * v2.f[i];
        e1 = e1 + v1.f[i] * v1.f[i];
        e2 = e2 + v2.f[i] * v2.f[i];
      }
    }
    if (e1 != 0.0f && e2 != 0.0f) dist_eigen = 1.0f - ec / (vm_sqrt(e1) * vm_sqrt(e2));
  }
  out[0] = dist_space; out[1] = dist_angle; out[2] = dist_stair; out[3] = dist_eigen; out[4] = dist_convx;
}
VGS_HD void vm_pair_distances(const VgsNode& v1, const VgsNode& v2, int svgs, float* out) { vm_pair_distances_t<false>(v1, v2, svgs, out); }

// VS:1736-1737 / SS:1900-1902.  Sigmas enter as reciprocals (computed once in float on the host).
VGS_HD float vm_distance_weight(const float* d, const VgsWeightParams& P) {
  float D;
  if (!P.svgs) {
    float s = d[0] * P.inv_sig_p, a = d[1] * P.inv_sig_n, t = d[2] * P.inv_sig_o, c = d[4] * P.inv_sig_c, e = d[3] * P.inv_sig_e;
    D = vm_sqrt((((s * s + a * a) + t * t) + c * c) + e * e);
  } else {
    // sqrt(ds^2/sig_p + da^2/sig_n + de^2/sig_e + dt^2/sig_o): sigma not squared, convexity unused
    D = vm_sqrt(((d[0] * d[0] * P.inv_sig_p + d[1] * d[1] * P.inv_sig_n) + d[3] * d[3] * P.inv_sig_e) + d[2] * d[2] * P.inv_sig_o);
  }
  return vm_exp((-0.5f * D) * P.inv_sig_w2);
}

VGS_HD float vm_pair_weight(const VgsNode& v1, const VgsNode& v2, const VgsWeightParams& P) {
  float d[5];
  vm_pair_distances(v1, v2, P.svgs, d);
  return vm_distance_weight(d, P);
}
// vm_pair_weight(v1, v2) and vm_pair_weight(v2, v1), bit for bit, for a little more than the price of one (see vm_pair_distances_t)
VGS_HD void vm_pair_weight_both(const VgsNode& v1, const VgsNode& v2, const VgsWeightParams& P, float* w12, float* w21) {
  float d[5], sw;
  vm_pair_distances_t<false, true>(v1, v2, P.svgs, d, &sw);
  *w12 = vm_distance_weight(d, P);
  d[4] = sw;
  *w21 = vm_distance_weight(d, P);
}
// the same value for records held in registers (see vm_pair_distances_t)
VGS_HD float vm_pair_weight_regs(const VgsNode& v1, const VgsNode& v2, const VgsWeightParams& P) {
  float d[5];
  vm_pair_distances_t<true>(v1, v2, P.svgs, d);
  return vm_distance_weight(d, P);
}

// Upper bounds of vm_pair_weight used by the lazy local cut (never part of a result, only of the schedule).
// Every float step of vm_distance_weight is monotone in each squared term, so dropping non-negative terms from
// the sum under the square root can only raise the weight; the 1e-6 factor covers vm_exp's <= 1 ulp error.
//   vm_weight_bound_d(d2):    every pair whose squared centroid distance is >= d2 weighs <= this
//   vm_weight_bound_da(a, b): this pair weighs <= this (proximity and normal-angle terms only; a NaN bound
//                             means "unknown": the caller then evaluates the full weight)
VGS_HD float vm_weight_bound_d(float d2, const VgsWeightParams& P) {
  const float d = vm_sqrt(d2);
  float D;
  if (!P.svgs) { const float s = d * P.inv_sig_p; D = vm_sqrt(s * s); }
  else D = vm_sqrt(d * d * P.inv_sig_p);
  return vm_exp((-0.5f * D) * P.inv_sig_w2) * 1.000001f;
}

// the weight of a pair whose other three distances are zero
VGS_HD float vm_weight_bound_sa(float dist_space, float dist_angle, const VgsWeightParams& P) {
  float D;
  if (!P.svgs) { const float s = dist_space * P.inv_sig_p, a = dist_angle * P.inv_sig_n; D = vm_sqrt(s * s + a * a); }
  else D = vm_sqrt(dist_space * dist_space * P.inv_sig_p + dist_angle * dist_angle * P.inv_sig_n);
  return vm_exp((-0.5f * D) * P.inv_sig_w2) * 1.000001f;
}

VGS_HD float vm_weight_bound_da(const VgsNode& v1, const VgsNode& v2, const VgsWeightParams& P) {
  float dist_space = 100.0f, dist_angle = 100.0f, d12 = 0.0f;
  if ((v1.flags & VGS_F_POS) && (v2.flags & VGS_F_POS)) {
    const float dx = v1.c[0] - v2.c[0], dy = v1.c[1] - v2.c[1], dz = v1.c[2] - v2.c[2];
    d12 = vm_sqrt((dx * dx + dy * dy) + dz * dz);
    dist_space = d12;
  }
  if ((v1.flags & VGS_F_NRM) && (v2.flags & VGS_F_NRM)) {
    const bool guard = P.svgs ? (d12 != 0.0f) : (dist_space != 0.0f);
    if (guard) dist_angle = vm_acos(vm_dot3(v1.n, v2.n));
  }
  return vm_weight_bound_sa(dist_space, dist_angle, P);
}

// Threshold of a segment in the local cut (VS:1968-1969): seg_int - cut/size, float.
VGS_HD float vm_cut_threshold(float seg_int, float cut, int size) { return seg_int - cut / (float)size; }

// ------------------------------------------------------------------------------------------
// Octree key / leaf order / centre (PCL OctreePointCloud; VS:2102-2109)
// ------------------------------------------------------------------------------------------
VGS_HD uint32_t vm_axis_key(float p, double min, double resolution) {
  return (uint32_t)(((double)p - min) / resolution);
}

// interleave 21-bit x,y,z: bit triple = (x<<2)|(y<<1)|z, x most significant
VGS_HD uint64_t vm_spread21(uint32_t v) {
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
VGS_HD uint64_t vm_morton(uint32_t kx, uint32_t ky, uint32_t kz) {
  return (vm_spread21(kx) << 2) | (vm_spread21(ky) << 1) | vm_spread21(kz);
}
VGS_HD uint32_t vm_compact21(uint64_t x) {
  x &= 0x1249249249249249ull;
  x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
  x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
  x = (x ^ (x >> 8)) & 0x1f0000ff0000ffull;
  x = (x ^ (x >> 16)) & 0x1f00000000ffffull;
  x = (x ^ (x >> 32)) & 0x1fffffull;
  return (uint32_t)x;
}
// VS:2106-2108 with the class's *float* resolution/min members (VS:1121-1123)
VGS_HD float vm_voxel_center(uint32_t key, float res_f, float min_f) {
  return (float)(((double)key + 0.5f) * res_f + min_f);
}

#endif  // VGS_MATH_H_
