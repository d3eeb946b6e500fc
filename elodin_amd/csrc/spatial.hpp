// spatial.hpp — device-side spatial algebra for the MI355X six_dof kernels.
//
// Restates (does not copy) the arithmetic of the reference's
//   libs/nox/src/quaternion.rs:141-155,268-305   (Hamilton product, inverse, normalize, q*v)
//   libs/nox/src/spatial.rs:353-361,530-593      (F / I, transform + motion, q * spatial)
// in forms chosen for CDNA4 rather than for a tracing compiler:
//   * a rotation is two cross products (18 FMA-class ops) instead of two Hamilton products plus a
//     recomputed inverse (~70 ops, 4 divides) — q v q^-1 is scale-invariant, so for the unit stage
//     quaternions the results agree to ~1 ulp;
//   * normalisation multiplies by one rsqrt instead of dividing four times by a sqrt;
//   * divides by mass / inertia become multiplies by reciprocals computed once per launch.
// These differ from the reference's rounding by O(1e-16) per operation; parity (<=1e-9 relative on
// f64 state over the tested horizons) is checked against the operation-order-exact CPU oracle.
// Quaternions are scalar-last [i,j,k,w], like the reference's columns.
#pragma once
#include <hip/hip_runtime.h>

namespace sixdof {

template <class T>
struct Vec3 {
    T x, y, z;
};
template <class T>
struct Quat {
    T i, j, k, w;
};
// SpatialMotion / SpatialForce: angular (or torque) part first, like the reference columns.
template <class T>
struct Spatial {
    Vec3<T> ang, lin;
};

template <class T>
__device__ __forceinline__ Vec3<T> operator+(Vec3<T> a, Vec3<T> b) {
    return {a.x + b.x, a.y + b.y, a.z + b.z};
}
template <class T>
__device__ __forceinline__ Vec3<T> operator-(Vec3<T> a, Vec3<T> b) {
    return {a.x - b.x, a.y - b.y, a.z - b.z};
}
template <class T>
__device__ __forceinline__ Vec3<T> operator*(T s, Vec3<T> a) {
    return {s * a.x, s * a.y, s * a.z};
}
template <class T>
__device__ __forceinline__ Vec3<T> hadamard(Vec3<T> a, Vec3<T> b) {
    return {a.x * b.x, a.y * b.y, a.z * b.z};
}
template <class T>
__device__ __forceinline__ Vec3<T> cross(Vec3<T> a, Vec3<T> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class T>
__device__ __forceinline__ T dot(Vec3<T> a, Vec3<T> b) {
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
// fused a + s*b
template <class T>
__device__ __forceinline__ Vec3<T> axpy(T s, Vec3<T> b, Vec3<T> a) {
    return {a.x + s * b.x, a.y + s * b.y, a.z + s * b.z};
}
template <class T>
__device__ __forceinline__ Spatial<T> axpy(T s, Spatial<T> b, Spatial<T> a) {
    return {axpy(s, b.ang, a.ang), axpy(s, b.lin, a.lin)};
}
template <class T>
__device__ __forceinline__ Spatial<T> operator+(Spatial<T> a, Spatial<T> b) {
    return {a.ang + b.ang, a.lin + b.lin};
}

// 1/x for the once-per-launch reciprocal mass / inertia: v_rcp_f64 + two Newton steps (each squares the error, so
// the result is correctly rounded or 1 ulp off) = 5 instructions against the 11 of an IEEE f64 divide.
// The two values a Newton step cannot handle (0 * inf) follow what the reference's calc_accel PRODUCES for them
// (six_dof.rs:137-146, a = q * ((q^-1 * f) / m)): m = +-inf (a static anchor) divides to 0 and stays 0 through the
// rotation -> the reciprocal is 0; m = 0 divides to +-inf (or 0/0), and rotating a vector with infinite components forms
// inf * u - inf * u' -> the reference's acceleration is NaN in every component, whatever the attitude -> the reciprocal is
// NaN (a bare 1/0 = inf would leave a +-inf acceleration where the reference has NaN).  Same rule in both dtypes.
__device__ __forceinline__ double recip(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r0, 1.0);
    double r = fma(e, r0, r0);
    r = fma(fma(-x, r, 1.0), r, r);
    return e == e ? r : (r0 == 0.0 ? r0 : __builtin_nan(""));
}
#ifdef SIXDOF_FAST_MATH   // generated f32 programs that opt in (codegen.py): the hardware's 1-ulp v_rcp / v_rsq / v_sqrt alone
__device__ __forceinline__ float recip(float x) { return x == 0.0f ? __builtin_nanf("") : __builtin_amdgcn_rcpf(x); }
#else
__device__ __forceinline__ float recip(float x) { return x == 0.0f ? __builtin_nanf("") : 1.0f / x; }
#endif
// 1/sqrt(x) for finite x > 0: hardware v_rsq_f64 seed plus one cubic correction (y0 (1 + e/2 + 3e^2/8), e = 1 - x y0^2):
// full f64 accuracy in 5 instructions, without the 0 / inf / denormal special-casing of the library rsqrt (4 more
// instructions and a v_cmp_class per call).  Arguments here are squared norms of quaternions (~1) and softened pair
// distances; x = 0 gives inf * 0 = NaN downstream, like the library form.
__device__ __forceinline__ double rsqrt_pos(double x) {
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y0, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}
__device__ __forceinline__ double fast_rsqrt(double x) { return rsqrt_pos(x); }
__device__ __forceinline__ double fast_sqrt(double x) { return sqrt(x); }
#ifdef SIXDOF_FAST_MATH
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }   // 1 instruction (rsqrtf: 5, sqrtf: 15)
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
__device__ __forceinline__ float fast_rsqrt(float x) { return rsqrtf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return sqrtf(x); }
#endif

// q * v for a UNIT quaternion:  v + 2w(u x v) + 2 u x (u x v)        (reference: quaternion.rs:283-305)
template <class T>
__device__ __forceinline__ Vec3<T> rotate(Quat<T> q, Vec3<T> v) {
    const Vec3<T> u = {q.i, q.j, q.k};
    const Vec3<T> t = T(2) * cross(u, v);
    return axpy(q.w, t, v) + cross(u, t);
}
// q^-1 * v for a UNIT quaternion (conjugate rotation)
template <class T>
__device__ __forceinline__ Vec3<T> rotate_inv(Quat<T> q, Vec3<T> v) {
    const Vec3<T> u = {q.i, q.j, q.k};
    const Vec3<T> t = T(2) * cross(u, v);
    return axpy(-q.w, t, v) + cross(u, t);
}
// scale-invariant forms for quaternions that may not be unit (first tick of semi-implicit on user data)
template <class T>
__device__ __forceinline__ Vec3<T> rotate_any(Quat<T> q, Vec3<T> v, T two_over_n2) {
    const Vec3<T> u = {q.i, q.j, q.k};
    const Vec3<T> t = two_over_n2 * cross(u, v);
    return axpy(q.w, t, v) + cross(u, t);
}
template <class T>
__device__ __forceinline__ Vec3<T> rotate_inv_any(Quat<T> q, Vec3<T> v, T two_over_n2) {
    const Vec3<T> u = {q.i, q.j, q.k};
    const Vec3<T> t = two_over_n2 * cross(u, v);
    return axpy(-q.w, t, v) + cross(u, t);
}

// SpatialTransform + SpatialMotion, angular part: normalize(q + (d/2, 0) (x) q)   (spatial.rs:530-549)
// `d` is the already-scaled angular increment (h * omega).
template <class T>
__device__ __forceinline__ Quat<T> integrate_world(Quat<T> q, Vec3<T> d, T* norm2 = nullptr) {
    const T hx = T(0.5) * d.x, hy = T(0.5) * d.y, hz = T(0.5) * d.z;
    Quat<T> r;
    r.i = q.i + (hx * q.w + hy * q.k - hz * q.j);
    r.j = q.j + (hy * q.w + hz * q.i - hx * q.k);
    r.k = q.k + (hz * q.w + hx * q.j - hy * q.i);
    r.w = q.w - (hx * q.i + hy * q.j + hz * q.k);
    const T n2 = r.i * r.i + r.j * r.j + r.k * r.k + r.w * r.w;
    if (norm2) *norm2 = n2;
    const T inv = fast_rsqrt(n2);
    return {r.i * inv, r.j * inv, r.k * inv, r.w * inv};
}

// `norm2` (optional) receives |q|^2 before normalisation: finite exactly when the quaternion is (0 * norm2 is the
// NaN-taint calc_accel uses, effectors.hpp).
template <class T>
__device__ __forceinline__ Quat<T> normalized(Quat<T> q, T* norm2 = nullptr) {
    const T n2 = q.i * q.i + q.j * q.j + q.k * q.k + q.w * q.w;
    if (norm2) *norm2 = n2;
    const T inv = fast_rsqrt(n2);
    return {q.i * inv, q.j * inv, q.k * inv, q.w * inv};
}

}  // namespace sixdof
