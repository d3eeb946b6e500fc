// effectors.hpp — per-entity effector pipes (the `sys` argument of six_dof) and calc_accel, device side.
//
// Reference: clear_forces | effectors | calc_accel, libs/nox-py/src/six_dof.rs:137-150,184-203.
// A pipe is either a compile-time list of op kinds (PipeStatic<...>, specialised kernels for the
// op lists the BASELINE workloads use) or the run-time interpreter (PipeGeneric, any list of up to
// kMaxOps ops).  Both produce the same wrench; the static form lets the compiler drop the
// wave-uniform branching and the dead halves of the wrench.
//
// The wrench keeps world-frame torque and BODY-frame torque apart: calc_accel needs the torque in
// the body frame (alpha = q * ((q^-1 * tau) / I)), so a body-frame torque effector
// (tau_world = q * tau_b, examples/apollo-lander/sim.py:396-398) contributes tau_b directly instead
// of being rotated to the world frame and straight back (q^-1 * (q * tau_b) = tau_b).  The world
// value is formed only when the `force` column is written.
#pragma once
#include <utility>

#include "kernels.hpp"
#include "spatial.hpp"
#include "../../include/sixdof_hip.h"

namespace sixdof {

template <class T>
struct Body {
    Quat<T> q;     // stage attitude (unit)
    Vec3<T> p;     // stage position
    Spatial<T> v;  // stage velocity
    T mass;
    Vec3<T> I;     // body-frame inertia diagonal (only read by generated pipes)
};

template <class T>
struct Wrench {
    Vec3<T> tau_w;  // world-frame torque
    Vec3<T> tau_b;  // body-frame torque
    Vec3<T> f;      // world-frame force
};

template <class T>
__device__ __forceinline__ Wrench<T> zero_wrench() {
    const Vec3<T> z = {T(0), T(0), T(0)};
    return {z, z, z};
}

template <int KIND, class T>
__device__ __forceinline__ void apply_one(const DevOp& op, const Vec3<T>& aux, const Body<T>& b, Wrench<T>& F) {
    if constexpr (KIND == SIXDOF_EFF_CONST_WRENCH) {
        F.tau_w = F.tau_w + Vec3<T>{T(op.p[0]), T(op.p[1]), T(op.p[2])};
        F.f = F.f + Vec3<T>{T(op.p[3]), T(op.p[4]), T(op.p[5])};
    } else if constexpr (KIND == SIXDOF_EFF_UNIFORM_GRAVITY) {
        F.f = axpy(b.mass, Vec3<T>{T(op.p[0]), T(op.p[1]), T(op.p[2])}, F.f);
    } else if constexpr (KIND == SIXDOF_EFF_BODY_TORQUE) {
        F.tau_b = F.tau_b + aux;
    } else if constexpr (KIND == SIXDOF_EFF_BODY_FORCE) {
        F.f = F.f + rotate(b.q, aux);
    } else if constexpr (KIND == SIXDOF_EFF_WORLD_TORQUE) {
        F.tau_w = F.tau_w + aux;
    } else if constexpr (KIND == SIXDOF_EFF_WORLD_FORCE) {
        F.f = F.f + aux;
    } else if constexpr (KIND == SIXDOF_EFF_BALL_DRAG) {
        // examples/ball/sim.py:96-116; el.SpatialForce(linear=...) drops whatever torque was there
        const Vec3<T> fl = aux - b.v.lin;
        const T v2 = dot(fl, fl);
        const T V = fast_sqrt(v2);
        const T drag = T(0.5) * ((T(op.p[0]) * T(op.p[1])) * v2 * T(op.p[2]));
        F.tau_w = Vec3<T>{T(0), T(0), T(0)};
        F.tau_b = Vec3<T>{T(0), T(0), T(0)};
        F.f = axpy(drag / V, fl, F.f);
    }
}

// Hooks for user systems piped AROUND six_dof (pre | six_dof(effectors) | post) that a generated program fuses
// into the same kernel.  Pipes without such systems inherit these no-ops.
struct NoModel {
    static constexpr bool kHasModel = false;
    static constexpr bool kWritesInertia = false;
    static constexpr bool kPreReadsAccel = false;   // a system in front of six_dof reads world_accel (the previous tick's)
    // a generated program WITHOUT six_dof whose systems touch no Body column (a whole-world StableHLO tick: every component of the
    // world, world_pos included, is a program column; the executor's Body columns are stand-ins): the kernel then neither loads nor
    // stores the Body slabs — 456 B per entity and tick at one tick per launch
    static constexpr bool kBodyDead = false;
    template <class T>
    struct Regs {};
    template <class T, int POL>
    __device__ static __forceinline__ void load(const StepParams&, uint32_t, bool, Regs<T>&) {}
    template <class T, int POL>
    __device__ static __forceinline__ void store(const StepParams&, uint32_t, const Regs<T>&) {}
    template <class T>
    __device__ static __forceinline__ void record(const StepParams&, size_t, uint32_t, const Regs<T>&) {}
    template <class T>
    __device__ static __forceinline__ void pre(const StepParams&, uint64_t, Regs<T>&, Quat<T>&, Vec3<T>&, Spatial<T>&,
                                               Vec3<T>&, T&, const Spatial<T>&) {}
    template <class T>
    __device__ static __forceinline__ void post(const StepParams&, uint64_t, Regs<T>&, Quat<T>&, Vec3<T>&, Spatial<T>&,
                                                Vec3<T>&, T&, const Spatial<T>&) {}
};

template <int KIND>
struct KindTraits {
    static constexpr bool reads_velocity = (KIND == SIXDOF_EFF_BALL_DRAG);
    static constexpr bool world_torque = (KIND == SIXDOF_EFF_CONST_WRENCH || KIND == SIXDOF_EFF_WORLD_TORQUE);
    static constexpr bool body_torque = (KIND == SIXDOF_EFF_BODY_TORQUE);
    static constexpr bool uses_aux =
        (KIND == SIXDOF_EFF_BODY_TORQUE || KIND == SIXDOF_EFF_BODY_FORCE || KIND == SIXDOF_EFF_BALL_DRAG ||
         KIND == SIXDOF_EFF_WORLD_TORQUE || KIND == SIXDOF_EFF_WORLD_FORCE);
};

// Compile-time op list.  Op k takes its constants from P.ops[k] and its column value from aux[k].
template <int... KINDS>
struct PipeStatic : NoModel {
    static constexpr int kOps = sizeof...(KINDS);
    static constexpr bool kStatic = true;
    static constexpr bool kWorldTorque = (false || ... || KindTraits<KINDS>::world_torque);
    static constexpr bool kBodyTorque = (false || ... || KindTraits<KINDS>::body_torque);
    static constexpr bool kReadsVelocity = (false || ... || KindTraits<KINDS>::reads_velocity);
    template <int K>
    static constexpr bool uses_aux() {
        constexpr bool t[sizeof...(KINDS) + 1] = {KindTraits<KINDS>::uses_aux..., false};
        return K < kOps && t[K];
    }
    __device__ static __forceinline__ bool vel_independent(const StepParams&) { return !kReadsVelocity; }
    template <class T, size_t... I>
    __device__ static __forceinline__ void apply_impl(const StepParams& P, const Vec3<T> (&aux)[kMaxOps],
                                                      const Body<T>& b, Wrench<T>& F, std::index_sequence<I...>) {
        (apply_one<KINDS>(P.ops[I], aux[I], b, F), ...);
    }
    template <class T, class R>
    __device__ static __forceinline__ void apply(const StepParams& P, const Vec3<T> (&aux)[kMaxOps], const R&,
                                                 const Body<T>& b, Wrench<T>& F) {
        apply_impl(P, aux, b, F, std::make_index_sequence<sizeof...(KINDS)>{});
    }
};

// Run-time interpreter: wave-uniform branches on kernel arguments.
struct PipeGeneric : NoModel {
    static constexpr int kOps = kMaxOps;
    static constexpr bool kStatic = false;
    static constexpr bool kWorldTorque = true;
    static constexpr bool kBodyTorque = true;
    template <int K>
    static constexpr bool uses_aux() { return true; }
    __device__ static __forceinline__ bool vel_independent(const StepParams& P) { return P.vel_independent != 0; }
    template <class T, class R>
    __device__ static __forceinline__ void apply(const StepParams& P, const Vec3<T> (&aux)[kMaxOps], const R&,
                                                 const Body<T>& b, Wrench<T>& F) {
#pragma unroll
        for (int k = 0; k < kMaxOps; k++) {
            if (k >= (int)P.n_ops) break;
            switch (P.ops[k].kind) {
            case SIXDOF_EFF_CONST_WRENCH: apply_one<SIXDOF_EFF_CONST_WRENCH>(P.ops[k], aux[k], b, F); break;
            case SIXDOF_EFF_UNIFORM_GRAVITY: apply_one<SIXDOF_EFF_UNIFORM_GRAVITY>(P.ops[k], aux[k], b, F); break;
            case SIXDOF_EFF_BODY_TORQUE: apply_one<SIXDOF_EFF_BODY_TORQUE>(P.ops[k], aux[k], b, F); break;
            case SIXDOF_EFF_BODY_FORCE: apply_one<SIXDOF_EFF_BODY_FORCE>(P.ops[k], aux[k], b, F); break;
            case SIXDOF_EFF_BALL_DRAG: apply_one<SIXDOF_EFF_BALL_DRAG>(P.ops[k], aux[k], b, F); break;
            case SIXDOF_EFF_WORLD_TORQUE: apply_one<SIXDOF_EFF_WORLD_TORQUE>(P.ops[k], aux[k], b, F); break;
            case SIXDOF_EFF_WORLD_FORCE: apply_one<SIXDOF_EFF_WORLD_FORCE>(P.ops[k], aux[k], b, F); break;
            default: break;
            }
        }
    }
};

// calc_accel (six_dof.rs:137-146): alpha = q * ((q^-1 * tau) / I_diag); a = q * ((q^-1 * f) / m) = f / m.
// The reference carries BOTH halves through the attitude, so a non-finite quaternion poisons the whole acceleration even
// where the rotation cancels algebraically.  `taint` (accel_taint below: +-0 for a finite attitude, NaN otherwise) keeps
// that propagation without the two rotations: the caller folds it into the reciprocal mass (inv_m_t = inv_m + taint,
// exact for +-0), so the product below is all it costs.
template <class PIPE, class T>
__device__ __forceinline__ Spatial<T> calc_accel(const Quat<T>& q, const Wrench<T>& F, const Vec3<T>& inv_I, T inv_m_t, T taint) {
    Vec3<T> bt = F.tau_b;
    if constexpr (PIPE::kWorldTorque) bt = bt + rotate_inv(q, F.tau_w);
    Spatial<T> a;
    if constexpr (PIPE::kWorldTorque || PIPE::kBodyTorque) a.ang = rotate(q, hadamard(bt, inv_I));
    else a.ang = Vec3<T>{taint, taint, taint};
    a.lin = inv_m_t * F.f;
    return a;
}
// norm2 = |q|^2 of the attitude calc_accel is about to see, before normalisation (integrate_world / normalized hand it out):
// NaN or inf exactly when the attitude is not finite.
//   RK4: every stage attitude is q0 (+) c*dt*v0.ang (the stage positions advance with the INITIAL velocity, rk4.rs:110-121),
//   so the four are finite together; the LAST stage's taint is enough: it poisons A_3 (the world_accel output and, through
//   sum(A_s), the new velocity), and the caller adds it to the (dt/6) factor of the position update, which the reference
//   poisons through v_s = v0 + c*dt*A_(s-1).  One multiply and two adds per tick instead of work in every stage.
//   Semi-implicit: the one calc_accel sees q0.
template <class T>
__device__ __forceinline__ T accel_taint(T norm2) {
#ifdef SIXDOF_AB_NO_TAINT
    return T(0);
#else
    return T(0) * norm2;
#endif
}

// The `force` column holds the world-frame wrench [tau, f] of the last stage.
template <class PIPE, class T>
__device__ __forceinline__ Spatial<T> world_wrench(const Quat<T>& q, const Wrench<T>& F) {
    Spatial<T> o;
    o.ang = F.tau_w;
    if constexpr (PIPE::kBodyTorque) o.ang = o.ang + rotate(q, F.tau_b);
    o.lin = F.f;
    return o;
}

}  // namespace sixdof
