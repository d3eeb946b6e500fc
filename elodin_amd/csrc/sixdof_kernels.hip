// sixdof_kernels.hip — built-in effector pipes of the fused per-entity six_dof step (kernel: step_kernel.hpp).
//
// Instantiates sixdof_step_kernel for the compile-time op lists the BASELINE workloads use and for the
// run-time interpreter, and dispatches a launch to the matching instantiation.
#include "step_kernel.hpp"

namespace sixdof {

// ---- dispatch ------------------------------------------------------------------------------------------

namespace {

using PipeNone = PipeStatic<>;
using PipeGravity = PipeStatic<SIXDOF_EFF_UNIFORM_GRAVITY>;
using PipeGravityTorque = PipeStatic<SIXDOF_EFF_UNIFORM_GRAVITY, SIXDOF_EFF_BODY_TORQUE>;
using PipeGravityDrag = PipeStatic<SIXDOF_EFF_UNIFORM_GRAVITY, SIXDOF_EFF_BALL_DRAG>;
using PipeGravityThrustTorque = PipeStatic<SIXDOF_EFF_UNIFORM_GRAVITY, SIXDOF_EFF_BODY_FORCE, SIXDOF_EFF_BODY_TORQUE>;

bool kinds_are(const StepParams& p, std::initializer_list<int> kinds) {
    if (p.n_ops != kinds.size()) return false;
    uint32_t k = 0;
    for (int kind : kinds)
        if (p.ops[k++].kind != kind) return false;
    return true;
}

}  // namespace

hipError_t launch_step(const StepParams& p, int integrator, int dtype, hipStream_t stream) {
    if (p.n == 0) return hipSuccess;
    const dim3 grid((p.n + kWave - 1) / kWave);
    // op lists of the BASELINE workloads get a compile-time pipe; anything else runs the interpreter
    if (kinds_are(p, {})) launch_p<PipeNone>(p, integrator, dtype, grid, stream);
    else if (kinds_are(p, {SIXDOF_EFF_UNIFORM_GRAVITY})) launch_p<PipeGravity>(p, integrator, dtype, grid, stream);
    else if (kinds_are(p, {SIXDOF_EFF_UNIFORM_GRAVITY, SIXDOF_EFF_BODY_TORQUE}))
        launch_p<PipeGravityTorque, true>(p, integrator, dtype, grid, stream);
    else if (kinds_are(p, {SIXDOF_EFF_UNIFORM_GRAVITY, SIXDOF_EFF_BALL_DRAG}))
        launch_p<PipeGravityDrag>(p, integrator, dtype, grid, stream);
    else if (kinds_are(p, {SIXDOF_EFF_UNIFORM_GRAVITY, SIXDOF_EFF_BODY_FORCE, SIXDOF_EFF_BODY_TORQUE}))
        launch_p<PipeGravityThrustTorque>(p, integrator, dtype, grid, stream);
    else launch_p<PipeGeneric>(p, integrator, dtype, grid, stream);
    return hipGetLastError();
}

}  // namespace sixdof
