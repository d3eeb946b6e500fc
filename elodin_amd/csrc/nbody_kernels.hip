// nbody_kernels.hip — built-in pairwise gravity folds (kernels: pair_kernel.hpp).
#include "pair_kernel.hpp"

namespace sixdof {

uint32_t pair_splits_for(uint32_t n) { return pair_splits_for_n(n); }

hipError_t launch_pair_small(const PairParams& p, int integrator, uint32_t n_ticks, hipStream_t stream,
                             uint64_t* launches) {
    // the complete-graph softened fold runs the tiled accumulate inside the small kernel too (PAIR unused there)
    if (p.pair_kind == SIXDOF_EFF_EDGE_GRAVITY_NEWTON) return launch_pair_small_t<PairNewton>(p, integrator, n_ticks, stream, launches);
    return launch_pair_small_t<PairSoftened>(p, integrator, n_ticks, stream, launches);
}

hipError_t launch_pair_ticks(const PairParams& p, int integrator, uint32_t n_ticks, hipStream_t stream, uint64_t* launches) {
    const bool packed = p.packed != 0;
    if (p.pair_kind == SIXDOF_EFF_ALLPAIRS_GRAVITY_SOFTENED) return launch_pair_ticks_t<PairSoftened, true>(p, integrator, n_ticks, packed, stream, launches);
    if (p.pair_kind == SIXDOF_EFF_EDGE_GRAVITY_NEWTON) return launch_pair_ticks_t<PairNewton, false>(p, integrator, n_ticks, packed, stream, launches);
    return launch_pair_ticks_t<PairSoftened, false>(p, integrator, n_ticks, packed, stream, launches);
}

}  // namespace sixdof
