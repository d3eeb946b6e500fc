// nbody_kernels.hip — pairwise (edge_fold) path. Placeholder until the tiled kernels land.
#include "kernels.hpp"
namespace sixdof {
hipError_t launch_pair_tick(const PairParams&, int, hipStream_t, uint64_t*) { return hipErrorNotSupported; }
}  // namespace sixdof
