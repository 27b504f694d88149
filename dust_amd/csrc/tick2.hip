// tick2.hip - second translation unit of libdust_amd.so: the owner-computes persistent tick kernel (tick2.hpp) and its launcher.
// Kept apart from dust_amd.hip so that the two compile in parallel (build() in __graft_entry__.py) and a change to this kernel does
// not rebuild the rest of the library.
#include "tick2.hpp"

namespace dust {

template <int MODEL, int MODE>
static int occ_of(size_t lds, int *occ) {
  hipError_t e = hipSuccess;
  if (lds > 64 * 1024)
    e = hipFuncSetAttribute((const void *)svmpc_tick2_kernel<MODEL, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, (const void *)svmpc_tick2_kernel<MODEL, MODE>, T2_NT, lds);
  return (int)e;
}

int tick2_occupancy(int model, int mode, size_t lds, int *occ) {
  *occ = 0;
  if (model == DUST_MODEL_PENDULUM) return mode == PAIR_IMQ ? occ_of<DUST_MODEL_PENDULUM, PAIR_IMQ>(lds, occ) : occ_of<DUST_MODEL_PENDULUM, PAIR_K1>(lds, occ);
  return mode == PAIR_IMQ ? occ_of<DUST_MODEL_PARTICLE, PAIR_IMQ>(lds, occ) : occ_of<DUST_MODEL_PARTICLE, PAIR_K1>(lds, occ);
}

int tick2_launch(const Tick2Args &f, int model, int mode, int grid, size_t lds, hipStream_t stream) {
  if (model == DUST_MODEL_PENDULUM) {
    if (mode == PAIR_IMQ) svmpc_tick2_kernel<DUST_MODEL_PENDULUM, PAIR_IMQ><<<grid, T2_NT, lds, stream>>>(f);
    else svmpc_tick2_kernel<DUST_MODEL_PENDULUM, PAIR_K1><<<grid, T2_NT, lds, stream>>>(f);
  } else {
    if (mode == PAIR_IMQ) svmpc_tick2_kernel<DUST_MODEL_PARTICLE, PAIR_IMQ><<<grid, T2_NT, lds, stream>>>(f);
    else svmpc_tick2_kernel<DUST_MODEL_PARTICLE, PAIR_K1><<<grid, T2_NT, lds, stream>>>(f);
  }
  return (int)hipGetLastError();
}

}  // namespace dust
