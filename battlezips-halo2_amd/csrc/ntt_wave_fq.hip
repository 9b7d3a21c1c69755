// k_ntt_pass_wave for FqParams (csrc/ntt_wave.cuh): its own translation unit so the build compiles the fields in parallel.
#include "ntt_wave.cuh"
namespace bzh {
template void launch_ntt_pass_wave<FqParams>(const NttPassArgs&, unsigned, unsigned, hipStream_t);
}
