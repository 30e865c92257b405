// kernels_377g1.hip -- the gfx950 kernels instantiated for Bls12_377_G1 (one translation unit per curve: see launch.hpp).
#include "launch_impl.hpp"

namespace msm {
template struct Launch<Bls12_377_G1::E>;
}
