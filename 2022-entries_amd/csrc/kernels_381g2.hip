// kernels_381g2.hip -- the gfx950 kernels instantiated for Bls12_381_G2 (one translation unit per curve: see launch.hpp).
#include "launch_impl.hpp"

namespace msm {
template struct Launch<Bls12_381_G2::E>;
}
