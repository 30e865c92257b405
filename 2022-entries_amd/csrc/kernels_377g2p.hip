// kernels_377g2p.hip -- the throughput kernels of Bls12_377_G2 with two lanes per point (fp2pair.hpp); its own translation unit so
// that it compiles beside kernels_377g2.hip.
#include "launch_pair_impl.hpp"

namespace msm {
template struct LaunchPair<Bls12_377_G2::E>;
}
