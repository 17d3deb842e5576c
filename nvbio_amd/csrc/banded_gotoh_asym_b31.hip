// explicit instantiation: band 31, direction-dependent gap costs (A16X / A32X)
#include "banded_gotoh_impl.h"
namespace nvb { template hipError_t launch_band_width_asym<31>(const GotohParams&, int, bool, hipStream_t); }
