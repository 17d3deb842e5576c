// explicit instantiation: band 7, direction-dependent gap costs (A16X / A32X)
#include "banded_gotoh_impl.h"
namespace nvb { template hipError_t launch_band_width_asym<7>(const GotohParams&, int, bool, hipStream_t); }
