// explicit instantiation: band 5, NoQual
#include "banded_gotoh_impl.h"
namespace nvb { template hipError_t launch_band_width<5, NoQual>(const GotohParams&, const NoQual&, int, bool, hipStream_t); }
