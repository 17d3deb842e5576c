// explicit instantiation: band 7, NoQual
#include "banded_gotoh_impl.h"
namespace nvb { template hipError_t launch_band_width<7, NoQual>(const GotohParams&, const NoQual&, int, bool, hipStream_t); }
