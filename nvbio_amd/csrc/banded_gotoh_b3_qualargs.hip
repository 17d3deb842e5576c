// explicit instantiation: band 3, QualArgs
#include "banded_gotoh_impl.h"
namespace nvb { template hipError_t launch_band_width<3, QualArgs>(const GotohParams&, const QualArgs&, int, bool, hipStream_t); }
