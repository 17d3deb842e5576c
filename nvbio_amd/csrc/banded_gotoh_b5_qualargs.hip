// explicit instantiation: band 5, QualArgs
#include "banded_gotoh_impl.h"
namespace nvb { template hipError_t launch_band_width<5, QualArgs>(const GotohParams&, const QualArgs&, int, bool, hipStream_t); }
