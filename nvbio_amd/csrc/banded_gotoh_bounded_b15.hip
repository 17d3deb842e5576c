// explicit instantiation: the bounded kernels (banded_gotoh_bounded.h), band 15
#include "banded_gotoh_bounded.h"
namespace nvb {
template hipError_t launch_band_width_bounded<15, QualArgs>(const GotohParams&, const QualArgs&, const BoundArgs&, int, bool, hipStream_t);
template hipError_t launch_band_width_bounded<15, NoQual>(const GotohParams&, const NoQual&, const BoundArgs&, int, bool, hipStream_t);
}
