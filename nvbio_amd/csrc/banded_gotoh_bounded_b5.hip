// explicit instantiation: the bounded kernels (banded_gotoh_bounded.h), band 5
#include "banded_gotoh_bounded.h"
namespace nvb {
template hipError_t launch_band_width_bounded<5, QualArgs>(const GotohParams&, const QualArgs&, const BoundArgs&, int, bool, hipStream_t);
template hipError_t launch_band_width_bounded<5, NoQual>(const GotohParams&, const NoQual&, const BoundArgs&, int, bool, hipStream_t);
}
