// compat/nvbio/basic/iterator.h -- iterator category vocabulary (nvbio/basic/iterator.h).  The reference's callers use
// thrust::host_vector / thrust::device_vector without including them (with CUDA's thrust they arrive through the headers
// nvbio/basic/types.h and iterator.h pull in); rocThrust ships with ROCm, so under hipcc this header brings them in.
#pragma once
#include "types.h"
#include <iterator>
#if defined(__HIPCC__)
#include <thrust/host_vector.h>
#include <thrust/device_vector.h>
#endif

namespace nvbio {

typedef std::input_iterator_tag          input_host_iterator_tag;
typedef std::output_iterator_tag         output_host_iterator_tag;
typedef std::forward_iterator_tag        forward_host_iterator_tag;
typedef std::bidirectional_iterator_tag  bidirectional_host_iterator_tag;
typedef std::random_access_iterator_tag  random_access_host_iterator_tag;
typedef std::random_access_iterator_tag  random_access_universal_iterator_tag;

} // namespace nvbio
