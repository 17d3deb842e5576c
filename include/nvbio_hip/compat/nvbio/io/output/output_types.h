// compat/nvbio/io/output/output_types.h -- the value types of the alignment output interface (nvbio/io/output/output_types.h:70-135):
// BNT (names / offsets of the reference sequences), the end / mate / score selectors, device and host CIGAR arrays.
#pragma once
#include "../../basic/types.h"
#include "../alignments.h"
#include "../sequence/sequence.h"
#include "../../basic/vector_array.h"
#include "output_utils.h"

namespace nvbio {
namespace io {

struct BNT
{
    uint32        n_seqs;
    const char*   names;
    const uint32* names_index;
    const uint32* sequence_index;
    BNT(const io::ConstSequenceDataView& reference)
        : n_seqs(reference.size()), names(reference.name_stream()), names_index(reference.name_index()), sequence_index(reference.sequence_index()) {}
};

typedef enum { SINGLE_END, PAIRED_END } AlignmentType;
typedef enum { MATE_1 = 0, MATE_2 = 1 } AlignmentMate;
typedef enum { BEST_SCORE, SECOND_BEST_SCORE } AlignmentScore;

#if defined(__HIPCC__)
struct DeviceCigarArray
{
    nvbio::DeviceVectorArray<io::Cigar>& array;
    thrust::device_vector<uint2>&        coords;
    DeviceCigarArray(nvbio::DeviceVectorArray<io::Cigar>& _array, thrust::device_vector<uint2>& _coords) : array(_array), coords(_coords) {}
};
struct HostCigarArray
{
    nvbio::HostVectorArray<io::Cigar> array;
    thrust::host_vector<uint2>        coords;
};
typedef nvbio::HostVectorArray<uint8> HostMdsArray;
#endif

} // namespace io
} // namespace nvbio
