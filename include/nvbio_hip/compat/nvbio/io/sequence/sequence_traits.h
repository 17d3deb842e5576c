// compat/nvbio/io/sequence/sequence_traits.h -- io::SequenceDataTraits<ALPHABET> (nvbio/io/sequence/sequence_traits.h):
// how sequence data of an alphabet is packed (big-endian, AlphabetTraits bits).
#pragma once
#include "../../basic/dna.h"

namespace nvbio {
namespace io {

/// how the mates of a paired-end read are oriented (io/sequence/sequence.h:188-196): F = forward, R = reverse
enum PairedEndPolicy { PE_POLICY_FF = 0, PE_POLICY_FR = 1, PE_POLICY_RF = 2, PE_POLICY_RR = 3 };

template <Alphabet ALPHABET>
struct SequenceDataTraits
{
    static const uint32 SEQUENCE_BITS       = AlphabetTraits<ALPHABET>::SYMBOL_SIZE;
    static const bool   SEQUENCE_BIG_ENDIAN = true;
    static const uint32 SEQUENCE_SYMBOLS_PER_WORD = 32u / SEQUENCE_BITS;
};

} // namespace io
} // namespace nvbio
