// compat/nvbio/io/sequence/sequence_traits.h -- io::SequenceDataTraits<ALPHABET> (nvbio/io/sequence/sequence_traits.h):
// how sequence data of an alphabet is packed (big-endian, AlphabetTraits bits).
#pragma once
#include "../../basic/dna.h"

namespace nvbio {
namespace io {

template <Alphabet ALPHABET>
struct SequenceDataTraits
{
    static const uint32 SEQUENCE_BITS       = AlphabetTraits<ALPHABET>::SYMBOL_SIZE;
    static const bool   SEQUENCE_BIG_ENDIAN = true;
    static const uint32 SEQUENCE_SYMBOLS_PER_WORD = 32u / SEQUENCE_BITS;
};

} // namespace io
} // namespace nvbio
