// compat/nvbio/basic/dna.h -- the alphabets the path uses (nvbio/basic/dna.h, alphabet.h): DNA = 2 bits (ACGT),
// DNA_N = 4 bits (ACGTN).
#pragma once
#include "types.h"

namespace nvbio {

enum Alphabet { DNA = 0u, DNA_N = 1u, DNA_IUPAC = 2u, PROTEIN = 3u, RNA = 4u, RNA_N = 5u, ASCII = 6u };
template <Alphabet A> struct AlphabetTraits {};
template <> struct AlphabetTraits<DNA>   { static const uint32 SYMBOL_SIZE = 2; static const uint32 SYMBOL_COUNT = 4; };
template <> struct AlphabetTraits<DNA_N> { static const uint32 SYMBOL_SIZE = 4; static const uint32 SYMBOL_COUNT = 5; };

/// symbol width of an alphabet, at run time (strings/alphabet.h: bits_per_symbol)
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 bits_per_symbol(const Alphabet a) { return a == DNA ? 2u : (a == DNA_N || a == DNA_IUPAC || a == RNA_N) ? 4u : a == RNA ? 2u : 8u; }

NVBIO_FORCEINLINE NVBIO_HOST_DEVICE char  dna_to_char(const uint8 c) { return c == 0 ? 'A' : c == 1 ? 'C' : c == 2 ? 'G' : c == 3 ? 'T' : 'N'; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint8 char_to_dna(const char c)  { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u; }
template <typename SymbolIterator>
NVBIO_HOST_DEVICE inline void dna_to_string(const SymbolIterator begin, const uint32 n, char* string)
{ for (uint32 i = 0; i < n; ++i) string[i] = dna_to_char(begin[i]); string[n] = '\0'; }
template <typename SymbolIterator>
NVBIO_HOST_DEVICE inline void dna_to_string(const SymbolIterator begin, const SymbolIterator end, char* string)
{ uint32 i = 0; for (SymbolIterator it = begin; it != end; ++it) string[i++] = dna_to_char(*it); string[i] = '\0'; }

/// ASCII -> DNA symbols, for a [begin, end) range or a NUL-terminated string (dna.h:165-190)
template <typename SymbolIterator>
NVBIO_HOST_DEVICE inline void string_to_dna(const char* begin, const char* end, SymbolIterator symbols)
{ for (uint32 i = 0; begin + i != end; ++i) symbols[i] = char_to_dna(begin[i]); }
template <typename SymbolIterator>
NVBIO_HOST_DEVICE inline void string_to_dna(const char* begin, SymbolIterator symbols)
{ for (uint32 i = 0; begin[i] != '\0'; ++i) symbols[i] = char_to_dna(begin[i]); }

} // namespace nvbio
