// compat/nvbio/io/output/output_utils.h -- small helpers over CIGARs and MD strings as the aligner stores them
// (nvbio/io/output/output_utils.h:38-124): CIGARs are held last operation first, MD strings as a byte program
// [len lo][len hi] then (op, payload) tokens.
#pragma once
#include "../alignments.h"
#include "../../basic/dna.h"

namespace nvbio {
namespace io {

/// where an alignment starts on the reference: window base + the low 16 bits of its sink
inline uint32 compute_cigar_pos(const uint32 sink, const uint32 alignment) { return alignment + (sink & 0xFFFFu); }

namespace priv {
/// sum of the lengths of the CIGAR operations accepted by `keep`
template <typename vector_type, typename Keep>
inline uint32 cigar_span(const vector_type cigar, const uint32 cigar_len, const Keep keep)
{
    uint32 total = 0;
    for (uint32 i = 0; i < cigar_len; ++i) if (keep(uint32(cigar[i].m_type))) total += cigar[i].m_len;
    return total;
}
struct on_reference { bool operator()(const uint32 op) const { return op == Cigar::SUBSTITUTION || op == Cigar::DELETION; } };
struct equal_to_op  { uint32 want; bool operator()(const uint32 op) const { return op == want; } };
} // namespace priv

/// reference bases covered by a CIGAR
template <typename vector_type>
inline uint32 reference_cigar_length(const vector_type cigar, const uint32 cigar_len) { return priv::cigar_span(cigar, cigar_len, priv::on_reference()); }
/// bases under operations of one type
template <typename vector_type>
inline uint32 count_symbols(const Cigar::Operation type, const vector_type cigar, const uint32 cigar_len)
{ const priv::equal_to_op keep = { uint32(type) }; return priv::cigar_span(cigar, cigar_len, keep); }

/// mismatches, gap opens and gap extensions of an MD program
template <typename vector_type>
inline void analyze_md_string(const vector_type mds, uint32& n_mm, uint32& n_gapo, uint32& n_gape)
{
    const uint32 end = uint32(mds[0]) | (uint32(mds[1]) << 8);
    n_mm = n_gapo = n_gape = 0;
    uint32 i = 2;
    while (i < end)
    {
        const uint8 op = mds[i++];
        if (op == MDS_MATCH)          { ++i; while (i < end && mds[i] == MDS_MATCH) ++i; }          // (a run's continuation tokens, as the reference walks them)
        else if (op == MDS_MISMATCH)  { ++n_mm; ++i; }
        else if (op == MDS_INSERTION || op == MDS_DELETION) { const uint8 l = mds[i++]; ++n_gapo; n_gape += l - 1u; i += l; }
    }
}

} // namespace io
} // namespace nvbio
