// compat/nvbio/io/alignments.h -- the alignment records nvBowtie's streams and drivers exchange (nvbio/io/alignments.h:46-330):
//   Cigar           a 16-bit CIGAR element {2-bit op, 14-bit length}
//   Alignment       64 bits: {score sign, |score|:17, edit distance (or DP sink offset):10, rc, mate, paired, discordant} + position
//   BestAlignments / PairedAlignments / BestPairedAlignments   the best-two bookkeeping of the best-mapping drivers
// Same bit layout as the product's own records (include/nvbio_hip/reduce.h, nvbio_hip_alignment_invalid()), so arrays of these
// can be handed to the C-ABI reduction / MAPQ entry points as uint64 words.
#pragma once
#include "../basic/types.h"
#include "../basic/pod.h"

namespace nvbio {
namespace io {

enum MDS_OP { MDS_MATCH = 0, MDS_MISMATCH = 1, MDS_INSERTION = 2, MDS_DELETION = 3, MDS_INVALID = 4 };

struct Cigar
{
    enum Operation { SUBSTITUTION = 0, INSERTION = 1, DELETION = 2, SOFT_CLIPPING = 3 };
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Cigar() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Cigar(const uint8 type, const uint16 len) : m_type(type), m_len(len) {}
    uint16 m_type:2, m_len:14;
};

struct Alignment
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint32 max_ed()    { return 255u; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static int32  max_score() { return (1 << 17) - 1; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static int32  min_score() { return -((1 << 17) - 1); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment(const uint32 pos, const uint32 ed, const int32 score, const uint32 rc, const uint32 mate = 0,
                                                  const bool paired = false, const bool discordant = false)
        : m_score_sgn(score < 0 ? 1u : 0u), m_score(score < 0 ? uint32(-score) : uint32(score)), m_ed(ed), m_rc(rc), m_mate(mate),
          m_paired(paired ? 1u : 0u), m_discordant(discordant ? 1u : 0u), m_align(pos) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32  score()         const { return m_score_sgn ? -int32(m_score) : int32(m_score); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_aligned()    const { return m_align != uint32(-1); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 alignment()     const { return m_align; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_rc()         const { return m_rc; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 ed()            const { return m_ed; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 sink()          const { return m_ed; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 mate()          const { return m_mate; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_paired()     const { return m_paired && is_aligned(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_unpaired()   const { return !m_paired && is_aligned(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_concordant() const { return m_paired && !m_discordant; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_discordant() const { return m_paired && m_discordant; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static Alignment invalid() { return Alignment(uint32(-1), max_ed(), max_score(), 0u, 0u, false); }
    uint32 m_score_sgn:1, m_score:17, m_ed:10, m_rc:1, m_mate:1, m_paired:1, m_discordant:1;
    uint32 m_align;
};

struct AlignmentCompare { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator()(const Alignment f, const Alignment s) const { return f.m_score > s.m_score; } };

struct BestAlignments
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BestAlignments() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BestAlignments(const Alignment& a1, const Alignment& a2) : m_a1(a1), m_a2(a2) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_aligned()           const { return m_a1.is_aligned(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   has_second()           const { return m_a2.is_aligned(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32  best_score()           const { return m_a1.score(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 best_ed()              const { return m_a1.ed(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 best_alignment_pos()   const { return m_a1.alignment(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32  second_score()         const { return m_a2.score(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 second_ed()            const { return m_a2.ed(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 second_alignment_pos() const { return m_a2.alignment(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const Alignment& best()        const { return m_a1; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const Alignment& second_best() const { return m_a2; }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const Alignment& alignment() const { return I == 0 ? m_a1 : m_a2; }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment& alignment() { return I == 0 ? m_a1 : m_a2; }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 alignment_pos() const { return I == 0 ? best_alignment_pos() : second_alignment_pos(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32  score() const { return I == 0 ? best_score() : second_score(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 ed() const { return I == 0 ? best_ed() : second_ed(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_aligned() const { return I == 0 ? m_a1.is_aligned() : m_a2.is_aligned(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_rc() const { return I == 0 ? m_a1.is_rc() : m_a2.is_rc(); }
    Alignment m_a1, m_a2;
};

struct PairedAlignments
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE PairedAlignments() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE PairedAlignments(const Alignment& a, const Alignment& o) : m_a(a), m_o(o) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool  is_aligned()    const { return m_a.is_aligned() && m_o.is_aligned(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool  is_paired()     const { return m_a.is_paired(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool  is_concordant() const { return m_a.is_concordant(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool  is_discordant() const { return m_a.is_discordant(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 score()         const { return m_a.score() + m_o.score(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 ed()            const { return int32(m_a.ed() + m_o.ed()); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment& mate(const uint32 m)       { return m == m_a.mate() ? m_a : m_o; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment  mate(const uint32 m) const { return m == m_a.mate() ? m_a : m_o; }
    Alignment m_a, m_o;
};

struct BestPairedAlignments
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BestPairedAlignments() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BestPairedAlignments(const BestAlignments& a, const BestAlignments& o) : m_a1(a.m_a1), m_a2(a.m_a2), m_o1(o.m_a1), m_o2(o.m_a2) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BestPairedAlignments(const BestAlignments& a) : m_a1(a.m_a1), m_a2(a.m_a2), m_o1(Alignment::invalid()), m_o2(Alignment::invalid()) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_paired()            const { return m_a1.is_paired(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_aligned()           const { return m_a1.is_aligned(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   has_second_paired()    const { return m_a2.is_paired(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   has_second_unpaired()  const { return m_a2.is_unpaired(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   has_second()           const { return is_paired() ? has_second_paired() : m_a2.is_aligned(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32  best_score()           const { return m_a1.score() + (is_paired() ? m_o1.score() : 0); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 best_ed()              const { return m_a1.ed() + (is_paired() ? m_o1.ed() : 0u); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 best_alignment_pos()   const { return m_a1.alignment(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32  second_score()         const { return m_a2.score() + (has_second_paired() ? m_o2.score() : 0); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 second_ed()            const { return m_a2.ed() + (has_second_paired() ? m_o2.ed() : 0u); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 second_alignment_pos() const { return m_a2.alignment(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BestAlignments best_anchor()   const { return BestAlignments(m_a1, m_a2); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BestAlignments best_opposite() const { return BestAlignments(m_o1, m_o2); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE PairedAlignments pair() const { return I == 0 ? PairedAlignments(m_a1, m_o1) : PairedAlignments(m_a2, m_o2); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const Alignment& alignment() const { return I == 0 ? m_a1 : m_a2; }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment& alignment() { return I == 0 ? m_a1 : m_a2; }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const Alignment& opposite_alignment() const { return I == 0 ? m_o1 : m_o2; }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment& opposite_alignment() { return I == 0 ? m_o1 : m_o2; }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 alignment_pos() const { return I == 0 ? best_alignment_pos() : second_alignment_pos(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 opposite_alignment_pos() const { return I == 0 ? m_o1.alignment() : m_o2.alignment(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32  score() const { return I == 0 ? best_score() : second_score(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 ed() const { return I == 0 ? best_ed() : second_ed(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_aligned() const { return I == 0 ? m_a1.is_aligned() : m_a2.is_aligned(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_rc() const { return I == 0 ? m_a1.is_rc() : m_a2.is_rc(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_opposite_rc() const { return I == 0 ? m_o1.is_rc() : m_o2.is_rc(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 anchor_mate()    const { return I == 0 ? m_a1.mate()  : m_a2.mate(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 opposite_mate()  const { return I == 0 ? m_o1.mate()  : m_o2.mate(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32  anchor_score()   const { return I == 0 ? m_a1.score() : m_a2.score(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 anchor_ed()      const { return I == 0 ? m_a1.ed()    : m_a2.ed(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32  opposite_score() const { return I == 0 ? m_o1.score() : m_o2.score(); }
    template <uint32 I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 opposite_ed()    const { return I == 0 ? m_o1.ed()    : m_o2.ed(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BestAlignments anchor()   const { return BestAlignments(m_a1, m_a2); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BestAlignments opposite() const { return BestAlignments(m_o1, m_o2); }
    /// the best and second-best slots of read file `m`, whichever side holds them
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BestAlignments mate(const uint32 m) const { return BestAlignments(m == m_a1.mate() ? m_a1 : m_o1, m == m_a2.mate() ? m_a2 : m_o2); }
    Alignment m_a1, m_a2, m_o1, m_o2;
};

/// predicates over the records, for thrust-style counting and partitioning (alignments.h:345-420)
struct has_second          { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator()(const BestAlignments& b) { return b.has_second(); }
                             NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator()(const BestPairedAlignments& b) { return b.has_second(); } };
struct has_second_paired   { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator()(const BestAlignments& b) { return b.has_second() && b.alignment<1>().is_paired(); }
                             NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator()(const BestPairedAlignments& b) { return b.has_second_paired(); } };
struct has_second_unpaired { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator()(const BestAlignments& b) { return b.has_second() && b.alignment<1>().is_unpaired(); }
                             NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator()(const BestPairedAlignments& b) { return b.has_second_unpaired(); } };
#define NVBIO_HIP_ALIGNMENT_PREDICATE(name, test)                                                                            \
    struct name { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator()(const BestAlignments& b) { const Alignment& a = b.alignment<0>(); return test; } \
                  NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator()(const Alignment& a) { return test; } };
NVBIO_HIP_ALIGNMENT_PREDICATE(is_paired,         a.is_paired())
NVBIO_HIP_ALIGNMENT_PREDICATE(is_unpaired,       a.is_unpaired())
NVBIO_HIP_ALIGNMENT_PREDICATE(is_concordant,     a.is_concordant())
NVBIO_HIP_ALIGNMENT_PREDICATE(is_discordant,     a.is_discordant())
NVBIO_HIP_ALIGNMENT_PREDICATE(is_not_concordant, a.is_concordant() == false)      // unpaired or discordant
NVBIO_HIP_ALIGNMENT_PREDICATE(is_aligned,        a.is_aligned())
#undef NVBIO_HIP_ALIGNMENT_PREDICATE

/// are two placements different?  Same strand and within `dist` of each other = the same placement (alignments_inl.h:34-136).
/// The Alignment / PairedAlignments forms compare END positions (alignment() + sink()): valid while the records still hold DP sinks.
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool distinct_alignments(const uint32 pos1, const bool rc1, const uint32 pos2, const bool rc2, const uint32 dist)
{
    if (rc1 != rc2) return true;
    const uint32 lo = pos2 - (pos2 < dist ? pos2 : dist);
    return !(pos1 >= lo && pos1 <= pos2 + dist);
}
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool distinct_alignments(const uint32 apos1, const uint32 opos1, const bool arc1, const bool orc1,
                                                             const uint32 apos2, const uint32 opos2, const bool arc2, const bool orc2)
{ return arc1 != arc2 || orc1 != orc2 || apos1 != apos2 || opos1 != opos2; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool distinct_alignments(const uint32 apos1, const uint32 opos1, const bool arc1, const bool orc1,
                                                             const uint32 apos2, const uint32 opos2, const bool arc2, const bool orc2, const uint32 dist)
{ return distinct_alignments(apos1, arc1, apos2, arc2, dist) || distinct_alignments(opos1, orc1, opos2, orc2, dist); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool distinct_alignments(const Alignment& p1, const Alignment& p2, const uint32 dist = 1)
{ return distinct_alignments(p1.alignment() + p1.sink(), p1.is_rc(), p2.alignment() + p2.sink(), p2.is_rc(), dist); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool distinct_alignments(const PairedAlignments& p1, const PairedAlignments& p2)
{
    const Alignment a1 = p1.mate(0), o1 = p1.mate(1), a2 = p2.mate(0), o2 = p2.mate(1);
    return distinct_alignments(a1.alignment() + a1.sink(), o1.alignment() + o1.sink(), a1.is_rc(), o1.is_rc(), a2.alignment() + a2.sink(), o2.alignment() + o2.sink(), a2.is_rc(), o2.is_rc());
}
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool distinct_alignments(const PairedAlignments& p1, const PairedAlignments& p2, const uint32 dist)
{
    const Alignment a1 = p1.mate(0), o1 = p1.mate(1), a2 = p2.mate(0), o2 = p2.mate(1);
    return distinct_alignments(a1.alignment() + a1.sink(), o1.alignment() + o1.sink(), a1.is_rc(), o1.is_rc(), a2.alignment() + a2.sink(), o2.alignment() + o2.sink(), a2.is_rc(), o2.is_rc(), dist);
}

} // namespace io

template <> struct pod_type<io::Alignment>      { typedef uint2 type; };
template <> struct pod_type<io::BestAlignments> { typedef uint4 type; };

} // namespace nvbio
