// compat/nvbio/fmindex/bwt.h -- gen_bwt_count_table (nvbio/fmindex/bwt.h:77-88).  The reference's rank4 counts bytes of a
// 2-bit text through a 256-entry table packed 4 x 8 bits; the table is generated here for callers that pass it along,
// although the rank queries of this build count with bit-planes and ignore it.
#pragma once
#include "../basic/types.h"

namespace nvbio {

inline void gen_bwt_count_table(uint32* count_table)
{
    for (uint32 b = 0; b < 256u; ++b)
    {
        uint32 x = 0;
        for (uint32 s = 0; s < 4u; ++s) x += 1u << (8u * ((b >> (2u * s)) & 3u));
        count_table[b] = x;
    }
}

} // namespace nvbio
