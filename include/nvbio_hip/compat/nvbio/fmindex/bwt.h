// compat/nvbio/fmindex/bwt.h -- suffix array / BWT construction for the reference's tests and small tools
// (nvbio/fmindex/bwt.h:36-88).  Conventions of the reference kept: SA has n+1 rows, SA[0] = n (the empty suffix '$'), rows
// 1..n are the suffixes of T in lexicographic order (a suffix that is a proper prefix of another sorts first);
// gen_bwt_from_sa writes the n BWT symbols with the '$' row dropped and returns `primary`, the row whose suffix is 0.
// The reference delegates the sort to the third-party sais library (contrib/sais.h, not vendored here); this header sorts by
// prefix doubling (Manber-Myers with comparison sorts): ranks by the first 4 symbols, then h = 4, 8, 16, ... until all ranks
// are distinct -- 3-4 rounds on random DNA, O(n log^2 n) worst case.  Index construction is not on the hot path.
#pragma once
#include "../basic/types.h"
#include <vector>
#include <algorithm>
#include <utility>

namespace nvbio {

template <typename StreamIterator>
uint32 gen_sa(const uint32 n, const StreamIterator T, int32* SA)
{
    SA[0] = int32(n);
    if (n == 0) return 0;
    std::vector< std::pair<uint64, uint32> > keyed(n);
    std::vector<uint32> rnk(n), tmp(n);
    // round 0: key = the first 4 symbols (8 bits each, +1 so that "past the end" = 0 sorts first)
    for (uint32 i = 0; i < n; ++i)
    {
        uint64 k = 0;
        for (uint32 j = 0; j < 4u; ++j) k = (k << 9) | (i + j < n ? uint64(uint8(T[i + j])) + 1u : 0u);
        keyed[i] = std::make_pair(k, i);
    }
    for (uint32 h = 4u;; h *= 2u)
    {
        std::sort(keyed.begin(), keyed.end());
        uint32 r = 0;
        for (uint32 i = 0; i < n; ++i)
        {
            if (i && keyed[i].first != keyed[i - 1].first) ++r;
            tmp[keyed[i].second] = r;
        }
        rnk.swap(tmp);
        if (r + 1u == n || h >= n) break;
        for (uint32 i = 0; i < n; ++i)
        {
            const uint32 s = keyed[i].second;
            keyed[i].first = (uint64(rnk[s]) + 1u) << 32 | (s + h < n ? uint64(rnk[s + h]) + 1u : 0u);
        }
    }
    for (uint32 i = 0; i < n; ++i) SA[1u + rnk[i]] = int32(i);
    return 0;
}

template <typename StreamIterator>
uint32 gen_bwt_from_sa(const uint32 n, const StreamIterator T, const int32* SA, StreamIterator bwt)
{
    uint32 primary = 0, out = 0;
    for (uint32 row = 0; row <= n; ++row)
    {
        if (SA[row] == 0) { primary = row; continue; }      // the '$' of the BWT is not stored
        bwt[out++] = T[SA[row] - 1];
    }
    return primary;
}

/// the reference's rank4 counts bytes of a 2-bit text through a 256-entry table packed 4 x 8 bits (bwt.h:77-88); generated here
/// for callers that pass it along, although the rank queries of this build count with bit-planes and ignore it
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
