// full_gotoh_striped.h -- parameters of the striped full-matrix sweep (full_gotoh_striped.hip), filled by full_gotoh.hip's entry points
#pragma once
#include "common.h"

namespace nvb {

struct StripeParams {
    StringSet      pat, txt;
    int32_t        match, mismatch;
    int32_t        e_go, e_ge;         // the move along the text (E)
    int32_t        f_go, f_ge;         // the move down the pattern (F)
    int32_t        col_go, col_ge;     // H(r, -1) = col_go + col_ge * r   (non-LOCAL)
    int32_t        row_go, row_ge;     // H(-1, c) = row_go + row_ge * c   (GLOBAL)
    int32_t        infimum;
    uint32_t       linear;             // SW / ED recurrences: E = H(left) + e_go, F = H(above) + f_go (no gap state: what the reference computes)
    uint32_t       trunc;              // model the int16 boundary column
    uint32_t       blk_log2;           // 3 Gotoh, 4 SW / ED
    uint32_t       pattern_blocking;   // LOCAL tie order only (the host admits pattern blocking without thresholds and inside int16)
    const int32_t* min_score;          // nullable; the Gotoh text-blocking exit test
    uint32_t       n;
    int32_t*       out_score; uint32_t* out_sink; uint8_t* out_ok;
    uint32_t       max_n;
    int32_t*       boundary;           // [waves][3][bnd_stride], nullptr when every pattern fits one stripe
    uint32_t       bnd_stride;
};

/// max_m: the longest pattern of the batch (> 1 024: the lines between stripes are allocated)
hipError_t launch_striped(StripeParams& p, int type, uint32_t max_m, hipStream_t s);

} // namespace nvb
