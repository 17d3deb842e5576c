// common.h -- device helpers shared by the gfx950 kernels of the hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nvbio_hip.h"

#define NVB_API extern "C" __attribute__((visibility("default")))

namespace nvb {

// thread-local name of the last kernel variant launched (introspection for tests/bench)
extern thread_local const char* g_last_kernel;

// Test switches (include/nvbio_hip.h, "Test switches"): alternative executions of the same results.  Each is an atomic int, seeded ONCE
// from the environment variable of the same name (std::call_once) and changed afterwards only through nvbio_hip_set_test_switch --
// no getenv on the launch paths, which run on several driver threads at a time.
enum TestSwitch { SW_FORCE_32BIT = 0, SW_NO_STAGING, SW_FULL_GENERIC, SW_ED_SWEEP, SW_FULL_SINGLE_JOB, SW_FULL_ROWS, SW_TRACEBACK_LANES, SW_SELECT_LANES, SW_COUNT };
int test_switch(TestSwitch which);       // the switch's integer value; 0 = default execution

// ---------------------------------------------------------------------------
// Packed-stream decoding.  The reference addresses symbols one at a time through
// PackedStream (nvbio/basic/packedstream_inl.h:336-400).  Here a thread pulls a
// whole group of 16 symbols with two or three dword loads and a funnel shift, and
// normalises it to a little-endian-in-register form (symbol r of the group at
// bits [W*r, W*r+W)), so the unrolled DP rows extract their symbol with one
// static v_bfe_u32.
// ---------------------------------------------------------------------------

// reverse the order of the 16 2-bit symbols of a word (big-endian -> canonical)
__device__ __forceinline__ uint32_t rev2(uint32_t x)
{
    x = __brev(x);
    return ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
}
// reverse the order of the 8 4-bit symbols of a word
__device__ __forceinline__ uint32_t rev4(uint32_t x)
{
    x = __builtin_bswap32(x);
    return ((x & 0xF0F0F0F0u) >> 4) | ((x & 0x0F0F0F0Fu) << 4);
}
// ({hi,lo} >> sh)[31:0], sh in [0,31]
__device__ __forceinline__ uint32_t funnel(uint32_t lo, uint32_t hi, uint32_t sh)
{
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

typedef const uint32_t __attribute__((address_space(3)))* lds_words_t;   // LDS pointer: ds_read, not flat

struct Stream {
    const uint32_t* words;
    uint64_t        n_words;     // loads are clamped to [0, n_words)
    uint32_t        bits;
    uint32_t        big_endian;
    // optional per-lane LDS copy of words [kb, kb + staged words): lane-interleaved, this lane's
    // word w at lds[w * 256] (the pointer already includes the lane's slot).  NULL = read HBM.
    lds_words_t     lds;
    uint64_t        kb;
};

__device__ __forceinline__ uint32_t ld_word_global(const Stream& s, uint64_t k)
{
    return s.words[k < s.n_words ? k : s.n_words - 1];
}
__device__ __forceinline__ uint32_t ld_word(const Stream& s, uint64_t k)
{
    return s.lds ? s.lds[(k - s.kb) * 256u] : ld_word_global(s, k);
}

// 16 symbols of a 2-bit stream starting at symbol index `sym`, canonical order
__device__ __forceinline__ uint32_t fetch16_2bit(const Stream& s, uint64_t sym)
{
    const uint64_t k  = sym >> 4;
    const uint32_t sh = (uint32_t(sym) & 15u) << 1;
    uint32_t w0 = ld_word(s, k), w1 = ld_word(s, k + 1);
    if (s.big_endian) { w0 = rev2(w0); w1 = rev2(w1); }
    return funnel(w0, w1, sh);
}
// 16 symbols of a 4-bit stream -> 64 bits canonical
__device__ __forceinline__ uint64_t fetch16_4bit(const Stream& s, uint64_t sym)
{
    const uint64_t k  = sym >> 3;
    const uint32_t sh = (uint32_t(sym) & 7u) << 2;
    uint32_t w0 = ld_word(s, k), w1 = ld_word(s, k + 1), w2 = ld_word(s, k + 2);
    if (s.big_endian) { w0 = rev4(w0); w1 = rev4(w1); w2 = rev4(w2); }
    const uint32_t lo = funnel(w0, w1, sh), hi = funnel(w1, w2, sh);
    return (uint64_t(hi) << 32) | lo;
}
// 16 symbols of an 8-bit stream as 16 nibbles.  The text is 2-bit, so all that matters of a pattern byte is WHICH text symbol it
// equals, if any: 0..3 stay, 255 -- the value the reference compares a text position past the end as (gotoh_banded_inl.h:580) --
// becomes 15, any other byte 4 (equal to nothing)
__device__ __forceinline__ uint64_t fetch16_8bit(const Stream& s, uint64_t sym)
{
    const uint64_t k  = sym >> 2;
    const uint32_t sh = (uint32_t(sym) & 3u) << 3;
    uint32_t w[5];
    #pragma unroll
    for (int i = 0; i < 5; ++i) { w[i] = ld_word(s, k + i); if (s.big_endian) w[i] = __builtin_bswap32(w[i]); }
    uint64_t out = 0;
    #pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const uint32_t b = funnel(w[i], w[i + 1], sh);
        const uint32_t t = b & 0xFCFCFCFCu;                                                    // bits that make a byte >= 4
        const uint32_t big = ((((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u) >> 7;      // 0x01 per byte >= 4
        const uint32_t n = ~b;
        const uint32_t ff = ((((n & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | n) & 0x80808080u) >> 7;       // 0x01 per byte != 255
        uint32_t v = ((b & 0x03030303u) & ~(big * 3u)) | (big << 2) | ((big & ~ff) * 15u);     // 255: 4 | 15 = 15
        v = (v | (v >> 4)) & 0x00FF00FFu;
        v = (v | (v >> 8)) & 0x0000FFFFu;
        out |= uint64_t(v) << (16 * i);
    }
    return out;
}
// spread 16 2-bit symbols to 16 nibbles
__device__ __forceinline__ uint64_t expand_2to4(uint32_t v)
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x <<  8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x <<  4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x <<  2)) & 0x3333333333333333ull;
    return x;
}
// one symbol (generic, used off the hot loops)
__device__ __forceinline__ uint32_t get_symbol(const Stream& s, uint64_t sym)
{
    if (s.bits == 2) {
        const uint32_t w = ld_word(s, sym >> 4), k = uint32_t(sym) & 15u;
        return (w >> (s.big_endian ? 30u - 2u * k : 2u * k)) & 3u;
    } else if (s.bits == 4) {
        const uint32_t w = ld_word(s, sym >> 3), k = uint32_t(sym) & 7u;
        return (w >> (s.big_endian ? 28u - 4u * k : 4u * k)) & 15u;
    } else {
        const uint32_t w = ld_word(s, sym >> 2), k = uint32_t(sym) & 3u;
        return (w >> (s.big_endian ? 24u - 8u * k : 8u * k)) & 255u;
    }
}

// A pattern with no rows under the text-blocking exit test (gotoh_inl.h:1203-1214): the column maximum stays at its initial value, so
// after each block of BLK text columns that is not the last one the test reads -2^30 + missing_cols * match < min_score.  Returns the
// first block that fires (the sink saw the row above the matrix up to that block's end), 0xFFFFFFFF if none does.
__host__ __device__ inline uint32_t empty_pattern_exit_block(uint32_t N, uint32_t BLK, int32_t match, int32_t min_score)
{
    const uint32_t nb = BLK * ((N + BLK - 1u) / BLK), end_block = nb > BLK ? nb : BLK;
    for (uint32_t block = 0; block + BLK < end_block; block += BLK)
        if (-(1 << 30) + int32_t(N - block - BLK) * match < min_score) return block;
    return 0xFFFFFFFFu;
}

struct StringSet {
    Stream          s;
    const uint64_t* begin;
    const uint32_t* length;
    uint32_t        fixed_length;
};

inline StringSet make_string_set(const nvbio_hip_string_set* h)
{
    StringSet d;
    d.s.words = h->words; d.s.n_words = h->n_words; d.s.bits = h->bits; d.s.big_endian = h->big_endian;
    d.s.lds = nullptr; d.s.kb = 0;
    d.begin = h->begin; d.length = h->length; d.fixed_length = h->fixed_length;
    return d;
}

inline hipStream_t to_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// blocks of a seeding kernel (map / locate) over n_items reads or rows, 256 lanes each
uint32_t seeding_grid(uint64_t n_items);

} // namespace nvb
