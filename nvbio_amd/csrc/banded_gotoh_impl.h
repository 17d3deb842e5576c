// banded_gotoh_impl.h -- kernel templates of the banded Gotoh score (see banded_gotoh.hip for the design notes)
#pragma once
#include "common.h"
#include <limits.h>
#include <algorithm>
#include <stdlib.h>

#include <type_traits>
namespace nvb {

struct GotohParams {
    StringSet pat, txt;
    int32_t   match, mismatch, gap_open, gap_ext;   // gap_* = the scheme's pattern_gap_open / _extension
    int32_t   txt_gap_open, txt_gap_ext;            // text_gap_* : GLOBAL row-zero init and the infimum only
    int32_t   f_gap_open, f_gap_ext;                // the asymmetric instances (A16X / A32X) only: the costs of the move from the previous row (F);
                                                    // gap_* are then the costs of the move along the row (E).  Linear SW gaps: open == ext
    uint32_t  n;
    uint32_t  len_lo, len_hi;     // this launch handles jobs with len_lo <= pattern_len <= len_hi
    uint32_t  plain16;            // the 16-bit launch runs the recurrence as the reference writes it (A16P): LOCAL jobs too long for the row frame's 16 bits
    uint32_t  stage_pw, stage_tw; // words of pattern / text each lane stages in LDS (0 = read HBM per block)
    int32_t*  out_score;
    uint32_t* out_sink;
    const uint32_t* n_dev;        // optional: the job count on the device (n is then the arrays' capacity) -- no host round trip to size the launch
    const uint32_t* out_index;    // optional: job i's results go to out_score[out_index[i]] / out_sink[out_index[i]] (a compacted batch writing back at its hits)
    const uint32_t* gate;         // optional: the launch does nothing unless *gate > gate_limit (a device-side choice between this kernel and the
    uint32_t        gate_limit;   //           wave-per-job kernel, which runs when *gate <= gate_limit: no host round trip to pick one)
};

// nvBowtie's quality-aware scheme (nvBowtie/bowtie2/cuda/scoring.h:283-293): the mismatch score is a
// function of the read symbol's quality byte, tabulated by the host for all 256 bytes.
struct QualArgs {
    const uint8_t* quals;        // quality of the pattern symbol at stream index begin + i
    uint64_t       n_quals;      // bytes allocated (loads are clamped)
    int32_t        lut[256];     // mismatch(q)
    // optional per-job view of the pattern (nvbio::io::ReadStream, nvbio/io/utils.h:100-330; how nvBowtie's streams hand out a read,
    // nvBowtie/bowtie2/cuda/alignment_utils.h:194-211): bit 0 = walk the stored symbols [begin, begin + M) backwards (symbol k of the
    // pattern is stored symbol begin + M - 1 - k, its quality likewise), bit 1 = complement the bases (c < 4 -> 3 - c).  NULL = none.
    const uint8_t* flags;
};
struct NoQual {};

template <int BAND> struct BandTraits {
    // bands 3,5,7,15: the reference's text cache is a plain uint32 array; any other band uses a
    // 2-bit PackedStream cache which truncates what it stores to 2 bits
    // (nvbio/alignment/alignment_base_inl.h:75-98, packedstream_inl.h:352-369)
    static constexpr bool QUIRK = !(BAND == 3 || BAND == 5 || BAND == 7 || BAND == 15);
    // The band's text symbols sit in a ring of registers indexed by (row + column) & MASK: with the rows of a block unrolled every index is
    // an instruction constant and no symbol is ever moved.  Bands up to 16 take a ring of 16 (a block of 16 rows brings it back to where it
    // started); wider bands a ring of 32, whose halves change places after each block of 16 rows (ring_advance: sixteen v_swap_b32).  Round 6
    // measured what the shifted array cost band 31 before: two v_mov_b32 per cell (the shift, and its copy for the lanes whose row is
    // masked off) = 62 of a row's 343 vector instructions.
    static constexpr int  MASK  = (BAND <= 16) ? 15 : 31;
    static constexpr int  ROWS  = 16;
    static constexpr int  NTC   = MASK + 1;
};

// ---------------------------------------------------------------------------
// arithmetic policies.  A value is an int32 (A32) or an int16 kept in the low
// half of a VGPR (A16; gfx9 16-bit VOP2 ops zero the high half).
// ---------------------------------------------------------------------------
struct A32 {
    typedef int32_t T;
    static constexpr bool ASYM = false;      // one pair of gap costs for both directions (the Gotoh schemes)
    static constexpr bool ROWTREND = true;   // the recurrence in the row frame (cell_rt, below)
    static __device__ __forceinline__ T   sub(T a, T b)        { return a - b; }
    static __device__ __forceinline__ T   add(T a, T b)        { return a + b; }
    static __device__ __forceinline__ T   mx(T a, T b)         { return max(a, b); }
    static __device__ __forceinline__ T   mx3(T a, T b, T c)   { return max(max(a, b), c); }
    static __device__ __forceinline__ T   clamp0(T a)          { return max(a, 0); }
    template <int J> static __device__ __forceinline__ T key(T hi) { return hi + J; }      // hi is a multiple of 32
    static __device__ __forceinline__ T   cnst(int32_t v)      { return v; }
    static __device__ __forceinline__ int32_t to_int(T a)      { return a; }
    static __device__ __forceinline__ uint32_t bits(T a)       { return uint32_t(a); }
    static __device__ __forceinline__ T   min_value()          { return INT_MIN; }
    // how text / pattern symbols are held for the substitution test: as they are
    static constexpr bool TABLE = false;
    static __device__ __forceinline__ uint32_t enc(uint32_t g)      { return g; }
    static __device__ __forceinline__ uint32_t enc_none()           { return 255u; }
    static __device__ __forceinline__ T subst(uint32_t, uint32_t, uint32_t) { return 0; }

    // one interior band cell (gotoh_banded_inl.h:520-577): F, H, E and (LOCAL) the row's sink key
    template <int TYPE, int J, bool FAST>
    static __device__ __forceinline__ void cell(T& Fj, const T Fnext, const T HGnext, T& HGj, T& E, T& rowkey,
                                                const uint32_t g, const uint32_t q, const T Go, const T Ge, const T sM, const T sX,
                                                const uint32_t, const uint32_t, const T = 0, const T = 0)
    {
        Fj = max(Fnext + Ge, HGnext);
        const T diag = HGj + (g == q ? sM : sX);
        T hi = max(max(Fj, E), diag);
        if (TYPE == NVBIO_HIP_LOCAL) { hi = max(hi, 0); rowkey = max(rowkey, hi + J); }
        HGj = hi + Go;
        E = max(E + Ge, HGj);
    }
    // The same cell in the ROW FRAME.  Every value of row i is held plus (i + 1) |G_e| -- X' = X - (i + 1) G_e, the row-zero values as they
    // are -- and the band keeps S = H' + G_o - G_e instead of H + G_o.  Moving down a row then costs F nothing:
    //   F'(i,j) = F(i,j) - (i+1) G_e = max(F(i-1,j+1) + G_e, H(i-1,j+1) + G_o) - (i+1) G_e = max(F'(i-1,j+1), S(i-1,j+1))
    // the diagonal keeps its pre-biased substitution score, H(i-1,j) + s - (i+1) G_e = S(i-1,j) + (s - G_o), and the gap along the row pays the
    // step it used to pay, E'(i,j+1) = max(E'(i,j), S(i,j)) + G_e.  One add less per cell (9 -> 8 instructions, LOCAL 12 -> 11); LOCAL's floor
    // is the row's own zero Z = (i + 1) |G_e| (one register per row), the sink key h' + j is compared inside the row as it is and loses Z once
    // per row.  Reachable values grow by M |G_e|: max_len_16bit (banded_gotoh.hip) counts that in.
    template <int TYPE, int J, bool FAST>
    static __device__ __forceinline__ void cell_rt(T& Fj, const T Fnext, const T Snext, T& Sj, T& E, T& rowkey,
                                                   const uint32_t g, const uint32_t q, const T GoE, const T Ge, const T sM, const T sX,
                                                   const uint32_t, const uint32_t, const T Z)
    {
        Fj = max(Fnext, Snext);
        const T diag = Sj + (g == q ? sM : sX);
        T hi = max(max(Fj, E), diag);
        if (TYPE == NVBIO_HIP_LOCAL) { hi = max(hi, Z); rowkey = max(rowkey, hi + J); }
        Sj = hi + GoE;
        E = max(E, Sj) + Ge;
    }
};
struct A16 {
    typedef uint32_t T;
    static constexpr bool ASYM = false;
    static constexpr bool ROWTREND = true;   // the recurrence in the row frame (A32::cell_rt has the derivation)
    static __device__ __forceinline__ T sub(T a, T b)      { T r; asm("v_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    static __device__ __forceinline__ T add(T a, T b)      { T r; asm("v_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    static __device__ __forceinline__ T mx(T a, T b)       { T r; asm("v_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    static __device__ __forceinline__ T mx3(T a, T b, T c) { return mx(mx(a, b), c); }
    static __device__ __forceinline__ T clamp0(T a)        { T r; asm("v_max_i16 %0, 0, %1" : "=v"(r) : "v"(a)); return r; }
    template <int J> static __device__ __forceinline__ T key(T hi) { T r; asm("v_add_u16 %0, %2, %1" : "=v"(r) : "v"(hi), "I"(J)); return r; }
    static __device__ __forceinline__ T cnst(int32_t v)    { return uint32_t(v) & 0xFFFFu; }
    static __device__ __forceinline__ int32_t to_int(T a)  { return int32_t(int16_t(a & 0xFFFFu)); }
    static __device__ __forceinline__ uint32_t bits(T a)   { return a & 0xFFFFu; }
    // The substitution score without a compare.  v_cmp + v_cndmask cost 6.6 issue cycles of a cell's 32 (VOPC runs at the 32-bit
    // rate, profiles/r02/valu_probe.txt); one v_perm_b32 costs 4.3 and needs no VCC.  A symbol is held as the byte selector that
    // picks ITS 16-bit entry out of an 8-byte table: 0x0C0C0100 + 0x0202 * g = {byte 2g, byte 2g+1, zero, zero}; the row's table
    // {lo, hi} holds entry v = (v == q ? sM : sX), built once per row for the band's cells.  A symbol past the end of the text (the
    // reference's 255, equal to nothing) has no entry: it is held as 0x0C0C0C0C, and a block of rows that can see one runs the
    // compare form below on the same selectors (pattern symbols encoded alike; N codes 4..15 give selectors no text symbol has).
    static constexpr bool TABLE = true;
    static __device__ __forceinline__ uint32_t enc(uint32_t g)      { uint32_t r; asm("v_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(g), "v"(0x0202u)); return r | 0x0C0C0100u; }
    static __device__ __forceinline__ uint32_t enc_none()           { return 0x0C0C0C0Cu; }
    static __device__ __forceinline__ T subst(uint32_t tlo, uint32_t thi, uint32_t sel) { return __builtin_amdgcn_perm(thi, tlo, sel); }

    // The same cell as one hand-scheduled instruction block: 2-cycle 16-bit VOP2 ops only (plus the
    // compare/select of the substitution score), ordered so that the serial E -> H -> HG -> E chain
    // (5 ops) is interleaved with the independent F / diagonal / key work.
    template <int TYPE, int J, bool FAST>
    static __device__ __forceinline__ void cell(T& Fj, const T Fnext, const T HGnext, T& HGj, T& E, T& rowkey,
                                                const uint32_t g, const uint32_t q, const T Go, const T Ge, const T sM, const T sX,
                                                const uint32_t tlo, const uint32_t thi, const T = 0, const T = 0)
    {
        T d, h, e2;
        if (FAST && TYPE == NVBIO_HIP_LOCAL)
            asm("v_perm_b32 %[d], %[thi], %[tlo], %[g]\n\t"
                "v_add_u16 %[f], %[fn], %[ge]\n\t"
                "v_max_i16 %[f], %[f], %[hgn]\n\t"
                "v_add_u16 %[d], %[hg], %[d]\n\t"
                "v_max_i16 %[h], %[f], %[e]\n\t"
                "v_add_u16 %[e2], %[e], %[ge]\n\t"
                "v_max_i16 %[h], %[h], %[d]\n\t"
                "v_max_i16 %[h], 0, %[h]\n\t"
                "v_add_u16 %[hg], %[h], %[go]\n\t"
                "v_add_u16 %[d], %[sj], %[h]\n\t"
                "v_max_i16 %[e], %[e2], %[hg]\n\t"
                "v_max_i16 %[rk], %[rk], %[d]"
                : [f] "=&v"(Fj), [d] "=&v"(d), [h] "=&v"(h), [e2] "=&v"(e2), [hg] "+v"(HGj), [e] "+v"(E), [rk] "+v"(rowkey)
                : [g] "v"(g), [tlo] "v"(tlo), [thi] "v"(thi), [fn] "v"(Fnext), [ge] "v"(Ge), [hgn] "v"(HGnext), [go] "v"(Go), [sj] "I"(J));
        else if (FAST)
            asm("v_perm_b32 %[d], %[thi], %[tlo], %[g]\n\t"
                "v_add_u16 %[f], %[fn], %[ge]\n\t"
                "v_max_i16 %[f], %[f], %[hgn]\n\t"
                "v_add_u16 %[d], %[hg], %[d]\n\t"
                "v_max_i16 %[h], %[f], %[e]\n\t"
                "v_add_u16 %[e2], %[e], %[ge]\n\t"
                "v_max_i16 %[h], %[h], %[d]\n\t"
                "v_add_u16 %[hg], %[h], %[go]\n\t"
                "v_max_i16 %[e], %[e2], %[hg]"
                : [f] "=&v"(Fj), [d] "=&v"(d), [h] "=&v"(h), [e2] "=&v"(e2), [hg] "+v"(HGj), [e] "+v"(E)
                : [g] "v"(g), [tlo] "v"(tlo), [thi] "v"(thi), [fn] "v"(Fnext), [ge] "v"(Ge), [hgn] "v"(HGnext), [go] "v"(Go));
        else if (TYPE == NVBIO_HIP_LOCAL)
            asm("v_cmp_eq_u32 vcc, %[g], %[q]\n\t"
                "v_add_u16 %[f], %[fn], %[ge]\n\t"
                "v_cndmask_b32 %[d], %[sx], %[sm], vcc\n\t"
                "v_max_i16 %[f], %[f], %[hgn]\n\t"
                "v_add_u16 %[d], %[hg], %[d]\n\t"
                "v_max_i16 %[h], %[f], %[e]\n\t"
                "v_add_u16 %[e2], %[e], %[ge]\n\t"
                "v_max_i16 %[h], %[h], %[d]\n\t"
                "v_max_i16 %[h], 0, %[h]\n\t"
                "v_add_u16 %[hg], %[h], %[go]\n\t"
                "v_add_u16 %[d], %[sj], %[h]\n\t"
                "v_max_i16 %[e], %[e2], %[hg]\n\t"
                "v_max_i16 %[rk], %[rk], %[d]"
                : [f] "=&v"(Fj), [d] "=&v"(d), [h] "=&v"(h), [e2] "=&v"(e2), [hg] "+v"(HGj), [e] "+v"(E), [rk] "+v"(rowkey)
                : [g] "v"(g), [q] "v"(q), [fn] "v"(Fnext), [ge] "v"(Ge), [sx] "v"(sX), [sm] "v"(sM), [hgn] "v"(HGnext), [go] "v"(Go), [sj] "I"(J)
                : "vcc");
        else
            asm("v_cmp_eq_u32 vcc, %[g], %[q]\n\t"
                "v_add_u16 %[f], %[fn], %[ge]\n\t"
                "v_cndmask_b32 %[d], %[sx], %[sm], vcc\n\t"
                "v_max_i16 %[f], %[f], %[hgn]\n\t"
                "v_add_u16 %[d], %[hg], %[d]\n\t"
                "v_max_i16 %[h], %[f], %[e]\n\t"
                "v_add_u16 %[e2], %[e], %[ge]\n\t"
                "v_max_i16 %[h], %[h], %[d]\n\t"
                "v_add_u16 %[hg], %[h], %[go]\n\t"
                "v_max_i16 %[e], %[e2], %[hg]"
                : [f] "=&v"(Fj), [d] "=&v"(d), [h] "=&v"(h), [e2] "=&v"(e2), [hg] "+v"(HGj), [e] "+v"(E)
                : [g] "v"(g), [q] "v"(q), [fn] "v"(Fnext), [ge] "v"(Ge), [sx] "v"(sX), [sm] "v"(sM), [hgn] "v"(HGnext), [go] "v"(Go)
                : "vcc");
    }
    // the row-frame cell (A32::cell_rt) as instruction blocks: S in `hg`, G_o - G_e in `goe`, the row's zero in `z`
    template <int TYPE, int J, bool FAST>
    static __device__ __forceinline__ void cell_rt(T& Fj, const T Fnext, const T Snext, T& Sj, T& E, T& rowkey,
                                                   const uint32_t g, const uint32_t q, const T GoE, const T Ge, const T sM, const T sX,
                                                   const uint32_t tlo, const uint32_t thi, const T Z)
    {
        T d, h;
        if (FAST && TYPE == NVBIO_HIP_LOCAL)
            asm("v_perm_b32 %[d], %[thi], %[tlo], %[g]\n\t"
                "v_max_i16 %[f], %[fn], %[hgn]\n\t"
                "v_add_u16 %[d], %[hg], %[d]\n\t"
                "v_max_i16 %[h], %[f], %[e]\n\t"
                "v_max_i16 %[h], %[h], %[d]\n\t"
                "v_max_i16 %[h], %[h], %[z]\n\t"
                "v_add_u16 %[hg], %[h], %[goe]\n\t"
                "v_add_u16 %[d], %[sj], %[h]\n\t"
                "v_max_i16 %[e], %[e], %[hg]\n\t"
                "v_add_u16 %[e], %[e], %[ge]\n\t"
                "v_max_i16 %[rk], %[rk], %[d]"
                : [f] "=&v"(Fj), [d] "=&v"(d), [h] "=&v"(h), [hg] "+v"(Sj), [e] "+v"(E), [rk] "+v"(rowkey)
                : [g] "v"(g), [tlo] "v"(tlo), [thi] "v"(thi), [fn] "v"(Fnext), [ge] "v"(Ge), [hgn] "v"(Snext), [goe] "v"(GoE), [z] "v"(Z), [sj] "I"(J));
        else if (FAST)
            asm("v_perm_b32 %[d], %[thi], %[tlo], %[g]\n\t"
                "v_max_i16 %[f], %[fn], %[hgn]\n\t"
                "v_add_u16 %[d], %[hg], %[d]\n\t"
                "v_max_i16 %[h], %[f], %[e]\n\t"
                "v_max_i16 %[h], %[h], %[d]\n\t"
                "v_add_u16 %[hg], %[h], %[goe]\n\t"
                "v_max_i16 %[e], %[e], %[hg]\n\t"
                "v_add_u16 %[e], %[e], %[ge]"
                : [f] "=&v"(Fj), [d] "=&v"(d), [h] "=&v"(h), [hg] "+v"(Sj), [e] "+v"(E)
                : [g] "v"(g), [tlo] "v"(tlo), [thi] "v"(thi), [fn] "v"(Fnext), [ge] "v"(Ge), [hgn] "v"(Snext), [goe] "v"(GoE));
        else if (TYPE == NVBIO_HIP_LOCAL)
            asm("v_cmp_eq_u32 vcc, %[g], %[q]\n\t"
                "v_max_i16 %[f], %[fn], %[hgn]\n\t"
                "v_cndmask_b32 %[d], %[sx], %[sm], vcc\n\t"
                "v_add_u16 %[d], %[hg], %[d]\n\t"
                "v_max_i16 %[h], %[f], %[e]\n\t"
                "v_max_i16 %[h], %[h], %[d]\n\t"
                "v_max_i16 %[h], %[h], %[z]\n\t"
                "v_add_u16 %[hg], %[h], %[goe]\n\t"
                "v_add_u16 %[d], %[sj], %[h]\n\t"
                "v_max_i16 %[e], %[e], %[hg]\n\t"
                "v_add_u16 %[e], %[e], %[ge]\n\t"
                "v_max_i16 %[rk], %[rk], %[d]"
                : [f] "=&v"(Fj), [d] "=&v"(d), [h] "=&v"(h), [hg] "+v"(Sj), [e] "+v"(E), [rk] "+v"(rowkey)
                : [g] "v"(g), [q] "v"(q), [fn] "v"(Fnext), [ge] "v"(Ge), [sx] "v"(sX), [sm] "v"(sM), [hgn] "v"(Snext), [goe] "v"(GoE), [z] "v"(Z), [sj] "I"(J)
                : "vcc");
        else
            asm("v_cmp_eq_u32 vcc, %[g], %[q]\n\t"
                "v_max_i16 %[f], %[fn], %[hgn]\n\t"
                "v_cndmask_b32 %[d], %[sx], %[sm], vcc\n\t"
                "v_add_u16 %[d], %[hg], %[d]\n\t"
                "v_max_i16 %[h], %[f], %[e]\n\t"
                "v_max_i16 %[h], %[h], %[d]\n\t"
                "v_add_u16 %[hg], %[h], %[goe]\n\t"
                "v_max_i16 %[e], %[e], %[hg]\n\t"
                "v_add_u16 %[e], %[e], %[ge]"
                : [f] "=&v"(Fj), [d] "=&v"(d), [h] "=&v"(h), [hg] "+v"(Sj), [e] "+v"(E)
                : [g] "v"(g), [q] "v"(q), [fn] "v"(Fnext), [ge] "v"(Ge), [sx] "v"(sX), [sm] "v"(sM), [hgn] "v"(Snext), [goe] "v"(GoE)
                : "vcc");
    }
    static __device__ __forceinline__ T min_value()        { return 0x8000u; }
};

// The policies with the recurrence as the reference writes it (H + G_o in the band, F and E paying their step): what the bounded kernel runs
// (banded_gotoh_bounded.h: its row bound reads the band's values as scores).
template <typename Base> struct Plain : Base { static constexpr bool ROWTREND = false; };
typedef Plain<A32> A32P;
typedef Plain<A16> A16P;

// Direction-dependent gap costs: SmithWatermanAligner with deletion != insertion (sw_banded_inl.h:378-379, :413, :432-433 -- the move from
// the previous row costs `deletion`, the move along the row `insertion`).  The same cell with F taking its own pair of costs:
//   F = max(F' + GeF, H' + GoF) = max(F' + GeF, HG' + dF),  dF = GoF - Go      (HG = H + Go serves E and the diagonal as before)
// one more add per cell than the symmetric block, left to the compiler's scheduler (the primitive ops are the same 16-bit / 32-bit ones).
template <typename Base>
struct Asym : Base {
    typedef typename Base::T T;
    static constexpr bool ASYM = true;
    static constexpr bool ROWTREND = false;  // (F has its own costs: it would need its own frame)
    template <int TYPE, int J, bool FAST>
    static __device__ __forceinline__ void cell(T& Fj, const T Fnext, const T HGnext, T& HGj, T& E, T& rowkey,
                                                const uint32_t g, const uint32_t q, const T Go, const T Ge, const T sM, const T sX,
                                                const uint32_t tlo, const uint32_t thi, const T GeF, const T dF)
    {
        const T sub = (FAST && Base::TABLE) ? Base::subst(tlo, thi, g) : (g == q ? sM : sX);
        Fj = Base::mx(Base::add(Fnext, GeF), Base::add(HGnext, dF));
        const T diag = Base::add(HGj, sub);
        T hi = Base::mx(Base::mx(Fj, E), diag);
        if (TYPE == NVBIO_HIP_LOCAL) { hi = Base::clamp0(hi); rowkey = Base::mx(rowkey, Base::template key<J>(hi)); }
        HGj = Base::add(hi, Go);
        E = Base::mx(Base::add(E, Ge), HGj);
    }
};
typedef Asym<A32> A32X;
typedef Asym<A16> A16X;

template <int BAND, typename A>
struct DPState {
    typename A::T HG[BAND];                  // (H + G_o) of the previous row  [x32 for LOCAL]
    typename A::T F[BAND - 1];               // F[BAND-1] is always `infimum`
    uint32_t      tc[BandTraits<BAND>::NTC]; // text symbols of the band
    typename A::T bestkey;                   // LOCAL: score*32 + j of the best cell so far
    uint32_t      besti;                     //        and its row
    typename A::T z;                         // A::ROWTREND: the row's zero, (i + 1) |G_e| (x32 for LOCAL)
    typename A::T infrow;                    //              the reference's infimum as the previous row's frame holds it: infimum + i |G_e|
};

template <typename A>
struct DPConsts {
    typename A::T Go, Ge, sM, sX, inf;       // sM/sX = match/mismatch - G_o ; inf = the infimum sentinel
    typename A::T GeF, dF;                   // A::ASYM: F's extension cost and (F's opening cost - G_o)
    typename A::T GoE, Zstep;                // A::ROWTREND: G_o - G_e, |G_e|
    bool bytes;                              // the patterns are 8-bit strings
    uint32_t sMM, sXX;                       // table arithmetic: sM / sX in both halves of a dword
};

// after a block of 16 rows: a ring of 32 has turned half way round -- its halves change places, so that the next block's row R again finds
// column j at (R + j) & 31.  (A ring of 16 is back where it started.)
template <int BAND, typename A>
__device__ __forceinline__ void ring_advance(DPState<BAND, A>& st)
{
    if constexpr (BandTraits<BAND>::MASK == 31)
    {
        #pragma unroll
        for (int k = 0; k < 16; ++k)
            asm("v_swap_b32 %0, %1" : "+v"(st.tc[k]), "+v"(st.tc[k + 16]));
    }
}

template <int BAND, int TYPE, typename A, bool FAST, int R, int J, int END>
struct CellLoop {
    __device__ __forceinline__ static void run(DPState<BAND, A>& st, const DPConsts<A>& k, const typename A::T sX,
                                               typename A::T& E, typename A::T& rowkey, const uint32_t q, const uint32_t tlo, const uint32_t thi)
    {
        typedef BandTraits<BAND> BT;
        typedef typename A::T T;
        const uint32_t g = st.tc[(R + J) & BT::MASK];                       // (:542: the reference shifts its cache here; the ring does not move)
        const T fnext = (J + 1 == BAND - 1) ? (A::ROWTREND ? st.infrow : k.inf) : st.F[J + 1 < BAND - 1 ? J + 1 : 0];
        if constexpr (A::ROWTREND) A::template cell_rt<TYPE, J, FAST>(st.F[J], fnext, st.HG[J + 1], st.HG[J], E, rowkey, g, q, k.GoE, k.Ge, k.sM, sX, tlo, thi, st.z);
        else                       A::template cell<TYPE, J, FAST>(st.F[J], fnext, st.HG[J + 1], st.HG[J], E, rowkey, g, q, k.Go, k.Ge, k.sM, sX, tlo, thi, k.GeF, k.dF);
        CellLoop<BAND, TYPE, A, FAST, R, J + 1, END>::run(st, k, sX, E, rowkey, q, tlo, thi);
    }
};
template <int BAND, int TYPE, typename A, bool FAST, int R, int END>
struct CellLoop<BAND, TYPE, A, FAST, R, END, END> {
    __device__ __forceinline__ static void run(DPState<BAND, A>&, const DPConsts<A>&, const typename A::T, typename A::T&, typename A::T&, const uint32_t,
                                               const uint32_t, const uint32_t) {}
};

// q, g_new, g_store and the cached symbols are in A's encoding (A::enc); FAST rows take substitution scores from the row's table
// {tlo, thi} and must not see a symbol past the text's end; g_store is what later rows read back for the entering symbol.
template <int BAND, int TYPE, typename A, bool FAST, int R>
__device__ __forceinline__ void dp_row(DPState<BAND, A>& st, const DPConsts<A>& k, const typename A::T sX,
                                       const uint32_t i, const uint32_t q, const uint32_t g_new, const uint32_t g_store,
                                       const uint32_t tlo, const uint32_t thi)
{
    typedef BandTraits<BAND> BT;
    typedef typename A::T T;
    T rowkey = A::min_value();
    T E;

    // j == 0  (gotoh_banded_inl.h:483-517)
    if constexpr (A::ROWTREND)
    {
        // F[BAND-1] is the reference's infimum at every row (:586) -- a value that takes part in the maxima once scores fall below it
        // (tests/test_banded_gpu.py::test_large_negative_scores_cross_infimum): in the previous row's frame it stands at infimum + i |G_e|
        st.infrow = A::add(k.inf, st.z);
        st.z = A::add(st.z, k.Zstep);                                          // this row's zero
        st.F[0] = A::mx((1 == BAND - 1) ? st.infrow : st.F[1 < BAND - 1 ? 1 : 0], st.HG[1]);
        const uint32_t g = st.tc[R & BT::MASK];
        const T diag = A::add(st.HG[0], FAST ? A::subst(tlo, thi, g) : (g == q ? k.sM : sX));
        T hi = A::mx(st.F[0], diag);
        if (TYPE == NVBIO_HIP_LOCAL) { hi = A::mx(hi, st.z); rowkey = hi; }
        st.HG[0] = A::add(hi, k.GoE);
        E = A::add(st.HG[0], k.Ge);
    }
    else
    {
        const T fnext = A::add((1 == BAND - 1) ? k.inf : st.F[1 < BAND - 1 ? 1 : 0], A::ASYM ? k.GeF : k.Ge);
        st.F[0] = A::mx(fnext, A::ASYM ? A::add(st.HG[1], k.dF) : st.HG[1]);
        const uint32_t g = st.tc[R & BT::MASK];
        const T diag = A::add(st.HG[0], FAST ? A::subst(tlo, thi, g) : (g == q ? k.sM : sX));
        T hi = A::mx(st.F[0], diag);
        if (TYPE == NVBIO_HIP_LOCAL) { hi = A::clamp0(hi); rowkey = hi; }
        st.HG[0] = A::add(hi, k.Go);
        E = st.HG[0];
    }
    // 1 <= j <= BAND-2  (:520-577)
    if constexpr (BAND > 16)
    {
        // wide bands: compile-time recursion over the cells.  (A `#pragma unroll` loop whose body is a 29-way switch of
        // asm blocks is not unrolled by the compiler at this size; the band state would then be indexed dynamically and
        // live in scratch memory.)
        CellLoop<BAND, TYPE, A, FAST, R, 1, BAND - 1>::run(st, k, sX, E, rowkey, q, tlo, thi);
    }
    else
    {
    #pragma unroll
    for (int j = 1; j < BAND - 1; ++j)
    {
        const uint32_t g = st.tc[(R + j) & BT::MASK];
        // F[BAND-1] is `infimum` at every row (:586), so the cell next to the band edge sees it as F[j+1]
        const T fnext = (j + 1 == BAND - 1) ? (A::ROWTREND ? st.infrow : k.inf) : st.F[j + 1 < BAND - 1 ? j + 1 : 0];
        switch (j) {   // the sink key's column is an instruction constant
            #define NVB_CELL(J) case J: if constexpr (A::ROWTREND) A::template cell_rt<TYPE, J, FAST>(st.F[j], fnext, st.HG[j + 1], st.HG[j], E, rowkey, g, q, k.GoE, k.Ge, k.sM, sX, tlo, thi, st.z); \
                                        else                       A::template cell<TYPE, J, FAST>(st.F[j], fnext, st.HG[j + 1], st.HG[j], E, rowkey, g, q, k.Go, k.Ge, k.sM, sX, tlo, thi, k.GeF, k.dF); break;
            NVB_CELL(1) NVB_CELL(2) NVB_CELL(3) NVB_CELL(4) NVB_CELL(5) NVB_CELL(6) NVB_CELL(7) NVB_CELL(8) NVB_CELL(9) NVB_CELL(10)
            NVB_CELL(11) NVB_CELL(12) NVB_CELL(13) NVB_CELL(14)
            #undef NVB_CELL
            default: break;
        }
    }
    }
    // the new text symbol enters the band (:580-581); the cached copy is what later rows see
    {
        st.tc[(R + BAND - 1) & BT::MASK] = g_store;
    }
    // j == BAND-1  (:584-614) -- compares against the raw symbol
    {
        const T diag = A::add(st.HG[BAND - 1], FAST ? A::subst(tlo, thi, g_new) : (g_new == q ? k.sM : sX));
        T hi = A::mx(E, diag);
        if (TYPE == NVBIO_HIP_LOCAL) { hi = A::ROWTREND ? A::mx(hi, st.z) : A::clamp0(hi); rowkey = A::mx(rowkey, A::template key<BAND - 1>(hi)); }
        st.HG[BAND - 1] = A::add(hi, A::ROWTREND ? k.GoE : k.Go);
    }
    if (TYPE == NVBIO_HIP_LOCAL)
    {
        // BestSink::report uses '<=' (sink_inl.h:57-68): a later cell with an equal score wins.
        // LOCAL keys are non-negative, so the comparison is the same in either width.  (Row frame: the row's keys lose its zero here.)
        if (A::ROWTREND) rowkey = A::sub(rowkey, st.z);
        const uint32_t rk = A::bits(rowkey), bk = A::bits(st.bestkey);
        const bool upd = (rk | 31u) >= bk;
        st.bestkey = upd ? rowkey : st.bestkey;
        st.besti   = upd ? i : st.besti;
    }
}

// FAST (table arithmetic only): the caller has checked that no row of this block lets a symbol past the text's end into the band
template <int BAND, int TYPE, typename A, bool QUAL, bool FAST, int R, int END>
struct RowUnrollN {
    __device__ __forceinline__ static void run(DPState<BAND, A>& st, const DPConsts<A>& k,
        const uint32_t i0, const uint32_t M, const uint32_t N, const uint64_t P, const uint32_t T,
        const uint4 Q, const typename A::T* lut, const uint2* masks)
    {
        typedef BandTraits<BAND> BT;
        const uint32_t i = i0 + R;
        if (i < M)
        {
            const uint32_t qr = uint32_t(P >> (4 * R)) & 15u;
            const uint32_t gr = (T >> (2 * R)) & 3u;
            const bool past = !FAST && (i + BAND - 1 >= N);
            // the entering symbol as this row compares it, and as later rows read it back from the reference's text cache
            const uint32_t g       = past ? A::enc_none() : A::enc(gr);
            const uint32_t g_store = past ? (BT::QUIRK ? A::enc(3u) : A::enc_none()) : g;        // the 2-bit cache keeps 255 & 3
            typename A::T sX = k.sX;
            if (QUAL) {
                const uint32_t w = (R >> 2) == 0 ? Q.x : (R >> 2) == 1 ? Q.y : (R >> 2) == 2 ? Q.z : Q.w;
                sX = lut[(w >> (8 * (R & 3))) & 255u];          // mismatch(quality of row i), LDS
            }
            uint32_t tlo = 0, thi = 0;
            if (FAST) {
                // entry v of the row's table: sM where v is the row's symbol, sX elsewhere (masks[q]: which 16-bit halves take sM)
                const uint2 m = masks[qr];
                const uint32_t xx = QUAL ? (uint32_t(sX) | (uint32_t(sX) << 16)) : k.sXX;
                tlo = (k.sMM & m.x) | (xx & ~m.x);
                thi = (k.sMM & m.y) | (xx & ~m.y);
            }
            // (an 8-bit pattern's byte 255 arrives as code 15, fetch16_8bit: it equals a text position past the end, and nothing else)
            dp_row<BAND, TYPE, A, FAST, R>(st, k, sX, i, FAST ? 0u : ((k.bytes && qr == 15u) ? A::enc_none() : A::enc(qr)), g, g_store, tlo, thi);
        }
        RowUnrollN<BAND, TYPE, A, QUAL, FAST, R + 1, END>::run(st, k, i0, M, N, P, T, Q, lut, masks);
    }
};
template <int BAND, int TYPE, typename A, bool QUAL, bool FAST, int END> struct RowUnrollN<BAND, TYPE, A, QUAL, FAST, END, END> {
    __device__ __forceinline__ static void run(DPState<BAND, A>&, const DPConsts<A>&, uint32_t, uint32_t, uint32_t, uint64_t, uint32_t,
                                               uint4, const typename A::T*, const uint2*) {}
};

// 16 quality bytes starting at byte `off` (unaligned dword loads, clamped to the array)
__device__ __forceinline__ uint32_t ld_qual_word(const QualArgs& qa, uint64_t off)
{
    const uint64_t last = qa.n_quals - 4u;
    const uint64_t lo = off < last ? off : last;
    const uint32_t sh = uint32_t(off - lo) * 8u;
    uint32_t w;
    __builtin_memcpy(&w, qa.quals + lo, 4);
    return sh >= 32u ? 0u : (w >> sh);
}
__device__ __forceinline__ uint4 fetch_quals16(const QualArgs& qa, uint64_t off)
{
    return make_uint4(ld_qual_word(qa, off), ld_qual_word(qa, off + 4), ld_qual_word(qa, off + 8), ld_qual_word(qa, off + 12));
}
__device__ __forceinline__ uint4 fetch_quals16(const NoQual&, uint64_t) { return make_uint4(0, 0, 0, 0); }

__device__ __forceinline__ uint64_t fetch_pattern16(const Stream& s, uint64_t sym)
{
    return (s.bits == 4) ? fetch16_4bit(s, sym) : (s.bits == 8) ? fetch16_8bit(s, sym) : expand_2to4(fetch16_2bit(s, sym));
}

__host__ __device__ __forceinline__ uint32_t stage_words_pattern(uint32_t off, uint32_t M, uint32_t bits);

// ---- pattern views (QualArgs::flags).  The 16 pattern symbols / qualities of rows i0 .. i0+15 of a job whose pattern is stored
// symbols [pb, pb + M).  A reversed view reads the group of stored symbols [last - i0 - 15, last - i0], last = pb + M - 1, and turns it
// round in registers: nothing is copied or staged per job, the DP rows see the same canonical group either way.
__device__ __forceinline__ uint32_t view_flags(const QualArgs& qa, uint32_t id) { return qa.flags ? uint32_t(qa.flags[id]) : 0u; }
__device__ __forceinline__ uint32_t view_flags(const NoQual&, uint32_t)        { return 0u; }
__device__ __forceinline__ uint64_t reverse_nibbles(uint64_t x)
{
    x = __builtin_bswap64(x);
    return ((x & 0xF0F0F0F0F0F0F0F0ull) >> 4) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
}
__device__ __forceinline__ uint64_t complement_nibbles(uint64_t v)        // c < 4 -> 3 - c, anything else (N) stays
{
    const uint64_t big = ((v >> 2) | (v >> 3)) & 0x1111111111111111ull;  // 1 where the nibble is >= 4
    return v ^ ((big ^ 0x1111111111111111ull) * 3ull);
}
// 16 quality bytes of stored positions s .. s+15 where s may be negative (a reversed read at the very start of the array): bytes
// before the array are zero -- they belong to rows past the pattern's end
__device__ __forceinline__ uint4 fetch_quals16_signed(const QualArgs& qa, int64_t s)
{
    if (s >= 0) return fetch_quals16(qa, uint64_t(s));
    uint32_t w[4] = { 0u, 0u, 0u, 0u };
    for (int r = 0; r < 16; ++r) {
        const int64_t pos = s + r;
        const uint32_t b = pos < 0 ? 0u : uint32_t(qa.quals[uint64_t(pos) < qa.n_quals ? uint64_t(pos) : qa.n_quals - 1u]);
        w[r >> 2] |= b << (8 * (r & 3));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void fetch_group(const Stream& ps, const QualArgs& qa, const uint64_t pb, const uint32_t M, const uint32_t fl, const uint32_t i0,
                                            uint64_t& P, uint4& Q)
{
    if ((fl & 1u) == 0u) { P = fetch_pattern16(ps, pb + i0); Q = fetch_quals16(qa, pb + i0); }
    else
    {
        const int64_t s = int64_t(pb + M - 1u) - int64_t(i0) - 15;
        const uint64_t raw = s >= 0 ? fetch_pattern16(ps, uint64_t(s)) : (s > -16 ? fetch_pattern16(ps, 0u) << (4u * uint32_t(-s)) : 0ull);
        const uint4 q = fetch_quals16_signed(qa, s);
        P = reverse_nibbles(raw);
        Q = make_uint4(__builtin_bswap32(q.w), __builtin_bswap32(q.z), __builtin_bswap32(q.y), __builtin_bswap32(q.x));
    }
    if (fl & 2u) P = complement_nibbles(P);
}
__device__ __forceinline__ void fetch_group(const Stream& ps, const NoQual&, const uint64_t pb, const uint32_t, const uint32_t, const uint32_t i0, uint64_t& P, uint4& Q)
{
    P = fetch_pattern16(ps, pb + i0); Q = make_uint4(0, 0, 0, 0);
}
// first word and word count of the pattern words a lane touches, for its LDS copy
__device__ __forceinline__ void pattern_words_span(const uint64_t pb, const uint32_t M, const uint32_t bits, const uint32_t fl, uint64_t& first_word, uint32_t& words)
{
    const uint32_t per = 32u / bits;
    if ((fl & 1u) == 0u) { first_word = pb / per; words = stage_words_pattern(uint32_t(pb % per), M, bits); }
    else
    {
        const uint32_t c16 = (M + 15u) & ~15u;
        const int64_t last = int64_t(pb + M - 1u);
        const int64_t lo = last - int64_t(c16) - 15, hi = last - 15;             // first symbol of the lowest / highest group fetched
        first_word = uint64_t(lo > 0 ? lo : 0) / per;
        words = uint32_t(uint64_t(hi > 0 ? hi : 0) / per + (bits == 4 ? 3u : 2u) - first_word);
    }
}

// the sentinel standing in for the reference's infimum (-32768 - max(G_o,G_e), :446-448).
// 32-bit: the reference's own number (scaled for LOCAL).  16-bit: the lowest value whose
// G_e step is still representable; the host only selects the 16-bit kernel when every reachable
// DP value stays far above it, in which case either sentinel loses every max it takes part in.
template <typename A> struct Sentinel {};
template <> struct Sentinel<A32> {
    static __device__ __forceinline__ int32_t get(int32_t Go, int32_t Ge, int32_t tGo, int32_t tGe, int sh)
    { return (-32768 - max(max(Go, Ge), max(tGo, tGe))) * (1 << sh); }
};
template <> struct Sentinel<A16> {   // infimum + G_e == -32768 exactly: the add cannot wrap
    static __device__ __forceinline__ uint32_t get(int32_t, int32_t Ge, int32_t, int32_t, int sh) { return A16::cnst(-32768 - Ge * (1 << sh)); }
};
template <> struct Sentinel<A32P> : Sentinel<A32> {};
template <> struct Sentinel<A16P> : Sentinel<A16> {};
template <> struct Sentinel<A32X> : Sentinel<A32> {};
template <> struct Sentinel<A16X> : Sentinel<A16> {};      // (the caller passes F's extension cost: the only step ever added to the sentinel)

// Words a lane touches over the whole DP: the last 16-symbol group fetched starts at ceil16(M) (the
// prefetch past the last block) for the pattern and at ceil16(M) + BAND - 1 for the text; a group
// fetch reads up to 24 (4-bit: 3 words) resp. 32 (2-bit: 2 words) symbols from its first word on.
__host__ __device__ __forceinline__ uint32_t stage_words_pattern(uint32_t off, uint32_t M, uint32_t bits)
{
    const uint32_t c16 = (M + 15u) & ~15u;
    return ((off + c16 + (bits == 4 ? 24u : bits == 8 ? 20u : 32u)) * bits + 31u) / 32u;
}
__host__ __device__ __forceinline__ uint32_t stage_words_text(uint32_t off, uint32_t M, uint32_t band)
{
    const uint32_t c16 = (M + 15u) & ~15u;
    return ((off + c16 + band - 1u + 32u) * 2u + 31u) / 32u;
}

template <typename QA> struct IsQual { static constexpr bool value = false; };
template <> struct IsQual<QualArgs> { static constexpr bool value = true; };

template <typename A> __device__ __forceinline__ void fill_lut(typename A::T* lut, const QualArgs& qa, int32_t Go, int sh)
{
    lut[threadIdx.x] = A::cnst((qa.lut[threadIdx.x] - Go) * (1 << sh));       // pre-biased by -G_o like sX
    __syncthreads();
}
template <typename A> __device__ __forceinline__ void fill_lut(typename A::T*, const NoQual&, int32_t, int) {}

// band 31 with per-base qualities (nvBowtie's default band: max_dist 15): with the band's text symbols in an array shifted every row the 16-bit
// instances took 173-189 VGPRs when left alone -- two waves per SIMD, a few registers over the 168 that allow three -- and this bound made the
// compiler fit them (profiles/r04/band31_occupancy.txt).  With the ring of 32 (BandTraits) every band-31 instance takes 158-160 VGPRs and no
// scratch; the bound stays as a statement of what the kernel is scheduled for.
template <typename A> struct occupancy_bound { static const int b31 = 1; };
template <> struct occupancy_bound<A16> { static const int b31 = 3; };
template <int BAND, int TYPE, typename A, typename QA>
__global__ void __launch_bounds__(256, (BAND == 31 ? occupancy_bound<A>::b31 : 1))
banded_gotoh_score_kernel(const GotohParams p, const QA qa)
{
    typedef BandTraits<BAND> BT;
    typedef typename A::T T;
    constexpr bool QUAL = IsQual<QA>::value;
    constexpr int SH = (TYPE == NVBIO_HIP_LOCAL) ? 5 : 0;      // LOCAL carries scores x32 (see header)
    __shared__ T s_lut[QUAL ? 256 : 1];
    __shared__ uint2 s_masks[16];            // table arithmetic: which halves of the row's {lo, hi} hold the match score, by pattern symbol
    if (A::TABLE && threadIdx.x < 16u)
        s_masks[threadIdx.x] = make_uint2(threadIdx.x == 0u ? 0x0000FFFFu : threadIdx.x == 1u ? 0xFFFF0000u : 0u,
                                          threadIdx.x == 2u ? 0x0000FFFFu : threadIdx.x == 3u ? 0xFFFF0000u : 0u);
    if (A::TABLE && !QUAL) __syncthreads();
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stage[];    // [stage_pw + stage_tw][256]
    fill_lut<A>(s_lut, qa, p.gap_open, SH);
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= (p.n_dev ? *p.n_dev : p.n)) return;
    if (p.gate && *p.gate <= p.gate_limit) return;

    const uint32_t M  = p.pat.length ? p.pat.length[id] : p.pat.fixed_length;
    if (M < p.len_lo || M > p.len_hi) return;                   // the other arithmetic width owns this job
    const uint64_t pb = p.pat.begin[id];
    const uint64_t tb = p.txt.begin[id];
    const uint32_t N  = p.txt.length ? p.txt.length[id] : p.txt.fixed_length;
    const uint32_t fl = view_flags(qa, id);

    int32_t  score = -(1 << 30);                 // BestSink<int32>() : numbers.h:832-835
    uint32_t sx = 0xFFFFFFFFu, sy = 0xFFFFFFFFu;

    if (N >= M)                                  // gotoh_banded_inl.h:431-432
    {
        // Stage this lane's read and reference-window words in LDS once (lane-interleaved, so the
        // later ds_read_b32 are conflict-free).  Without it the lane re-touches its 128-byte lines
        // every 16 rows, and the ~4 MB working set of an XCD's in-flight lanes is evicted from the
        // 4 MB L2 in between: measured 1.9x the algorithmic fabric traffic.  A lane whose strings
        // are longer than the launch was sized for simply keeps reading HBM.
        Stream ps = p.pat.s, ts = p.txt.s;
        if (p.stage_pw != 0u)
        {
            uint64_t kbp; uint32_t pwords;
            pattern_words_span(pb, M, ps.bits, fl, kbp, pwords);
            const uint64_t kbt = tb >> 4;
            if (pwords <= p.stage_pw &&
                stage_words_text(uint32_t(tb & 15u), M, BAND) <= p.stage_tw)
            {
                uint32_t* lp = s_stage + threadIdx.x;
                uint32_t* lt = s_stage + p.stage_pw * 256u + threadIdx.x;
                // stage_pw / stage_tw are multiples of 4: four loads in flight per step (clamped, so the
                // round-up over-read is harmless)
                for (uint32_t w = 0; w < p.stage_pw; w += 4u) {
                    const uint32_t a = ld_word_global(ps, kbp + w), b = ld_word_global(ps, kbp + w + 1u),
                                   c = ld_word_global(ps, kbp + w + 2u), d = ld_word_global(ps, kbp + w + 3u);
                    lp[w * 256u] = a; lp[(w + 1u) * 256u] = b; lp[(w + 2u) * 256u] = c; lp[(w + 3u) * 256u] = d;
                }
                for (uint32_t w = 0; w < p.stage_tw; w += 4u) {
                    const uint32_t a = ld_word_global(ts, kbt + w), b = ld_word_global(ts, kbt + w + 1u),
                                   c = ld_word_global(ts, kbt + w + 2u), d = ld_word_global(ts, kbt + w + 3u);
                    lt[w * 256u] = a; lt[(w + 1u) * 256u] = b; lt[(w + 2u) * 256u] = c; lt[(w + 3u) * 256u] = d;
                }
                ps.lds = (lds_words_t)lp; ps.kb = kbp;
                ts.lds = (lds_words_t)lt; ts.kb = kbt;
            }
        }

        DPConsts<A> k;
        k.Go = A::cnst(p.gap_open * (1 << SH)); k.Ge = A::cnst(p.gap_ext * (1 << SH));
        k.sM = A::cnst((p.match - p.gap_open) * (1 << SH)); k.sX = A::cnst((p.mismatch - p.gap_open) * (1 << SH));
        k.inf = Sentinel<A>::get(p.gap_open, A::ASYM ? p.f_gap_ext : p.gap_ext, p.txt_gap_open, p.txt_gap_ext, SH);
        k.GeF = A::cnst((A::ASYM ? p.f_gap_ext : 0) * (1 << SH)); k.dF = A::cnst((A::ASYM ? p.f_gap_open - p.gap_open : 0) * (1 << SH));
        k.GoE = A::cnst((p.gap_open - p.gap_ext) * (1 << SH)); k.Zstep = A::cnst(-p.gap_ext * (1 << SH));
        constexpr int32_t RT = A::ROWTREND ? 1 : 0;                        // row frame: the band holds S = H + G_o - G_e, F starts below everything
        k.bytes = (p.pat.s.bits == 8u);
        k.sMM = (uint32_t(k.sM) & 0xFFFFu) * 0x10001u; k.sXX = (uint32_t(k.sX) & 0xFFFFu) * 0x10001u;
        const T infimum = k.inf;

        DPState<BAND, A> st;
        // init_row_zero (:46-77), stored as H + G_o (row frame: H + G_o - G_e)
        st.HG[0] = A::ROWTREND ? k.GoE : k.Go;
        #pragma unroll
        for (int j = 1; j < BAND; ++j)
            st.HG[j] = A::cnst(((TYPE == NVBIO_HIP_GLOBAL ? p.txt_gap_open + (j - 1) * p.txt_gap_ext : 0) + p.gap_open - RT * p.gap_ext) * (1 << SH));
        #pragma unroll
        for (int j = 0; j < BAND - 1; ++j) st.F[j] = infimum;
        st.bestkey = A::cnst(0); st.besti = 0; st.z = A::cnst(0); st.infrow = infimum;

        // first band of text (:441-442): symbols 0..BAND-2, no bounds check in the reference either
        {
            #pragma unroll
            for (int b = 0; b < BAND - 1; b += 16)
            {
                const uint32_t T0 = fetch16_2bit(ts, tb + b);
                #pragma unroll
                for (int j = b; j < BAND - 1 && j < b + 16; ++j)
                    st.tc[j & BT::MASK] = A::enc((T0 >> (2 * (j - b))) & 3u);
            }
        }

        uint64_t P; uint4 Q;
        fetch_group(ps, qa, pb, M, fl, 0u, P, Q);
        uint32_t Tx = fetch16_2bit(ts, tb + BAND - 1);
        for (uint32_t i0 = 0; i0 < M; i0 += BT::ROWS)
        {
            // prefetch the next block's symbols while this one computes
            uint64_t Pn; uint4 Qn;
            fetch_group(ps, qa, pb, M, fl, i0 + BT::ROWS, Pn, Qn);
            const uint32_t Tn = fetch16_2bit(ts, tb + i0 + BT::ROWS + BAND - 1);
            // a block none of whose rows lets a symbol past the text's end into the band runs on table arithmetic
            // (the block's last row that exists: rows past the pattern's end are skipped, so they need no symbol)
            if (A::TABLE && (i0 + BT::ROWS < M ? i0 + BT::ROWS : M) - 1u + BAND - 1u < N)
                RowUnrollN<BAND, TYPE, A, QUAL, true, 0, BT::ROWS>::run(st, k, i0, M, N, P, Tx, Q, s_lut, s_masks);
            else
                RowUnrollN<BAND, TYPE, A, QUAL, false, 0, BT::ROWS>::run(st, k, i0, M, N, P, Tx, Q, s_lut, s_masks);
            ring_advance<BAND, A>(st);
            P = Pn; Tx = Tn; Q = Qn;
        }

        if (TYPE == NVBIO_HIP_LOCAL)
        {
            if (M > 0) {
                const uint32_t key = uint32_t(A::to_int(st.bestkey));          // >= 0
                const uint32_t j = key & 31u;
                score = int32_t(key >> 5);
                sx = st.besti + j + 1; sy = st.besti + 1;
            }
        }
        else if (TYPE == NVBIO_HIP_GLOBAL)
        {
            score = A::to_int(st.HG[BAND - 1]) - p.gap_open + RT * int32_t(M + 1u) * p.gap_ext;      // :641-642 (row frame: out of row M-1's)
            sx = M + BAND - 1; sy = M;
        }
        else
        {
            // :643-655
            const uint32_t a = M + BAND - 1u;
            const uint32_t m = (a < N ? a : N) - (M - 1u);
            #pragma unroll
            for (int j = 0; j < BAND; ++j)
            {
                const int32_t h = A::to_int(st.HG[j]) - p.gap_open + RT * int32_t(M + 1u) * p.gap_ext;
                if ((j == 0 || uint32_t(j) < m) && score <= h) { score = h; sx = M + j; sy = M; }
            }
        }
    }
    const uint32_t o = p.out_index ? p.out_index[id] : id;
    p.out_score[o] = score;
    reinterpret_cast<uint2*>(p.out_sink)[o] = make_uint2(sx, sy);
}

template <int BAND, typename A, typename QA>
hipError_t launch_band(const GotohParams& p, const QA& qa, int type, hipStream_t stream)
{
    const dim3 grid((p.n + 255u) / 256u), block(256);
    const unsigned lds_pad = (p.stage_pw + p.stage_tw) * 256u * 4u;      // the lanes' staged words
    if constexpr (std::is_same<A, A16P>::value)                         // (instantiated for LOCAL only: the other types' row-frame limits are far out)
    {
        if (type != NVBIO_HIP_LOCAL) return hipErrorInvalidValue;
        hipLaunchKernelGGL((banded_gotoh_score_kernel<BAND, NVBIO_HIP_LOCAL, A, QA>), grid, block, lds_pad, stream, p, qa);
        return hipGetLastError();
    }
    else
    switch (type) {
    case NVBIO_HIP_GLOBAL:      hipLaunchKernelGGL((banded_gotoh_score_kernel<BAND, NVBIO_HIP_GLOBAL, A, QA>),      grid, block, lds_pad, stream, p, qa); break;
    case NVBIO_HIP_LOCAL:       hipLaunchKernelGGL((banded_gotoh_score_kernel<BAND, NVBIO_HIP_LOCAL, A, QA>),       grid, block, lds_pad, stream, p, qa); break;
    case NVBIO_HIP_SEMI_GLOBAL: hipLaunchKernelGGL((banded_gotoh_score_kernel<BAND, NVBIO_HIP_SEMI_GLOBAL, A, QA>), grid, block, lds_pad, stream, p, qa); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// One explicit instantiation per (band, scheme kind) lives in its own translation unit
// (banded_gotoh_inst.hip.inc) so that the 60 kernels build in parallel.
template <int BAND, typename QA>
hipError_t launch_band_width(const GotohParams& p, const QA& qa, int type, bool width16, hipStream_t s)
{
    return width16 ? (p.plain16 ? launch_band<BAND, A16P, QA>(p, qa, type, s) : launch_band<BAND, A16, QA>(p, qa, type, s)) : launch_band<BAND, A32, QA>(p, qa, type, s);
}

// the asymmetric instances: no qualities (SimpleSmithWatermanScheme)
template <int BAND>
hipError_t launch_band_width_asym(const GotohParams& p, int type, bool width16, hipStream_t s)
{
    return width16 ? launch_band<BAND, A16X, NoQual>(p, NoQual(), type, s) : launch_band<BAND, A32X, NoQual>(p, NoQual(), type, s);
}

} // namespace nvb
