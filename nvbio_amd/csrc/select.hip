// select.hip -- nvBowtie's hit-selection stage and the per-round glue kernels of its best-approx extension
// loop, on gfx950.
//   select_init_kernel                     nvBowtie/bowtie2/cuda/select.cu:36-103
//   select_kernel / rand_select_kernel / select_multi_kernel / rand_select_multi_kernel,
//   randomized_select                      nvBowtie/bowtie2/cuda/select_inl.h:74-607
//   SumTree<float*> / sample()             nvbio/basic/sum_tree_inl.h:38-178
//   locate_kernel (index direction)        nvBowtie/bowtie2/cuda/locate_inl.h:53-143
//   BestScoreStream::init_context          nvBowtie/bowtie2/cuda/score_best_inl.h:95-126
//
// One lane per active read: a read's selection is a short sequential walk over its own deque / probability
// tree (order dependent: LCG draws, pop_front, tree updates), reads are independent.  Where the reference takes
// output slots with warp-aggregated atomics (so the order of its output queues changes from run to run), the
// lanes here stage their picks, one scan turns the per-read counts into offsets, and a second kernel compacts:
// active reads and their hits come out in input-queue order, every run.  The multi-hit index (ReadHitsIndex
// links, scoring_queues.h:64-130) becomes a CSR offset array -- a read's hits are contiguous.
//
// Single-precision arithmetic is done operation by operation (no contraction), matching the host compile of
// the reference's HOST_DEVICE code that the oracle restates.
#include "fmindex_device.h"
#include "hit_deque.h"
#include <hipcub/hipcub.hpp>
#include <mutex>
#include <limits.h>
#include <stdlib.h>

namespace nvb {

__device__ __forceinline__ uint32_t ilog2(uint32_t n) { return 31u - uint32_t(__clz(int(n | 1u))); }       // nvbio::log2 (floor; log2(0) = 0)
__device__ __forceinline__ uint32_t st_padded(const uint32_t size) { const uint32_t l = ilog2(size); return (1u << l) < size ? 1u << (l + 1u) : 1u << l; }

// Where a tree's nodes live.  HBM holds only the LEAVES of a read's tree (leaf i at probs[r * probs_stride + i], i < the read's hit count):
// every internal node of a SumTree is the float sum of its two children, left + right (setup() builds it so, set() keeps it so), i.e. a
// function of the leaves -- so a selection call loads the leaves, rebuilds the sums on chip (the same pairwise adds, the same values bit for
// bit) and writes back the one leaf a pick zeroes.  The reference keeps all 2 * padded - 1 nodes in memory (SeedHitDequeArray::m_probs)
// and rewrites a zeroed leaf's ancestors; at 10 M reads that is the second 128-byte line of every read on every call and four more writes
// per exhausted hit (profiles/r03/select_coop.txt: the stage is bound by exactly those lines).
//   RowCells   nodes in a row of global memory (wide rows: hits_stride > 32; the row needs room for all the nodes, internal ones are scratch)
//   LdsCells   nodes in LDS, lane-interleaved (node i of this lane at s_tree[i * 256 + lane]: conflict-free)
struct RowCells { float* p; __device__ __forceinline__ float& operator[](const uint32_t i) const { return p[i]; } };
struct LdsCells { float* p; __device__ __forceinline__ float& operator[](const uint32_t i) const { return p[i * 256u]; } };

template <typename Cells>
struct SumTreeT
{
    Cells    c;
    uint32_t size, padded;
    __device__ __forceinline__ SumTreeT(const Cells cells, const uint32_t n) : c(cells), size(n), padded(st_padded(n)) {}
    __device__ __forceinline__ float sum() const { return c[padded * 2u - 2u]; }
    /// the sums above leaves that are already in place (cells [0, size)): padding leaves zeroed, then level by level
    __device__ void setup() const
    {
        for (uint32_t i = size; i < padded; ++i) c[i] = 0.0f;
        uint32_t src = 0;
        for (uint32_t n = padded; n >= 2u; n >>= 1) {
            const uint32_t dst = src + n, m = n >> 1;
            for (uint32_t i = 0; i < m; ++i) c[dst + i] = __fadd_rn(c[src + i * 2u], c[src + i * 2u + 1u]);
            src += n;
        }
    }
    __device__ void set(const uint32_t i, const float v) const
    {
        c[i] = v;
        uint32_t prev = 0u, base = padded, parent = i >> 1;
        for (uint32_t m = padded >> 1; base + parent < padded * 2u - 1u; m >>= 1) {
            c[base + parent] = __fadd_rn(c[prev + parent * 2u], c[prev + parent * 2u + 1u]);
            prev = base; base += m; parent >>= 1;
        }
    }
    __device__ uint32_t sample(const float value) const
    {
        uint32_t base = padded * 2u - 4u, node = 0;
        float v = value;
        for (uint32_t m = 2u; m < padded; m *= 2u) {
            const float l = c[base + node], r = c[base + node + 1u];
            const float s = __fadd_rn(l, r);
            if (s == 0.0f) node *= 2u;
            else {
                const float vs = __fmul_rn(v, s);
                if (vs < l || r == 0.0f) { node = node * 2u; const float q = __fdiv_rn(vs, l); v = q < 1.0f ? q : 1.0f; }
                else { node = (node + 1u) * 2u; const float q = __fdiv_rn(__fsub_rn(vs, l), r); v = q < 1.0f ? q : 1.0f; }
            }
            base -= m * 2u;
        }
        const float l = node < size ? c[node] : 0.0f, r = node + 1u < size ? c[node + 1u] : 0.0f;
        const float vs = __fmul_rn(v, __fadd_rn(l, r));
        node = (vs < l || r == 0.0f) ? node : node + 1u;
        return node < size ? node : size - 1u;
    }
};

__device__ __forceinline__ uint32_t hit_delta(const uint2 h) { return h.y & 0xFFFFFu; }
__device__ __forceinline__ uint32_t hit_pop_front(uint2* h)              // SeedHit::pop_front (seed_hit.h:136-142)
{
    uint2 v = *h;
    const uint32_t r = v.x;
    v.x = r + 1u;
    v.y = (v.y & ~0xFFFFFu) | ((v.y - 1u) & 0xFFFFFu);
    *h = v;
    return r;
}
__device__ __forceinline__ uint32_t packed_seed_of(const uint2 h, const uint32_t top_flag)      // defs.h:171-181
{
    return ((h.y >> 20) & 0x3FFu) | (((h.y >> 31) & 1u) << 12) | (((h.y >> 30) & 1u) << 13) | ((top_flag & 1u) << 14);
}

__global__ void __launch_bounds__(256)
select_init_kernel(uint32_t n_reads, const char* __restrict__ names, const uint32_t* __restrict__ names_idx,
                   const uint2* __restrict__ hits, uint32_t hits_stride, const uint32_t* __restrict__ counts,
                   float* __restrict__ probs, uint32_t probs_stride, uint32_t* __restrict__ trys, uint32_t* __restrict__ rseeds,
                   uint32_t max_effort_init, int randomized, int top_seed, int build_tree, const uint32_t* __restrict__ queue)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_reads) return;
    const uint32_t r = queue ? queue[t] : t;          // queued form: only the reads of the seeding pass's queue
    if (trys) trys[r] = max_effort_init;
    if (!randomized) return;
    if (names) {
        const uint32_t off = names_idx[r], len = names_idx[r + 1u] - off;
        uint32_t hash = 5381u;
        for (uint32_t i = 0; i < len && names[off + i]; ++i) hash = ((hash << 5) + hash) ^ uint32_t(int32_t((signed char)names[off + i]));
        rseeds[r] = hash;
    }
    if (!build_tree) return;
    const uint32_t n = counts[r];
    if (n == 0u) return;
    float* pr = probs + uint64_t(r) * probs_stride;
    const uint2* h = hits + uint64_t(r) * hits_stride;
    for (uint32_t i = 0; i < n; ++i) { const float d = __uint2float_rn(hit_delta(h[i])); pr[i] = __fdiv_rn(1.0f, __fmul_rn(d, d)); }
    if (top_seed) pr[0] = 0.0f;              // the leaves are the tree (see RowCells / LdsCells above)
}

// A read's hit row and probability tree held by a group of G lanes, four hits / leaves per lane (G = 4: rows of <= 16 hits, G = 8: <= 32).
// Every internal node of a SumTree is the float sum of its two children, left + right (setup() builds it so, set() keeps it so), so the
// node values are a function of the leaves: blocks of 2 and 4 leaves are sums inside the lane, blocks of 8, 16 and 32 a butterfly over the
// group -- the same pairwise adds, the same values bit for bit (float add commutes).  A lane then holds every ancestor of its own leaves.
template <int G>
struct TreeQuad
{
    float lf[4];                    // leaves 4 j .. 4 j + 3
    float a, b, c, d, e, f;         // the blocks of 2 (left, right), 4, 8, 16, 32 leaves this lane belongs to
    __device__ __forceinline__ void rebuild()
    {
        a = __fadd_rn(lf[0], lf[1]); b = __fadd_rn(lf[2], lf[3]); c = __fadd_rn(a, b);
        d = __fadd_rn(c, __shfl_xor(c, 1, G));
        e = __fadd_rn(d, __shfl_xor(d, 2, G));
        f = G > 4 ? __fadd_rn(e, __shfl_xor(e, 4, G)) : 0.0f;
    }
    // the node of level k (blocks of 2^k leaves) that starts at leaf `first` (a multiple of 2^k, the same in every lane of the group)
    __device__ __forceinline__ float node(const uint32_t k, const uint32_t first) const
    {
        const uint32_t q = first & 3u;
        float v = q == 0u ? lf[0] : q == 1u ? lf[1] : q == 2u ? lf[2] : lf[3];
        v = k == 1u ? (q ? b : a) : v;
        v = k == 2u ? c : v; v = k == 3u ? d : v; v = k == 4u ? e : v; v = k == 5u ? f : v;
        return __shfl(v, int(first >> 2), G);
    }
    // this lane's own ancestor of level k >= 1 above its leaf slot q
    __device__ __forceinline__ float own(const uint32_t k, const uint32_t q) const
    {
        return k == 1u ? (q >= 2u ? b : a) : k == 2u ? c : k == 3u ? d : k == 4u ? e : f;
    }
};

// The leaves of select_init, four per lane: leaf = 1 / delta^2 (0 for the first hit under top_seed).  A read's hit row is read and its leaves
// written in contiguous pieces -- where one lane per read walks 16 scattered 8-byte loads and 16 scattered 4-byte stores.
template <int G>
__global__ void __launch_bounds__(256)
select_init_tree_kernel(uint32_t n_reads, const uint2* __restrict__ hits, uint32_t hits_stride, const uint32_t* __restrict__ counts,
                        float* __restrict__ probs, uint32_t probs_stride, int top_seed, const uint32_t* __restrict__ queue)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t slot = t / G, j = t % G;
    if (slot >= n_reads) return;
    const uint32_t r = queue ? queue[slot] : slot;
    const uint32_t n = counts[r];
    if (4u * j >= n) return;
    float* pr = probs + uint64_t(r) * probs_stride;
    const uint2* hrow = hits + uint64_t(r) * hits_stride;
    float v[4];
    #pragma unroll
    for (uint32_t q = 0; q < 4u; ++q)
    {
        const uint32_t i = 4u * j + q;
        v[q] = 0.0f;
        if (i < n) { const float dl = __uint2float_rn(hit_delta(hrow[i])); v[q] = __fdiv_rn(1.0f, __fmul_rn(dl, dl)); }
        if (top_seed && i == 0u) v[q] = 0.0f;
    }
    if ((probs_stride & 3u) == 0u && 4u * j + 4u <= probs_stride) *reinterpret_cast<float4*>(pr + 4u * j) = make_float4(v[0], v[1], v[2], v[3]);
    else { for (uint32_t q = 0; q < 4u; ++q) if (4u * j + q < n) pr[4u * j + q] = v[q]; }
}

template <typename Tree>
__device__ uint32_t randomized_select(const Tree& tree, const uint2* h, uint32_t* rseed)
{
    uint32_t s = *rseed;
    uint32_t pick = 0u;
    bool found = false;
    for (uint32_t i = 0; i < 10u && !found; ++i) {
        s = 1664525u * s + 1013904223u;
        const float rf = __fdiv_rn(__uint2float_rn(s), 4294967296.0f);          // float(0xFFFFFFFFu) rounds to 2^32
        const uint32_t id = tree.sample(rf);
        if (hit_delta(h[id]) != 0u) { pick = id; found = true; }
    }
    *rseed = s;
    return pick;
}

// ---- make() by table.  HitDequeT::make() is a comparison sort-like procedure: what it does to a row depends only on how the hits' range sizes
// compare, never on their values.  A row whose range sizes take at most TWO distinct values -- the common case: a read's seeds are unique in the
// genome (all sizes 1), then some of them are used up (size 0) -- is therefore one of 2^n patterns ("which hits hold the larger value"), and the
// arrangement make() leaves is a fixed permutation per (n, pattern).  The table holds it for every n <= 16: entry (2^n - 2 + pattern) = sixteen
// nibbles, nibble k = the slot whose hit make() moves to slot k.  Built once per device by running make() itself over every pattern (1 MB), so a
// looked-up arrangement IS make()'s.  A selection call then replaces the ~n log n dependent LDS exchanges of the construction by one 8-byte load
// and the moves it names (usually none).  Rows with three or more distinct sizes, or more than 16 hits, run make() as before.
struct LocalKeys
{
    typedef uint32_t value_type;
    uint32_t* p;
    __device__ __forceinline__ uint32_t& operator[](const int i) const { return p[i]; }
    __device__ __forceinline__ static bool before(const uint32_t f, const uint32_t s) { return (f >> 8) > (s >> 8); }
};
__global__ void __launch_bounds__(256)
select_make_table_kernel(uint64_t* __restrict__ table)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= (1u << 17) - 2u) return;
    const uint32_t n = 31u - uint32_t(__clz(int(t + 2u)));          // 2^n - 2 <= t < 2^(n+1) - 2
    const uint32_t pattern = t - ((1u << n) - 2u);
    uint32_t keys[16];
    for (uint32_t i = 0; i < 16u; ++i) keys[i] = (((pattern >> i) & 1u) << 8) | i;
    const HitDequeT<LocalKeys> d = { { keys } };
    d.make(int(n));
    uint64_t perm = 0;
    for (uint32_t k = 0; k < 16u; ++k) perm |= uint64_t(k < n ? (keys[k] & 255u) : k) << (4u * k);
    table[t] = perm;
}

// stage 1: every active read makes its picks into its own staging slots.
// PADDED (randomized only): 0 = the tree's nodes in the read's row of global memory (wide rows), 16 / 32 = in LDS, rebuilt from the leaves
template <bool RANDOMIZED, int PADDED>
__global__ void __launch_bounds__(256)
select_kernel(uint32_t n_multi, const uint32_t* __restrict__ active_in, uint32_t n_active,
              uint2* __restrict__ hits, uint32_t hits_stride, uint32_t* __restrict__ counts,
              float* __restrict__ probs, uint32_t probs_stride, uint32_t* __restrict__ rseeds, const uint32_t* __restrict__ trys,
              uint32_t* __restrict__ stage_read, uint32_t* __restrict__ stage_loc, uint32_t* __restrict__ stage_seed, uint64_t* __restrict__ key,
              const uint64_t* __restrict__ make_table)
{
    __shared__ float s_tree[(RANDOMIZED && PADDED > 0) ? (2 * PADDED - 1) * 256 : 1];
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t > n_active) return;
    if (t == n_active) { key[t] = 0ull; return; }
    const uint32_t read_id = active_in[t] & 0x7FFFFFFFu;
    uint32_t top_flag = active_in[t] >> 31;
    uint32_t n_sel = 0u;
    uint32_t n = (trys[read_id] == 0u) ? 0u : counts[read_id];                  // SelectBestApproxContext::stop, empty deque
    if (n != 0u)
    {
        uint2* h = hits + uint64_t(read_id) * hits_stride;
        uint32_t* out_loc = stage_loc + uint64_t(t) * n_multi;
        uint32_t* out_seed = stage_seed + uint64_t(t) * n_multi;
        const HitDeque deque = { h };
        // hits[ read_id ] rebuilds the heap (hit_deque.h: make()).  For rows that fit the tree's LDS cells (not yet loaded) the construction never
        // touches memory exchange by exchange:
        //   * a row of <= 16 hits is read ONCE into registers (eight 16-byte loads, independent of each other);
        //   * its arrangement is looked up (at most two distinct range sizes: select_make_table_kernel) or built by make() on one key per hit in LDS;
        //   * the arrangement is a permutation of <= 16 slots held in one 64-bit register; the row goes through the LDS cells once (x words, then
        //     y words) and comes back permuted, and only the slots that changed are stored -- independent stores.
        // Round 5 moved the hits in global memory cycle by cycle -- h[j] = h[src] with the next step waiting for the store: a chain of ~n memory
        // round trips per read, and with equal range sizes (a read whose seeds are unique) make() moves every hit at every round.  That chain, not
        // the construction, was the stage's 12.5 ms per 10 M reads (the table alone changed nothing).  Rows of 17 .. 32 hits keep the old form.
        // rows of <= RMAX = 16 hits live in registers from here to the end of the call: hx / hy, `dirty` = the slots to store back
        constexpr uint32_t RMAX = 16u;       // (32 registers of row per lane; at 32 hits per row the 32-way selects cost more than the loads they replace: measured, profiles/r06/README.md)
        uint32_t hx[RMAX], hy[RMAX];
        uint32_t dirty = 0u;
        bool in_regs = false;
        if constexpr (PADDED > 0)
        {
            uint32_t* cell = reinterpret_cast<uint32_t*>(s_tree) + threadIdx.x;          // this lane's cells: cell[i * 256]
            const LdsKeys keys = { cell };
            if (n <= RMAX && (hits_stride & 1u) == 0u)
            {
                in_regs = true;
                #pragma unroll
                for (uint32_t i = 0; i < RMAX; i += 2u)
                {
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (i < n) v = *reinterpret_cast<const uint4*>(h + i);
                    hx[i] = v.x; hy[i] = v.y; hx[i + 1u] = v.z; hy[i + 1u] = v.w;
                }
                // at most two distinct range sizes (and a row the table covers)?
                uint32_t dmin = 0xFFFFFFFFu, dmax = 0u;
                #pragma unroll
                for (uint32_t i = 0; i < RMAX; ++i)
                    if (i < n) { const uint32_t d = hy[i] & 0xFFFFFu; dmin = d < dmin ? d : dmin; dmax = d > dmax ? d : dmax; }
                uint32_t pattern = 0u; bool two = make_table != nullptr && n <= 16u;
                #pragma unroll
                for (uint32_t i = 0; i < 16u && i < RMAX; ++i)
                    if (i < n) { const uint32_t d = hy[i] & 0xFFFFFu; two = two && (d == dmin || d == dmax); pattern |= ((d == dmax && dmax != dmin) ? 1u : 0u) << i; }
                uint32_t src[RMAX];                                                  // slot k takes what was in slot src[k]
                bool moved = false;
                if (two)
                {
                    const uint64_t perm = make_table[((1u << n) - 2u) + pattern];
                    #pragma unroll
                    for (uint32_t k = 0; k < RMAX; ++k) { src[k] = (k < 16u && k < n) ? (uint32_t(perm >> (4u * (k & 15u))) & 15u) : k; moved = moved || src[k] != k; }
                }
                else
                {
                    #pragma unroll
                    for (uint32_t i = 0; i < RMAX; ++i) if (i < n) keys[int(i)] = ((hy[i] & 0xFFFFFu) << 8) | i;
                    const HitDequeT<LdsKeys> kd = { keys };
                    kd.make(int(n));
                    #pragma unroll
                    for (uint32_t k = 0; k < RMAX; ++k) { src[k] = k < n ? (keys[int(k)] & 255u) : k; moved = moved || src[k] != k; }
                }
                if (moved)
                {
                    uint32_t nx[RMAX], ny[RMAX];
                    #pragma unroll
                    for (uint32_t i = 0; i < RMAX; ++i) if (i < n) cell[i * 256u] = hx[i];
                    #pragma unroll
                    for (uint32_t k = 0; k < RMAX; ++k) nx[k] = k < n ? cell[src[k] * 256u] : 0u;
                    #pragma unroll
                    for (uint32_t i = 0; i < RMAX; ++i) if (i < n) cell[i * 256u] = hy[i];
                    #pragma unroll
                    for (uint32_t k = 0; k < RMAX; ++k) ny[k] = k < n ? cell[src[k] * 256u] : 0u;
                    #pragma unroll
                    for (uint32_t k = 0; k < RMAX; ++k)
                        if (k < n && src[k] != k) { hx[k] = nx[k]; hy[k] = ny[k]; dirty |= 1u << k; }
                }
            }
            else
            {
            if ((hits_stride & 1u) == 0u)
                for (uint32_t i = 0; i < n; i += 2u)
                {
                    const uint4 v = *reinterpret_cast<const uint4*>(h + i);
                    keys[int(i)] = ((v.y & 0xFFFFFu) << 8) | i;
                    if (i + 1u < n) keys[int(i + 1u)] = ((v.w & 0xFFFFFu) << 8) | (i + 1u);
                }
            else
                for (uint32_t i = 0; i < n; ++i) keys[int(i)] = (hit_delta(h[i]) << 8) | i;
            const HitDequeT<LdsKeys> kd = { keys };
            kd.make(int(n));
            for (uint32_t k = 0; k < n; ++k)                                    // slot k now holds what was in slot keys[k] & 255
            {
                if ((keys[int(k)] & 255u) == k) continue;
                const uint2 first = h[k];
                uint32_t j = k;
                for (;;)
                {
                    const uint32_t src = keys[int(j)] & 255u;
                    keys[int(j)] = (keys[int(j)] & ~255u) | j;                  // placed
                    if (src == k) { h[j] = first; break; }
                    h[j] = h[src];
                    j = src;
                }
            }
            }
        }
        else deque.make(int(n));
        if constexpr (!RANDOMIZED)
        {
            for (uint32_t i = 0; i < n_multi; ++i)
            {
                uint32_t top = uint32_t(deque.top(int(n)));
                if (hit_delta(h[top]) == 0u) {                                  // the top range ran out: next one
                    deque.pop_top(int(n)); --n;
                    if (n == 0u) break;
                    top = uint32_t(deque.top(int(n)));
                    top_flag = 0u;
                }
                out_loc[n_sel] = hit_pop_front(&h[top]);
                out_seed[n_sel] = packed_seed_of(h[top], top_flag);
                ++n_sel;
            }
            counts[read_id] = n;                                                // ~SeedHitDequeReference
        }
        else
        {
            float* leaves = probs + uint64_t(read_id) * probs_stride;
            auto picks = [&](const auto& tree)
            {
                for (uint32_t i = 0; i < n_multi; ++i)
                {
                    if (tree.sum() <= 0.0f) break;                              // (erase() here is undone by the reference's destructor)
                    if (top_flag && hit_delta(h[0]) == 0u) top_flag = 0u;
                    const uint32_t id = top_flag ? 0u : randomized_select(tree, h, &rseeds[read_id]);
                    if (hit_delta(h[id]) == 0u) { if (n_multi > 1u) continue; else break; }
                    out_loc[n_sel] = hit_pop_front(&h[id]);
                    if (hit_delta(h[id]) == 0u) { tree.set(id, 0.0f); leaves[id] = 0.0f; }     // on chip: the leaf and its ancestors; in memory: the leaf
                    out_seed[n_sel] = packed_seed_of(h[id], top_flag);
                    ++n_sel;
                }
            };
            if constexpr (PADDED > 0)
            {
                const LdsCells cells = { s_tree + threadIdx.x };
                const SumTreeT<LdsCells> tree(cells, n);
                const uint32_t n4 = (n + 3u) & ~3u;
                if ((probs_stride & 3u) == 0u && n4 <= probs_stride)
                    for (uint32_t i = 0; i < n4; i += 4u)
                    {
                        const float4 v = *reinterpret_cast<const float4*>(leaves + i);
                        cells[i] = v.x; cells[i + 1u] = i + 1u < n ? v.y : 0.0f; cells[i + 2u] = i + 2u < n ? v.z : 0.0f; cells[i + 3u] = i + 3u < n ? v.w : 0.0f;
                    }
                else
                    for (uint32_t i = 0; i < n; ++i) cells[i] = leaves[i];
                tree.setup();
                if (!in_regs) picks(tree);
                else
                {
                    // the same picks on the row in registers: a hit is looked up / popped by an RMAX-way select instead of a load of h[id] -- with several
                    // picks per read (the multi-hit rounds) those loads missed L2 again and again (128 fabric reads per read at 20 picks per read,
                    // profiles/r06/pmc_select.txt); the LCG state stays in a register; the slots that changed are stored once, at the end
                    auto y_at = [&](const uint32_t id) { uint32_t r = 0u;
                        #pragma unroll
                        for (uint32_t i = 0; i < RMAX; ++i) r = (i == id) ? hy[i] : r;
                        return r; };
                    auto x_at = [&](const uint32_t id) { uint32_t r = 0u;
                        #pragma unroll
                        for (uint32_t i = 0; i < RMAX; ++i) r = (i == id) ? hx[i] : r;
                        return r; };
                    uint32_t sd = rseeds[read_id];
                    const uint32_t sd_in = sd;
                    bool drew = false;
                    for (uint32_t i = 0; i < n_multi; ++i)
                    {
                        if (tree.sum() <= 0.0f) break;
                        if (top_flag && (hy[0] & 0xFFFFFu) == 0u) top_flag = 0u;
                        uint32_t id = 0u;
                        if (!top_flag)
                        {
                            bool found = false;                                     // randomized_select
                            for (uint32_t tr = 0; tr < 10u && !found; ++tr) {
                                sd = 1664525u * sd + 1013904223u;
                                const float rf = __fdiv_rn(__uint2float_rn(sd), 4294967296.0f);
                                const uint32_t cand = tree.sample(rf);
                                if ((y_at(cand) & 0xFFFFFu) != 0u) { id = cand; found = true; }
                            }
                            drew = true;
                        }
                        const uint32_t y = y_at(id);
                        if ((y & 0xFFFFFu) == 0u) { if (n_multi > 1u) continue; else break; }
                        const uint32_t x = x_at(id);
                        const uint32_t y1 = (y & ~0xFFFFFu) | ((y - 1u) & 0xFFFFFu);            // pop_front
                        #pragma unroll
                        for (uint32_t k = 0; k < RMAX; ++k) if (k == id) { hx[k] = x + 1u; hy[k] = y1; }
                        dirty |= 1u << id;
                        out_loc[n_sel] = x;
                        if ((y1 & 0xFFFFFu) == 0u) { tree.set(id, 0.0f); leaves[id] = 0.0f; }
                        out_seed[n_sel] = packed_seed_of(make_uint2(x + 1u, y1), top_flag);
                        ++n_sel;
                    }
                    if (drew || sd != sd_in) rseeds[read_id] = sd;
                }
            }
            else
            {
                const RowCells cells = { leaves };
                const SumTreeT<RowCells> tree(cells, n);
                tree.setup();                                                    // the row's internal nodes are scratch: rebuilt on entry
                picks(tree);
            }
        }
        if (in_regs && dirty != 0u)
        {
            #pragma unroll
            for (uint32_t k = 0; k < RMAX; ++k) if ((dirty >> k) & 1u) h[k] = make_uint2(hx[k], hy[k]);
        }
    }
    stage_read[t] = read_id | (top_flag << 31);
    key[t] = (uint64_t(n_sel != 0u ? 1u : 0u) << 32) | n_sel;
}

// Randomized selection with a read's hit row and tree in a group of G lanes, four hits / leaves per lane (TreeQuad).  One lane per read
// walks its tree and row with ~25 scattered 4/8-byte accesses per pick; here the row and the LEAVES are loaded once in contiguous pieces and
// everything else happens in registers:
//   * SumTree::sample()'s descent (sum_tree_inl.h:120-178) runs redundantly in the lanes of the group, fetching the two children of the
//     current node by shuffle; the LCG, the float multiply / divide / min sequence and the 10-try loop of randomized_select are unchanged;
//   * pop_front touches one slot of the lane that owns the picked hit; an exhausted hit zeroes that lane's leaf, and the lane stores the
//     hit, the leaf, the leaf's ancestors (all its own values) and the staged pick.
// Same picks, same state, bit for bit (tests/test_select_gpu.py against the oracle's one-lane restatement; NVBIO_HIP_SELECT_LANES=1 runs
// the one-lane form, =4 this one).  NOT the default: on config 4 it is slower than one lane per read (586 vs 414 us per call), and so was
// a form with sixteen lanes per read and one leaf each (536 us): the stage is bound by moving each active read's two rows in and out of
// HBM, not by the number of requests (profiles/r03/select_coop.txt).
template <int G>
__global__ void __launch_bounds__(256)
select_rand_quad_kernel(uint32_t n_multi, const uint32_t* __restrict__ active_in, uint32_t n_active,
                        uint2* __restrict__ hits, uint32_t hits_stride, uint32_t* __restrict__ counts,
                        float* __restrict__ probs, uint32_t probs_stride, uint32_t* __restrict__ rseeds, const uint32_t* __restrict__ trys,
                        uint32_t* __restrict__ stage_read, uint32_t* __restrict__ stage_loc, uint32_t* __restrict__ stage_seed, uint64_t* __restrict__ key)
{
    const uint32_t gt = blockIdx.x * 256u + threadIdx.x;
    const uint32_t t = gt / G, j = gt % G;
    if (t > n_active) return;
    if (t == n_active) { if (j == 0u) key[t] = 0ull; return; }
    const uint32_t read_id = active_in[t] & 0x7FFFFFFFu;
    uint32_t top_flag = active_in[t] >> 31;
    uint32_t n_sel = 0u;
    const uint32_t n = (trys[read_id] == 0u) ? 0u : counts[read_id];
    if (n != 0u)
    {
        uint2* hrow = hits + uint64_t(read_id) * hits_stride;
        float* pr = probs + uint64_t(read_id) * probs_stride;
        const uint32_t padded = st_padded(n), lg = ilog2(padded);
        if (j == 0u) { const HitDeque deque = { hrow }; deque.make(int(n)); }          // hits[ read_id ] rebuilds the heap; the group's lanes
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                         // are in one wavefront and read the row after it
        uint32_t hx[4], hy[4];
        TreeQuad<G> tq;
        #pragma unroll
        for (uint32_t q = 0; q < 4u; ++q)
        {
            const uint32_t i = 4u * j + q;
            const uint2 h = (i < n) ? hrow[i] : make_uint2(0u, 0u);
            hx[q] = h.x; hy[q] = h.y;
            tq.lf[q] = (i < n) ? pr[i] : 0.0f;
        }
        // delta of hit i (the same i in every lane of the group)
        auto delta_of = [&](const uint32_t i) -> uint32_t {
            const uint32_t q = i & 3u;
            const uint32_t y = q == 0u ? hy[0] : q == 1u ? hy[1] : q == 2u ? hy[2] : hy[3];
            return uint32_t(__shfl(int32_t(y & 0xFFFFFu), int(i >> 2), G));
        };
        uint32_t s = rseeds[read_id];
        const uint32_t s_in = s;
        tq.rebuild();
        for (uint32_t i = 0; i < n_multi; ++i)
        {
            if (tq.node(lg, 0u) <= 0.0f) break;                                        // tree.sum(): the root
            if (top_flag && delta_of(0u) == 0u) top_flag = 0u;
            uint32_t id = 0u;
            if (!top_flag)
            {
                // randomized_select: up to 10 draws, the first whose hit still has rows
                bool found = false;
                for (uint32_t tr = 0; tr < 10u && !found; ++tr)
                {
                    s = 1664525u * s + 1013904223u;
                    float v = __fdiv_rn(__uint2float_rn(s), 4294967296.0f);
                    // SumTree::sample: from the two children of the root down to a pair of leaves
                    uint32_t nd = 0u;                                                  // the left node of the current pair, index in its level
                    for (uint32_t k = lg; k >= 2u; --k)                                // nodes of this level cover 2^(k-1) leaves
                    {
                        const uint32_t B = 1u << (k - 1u);
                        const float l = tq.node(k - 1u, nd * B), r = tq.node(k - 1u, (nd + 1u) * B);
                        const float ss = __fadd_rn(l, r);
                        if (ss == 0.0f) nd *= 2u;
                        else {
                            const float vs = __fmul_rn(v, ss);
                            if (vs < l || r == 0.0f) { nd = nd * 2u; const float qq = __fdiv_rn(vs, l); v = qq < 1.0f ? qq : 1.0f; }
                            else { nd = (nd + 1u) * 2u; const float qq = __fdiv_rn(__fsub_rn(vs, l), r); v = qq < 1.0f ? qq : 1.0f; }
                        }
                    }
                    const float l = nd < n ? tq.node(0u, nd) : 0.0f, r = nd + 1u < n ? tq.node(0u, nd + 1u) : 0.0f;
                    const float vs = __fmul_rn(v, __fadd_rn(l, r));
                    nd = (vs < l || r == 0.0f) ? nd : nd + 1u;
                    const uint32_t pick = nd < n ? nd : n - 1u;
                    if (delta_of(pick) != 0u) { id = pick; found = true; }
                }
            }
            const uint32_t dsel = delta_of(id);
            if (dsel == 0u) { if (n_multi > 1u) continue; else break; }
            if (j == (id >> 2))                                                        // the lane that owns the picked hit
            {
                const uint32_t q = id & 3u;
                const uint32_t x = q == 0u ? hx[0] : q == 1u ? hx[1] : q == 2u ? hx[2] : hx[3];
                uint32_t y = q == 0u ? hy[0] : q == 1u ? hy[1] : q == 2u ? hy[2] : hy[3];
                y = (y & ~0xFFFFFu) | ((y - 1u) & 0xFFFFFu);                             // pop_front
                #pragma unroll
                for (uint32_t w = 0; w < 4u; ++w) if (w == q) { hx[w] = x + 1u; hy[w] = y; }
                hrow[id] = make_uint2(x + 1u, y);
                stage_loc[uint64_t(t) * n_multi + n_sel] = x;
                stage_seed[uint64_t(t) * n_multi + n_sel] = packed_seed_of(make_uint2(0u, y), top_flag);
                if (dsel == 1u)                                                        // the hit ran out: tree.set(id, 0), leaf first
                {
                    #pragma unroll
                    for (uint32_t w = 0; w < 4u; ++w) if (w == q) tq.lf[w] = 0.0f;
                    pr[id] = 0.0f;
                }
            }
            if (dsel == 1u) tq.rebuild();                                              // ... its ancestors exist in registers only
            ++n_sel;
        }
        if (j == 0u && s != s_in) rseeds[read_id] = s;
    }
    if (j == 0u)
    {
        stage_read[t] = read_id | (top_flag << 31);
        key[t] = (uint64_t(n_sel != 0u ? 1u : 0u) << 32) | n_sel;
    }
}

// stage 2: compaction in queue order
__global__ void __launch_bounds__(256)
select_compact_kernel(uint32_t n_multi, uint32_t n_active, const uint64_t* __restrict__ key, const uint64_t* __restrict__ off,
                      const uint32_t* __restrict__ stage_read, const uint32_t* __restrict__ stage_loc, const uint32_t* __restrict__ stage_seed,
                      uint32_t* __restrict__ active_out, uint64_t* __restrict__ hit_begin, uint32_t* __restrict__ hit_read_id,
                      uint32_t* __restrict__ hit_loc, uint32_t* __restrict__ hit_seed, uint32_t* __restrict__ out_sizes)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t > n_active) return;
    const uint64_t o = off[t];
    if (t == n_active) {
        hit_begin[uint32_t(o >> 32)] = o & 0xFFFFFFFFull;
        out_sizes[0] = uint32_t(o >> 32); out_sizes[1] = uint32_t(o);
        return;
    }
    const uint32_t n_sel = uint32_t(key[t]);
    if (n_sel == 0u) return;
    const uint32_t slot = uint32_t(o >> 32), hb = uint32_t(o);
    active_out[slot] = stage_read[t];
    hit_begin[slot] = hb;
    const uint32_t read_id = stage_read[t] & 0x7FFFFFFFu;
    for (uint32_t i = 0; i < n_sel; ++i) {
        hit_read_id[hb + i] = read_id;
        hit_loc[hb + i] = stage_loc[uint64_t(t) * n_multi + i];
        hit_seed[hb + i] = stage_seed[uint64_t(t) * n_multi + i];
    }
}

__global__ void __launch_bounds__(256)
locate_hits_kernel(const Fmi f, const Fmi rf, uint32_t n, uint32_t* __restrict__ hit_loc, const uint32_t* __restrict__ hit_seed)
{
    // grid-stride: see map_exact_kernel
    for (uint64_t i = uint64_t(blockIdx.x) * 256u + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256u)
    {
        const uint32_t seed = hit_seed[i], dir = (seed >> 12) & 1u, pir = seed & 0xFFFu;
        // (two code paths, not `const Fmi& x = dir ? rf : f`: a reference picked per lane makes the compiler park both index descriptors in
        // scratch memory -- 296 bytes written and re-read per lane, three times the kernel's useful traffic)
        uint32_t g;
        if (dir) { const uint2 it = fm_locate_it(rf, hit_loc[i]); g = rf.length - 1u - (rf.ssa[it.x / rf.sa_int] + it.y); }
        else     { const uint2 it = fm_locate_it(f,  hit_loc[i]); g = f.ssa[it.x / f.sa_int] + it.y; }
        hit_loc[i] = g - pir;
    }
}

constexpr uint32_t SETUP_TILES = 4u;            // tiles of 256 hits a block of score_best_setup_kernel describes
__global__ void __launch_bounds__(256)
score_best_setup_kernel(uint32_t n, const uint32_t* __restrict__ hit_read_id, const uint32_t* __restrict__ hit_loc, const uint32_t* __restrict__ hit_seed,
                        const uint64_t* __restrict__ read_begin, const uint32_t* __restrict__ read_len, uint32_t fixed_len, uint64_t rc_offset,
                        uint32_t band_len, uint32_t genome_len, const uint2* __restrict__ best, uint32_t best_stride, int32_t score_limit,
                        uint64_t* __restrict__ pat_begin, uint32_t* __restrict__ pat_len,
                        uint64_t* __restrict__ text_begin, uint32_t* __restrict__ text_len, int32_t* __restrict__ min_score, int32_t* __restrict__ known_score,
                        uint32_t* __restrict__ job_count, uint32_t* __restrict__ job_hit)
{
    // a block describes SETUP_TILES x 256 hits: the compacted form below takes ONE slot range per block from the job counter, and with a block per
    // 256 hits a round of 650 k hits queued 2 500 atomics on that one address -- most of the kernel's 38 us (profiles/r06); four tiles per block
    // queue a quarter of them
    __shared__ uint32_t wave_need[SETUP_TILES][4], block_base;
    uint32_t r_[SETUP_TILES], seed_[SETUP_TILES], len_[SETUP_TILES], gb_[SETUP_TILES], ge_[SETUP_TILES], w2_[SETUP_TILES];
    int32_t  known_[SETUP_TILES];
    bool     live_[SETUP_TILES];
    #pragma unroll
    for (uint32_t k = 0; k < SETUP_TILES; ++k)
    {
        const uint32_t i = (blockIdx.x * SETUP_TILES + k) * 256u + threadIdx.x;
        const bool live = i < n;
        const uint32_t r = live ? hit_read_id[i] : 0u, g = live ? hit_loc[i] : 0u;
        const uint32_t len = live ? (read_len ? read_len[r] : fixed_len) : 0u;
        const uint32_t gb = g > band_len / 2u ? g - band_len / 2u : 0u;
        const uint32_t sum = gb + band_len + len;
        const uint32_t ge = sum < genome_len ? sum : genome_len;
        // A hit at a placement the read already recorded (same strand, same start) would be scored over the very same window again, and its
        // score is the recorded one: hand that score to the reduction instead (which usually skips the hit anyway, reduce_inl.h:111-114) and
        // give the job an empty window.  Most seeds of a read point at one placement, so this removes most of the extension work.
        int32_t known = INT32_MIN;
        const uint32_t seed = live ? hit_seed[i] : 0u;
        uint32_t w2 = 0u;
        if (live) {
            const uint2 a2 = best[r + best_stride];
            w2 = a2.x;
            if (known_score) {
                const uint32_t rc = (seed >> 13) & 1u;
                const uint2 a1 = best[r];
                if (((a1.x >> 28) & 1u) == rc && a1.y == g)      { const int32_t m = int32_t((a1.x >> 1) & 0x1FFFFu); known = (a1.x & 1u) ? -m : m; }
                else if (((a2.x >> 28) & 1u) == rc && a2.y == g) { const int32_t m = int32_t((a2.x >> 1) & 0x1FFFFu); known = (a2.x & 1u) ? -m : m; }
                known_score[i] = known;
            }
        }
        r_[k] = r; seed_[k] = seed; len_[k] = len; gb_[k] = gb; ge_[k] = ge; w2_[k] = w2; known_[k] = known; live_[k] = live;
    }
    // compacted form: only the hits that still need a DP become jobs, job_hit[slot] = the hit.  One atomic per BLOCK (the waves' counts
    // meet in LDS): one per wavefront made 156 k same-address atomics per 10 M hits, which is what the kernel then waited for.  The slot
    // order varies from run to run, the scores scattered back through job_hit do not.
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint64_t mask[SETUP_TILES];
    if (job_hit) {
        #pragma unroll
        for (uint32_t k = 0; k < SETUP_TILES; ++k) {
            mask[k] = __ballot(live_[k] && known_[k] == INT32_MIN);
            if (lane == 0u) wave_need[k][wv] = uint32_t(__popcll(mask[k]));
        }
        __syncthreads();
        if (threadIdx.x == 0u) {
            uint32_t tot = 0u;
            for (uint32_t k = 0; k < SETUP_TILES; ++k) tot += wave_need[k][0] + wave_need[k][1] + wave_need[k][2] + wave_need[k][3];
            block_base = tot ? atomicAdd(job_count, tot) : 0u;
        }
        __syncthreads();
    }
    uint32_t tile_base = job_hit ? block_base : 0u;
    #pragma unroll
    for (uint32_t k = 0; k < SETUP_TILES; ++k)
    {
        const uint32_t i = (blockIdx.x * SETUP_TILES + k) * 256u + threadIdx.x;
        uint32_t o = i;
        bool write = live_[k];
        if (job_hit) {
            const bool need = live_[k] && known_[k] == INT32_MIN;
            uint32_t base = tile_base;
            for (uint32_t q = 0; q < wv; ++q) base += wave_need[k][q];
            o = base + uint32_t(__popcll(mask[k] & ((1ull << lane) - 1ull)));
            write = need;
            if (need) job_hit[o] = i;
            tile_base += wave_need[k][0] + wave_need[k][1] + wave_need[k][2] + wave_need[k][3];
        }
        if (!write) continue;
        text_begin[o] = gb_[k];
        text_len[o] = (ge_[k] > gb_[k] && known_[k] == INT32_MIN) ? ge_[k] - gb_[k] : 0u;                // (a wrapped read start: empty window, the alignment fails)
        pat_begin[o] = (read_begin ? read_begin[r_[k]] : uint64_t(r_[k]) * fixed_len) + (((seed_[k] >> 13) & 1u) ? rc_offset : 0ull);
        if (pat_len) pat_len[o] = len_[k];
        const int32_t m2 = int32_t((w2_[k] >> 1) & 0x1FFFFu), s2 = (w2_[k] & 1u) ? -m2 : m2;
        min_score[o] = s2 > score_limit ? s2 : score_limit;
    }
}

// one lane per program of deque operations (0 push with the mappers' "full: pop_bottom first" rule, 1 pop_top,
// 2 pop_bottom, 3 shrink the range in slot val & 2^32-1 to val >> 32 in place, 4 rebuild the heap), the deque's array written out after every operation: lets a test replay the programs recorded
// from the reference's heap (tests/golden/hit_deque_vectors.npz) through the device implementation
__global__ void __launch_bounds__(64)
hit_deque_replay_kernel(uint32_t n_cases, const uint32_t* __restrict__ case_start, const uint8_t* __restrict__ ops, const uint64_t* __restrict__ vals,
                        const uint32_t* __restrict__ caps, const uint64_t* __restrict__ state_start, uint2* __restrict__ scratch, uint32_t scratch_stride,
                        uint64_t* __restrict__ out_states)
{
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    if (c >= n_cases) return;
    const HitDeque d = { scratch + uint64_t(c) * scratch_stride };
    int n = 0;
    uint64_t o = state_start[c];
    for (uint32_t i = case_start[c]; i < case_start[c + 1u]; ++i)
    {
        if (ops[i] == 0u) {
            if (uint32_t(n) == caps[i]) { d.pop_bottom(n); --n; }
            d.a[n++] = make_uint2(uint32_t(vals[i]), uint32_t(vals[i] >> 32));
            d.push(n);
        }
        else if (ops[i] == 1u) { d.pop_top(n); --n; }
        else if (ops[i] == 2u) { d.pop_bottom(n); --n; }
        else if (ops[i] == 3u) { uint2& w = d.a[uint32_t(vals[i])]; w.y = (w.y & ~0xFFFFFu) | (uint32_t(vals[i] >> 32) & 0xFFFFFu); }
        else d.make(n);
        for (int k = 0; k < n; ++k) out_states[o++] = (uint64_t(d.a[k].y) << 32) | d.a[k].x;
    }
}

static inline dim3 grid_for(uint64_t n) { return dim3(uint32_t((n + 255u) / 256u)); }
static inline uint64_t align256(uint64_t x) { return (x + 255ull) & ~255ull; }
static size_t select_scan_bytes(uint32_t n)
{
    size_t scan = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (const uint64_t*)nullptr, (uint64_t*)nullptr, int(n) + 1);    // size query only
    return scan;
}

} // namespace nvb

using namespace nvb;

NVB_API uint32_t nvbio_hip_sum_tree_node_count(uint32_t size)
{
    uint32_t l = 0; for (uint32_t n = size; n > 1u; n >>= 1) ++l;
    const uint32_t padded = (1u << l) < size ? 1u << (l + 1u) : 1u << l;
    return padded * 2u - 1u;
}

// floats a read's row of `probs` needs: its leaves (one per hit slot); wide rows (> 32 slots) also hold the rebuilt sums, as scratch
static uint32_t min_probs_stride(const uint32_t hits_stride) { return hits_stride <= 32u ? hits_stride : nvbio_hip_sum_tree_node_count(hits_stride); }

static int select_init_impl(uint32_t n_reads, const uint32_t* queue, const char* read_names, const uint32_t* read_names_idx,
                            const uint64_t* hits, uint32_t hits_stride, const uint32_t* hit_counts,
                            float* probs, uint32_t probs_stride, uint32_t* trys, uint32_t* rseeds,
                            uint32_t max_effort_init, int32_t randomized, int32_t top_seed, void* stream)
{
    if (n_reads == 0) return hipSuccess;
    if (randomized) {
        if (!hits || !hit_counts || !probs || !rseeds || hits_stride == 0) return hipErrorInvalidValue;
        if (probs_stride < min_probs_stride(hits_stride)) return hipErrorInvalidValue;
        if (read_names && !read_names_idx) return hipErrorInvalidValue;
    }
    g_last_kernel = "select_init_kernel";
    // trees: 4 or 8 lanes per read (four leaves each) when a read's hit slots fit, one lane per read otherwise
    const bool one_lane = test_switch(SW_SELECT_LANES) == 1;             // NVBIO_HIP_SELECT_LANES = 1: one lane per read
    const int G = (!randomized || one_lane) ? 0 : hits_stride <= 16u ? 4 : hits_stride <= 32u ? 8 : 0;
    hipLaunchKernelGGL(select_init_kernel, grid_for(n_reads), dim3(256), 0, to_stream(stream), n_reads, read_names, read_names_idx,
                       reinterpret_cast<const uint2*>(hits), hits_stride, hit_counts, probs, probs_stride, trys, rseeds, max_effort_init,
                       int(randomized), int(top_seed), G == 0 ? 1 : 0, queue);
    if (G == 4) hipLaunchKernelGGL(select_init_tree_kernel<4>, grid_for(uint64_t(n_reads) * 4u), dim3(256), 0, to_stream(stream), n_reads,
                                    reinterpret_cast<const uint2*>(hits), hits_stride, hit_counts, probs, probs_stride, int(top_seed), queue);
    if (G == 8) hipLaunchKernelGGL(select_init_tree_kernel<8>, grid_for(uint64_t(n_reads) * 8u), dim3(256), 0, to_stream(stream), n_reads,
                                    reinterpret_cast<const uint2*>(hits), hits_stride, hit_counts, probs, probs_stride, int(top_seed), queue);
    return hipGetLastError();
}

NVB_API int nvbio_hip_select_init(uint32_t n_reads, const char* read_names, const uint32_t* read_names_idx,
                                  const uint64_t* hits, uint32_t hits_stride, const uint32_t* hit_counts,
                                  float* probs, uint32_t probs_stride, uint32_t* trys, uint32_t* rseeds,
                                  uint32_t max_effort_init, int32_t randomized, int32_t top_seed, void* stream)
{
    return select_init_impl(n_reads, nullptr, read_names, read_names_idx, hits, hits_stride, hit_counts, probs, probs_stride, trys, rseeds,
                            max_effort_init, randomized, top_seed, stream);
}
// The same for the reads of a queue only (queue[t] = read id): what a re-seeding pass needs -- every other read has no hits in this pass and
// is not selected from until the next select_init reaches it.  (The reference's kernel always runs over the whole batch, select.cu:36-103;
// the state it writes for reads outside the queue is never read.)
NVB_API int nvbio_hip_select_init_queued(uint32_t n_queue, const uint32_t* queue, const char* read_names, const uint32_t* read_names_idx,
                                         const uint64_t* hits, uint32_t hits_stride, const uint32_t* hit_counts,
                                         float* probs, uint32_t probs_stride, uint32_t* trys, uint32_t* rseeds,
                                         uint32_t max_effort_init, int32_t randomized, int32_t top_seed, void* stream)
{
    if (n_queue != 0 && !queue) return hipErrorInvalidValue;
    return select_init_impl(n_queue, queue, read_names, read_names_idx, hits, hits_stride, hit_counts, probs, probs_stride, trys, rseeds,
                            max_effort_init, randomized, top_seed, stream);
}

NVB_API uint64_t nvbio_hip_select_temp_bytes(uint32_t n_active, uint32_t n_multi)
{
    const uint64_t n = n_active, m = n_multi ? n_multi : 1u;
    return 2u * align256(n * m * 4u) + align256(n * 4u) + 2u * align256((n + 1u) * 8u) + align256(select_scan_bytes(n_active)) + 256u;
}

namespace nvb {
// the table of make() arrangements of the calling thread's device, built on first use (1 MB; never freed)
static const uint64_t* select_make_table(hipStream_t s)
{
    static std::mutex mtx;
    static uint64_t* tables[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mtx);
    if (!tables[dev])
    {
        uint64_t* t = nullptr;
        const uint32_t entries = (1u << 17) - 2u;
        if (hipMalloc(reinterpret_cast<void**>(&t), uint64_t(entries) * 8u) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        hipLaunchKernelGGL(select_make_table_kernel, dim3((entries + 255u) / 256u), dim3(256), 0, s, t);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(t); return nullptr; }   // every stream may use it from here on
        tables[dev] = t;
    }
    return tables[dev];
}
} // namespace nvb

NVB_API int nvbio_hip_select(int32_t randomized, uint32_t n_multi, const uint32_t* active_in, uint32_t n_active,
                             uint64_t* hits, uint32_t hits_stride, uint32_t* hit_counts,
                             float* probs, uint32_t probs_stride, uint32_t* rseeds, const uint32_t* trys,
                             uint32_t* active_out, uint64_t* hit_begin, uint32_t* hit_read_id, uint32_t* hit_loc, uint32_t* hit_seed,
                             uint32_t* out_sizes, void* temp, uint64_t temp_bytes, void* stream)
{
    if (!out_sizes || !hit_begin) return hipErrorInvalidValue;
    if (n_multi == 0 || n_multi > 4096u) return hipErrorInvalidValue;          // the reference encodes the per-read hit index in 12 bits
    if (n_active != 0 && (!active_in || !hits || !hit_counts || !trys || !active_out || !hit_read_id || !hit_loc || !hit_seed || hits_stride == 0))
        return hipErrorInvalidValue;
    if (randomized && n_active != 0 && (!probs || !rseeds || probs_stride < min_probs_stride(hits_stride))) return hipErrorInvalidValue;
    if (!temp || temp_bytes < nvbio_hip_select_temp_bytes(n_active, n_multi)) return hipErrorInvalidValue;
    const uint64_t n = n_active;
    uint8_t* p = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(temp) + 255u) & ~uintptr_t(255));
    uint32_t* stage_loc  = reinterpret_cast<uint32_t*>(p); p += align256(n * n_multi * 4u);
    uint32_t* stage_seed = reinterpret_cast<uint32_t*>(p); p += align256(n * n_multi * 4u);
    uint32_t* stage_read = reinterpret_cast<uint32_t*>(p); p += align256(n * 4u);
    uint64_t* key        = reinterpret_cast<uint64_t*>(p); p += align256((n + 1u) * 8u);
    uint64_t* off        = reinterpret_cast<uint64_t*>(p); p += align256((n + 1u) * 8u);
    size_t scan_bytes = select_scan_bytes(n_active);
    hipStream_t s = to_stream(stream);
    g_last_kernel = randomized ? "select_kernel<rand>" : "select_kernel";
    // One lane per read is the default: the stage moves every active read's hit row and tree row in and out of HBM once per call and
    // sits near that floor; the four-leaves-per-lane form (NVBIO_HIP_SELECT_LANES=4, same results) measured slower on config 4
    // (profiles/r03/select_coop.txt).
    const bool quad = randomized && hits_stride <= 32u && test_switch(SW_SELECT_LANES) == 4;
    // make() by table (see select_make_table_kernel): NVBIO_HIP_SELECT_LANES=2 runs without it (same results; the parity suite covers both)
    const uint64_t* table = (randomized && hits_stride <= 32u && test_switch(SW_SELECT_LANES) != 2) ? select_make_table(s) : nullptr;
    if (quad && hits_stride <= 16u)
        hipLaunchKernelGGL(select_rand_quad_kernel<4>, grid_for((uint64_t(n) + 1u) * 4u), dim3(256), 0, s, n_multi, active_in, n_active, reinterpret_cast<uint2*>(hits), hits_stride,
                           hit_counts, probs, probs_stride, rseeds, trys, stage_read, stage_loc, stage_seed, key);
    else if (quad)
        hipLaunchKernelGGL(select_rand_quad_kernel<8>, grid_for((uint64_t(n) + 1u) * 8u), dim3(256), 0, s, n_multi, active_in, n_active, reinterpret_cast<uint2*>(hits), hits_stride,
                           hit_counts, probs, probs_stride, rseeds, trys, stage_read, stage_loc, stage_seed, key);
    else if (randomized && hits_stride <= 16u)
        hipLaunchKernelGGL((select_kernel<true, 16>), grid_for(n + 1u), dim3(256), 0, s, n_multi, active_in, n_active, reinterpret_cast<uint2*>(hits), hits_stride,
                           hit_counts, probs, probs_stride, rseeds, trys, stage_read, stage_loc, stage_seed, key, table);
    else if (randomized && hits_stride <= 32u)
        hipLaunchKernelGGL((select_kernel<true, 32>), grid_for(n + 1u), dim3(256), 0, s, n_multi, active_in, n_active, reinterpret_cast<uint2*>(hits), hits_stride,
                           hit_counts, probs, probs_stride, rseeds, trys, stage_read, stage_loc, stage_seed, key, table);
    else if (randomized)
        hipLaunchKernelGGL((select_kernel<true, 0>), grid_for(n + 1u), dim3(256), 0, s, n_multi, active_in, n_active, reinterpret_cast<uint2*>(hits), hits_stride,
                           hit_counts, probs, probs_stride, rseeds, trys, stage_read, stage_loc, stage_seed, key, nullptr);
    else
        hipLaunchKernelGGL((select_kernel<false, 0>), grid_for(n + 1u), dim3(256), 0, s, n_multi, active_in, n_active, reinterpret_cast<uint2*>(hits), hits_stride,
                           hit_counts, probs, probs_stride, rseeds, trys, stage_read, stage_loc, stage_seed, key, nullptr);
    if (hipError_t e = hipcub::DeviceScan::ExclusiveSum(p, scan_bytes, key, off, int(n_active) + 1, s)) return e;
    hipLaunchKernelGGL(select_compact_kernel, grid_for(n + 1u), dim3(256), 0, s, n_multi, n_active, key, off, stage_read, stage_loc, stage_seed,
                       active_out, hit_begin, hit_read_id, hit_loc, hit_seed, out_sizes);
    return hipGetLastError();
}

NVB_API int nvbio_hip_locate_hits(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n,
                                  uint32_t* hit_loc, const uint32_t* hit_seed, void* stream)
{
    if (!fmi || !fmi->bwt_occ || !fmi->ssa) return hipErrorInvalidValue;
    if (fmi->sa_int == 0 || (fmi->sa_int & (fmi->sa_int - 1)) != 0) return hipErrorInvalidValue;
    if (rfmi && (!rfmi->bwt_occ || !rfmi->ssa || rfmi->sa_int == 0 || (rfmi->sa_int & (rfmi->sa_int - 1)) != 0)) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    if (!hit_loc || !hit_seed) return hipErrorInvalidValue;
    g_last_kernel = "locate_hits_kernel";
    hipLaunchKernelGGL(locate_hits_kernel, dim3(seeding_grid(n)), dim3(256), 0, to_stream(stream), make_fmi(fmi), make_fmi(rfmi ? rfmi : fmi), n, hit_loc, hit_seed);
    return hipGetLastError();
}

NVB_API int nvbio_hip_score_best_setup(uint32_t n_hits, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed,
                                       const uint64_t* read_begin, const uint32_t* read_len, uint32_t fixed_read_len, uint64_t rc_offset,
                                       uint32_t band_len, uint32_t genome_length, const uint64_t* best_alignments, uint32_t best_stride,
                                       int32_t score_limit, uint64_t* pattern_begin, uint32_t* pattern_len,
                                       uint64_t* text_begin, uint32_t* text_len, int32_t* min_score, int32_t* known_score,
                                       uint32_t* job_count, uint32_t* job_hit, void* stream)
{
    if ((job_count != nullptr) != (job_hit != nullptr) || (job_hit && !known_score)) return hipErrorInvalidValue;
    if (job_count) { const hipError_t e = hipMemsetAsync(job_count, 0, sizeof(uint32_t), to_stream(stream)); if (e != hipSuccess) return e; }
    if (n_hits == 0) return hipSuccess;
    if (!hit_read_id || !hit_loc || !hit_seed || !best_alignments || best_stride == 0 || !pattern_begin || !text_begin || !text_len || !min_score)
        return hipErrorInvalidValue;
    if (!read_len && fixed_read_len == 0) return hipErrorInvalidValue;
    if (read_len && !pattern_len) return hipErrorInvalidValue;
    g_last_kernel = "score_best_setup_kernel";
    hipLaunchKernelGGL(score_best_setup_kernel, grid_for((uint64_t(n_hits) + SETUP_TILES - 1u) / SETUP_TILES), dim3(256), 0, to_stream(stream), n_hits, hit_read_id, hit_loc, hit_seed,
                       read_begin, read_len, fixed_read_len, rc_offset, band_len, genome_length, reinterpret_cast<const uint2*>(best_alignments),
                       best_stride, score_limit, pattern_begin, pattern_len, text_begin, text_len, min_score, known_score, job_count, job_hit);
    return hipGetLastError();
}

NVB_API int nvbio_hip_hit_deque_replay(uint32_t n_cases, const uint32_t* case_start, const uint8_t* ops, const uint64_t* vals, const uint32_t* caps,
                                       const uint64_t* state_start, uint64_t* scratch, uint32_t scratch_stride, uint64_t* out_states, void* stream)
{
    if (n_cases == 0) return hipSuccess;
    if (!case_start || !ops || !vals || !caps || !state_start || !scratch || !out_states || scratch_stride == 0) return hipErrorInvalidValue;
    g_last_kernel = "hit_deque_replay_kernel";
    hipLaunchKernelGGL(hit_deque_replay_kernel, dim3((n_cases + 63u) / 64u), dim3(64), 0, to_stream(stream), n_cases, case_start, ops, vals, caps,
                       state_start, reinterpret_cast<uint2*>(scratch), scratch_stride, out_states);
    return hipGetLastError();
}

// ------------------------------------------------------------------ driver utilities
// What the reference's host drivers do with thrust / nvbio primitives between the stages:
//   mark_unaligned_kernel  aligner_init.cu:421-436 ; nvbio::copy_flagged  (nvbio/basic/primitives.h) -> hipCUB select
//   BestTracebackStream::init_context  traceback_inl.h:104-136 (window and pattern of every best alignment)
namespace nvb {

__global__ void __launch_bounds__(256)
mark_unaligned_kernel(uint32_t n_active, const uint32_t* __restrict__ active, const uint2* __restrict__ best, uint8_t* __restrict__ reseed)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_active) return;
    if (best[active[t]].y == 0xFFFFFFFFu) reseed[t] = 1u;
}

// anchor-style tracebacks (banded): window = alignment - band/2 (clamped), band + read_len long; opposite-style (full matrix,
// concordant opposite mates): [alignment, alignment + sink).  The mate (which read set) and the strand come from the alignment.
// valid[i] = 0 for unaligned entries (the stream's init_context returns false) or when `want` does not match:
// want 0: every aligned entry -> banded; 1: concordant entries -> full; 2: aligned, not concordant -> banded
__global__ void __launch_bounds__(256)
traceback_best_setup_kernel(uint32_t n, const uint32_t* __restrict__ idx, const uint2* __restrict__ best, uint32_t band_len, uint32_t genome_len,
                            const uint64_t* __restrict__ read_begin, const uint32_t* __restrict__ read_len, uint32_t fixed_len, uint64_t rc_offset,
                            const uint64_t* __restrict__ read_begin1, const uint32_t* __restrict__ read_len1, uint32_t fixed_len1, uint64_t rc_offset1, int two_mates,
                            uint64_t mate_offset, int want,
                            uint8_t* __restrict__ valid, uint64_t* __restrict__ pat_begin, uint32_t* __restrict__ pat_len,
                            uint64_t* __restrict__ text_begin, uint32_t* __restrict__ text_len)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = idx ? idx[i] : i;
    const uint2 a = best[r];
    const bool aligned = a.y != 0xFFFFFFFFu;
    const bool concordant = ((a.x >> 30) & 1u) && !((a.x >> 31) & 1u);
    const bool ok = aligned && (want == 0 || (want == 1 ? concordant : !concordant));
    const uint32_t rc = (a.x >> 28) & 1u, mate = (a.x >> 29) & 1u, sink = (a.x >> 18) & 0x3FFu;
    // (two_mates: mate 1's reads have their own begins / lengths / rc offset inside their half of the stream -- mates of different lengths)
    const bool m1 = two_mates && mate;
    const uint32_t len = m1 ? (read_len1 ? read_len1[r] : fixed_len1) : (read_len ? read_len[r] : fixed_len);
    uint32_t gb, ge;
    if (want == 1) { gb = a.y; ge = gb + sink; }
    else { gb = a.y > band_len / 2u ? a.y - band_len / 2u : 0u; ge = gb + band_len + len; }
    ge = ge < genome_len ? ge : genome_len;
    valid[i] = ok ? 1u : 0u;
    text_begin[i] = gb;
    text_len[i] = (ok && ge > gb) ? ge - gb : 0u;
    pat_begin[i] = m1 ? (read_begin1 ? read_begin1[r] : uint64_t(r) * fixed_len1) + (rc ? rc_offset1 : 0ull) + mate_offset
                      : (read_begin ? read_begin[r] : uint64_t(r) * fixed_len) + (rc ? rc_offset : 0ull) + (mate ? mate_offset : 0ull);
    if (pat_len) pat_len[i] = len;
}

// active_read_queues.in_queue = pack_read(top_seed) of the seed queue (defs.h:185-205): read id in the low 31 bits, the flag on top
__global__ void __launch_bounds__(256)
pack_read_queue_kernel(uint32_t n, const uint32_t* __restrict__ queue, uint32_t top_flag, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = ((queue ? queue[i] : i) & 0x7FFFFFFFu) | (top_flag << 31);
}

} // namespace nvb

NVB_API int nvbio_hip_pack_read_queue(uint32_t n, const uint32_t* queue, uint32_t top_flag, uint32_t* out, void* stream)
{
    if (n == 0) return hipSuccess;
    if (!out) return hipErrorInvalidValue;
    g_last_kernel = "pack_read_queue_kernel";
    hipLaunchKernelGGL(pack_read_queue_kernel, grid_for(n), dim3(256), 0, to_stream(stream), n, queue, top_flag & 1u, out);
    return hipGetLastError();
}

NVB_API int nvbio_hip_mark_unaligned(uint32_t n_active, const uint32_t* active_reads, const uint64_t* best_alignments, uint8_t* reseed, void* stream)
{
    if (n_active == 0) return hipSuccess;
    if (!active_reads || !best_alignments || !reseed) return hipErrorInvalidValue;
    g_last_kernel = "mark_unaligned_kernel";
    hipLaunchKernelGGL(mark_unaligned_kernel, grid_for(n_active), dim3(256), 0, to_stream(stream), n_active, active_reads,
                       reinterpret_cast<const uint2*>(best_alignments), reseed);
    return hipGetLastError();
}

NVB_API uint64_t nvbio_hip_copy_flagged_temp_bytes(uint32_t n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceSelect::Flagged(nullptr, bytes, (const uint32_t*)nullptr, (const uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, int(n));   // size query only
    return align256(bytes) + 256u;
}

NVB_API int nvbio_hip_copy_flagged(uint32_t n, const uint32_t* in, const uint8_t* flags, uint32_t* out, uint32_t* out_count, void* temp, uint64_t temp_bytes, void* stream)
{
    if (!out_count) return hipErrorInvalidValue;
    if (n == 0) return hipMemsetAsync(out_count, 0, 4, to_stream(stream));
    if (!in || !flags || !out || !temp || temp_bytes < nvbio_hip_copy_flagged_temp_bytes(n)) return hipErrorInvalidValue;
    uint8_t* p = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(temp) + 255u) & ~uintptr_t(255));
    size_t bytes = size_t(temp_bytes - 256u);
    g_last_kernel = "hipcub::DeviceSelect::Flagged";
    return hipcub::DeviceSelect::Flagged(p, bytes, in, flags, out, out_count, int(n), to_stream(stream));
}

NVB_API int nvbio_hip_traceback_best_setup(uint32_t n, const uint32_t* idx, const uint64_t* best_alignments, uint32_t band_len, uint32_t genome_length,
                                           const uint64_t* read_begin, const uint32_t* read_len, uint32_t fixed_read_len, uint64_t rc_offset, uint64_t mate_offset,
                                           int32_t want, uint8_t* out_valid, uint64_t* pattern_begin, uint32_t* pattern_len, uint64_t* text_begin, uint32_t* text_len,
                                           void* stream)
{
    if (n == 0) return hipSuccess;
    if (!best_alignments || !out_valid || !pattern_begin || !text_begin || !text_len || want < 0 || want > 2) return hipErrorInvalidValue;
    if ((!read_len && fixed_read_len == 0) || (read_len && !pattern_len)) return hipErrorInvalidValue;
    g_last_kernel = "traceback_best_setup_kernel";
    hipLaunchKernelGGL(traceback_best_setup_kernel, grid_for(n), dim3(256), 0, to_stream(stream), n, idx, reinterpret_cast<const uint2*>(best_alignments),
                       band_len, genome_length, read_begin, read_len, fixed_read_len, rc_offset, nullptr, nullptr, 0u, 0ull, 0, mate_offset, int(want), out_valid, pattern_begin, pattern_len,
                       text_begin, text_len);
    return hipGetLastError();
}

// ... for pairs whose mates have their own lengths: mate m's reads at read_begin[m][r] (or r * fixed_read_len[m]) of ITS half of the pattern stream, its
// reverse complements rc_offset[m] further; mate 1's half starts mate_offset symbols into the stream
NVB_API int nvbio_hip_traceback_best_setup_mates(uint32_t n, const uint32_t* idx, const uint64_t* best_alignments, uint32_t band_len, uint32_t genome_length,
                                                 const uint64_t* const read_begin[2], const uint32_t* const read_len[2], const uint32_t fixed_read_len[2], const uint64_t rc_offset[2],
                                                 uint64_t mate_offset, int32_t want, uint8_t* out_valid, uint64_t* pattern_begin, uint32_t* pattern_len, uint64_t* text_begin,
                                                 uint32_t* text_len, void* stream)
{
    if (n == 0) return hipSuccess;
    if (!best_alignments || !out_valid || !pattern_begin || !pattern_len || !text_begin || !text_len || want < 0 || want > 2) return hipErrorInvalidValue;
    if (!read_begin || !read_len || !fixed_read_len || !rc_offset) return hipErrorInvalidValue;
    for (int m = 0; m < 2; ++m) if (!read_len[m] && fixed_read_len[m] == 0) return hipErrorInvalidValue;
    g_last_kernel = "traceback_best_setup_kernel";
    hipLaunchKernelGGL(traceback_best_setup_kernel, grid_for(n), dim3(256), 0, to_stream(stream), n, idx, reinterpret_cast<const uint2*>(best_alignments),
                       band_len, genome_length, read_begin[0], read_len[0], fixed_read_len[0], rc_offset[0], read_begin[1], read_len[1], fixed_read_len[1], rc_offset[1], 1,
                       mate_offset, int(want), out_valid, pattern_begin, pattern_len, text_begin, text_len);
    return hipGetLastError();
}

// ------------------------------------------------------------------ finish_alignment
// finish_alignment_kernel (traceback_inl.h:523-722) + BestTracebackStream::finish (:177-189): one lane replays its alignment's
// CIGAR (stored end first) over the read and its genome window and emits the MD string in nvbio's byte code (io::MDS_OP:
// [len lo, len hi] then {MATCH, run <= 255} / {MISMATCH, read symbol} / {INSERTION | DELETION, length byte, symbols...}), the
// edit distance (clips not counted) and the final score = scoring_scheme.score() summed over the SUBSTITUTION columns (scoring.h:301-311)
// minus cumulative_deletion(l) per INSERTION run and cumulative_insertion(l) per DELETION run (traceback_inl.h:664-665), then rewrites the alignment: m_align = window begin, m_ed, m_score.  A streaming pass over ~100 bytes
// per read.  A CIGAR that overflowed its slots cannot be replayed: skipped (mds_len 0, alignment untouched).
namespace nvb {

struct FinishParams {
    uint32_t n; const uint8_t* valid; StringSet pat, txt; const uint8_t* quals; uint64_t n_quals;
    const uint16_t* cigar; uint32_t cigar_stride; const uint32_t* cigar_len; const uint2* cigar_source;
    int32_t match, n_penalty; int32_t gaps[4]; int32_t mismatch[256];
    const uint32_t* idx; uint2* best; uint8_t* mds; uint32_t mds_stride; uint32_t* mds_len;
};

__global__ void __launch_bounds__(256) finish_alignment_kernel(const FinishParams p)
{
    __shared__ int32_t mm[256];
    mm[threadIdx.x] = p.mismatch[threadIdx.x];
    __syncthreads();
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= p.n) return;
    const uint32_t clen = p.cigar_len[w];
    if (!p.valid[w] || clen == 0u || clen > p.cigar_stride) { p.mds_len[w] = 0u; return; }
    const uint16_t* cv = p.cigar + uint64_t(w) * p.cigar_stride;
    uint8_t* mds = p.mds + uint64_t(w) * p.mds_stride;
    const uint64_t pb = p.pat.begin[w], tb = p.txt.begin[w];
    uint32_t mds_len = 2u, mds_op = 4u, ed = 0u, run = 0u /* counter of the open MATCH run (kept in a register) */, run_at = 0u;
    int32_t  score = 0;
    uint32_t j = 0u, k = p.cigar_source[w].x & 0xFFFFu;
    auto push = [&](const uint32_t v) { if (mds_len < p.mds_stride) mds[mds_len] = uint8_t(v); ++mds_len; };
    // the walk moves one symbol at a time: keep the packed word under each cursor in a register (one load per 8 / 16 symbols)
    uint64_t pw_at = ~0ull, tw_at = ~0ull; uint32_t pw = 0u, tw = 0u;
    const uint32_t pshift = p.pat.s.bits == 2 ? 4u : 3u;
    auto pat_sym = [&](const uint64_t sym) -> uint32_t {
        const uint64_t wi = sym >> pshift;
        if (wi != pw_at) { pw = ld_word_global(p.pat.s, wi); pw_at = wi; }
        if (p.pat.s.bits == 2) { const uint32_t c = uint32_t(sym) & 15u; return (pw >> (p.pat.s.big_endian ? 30u - 2u * c : 2u * c)) & 3u; }
        const uint32_t c = uint32_t(sym) & 7u; return (pw >> (p.pat.s.big_endian ? 28u - 4u * c : 4u * c)) & 15u;
    };
    auto txt_sym = [&](const uint64_t sym) -> uint32_t {
        const uint64_t wi = sym >> 4;
        if (wi != tw_at) { tw = ld_word_global(p.txt.s, wi); tw_at = wi; }
        const uint32_t c = uint32_t(sym) & 15u; return (tw >> (p.txt.s.big_endian ? 30u - 2u * c : 2u * c)) & 3u;
    };
    auto close_run = [&]() { if (run && run_at < p.mds_stride) mds[run_at] = uint8_t(run); run = 0u; };
    for (uint32_t i = 0; i < clen; ++i)
    {
        const uint32_t word = cv[clen - i - 1u], t = word & 3u, l = word >> 2;
        if (t != 0u) { close_run(); mds_op = (t == 2u) ? 3u : 2u; push(mds_op); push(l); }
        for (uint32_t x = 0; x < l; ++x)
        {
            j += (t != 2u) ? 1u : 0u;
            k += (t == 0u || t == 2u) ? 1u : 0u;
            const uint32_t readc = j > 0u ? pat_sym(pb + j - 1u) : 255u;
            const uint32_t refc  = k > 0u ? txt_sym(tb + k - 1u) : 255u;
            if (t == 0u) {
                if (readc == refc) {
                    if (mds_op == 0u && run < 255u) ++run;
                    else { close_run(); mds_op = 0u; push(0u); run_at = mds_len; push(1u); run = 1u; }
                } else { close_run(); mds_op = 1u; push(1u); push(readc); ++ed; }
                const uint32_t ref_mask = (1u << (refc & 31u)) & 0xFFu;
                if (readc > 3u || ref_mask > 15u) score -= p.n_penalty;
                else if (ref_mask & (1u << readc)) score += p.match;
                else score += mm[p.quals ? p.quals[min(pb + j - 1u, p.n_quals - 1u)] : 0u];       // the quality matters on a mismatch only
            } else {
                push(t == 2u ? refc : readc);
                if (t != 3u) ++ed;
            }
        }
        // a run of l inserted read symbols costs cumulative_deletion(l) = ref_gap_const + ref_gap_coeff * l = -(text_gap_open + (l - 1) * text_gap_ext),
        // a run of l deleted genome symbols cumulative_insertion(l), the pattern-gap twin (scoring.h:319-329; the names are crossed in the reference)
        if (t == 1u && l)      score += p.gaps[2] + int32_t(l - 1u) * p.gaps[3];
        else if (t == 2u && l) score += p.gaps[0] + int32_t(l - 1u) * p.gaps[1];
    }
    close_run();
    if (p.mds_stride >= 2u) { mds[0] = uint8_t(mds_len & 0xFFu); mds[1] = uint8_t(mds_len >> 8); }
    p.mds_len[w] = mds_len;
    const uint32_t r = p.idx ? p.idx[w] : w;
    const uint32_t mag = score < 0 ? uint32_t(-score) : uint32_t(score);
    const uint32_t a = (p.best[r].x & 0xF0000000u) | (score < 0 ? 1u : 0u) | ((mag & 0x1FFFFu) << 1) | ((ed & 0x3FFu) << 18);
    p.best[r] = make_uint2(a, uint32_t(tb));
}

} // namespace nvb

NVB_API int nvbio_hip_finish_alignment(uint32_t n, const uint8_t* valid, const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals,
                                       const nvbio_hip_string_set* texts, const uint16_t* cigar, uint32_t cigar_stride, const uint32_t* cigar_len,
                                       const uint32_t* cigar_source, int32_t match, const int32_t* mismatch_by_quality, int32_t n_penalty, const int32_t* gap_costs,
                                       const uint32_t* idx, uint64_t* best_alignments, uint8_t* out_mds, uint32_t mds_stride, uint32_t* out_mds_len, void* stream)
{
    if (n == 0) return hipSuccess;
    if (!valid || !patterns || !texts || !cigar || !cigar_len || !cigar_source || !mismatch_by_quality || !gap_costs || !best_alignments || !out_mds || !out_mds_len ||
        cigar_stride == 0 || mds_stride < 2) return hipErrorInvalidValue;
    if (!(patterns->bits == 2 || patterns->bits == 4) || texts->bits != 2) return hipErrorNotSupported;
    if (!patterns->words || !patterns->begin || !texts->words || !texts->begin) return hipErrorInvalidValue;
    FinishParams p;
    p.n = n; p.valid = valid; p.pat = make_string_set(patterns); p.txt = make_string_set(texts); p.quals = quals; p.n_quals = n_quals;
    p.cigar = cigar; p.cigar_stride = cigar_stride; p.cigar_len = cigar_len; p.cigar_source = reinterpret_cast<const uint2*>(cigar_source);
    p.match = match; p.n_penalty = n_penalty;
    for (int i = 0; i < 4; ++i) p.gaps[i] = gap_costs[i];
    for (int i = 0; i < 256; ++i) p.mismatch[i] = mismatch_by_quality[i];
    p.idx = idx; p.best = reinterpret_cast<uint2*>(best_alignments); p.mds = out_mds; p.mds_stride = mds_stride; p.mds_len = out_mds_len;
    g_last_kernel = "finish_alignment_kernel";
    hipLaunchKernelGGL(finish_alignment_kernel, grid_for(n), dim3(256), 0, to_stream(stream), p);
    return hipGetLastError();
}

// copies row i of src (row_bytes bytes, a multiple of 4) to row idx[i] of dst: puts the results of a compacted batch (e.g. the
// full-matrix tracebacks of the concordant opposite mates) back at their reads
namespace nvb {
__global__ void __launch_bounds__(256) scatter_rows_kernel(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t row_words)
{
    const uint64_t t = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (t >= uint64_t(n) * row_words) return;
    const uint32_t i = uint32_t(t / row_words), w = uint32_t(t % row_words);
    dst[uint64_t(idx[i]) * row_words + w] = src[t];
}
}
NVB_API int nvbio_hip_scatter_rows(uint32_t n, const uint32_t* idx, const void* src, void* dst, uint32_t row_bytes, void* stream)
{
    if (n == 0) return hipSuccess;
    if (!idx || !src || !dst || row_bytes == 0 || (row_bytes & 3u)) return hipErrorInvalidValue;
    g_last_kernel = "scatter_rows_kernel";
    hipLaunchKernelGGL(scatter_rows_kernel, grid_for(uint64_t(n) * (row_bytes / 4u)), dim3(256), 0, to_stream(stream), n, idx,
                       static_cast<const uint32_t*>(src), static_cast<uint32_t*>(dst), row_bytes / 4u);
    return hipGetLastError();
}

// dst row i = src row idx[i] (the twin of scatter_rows): compacts accepted alignments through an index list
namespace nvb {
__global__ void __launch_bounds__(256) gather_rows_kernel(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t row_words)
{
    const uint64_t t = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (t >= uint64_t(n) * row_words) return;
    const uint32_t i = uint32_t(t / row_words), w = uint32_t(t % row_words);
    dst[t] = src[uint64_t(idx[i]) * row_words + w];
}
}
NVB_API int nvbio_hip_gather_rows(uint32_t n, const uint32_t* idx, const void* src, void* dst, uint32_t row_bytes, void* stream)
{
    if (n == 0) return hipSuccess;
    if (!idx || !src || !dst || row_bytes == 0 || (row_bytes & 3u)) return hipErrorInvalidValue;
    g_last_kernel = "gather_rows_kernel";
    hipLaunchKernelGGL(gather_rows_kernel, grid_for(uint64_t(n) * (row_bytes / 4u)), dim3(256), 0, to_stream(stream), n, idx,
                       static_cast<const uint32_t*>(src), static_cast<uint32_t*>(dst), row_bytes / 4u);
    return hipGetLastError();
}
