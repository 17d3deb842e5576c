// compat/nvbio/basic/sum_tree.h -- SumTree<Iterator>: a complete binary tree of partial sums laid out level by level in one array
// (leaves first, padded to a power of two, root last), and sample(): the leaf a value in [0,1] falls on when leaves are weighted by
// their cells (nvbio/basic/sum_tree.h:62-181, sum_tree_inl.h).  nvBowtie draws its seed hits through this; the float operations below
// are in the reference's order so that a draw lands on the same leaf, bit for bit (the reference's own nvbio-test/sum_tree_test.cpp
// runs against this header in tools/ref_bind_check.py).
#pragma once
#include "types.h"
#include "numbers.h"
#include <iterator>

namespace nvbio {

template <typename Iterator>
struct SumTree
{
    typedef Iterator                                             iterator_type;
    typedef typename std::iterator_traits<Iterator>::value_type  value_type;

    /// leaves rounded up to a power of two
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint32 padded(const uint32 size) { const uint32 p = 1u << nvbio::log2(size); return p < size ? p * 2u : p; }
    /// cells a tree of `size` leaves occupies
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint32 node_count(const uint32 size) { return padded(size) * 2u - 1u; }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE SumTree(const uint32 size, iterator_type cells) : m_cells(cells), m_size(size), m_padded_size(padded(size)) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size()        const { return m_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 padded_size() const { return m_padded_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 nodes()       const { return m_padded_size * 2u - 1u; }

    /// build every level above the leaves (padding leaves are set to `zero` first)
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void setup(const value_type zero = value_type(0))
    {
        for (uint32 i = m_size; i < m_padded_size; ++i) m_cells[i] = zero;
        for (uint32 level = 0, width = m_padded_size; width >= 2u; level += width, width >>= 1)
            for (uint32 i = 0; i < width / 2u; ++i)
                m_cells[level + width + i] = m_cells[level + 2u * i] + m_cells[level + 2u * i + 1u];
    }
    /// add v to leaf i and to every sum above it
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void add(const uint32 i, const value_type v)
    {
        for (uint32 level = 0, width = m_padded_size, j = i; width >= 2u; level += width, width >>= 1, j >>= 1)
            m_cells[level + j] += v;
    }
    /// set leaf i to v and recompute every sum above it from its two children
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void set(const uint32 i, const value_type v)
    {
        m_cells[i] = v;
        uint32 below = 0u, level = m_padded_size, j = i >> 1;
        for (uint32 width = m_padded_size >> 1; level + j < nodes(); width >>= 1, j >>= 1)
        {
            m_cells[level + j] = m_cells[below + 2u * j] + m_cells[below + 2u * j + 1u];
            below = level; level += width;
        }
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type sum() const { return m_cells[m_padded_size * 2u - 2u]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type cell(const uint32 i) const { return m_cells[i]; }

private:
    iterator_type m_cells;
    uint32        m_size;
    uint32        m_padded_size;
};

/// walk from the root to a leaf: at each pair (l, r) go left when value * (l + r) < l or r is empty, rescaling value into the chosen child
template <typename Iterator>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 sample(const SumTree<Iterator>& tree, const float value)
{
    const uint32 padded_size = tree.padded_size(), size = tree.size();
    uint32 level = padded_size * 2u - 4u;           // the pair below the root
    uint32 node  = 0;
    float  v     = value;
    for (uint32 width = 2; width < padded_size; width *= 2)
    {
        const float l = float(tree.cell(level + node)), r = float(tree.cell(level + node + 1u));
        const float sum = float(l + r);
        if (sum == 0.0f) node *= 2u;
        else if (v * sum < l || r == 0.0f) { node = node * 2u;        v = nvbio::min(v * sum / l, 1.0f); }
        else                               { node = (node + 1u) * 2u; v = nvbio::min((v * sum - l) / r, 1.0f); }
        level -= width * 2u;
    }
    const float l = node < size ? float(tree.cell(node)) : 0.0f, r = node + 1u < size ? float(tree.cell(node + 1u)) : 0.0f;
    const float sum = float(l + r);
    if (!(v * sum < l || r == 0.0f)) node += 1u;
    return node < size ? node : size - 1u;
}

} // namespace nvbio
