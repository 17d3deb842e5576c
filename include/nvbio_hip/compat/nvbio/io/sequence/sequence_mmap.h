// compat/nvbio/io/sequence/sequence_mmap.h -- sequence data published by a separate server process through named shared memory
// (nvbio/io/sequence/sequence_mmap.h:44-152; the nvFM-server workflow).  This layer runs no such server: map_sequence_file() finds
// nothing to map and returns NULL, SequenceDataMMAP::load() returns false, and callers take their load-from-disk branch
// (nvBowtie.cpp:506-530 does exactly that when mapping fails).
#pragma once
#include "sequence.h"
#include <string>

namespace nvbio {
namespace io {

struct SequenceDataMMAP : public SequenceData
{
    typedef SequenceDataView       plain_view_type;
    typedef ConstSequenceDataView  const_plain_view_type;
    SequenceDataMMAP() : m_sequence_ptr(NULL), m_sequence_index_ptr(NULL), m_qual_ptr(NULL), m_name_ptr(NULL), m_name_index_ptr(NULL) {}
    bool load(const char*) { return false; }
    operator plain_view_type()             { return plain_view_type(*this, m_sequence_ptr, m_sequence_index_ptr, m_qual_ptr, m_name_ptr, m_name_index_ptr); }
    operator const_plain_view_type() const { return const_plain_view_type(*this, m_sequence_ptr, m_sequence_index_ptr, m_qual_ptr, m_name_ptr, m_name_index_ptr); }
    uint32* m_sequence_ptr;
    uint32* m_sequence_index_ptr;
    char*   m_qual_ptr;
    char*   m_name_ptr;
    uint32* m_name_index_ptr;
};
inline SequenceDataMMAP* map_sequence_file(const char*) { return NULL; }

} // namespace io
} // namespace nvbio
