// compat/nvbio/io/output/output_file.h -- where an aligner sends its results (nvbio/io/output/output_file.h:75-164, output_file.cpp):
// OutputFile is the null sink and the base class; OutputFile::open() picks the writer from the file name.  This layer writes SAM
// (output_sam.h) and BAM (output_bam.h); a ".dbg" name -- DebugOutput in the reference -- gets the SAM writer, with a warning.
#pragma once
#include "output_types.h"
#include "output_stats.h"
#include "../sequence/sequence.h"
#include "../../basic/console.h"
#include <stdio.h>
#include <string.h>
#include <string>
#if defined(__HIPCC__)

namespace nvbio {
namespace io {

struct HostOutputBatchSE;
struct HostOutputBatchPE;

struct OutputFile
{
protected:
    OutputFile(const char* _file_name, AlignmentType _alignment_type, BNT _bnt) : file_name(_file_name), alignment_type(_alignment_type), bnt(_bnt), mapq_filter(-1) {}

public:
    virtual ~OutputFile() {}

    void set_program(const char* _pg_id, const char* _pg_name, const char* _pg_version, const char* _pg_args)
    { pg_id = _pg_id ? _pg_id : ""; pg_name = _pg_name ? _pg_name : ""; pg_version = _pg_version ? _pg_version : ""; pg_args = _pg_args ? _pg_args : ""; }
    void set_rg(const char* _rg_id, const char* _rg_string) { rg_id = _rg_id ? _rg_id : ""; rg_string = _rg_string ? _rg_string : ""; }

    virtual void header() {}
    virtual void configure_mapq_evaluator(int _mapq_filter) { mapq_filter = _mapq_filter; }
    virtual void process(struct HostOutputBatchSE& batch) {}
    virtual void process(struct HostOutputBatchPE& batch) {}
    virtual void close(void) {}
    virtual IOStats& get_aggregate_statistics(void) { return iostats; }

protected:
    const char*   file_name;
    AlignmentType alignment_type;
    BNT           bnt;
    int           mapq_filter;
    IOStats       iostats;
    std::string   pg_id, pg_name, pg_version, pg_args, rg_id, rg_string;

public:
    /// "" -> SAM on stdout; "/dev/null" -> the null sink; otherwise by extension
    static inline OutputFile* open(const char* file_name, AlignmentType aln_type, BNT bnt);
};

} // namespace io
} // namespace nvbio

#include "output_sam.h"
#include "output_bam.h"

namespace nvbio {
namespace io {

namespace priv { struct NullOutput : public OutputFile { NullOutput(const char* n, AlignmentType t, BNT b) : OutputFile(n, t, b) {} }; }

inline OutputFile* OutputFile::open(const char* file_name, AlignmentType aln_type, BNT bnt)
{
    const size_t len = strlen(file_name);
    if (len == 0) return new SamOutput(NULL, aln_type, bnt);
    if (strcmp(file_name, "/dev/null") == 0) return new priv::NullOutput(file_name, aln_type, bnt);
    const char* ext = len >= 4 ? file_name + len - 4 : "";
    if (strcmp(ext, ".sam") == 0) return new SamOutput(file_name, aln_type, bnt);
    if (strcmp(ext, ".bam") == 0) return new BamOutput(file_name, aln_type, bnt);
    if (strcmp(ext, ".dbg") == 0) log_warning(stderr, "%s output is not written by this layer; writing SAM text to %s\n", ext + 1, file_name);
    else                                                     log_warning(stderr, "could not determine file type for %s; guessing SAM\n", file_name);
    return new SamOutput(file_name, aln_type, bnt);
}

} // namespace io
} // namespace nvbio
#endif
