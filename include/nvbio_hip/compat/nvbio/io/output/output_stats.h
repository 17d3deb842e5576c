// compat/nvbio/io/output/output_stats.h -- counters an output file keeps about itself (nvbio/io/output/output_stats.h:38-60)
#pragma once
#include "output_types.h"
#include "../../basic/timer.h"

namespace nvbio {
namespace io {

struct IOStats
{
    int        alignments_DtoH_count;       ///< reads copied device -> host
    float      alignments_DtoH_time;        ///< seconds spent in those copies
    uint32     n_reads;                     ///< reads written
    TimeSeries output_process_timings;      ///< one entry per OutputFile::process() call
    IOStats() : alignments_DtoH_count(0), alignments_DtoH_time(0.0f), n_reads(0) {}
};

} // namespace io
} // namespace nvbio
