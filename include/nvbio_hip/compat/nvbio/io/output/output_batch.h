// compat/nvbio/io/output/output_batch.h -- one pass of alignment results on its way to an output file
// (nvbio/io/output/output_batch.h:44-135, output_batch.cpp): DeviceOutputBatchSE references the aligner's device buffers (alignments,
// CIGARs + their coordinates, MD programs, mapping qualities, optional read ids); HostOutputBatchSE / PE own host copies, one set per mate.
#pragma once
#include "output_types.h"
#include "output_stats.h"
#include "../sequence/sequence.h"
#include "../../basic/vector_array.h"
#if defined(__HIPCC__)

namespace nvbio {
namespace io {

struct DeviceOutputBatchSE
{
    uint32                                  count;
    thrust::device_vector<io::Alignment>&   alignments;
    DeviceCigarArray                        cigar;
    nvbio::DeviceVectorArray<uint8>&        mds;
    thrust::device_vector<uint8>&           mapq;
    thrust::device_vector<uint32>*          read_ids;

    DeviceOutputBatchSE(uint32 _count, thrust::device_vector<io::Alignment>& _alignments, DeviceCigarArray _cigar, nvbio::DeviceVectorArray<uint8>& _mds,
                        thrust::device_vector<uint8>& _mapq, thrust::device_vector<uint32>* _read_ids = NULL)
        : count(_count), alignments(_alignments), cigar(_cigar), mds(_mds), mapq(_mapq), read_ids(_read_ids) {}

    // device -> host copies of each part
    void readback_scores(thrust::host_vector<io::Alignment>& host_alignments) const { host_alignments = alignments; }
    void readback_cigars(HostCigarArray& host_cigars) const { host_cigars.array = cigar.array; host_cigars.coords = cigar.coords; }
    void readback_mds(nvbio::HostVectorArray<uint8>& host_mds) const { host_mds = mds; }
    void readback_mapq(thrust::host_vector<uint8>& host_mapq) const { host_mapq = mapq; }
    void readback_ids(thrust::host_vector<uint32>& host_ids) const { if (read_ids) host_ids = *read_ids; else host_ids.resize(0); }
};

struct HostOutputBatchSE
{
    uint32                              count;
    thrust::host_vector<io::Alignment>  alignments;
    HostCigarArray                      cigar;
    HostMdsArray                        mds;
    thrust::host_vector<uint8>          mapq;
    thrust::host_vector<uint32>         read_ids;
    const io::SequenceDataHost*         read_data;

    HostOutputBatchSE() : count(0), read_data(NULL) {}
    void readback(const DeviceOutputBatchSE batch)
    {
        count = batch.count;
        batch.readback_scores(alignments); batch.readback_cigars(cigar); batch.readback_mds(mds); batch.readback_mapq(mapq); batch.readback_ids(read_ids);
    }
};

struct HostOutputBatchPE
{
    uint32                              count;
    thrust::host_vector<io::Alignment>  alignments[2];
    HostCigarArray                      cigar[2];
    HostMdsArray                        mds[2];
    thrust::host_vector<uint8>          mapq[2];
    thrust::host_vector<uint32>         read_ids;
    const io::SequenceDataHost*         read_data[2];

    HostOutputBatchPE() : count(0) { read_data[0] = read_data[1] = NULL; }
    /// slot set `mate` (0: the anchors' pass, 1: the opposite mates') from one device pass; the ids come with the first
    void readback(const DeviceOutputBatchSE batch, const AlignmentMate mate)
    {
        count = batch.count;
        batch.readback_scores(alignments[mate]); batch.readback_cigars(cigar[mate]); batch.readback_mds(mds[mate]); batch.readback_mapq(mapq[mate]);
        if (mate == MATE_1) batch.readback_ids(read_ids);
    }
};

} // namespace io
} // namespace nvbio
#endif
