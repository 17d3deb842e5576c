// tests/compat/io_callers.hip -- a caller of the drop-in layer's alignment writers (compat/nvbio/io/output): host code only, so the CPU
// suite can run it.  The test hands over one batch of alignments as flat arrays; this file builds what an aligner would hand to
// io::OutputFile -- the reads as the loader stores them (io::REVERSE), the reference's names and offsets, HostOutputBatchSE / PE with
// io::Alignment words, CIGAR and MD arenas -- and lets OutputFile::open() pick SamOutput or BamOutput by the file name.
#include <nvbio/io/output/output_file.h>
#include <nvbio/io/output/output_batch.h>
#include <nvbio/io/sequence/sequence.h>
#include <string.h>
#include <string>

using namespace nvbio;
#define API extern "C" __attribute__((visibility("default")))

struct SlotSet                                   // one set of per-read results (single-end: one; pairs: anchors and opposite mates)
{
    const uint32* words;    const uint32* aligns;            // io::Alignment: packed word, m_align
    const uint32* cigar_offsets; const uint16* cigar_ops;    // ops (m_type | m_len << 2) stored LAST operation first; offsets [n + 1]
    const uint32* cigar_source;                              // coords.x: offset of the alignment's first reference base from m_align
    const uint32* mds_offsets;   const uint8*  mds_bytes;    // MD programs; offsets [n + 1]
    const uint8*  mapq;
};
struct WriteArgs
{
    const char*   path;  uint32 paired;
    uint32        n_ref; const char* ref_names; const uint32* ref_lengths;         // names '\0'-separated
    uint32        n, read_len;
    const char*   names[2]; const uint8* bases[2]; const uint8* quals[2];          // per mate: names '\0'-separated, ASCII bases / qualities, forward
    SlotSet       slots[2];
};

static void fill(const SlotSet& s, const uint32 n, thrust::host_vector<io::Alignment>& aln, io::HostCigarArray& cig, io::HostMdsArray& mds, thrust::host_vector<uint8>& mapq)
{
    aln.resize(n); mapq.resize(n); cig.coords.resize(n);
    cig.array.resize(n, s.cigar_offsets[n] + 1u); mds.resize(n, s.mds_offsets[n] + 1u);
    VectorArrayView<io::Cigar> cv = cig.array.plain_view(); VectorArrayView<uint8> mv = mds.plain_view();
    for (uint32 i = 0; i < n; ++i)
    {
        const uint32 w[2] = { s.words[i], s.aligns[i] };
        memcpy(&aln[i], w, 8u);
        mapq[i] = s.mapq[i];
        const uint32 nc = s.cigar_offsets[i + 1] - s.cigar_offsets[i];
        io::Cigar* c = cv.alloc(i, nc);
        for (uint32 k = 0; k < nc; ++k) { const uint16 op = s.cigar_ops[s.cigar_offsets[i] + k]; c[k] = io::Cigar(uint8(op & 3u), uint16(op >> 2)); }
        cig.coords[i] = make_uint2(s.cigar_source[i], nc);
        const uint32 nm = s.mds_offsets[i + 1] - s.mds_offsets[i];
        uint8* m = mv.alloc(i, nm);
        memcpy(m, s.mds_bytes + s.mds_offsets[i], nm);
    }
}

API int write_alignments(const WriteArgs* a)
{
    // the reference: names and lengths are all the writers look at
    io::SequenceDataHost ref;
    {
        io::SequenceDataEncoder enc(DNA, &ref); enc.begin_batch();
        const char* name = a->ref_names;
        for (uint32 k = 0; k < a->n_ref; ++k)
        {
            const std::string bases(a->ref_lengths[k], 'A');
            enc.push_back(a->ref_lengths[k], name, reinterpret_cast<const uint8*>(bases.data()), NULL, io::Phred33, uint32(-1), 0u, 0u, io::SequenceDataEncoder::NO_OP);
            name += strlen(name) + 1u;
        }
        enc.end_batch();
    }
    // the reads, stored reversed as nvBowtie loads them (nvBowtie.cpp:579)
    io::SequenceDataHost reads[2];
    for (uint32 m = 0; m < (a->paired ? 2u : 1u); ++m)
    {
        io::SequenceDataEncoder enc(DNA_N, &reads[m]); enc.begin_batch();
        const char* name = a->names[m];
        for (uint32 i = 0; i < a->n; ++i)
        {
            enc.push_back(a->read_len, name, a->bases[m] + uint64(i) * a->read_len, a->quals[m] + uint64(i) * a->read_len, io::Phred33, uint32(-1), 0u, 0u, io::SequenceDataEncoder::REVERSE_OP);
            name += strlen(name) + 1u;
        }
        enc.end_batch();
    }
    const io::ConstSequenceDataView ref_view(ref);
    io::OutputFile* out = io::OutputFile::open(a->path, a->paired ? io::PAIRED_END : io::SINGLE_END, io::BNT(ref_view));
    if (out == NULL) return 1;
    out->set_program("id", "test", "0.1", "args");
    out->header();
    // (a timing aid: NVBIO_IO_CALLERS_REPEAT=k hands the same batch to the writer k times)
    const char* rep_env = getenv("NVBIO_IO_CALLERS_REPEAT");
    const int   repeat  = rep_env ? std::max(atoi(rep_env), 1) : 1;
    if (a->paired)
    {
        io::HostOutputBatchPE batch; batch.count = a->n; batch.read_data[0] = &reads[0]; batch.read_data[1] = &reads[1];
        for (uint32 s = 0; s < 2u; ++s) fill(a->slots[s], a->n, batch.alignments[s], batch.cigar[s], batch.mds[s], batch.mapq[s]);
        for (int r = 0; r < repeat; ++r) out->process(batch);
    }
    else
    {
        io::HostOutputBatchSE batch; batch.count = a->n; batch.read_data = &reads[0];
        fill(a->slots[0], a->n, batch.alignments, batch.cigar, batch.mds, batch.mapq);
        for (int r = 0; r < repeat; ++r) out->process(batch);
    }
    out->close();
    delete out;
    return 0;
}
