"""nvbio_amd -- MI355X (gfx950) native seed-and-extend hot path with nvbio's interface.

The product is the C-ABI library `lib/libnvbio_hip.so` (hand-written HIP kernels,
declared in include/nvbio_hip.h).  This Python package is a thin host layer over
that ABI which mirrors the reference's names (nvbio::aln / nvbio::fm_index /
FMIndexFilter) and uses torch only for device memory and streams.  There is no
CPU fallback: importing the package without the built library raises.
"""
from ._lib import lib, LIB_PATH, check, set_test_switch, test_switch  # noqa: F401
from .strings import PackedStringSet, pack_symbols  # noqa: F401
from .alignment import (GLOBAL, LOCAL, SEMI_GLOBAL, PATTERN_BLOCKING, TEXT_BLOCKING, SimpleGotohScheme, SmithWatermanScoringScheme, GotohAligner,  # noqa: F401
                        make_gotoh_aligner, BatchedBandedAlignmentScore, batch_banded_alignment_score,
                        BatchedAlignmentScore, batch_alignment_score,
                        BatchedBandedAlignmentTraceback, batch_banded_alignment_traceback, batch_alignment_traceback,
                        SimpleSmithWatermanScheme, SmithWatermanAligner, EditDistanceAligner,
                        make_smith_waterman_aligner, make_edit_distance_aligner)
from .fmindex import FMIndexDevice, FMIndexFilter, rank, rank4, rank_range, match, locate, \
    locate_ssa_iterator, lookup_ssa_iterator, build_bwt_occ  # noqa: F401
from .mapping import MappingParams, map_exact, map_seeds, unpack_seed_hits  # noqa: F401
from .reduce import BestAlignments, score_reduce, score_reduce_paired, mapq, mapq_paired, opposite_mate_windows  # noqa: F401
