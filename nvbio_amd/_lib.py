"""ctypes binding of include/nvbio_hip.h.  Fails loudly when the HIP library is missing."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libnvbio_hip.so")

# every symbol include/nvbio_hip.h declares
SYMBOLS = [
    "nvbio_hip_banded_gotoh_score", "nvbio_hip_banded_gotoh_score_qual", "nvbio_hip_banded_gotoh_score_qual_views", "nvbio_hip_banded_gotoh_score_qual_bounded", "nvbio_hip_banded_gotoh_score_qual_wave", "nvbio_hip_gotoh_score", "nvbio_hip_banded_sw_score", "nvbio_hip_sw_score", "nvbio_hip_alignment_score", "nvbio_hip_alignment_score_qual", "nvbio_hip_alignment_score_qual_jobs",
    "nvbio_hip_banded_gotoh_traceback_temp_bytes", "nvbio_hip_banded_gotoh_traceback", "nvbio_hip_banded_gotoh_traceback_qual",
    "nvbio_hip_gotoh_traceback_temp_bytes", "nvbio_hip_gotoh_traceback", "nvbio_hip_gotoh_traceback_qual", "nvbio_hip_gotoh_traceback_known_score", "nvbio_hip_gotoh_traceback_qual_known_score", "nvbio_hip_known_score_redone", "nvbio_hip_banded_sw_traceback", "nvbio_hip_sw_traceback",
    "nvbio_hip_fm_rank", "nvbio_hip_fm_rank4", "nvbio_hip_fm_rank_range",
    "nvbio_hip_fm_match", "nvbio_hip_fm_build_ktab", "nvbio_hip_fm_dense_ssa_entries", "nvbio_hip_fm_build_dense_ssa",
    "nvbio_hip_banded_gotoh_score_host", "nvbio_hip_banded_sw_score_host", "nvbio_hip_alignment_score_host",
    "nvbio_hip_fm_rank_host", "nvbio_hip_fm_match_host", "nvbio_hip_fm_locate_host",
    "nvbio_hip_fm_dimer_index_bytes", "nvbio_hip_fm_build_dimer_index_temp_bytes", "nvbio_hip_fm_build_dimer_index", "nvbio_hip_fm_attach_dimer_index",
    "nvbio_hip_fm_trimer_index_bytes", "nvbio_hip_fm_build_trimer_index_temp_bytes", "nvbio_hip_fm_build_trimer_index", "nvbio_hip_fm_attach_trimer_index", "nvbio_hip_map_exact", "nvbio_hip_map",
    "nvbio_hip_alignment_invalid", "nvbio_hip_init_alignments", "nvbio_hip_score_reduce", "nvbio_hip_score_reduce_paired", "nvbio_hip_opposite_mate_windows", "nvbio_hip_mapq", "nvbio_hip_mapq_paired", "nvbio_hip_fm_locate",
    "nvbio_hip_sum_tree_node_count", "nvbio_hip_select_init", "nvbio_hip_select_init_queued", "nvbio_hip_select_temp_bytes", "nvbio_hip_select", "nvbio_hip_locate_hits", "nvbio_hip_hit_deque_replay",
    "nvbio_hip_score_best_setup", "nvbio_hip_score_reduce_best_approx",
    "nvbio_hip_anchor_score_setup", "nvbio_hip_anchor_score_finish", "nvbio_hip_anchor_memo_mark", "nvbio_hip_anchor_score_finish_memo", "nvbio_hip_anchor_memo_update", "nvbio_hip_opposite_score_setup", "nvbio_hip_opposite_score_finish",
    "nvbio_hip_score_reduce_paired_best_approx", "nvbio_hip_mark_discordant",
    "nvbio_hip_pack_read_queue", "nvbio_hip_mark_unaligned", "nvbio_hip_copy_flagged_temp_bytes", "nvbio_hip_copy_flagged", "nvbio_hip_traceback_best_setup", "nvbio_hip_traceback_best_setup_mates", "nvbio_hip_finish_alignment", "nvbio_hip_scatter_rows",
    "nvbio_hip_gather_ranges", "nvbio_hip_select_all", "nvbio_hip_mark_straddling", "nvbio_hip_score_all_setup", "nvbio_hip_score_all_output",
    "nvbio_hip_traceback_all_setup", "nvbio_hip_all_mapping_temp_bytes", "nvbio_hip_inclusive_scan_u32", "nvbio_hip_inclusive_scan_u64", "nvbio_hip_sort_hi_bits",
    "nvbio_hip_sort_hits", "nvbio_hip_sort_hits_pingpong", "nvbio_hip_gather_rows", "nvbio_hip_list_flagged", "nvbio_hip_opposite_memo_lookup", "nvbio_hip_opposite_memo_update", "nvbio_hip_traceback_best_known", "nvbio_hip_banded_gotoh_traceback_qual_known",
    "nvbio_hip_fm_locate_ssa_iterator", "nvbio_hip_fm_lookup_ssa_iterator",
    "nvbio_hip_fm_filter_temp_bytes", "nvbio_hip_fm_filter_rank", "nvbio_hip_fm_filter_locate",
    "nvbio_hip_build_bwt_occ_temp_bytes", "nvbio_hip_build_bwt_occ",
    "nvbio_hip_device_malloc", "nvbio_hip_device_free", "nvbio_hip_device_free_ordered", "nvbio_hip_device_free_after", "nvbio_hip_device_trim", "nvbio_hip_device_mem_info", "nvbio_hip_memcpy", "nvbio_hip_memset",
    "nvbio_hip_stream_synchronize", "nvbio_hip_stream_query", "nvbio_hip_host_malloc", "nvbio_hip_host_free", "nvbio_hip_stream_create", "nvbio_hip_stream_destroy",
    "nvbio_hip_comm_available", "nvbio_hip_device_count", "nvbio_hip_set_device", "nvbio_hip_get_device", "nvbio_hip_comm_unique_id", "nvbio_hip_comm_init_rank",
    "nvbio_hip_comm_init_all", "nvbio_hip_comm_destroy", "nvbio_hip_comm_rank", "nvbio_hip_gather_records", "nvbio_hip_comm_abort", "nvbio_hip_comm_set_transport",
    "nvbio_hip_abi_version", "nvbio_hip_arch", "nvbio_hip_last_kernel", "nvbio_hip_set_test_switch", "nvbio_hip_get_test_switch", "nvbio_hip_test_switch_name",
]


class PeParamsStruct(C.Structure):       # nvbio_hip_pe_params
    _fields_ = [("pe_policy", C.c_int32), ("min_frag_len", C.c_int32), ("max_frag_len", C.c_int32), ("pe_overlap", C.c_int32),
                ("score_limit", C.c_int32), ("anchor", C.c_uint32), ("genome_length", C.c_uint32)]


class StringSetStruct(C.Structure):      # nvbio_hip_string_set
    _fields_ = [("words", C.c_void_p), ("n_words", C.c_uint64), ("bits", C.c_uint32), ("big_endian", C.c_uint32),
                ("begin", C.c_void_p), ("length", C.c_void_p), ("fixed_length", C.c_uint32), ("_pad", C.c_uint32)]


class GotohSchemeStruct(C.Structure):    # nvbio_hip_gotoh_scheme
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap_open", C.c_int32), ("gap_ext", C.c_int32)]


class GotohQualSchemeStruct(C.Structure):  # nvbio_hip_gotoh_qual_scheme
    _fields_ = [("match", C.c_int32), ("pattern_gap_open", C.c_int32), ("pattern_gap_ext", C.c_int32),
                ("text_gap_open", C.c_int32), ("text_gap_ext", C.c_int32), ("mismatch", C.c_int32 * 256)]


class MapParamsStruct(C.Structure):       # nvbio_hip_map_params
    _fields_ = [(k, C.c_uint32) for k in ("seed_len", "min_read_len", "max_hits", "max_reseed", "retry", "rep_seeds", "fw", "rc")]


class FMIndexStruct(C.Structure):        # nvbio_hip_fmindex
    _fields_ = [("length", C.c_uint32), ("primary", C.c_uint32), ("L2", C.c_uint32 * 5), ("sa_int", C.c_uint32),
                ("bwt_occ", C.c_void_p), ("ssa", C.c_void_p), ("ktab", C.c_void_p), ("ktab_k", C.c_uint32), ("_pad", C.c_uint32),
                ("dimer", C.c_void_p), ("dimer_p1", C.c_uint32), ("dimer_fill1", C.c_uint32),
                ("dimer_S", C.c_uint32 * 4), ("dimer_T", C.c_uint32 * 4), ("trimer", C.c_void_p)]


_lib = None
ABI_VERSION = 2


def lib():
    """The loaded C-ABI library.  No fallback: a missing build is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "nvbio_amd: %s is missing -- build it with `python -m nvbio_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        # the version first: a stale library fails here with the message that says what to do, not on a symbol it lacks
        try:
            L.nvbio_hip_abi_version.restype = C.c_int
            found = L.nvbio_hip_abi_version()
        except AttributeError:
            found = None
        if found != ABI_VERSION:
            raise RuntimeError("nvbio_amd: lib/libnvbio_hip.so has ABI version %s, this package needs %d -- rebuild it (python -m nvbio_amd.build)" % (found, ABI_VERSION))
        for s in SYMBOLS:
            getattr(L, s)   # AttributeError if the library does not export the ABI
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
        P = C.POINTER
        L.nvbio_hip_banded_gotoh_score.argtypes = [P(GotohSchemeStruct), i32, u32, P(StringSetStruct), P(StringSetStruct), u32, u32, u32, vp, vp, vp]
        L.nvbio_hip_banded_gotoh_score_qual.argtypes = [P(GotohQualSchemeStruct), i32, u32, P(StringSetStruct), vp, u64, P(StringSetStruct), u32, u32, u32, vp, vp, vp]
        L.nvbio_hip_banded_gotoh_score_qual_views.argtypes = [P(GotohQualSchemeStruct), i32, u32, P(StringSetStruct), vp, u64, vp, P(StringSetStruct), u32, u32, u32, vp, vp, vp]
        L.nvbio_hip_banded_gotoh_score_qual_bounded.argtypes = [P(GotohQualSchemeStruct), i32, u32, P(StringSetStruct), vp, u64, vp, P(StringSetStruct), u32, u32, u32, vp, vp, vp, vp, vp, u32, vp, vp, vp]
        L.nvbio_hip_banded_gotoh_score_qual_wave.argtypes = [P(GotohQualSchemeStruct), i32, u32, P(StringSetStruct), vp, u64, P(StringSetStruct), u32, u32, vp, vp, vp, vp, u32, vp, vp, vp]
        L.nvbio_hip_banded_gotoh_traceback_temp_bytes.argtypes = [u32, u32, u32]
        L.nvbio_hip_banded_gotoh_traceback_temp_bytes.restype = u64
        L.nvbio_hip_banded_gotoh_traceback.argtypes = [P(GotohSchemeStruct), i32, u32, P(StringSetStruct), P(StringSetStruct), u32, u32, u32,
                                                       vp, vp, vp, vp, u32, vp, vp, u64, vp]
        L.nvbio_hip_banded_gotoh_traceback_qual.argtypes = [P(GotohQualSchemeStruct), i32, u32, P(StringSetStruct), vp, u64, P(StringSetStruct), u32, u32, u32,
                                                            vp, vp, vp, vp, u32, vp, vp, u64, vp]
        L.nvbio_hip_banded_gotoh_traceback_qual_known.argtypes = [P(GotohQualSchemeStruct), i32, u32, P(StringSetStruct), vp, u64, P(StringSetStruct), u32, u32, u32,
                                                            vp, vp, vp, vp, u32, vp, vp, u64, vp]
        L.nvbio_hip_banded_sw_score.argtypes = [P(GotohSchemeStruct), i32, u32, P(StringSetStruct), P(StringSetStruct), u32, u32, u32, vp, vp, vp]
        L.nvbio_hip_sw_score.argtypes = [P(GotohSchemeStruct), i32, P(StringSetStruct), P(StringSetStruct), u32, u32, u32, vp, vp, vp]
        L.nvbio_hip_alignment_score.argtypes = [i32, i32, vp, i32, P(StringSetStruct), P(StringSetStruct), u32, u32, vp, u32, vp, vp, vp, vp]
        L.nvbio_hip_gotoh_traceback_temp_bytes.argtypes = [u32, u32, u32]
        L.nvbio_hip_gotoh_traceback_temp_bytes.restype = u64
        L.nvbio_hip_gotoh_traceback.argtypes = [P(GotohSchemeStruct), i32, P(StringSetStruct), P(StringSetStruct), u32, u32, u32,
                                                vp, vp, vp, vp, u32, vp, vp, u64, vp]
        L.nvbio_hip_gotoh_traceback_qual.argtypes = [P(GotohQualSchemeStruct), i32, P(StringSetStruct), vp, u64, P(StringSetStruct), u32, u32, u32,
                                                     vp, vp, vp, vp, u32, vp, vp, u64, vp]
        L.nvbio_hip_gotoh_traceback_known_score.argtypes = [P(GotohSchemeStruct), i32, P(StringSetStruct), P(StringSetStruct), vp, u32, u32, u32,
                                                            vp, vp, vp, vp, u32, vp, vp, u64, vp]
        L.nvbio_hip_gotoh_traceback_qual_known_score.argtypes = [P(GotohQualSchemeStruct), i32, P(StringSetStruct), vp, u64, P(StringSetStruct), vp, u32, u32, u32,
                                                                 vp, vp, vp, vp, u32, vp, vp, u64, vp]
        L.nvbio_hip_known_score_redone.argtypes = []; L.nvbio_hip_known_score_redone.restype = u64
        L.nvbio_hip_banded_sw_traceback.argtypes = [P(GotohSchemeStruct), i32, u32, P(StringSetStruct), P(StringSetStruct), u32, u32, u32,
                                                    vp, vp, vp, vp, u32, vp, vp, u64, vp]
        L.nvbio_hip_sw_traceback.argtypes = [P(GotohSchemeStruct), i32, P(StringSetStruct), P(StringSetStruct), u32, u32, u32,
                                             vp, vp, vp, vp, u32, vp, vp, u64, vp]
        L.nvbio_hip_alignment_score_qual.argtypes = [P(GotohQualSchemeStruct), i32, i32, P(StringSetStruct), vp, u64, P(StringSetStruct), u32, u32, vp, u32, vp, vp, vp, vp]
        L.nvbio_hip_gotoh_score.argtypes = [P(GotohSchemeStruct), i32, P(StringSetStruct), P(StringSetStruct), u32, u32, vp, u32, vp, vp, vp, vp]
        L.nvbio_hip_fm_rank.argtypes = [P(FMIndexStruct), vp, vp, u32, vp, vp]
        L.nvbio_hip_fm_rank4.argtypes = [P(FMIndexStruct), vp, u32, vp, vp]
        L.nvbio_hip_fm_rank_range.argtypes = [P(FMIndexStruct), vp, vp, u32, vp, vp]
        L.nvbio_hip_fm_match.argtypes = [P(FMIndexStruct), P(StringSetStruct), u32, vp, vp]
        L.nvbio_hip_fm_build_ktab.argtypes = [P(FMIndexStruct), u32, vp, vp]
        L.nvbio_hip_banded_gotoh_score_host.argtypes = [P(GotohSchemeStruct), i32, u32, P(StringSetStruct), P(StringSetStruct), u32, vp, vp, i32]
        L.nvbio_hip_banded_sw_score_host.argtypes = [P(GotohSchemeStruct), i32, u32, P(StringSetStruct), P(StringSetStruct), u32, vp, vp, i32]
        L.nvbio_hip_alignment_score_host.argtypes = [i32, i32, vp, i32, P(StringSetStruct), P(StringSetStruct), vp, u32, vp, vp, vp, i32]
        L.nvbio_hip_fm_rank_host.argtypes = [P(FMIndexStruct), vp, vp, u32, vp, i32]
        L.nvbio_hip_fm_match_host.argtypes = [P(FMIndexStruct), P(StringSetStruct), u32, vp, i32]
        L.nvbio_hip_fm_locate_host.argtypes = [P(FMIndexStruct), vp, u32, vp, i32]
        L.nvbio_hip_fm_dimer_index_bytes.argtypes = [u32]; L.nvbio_hip_fm_dimer_index_bytes.restype = u64
        L.nvbio_hip_fm_build_dimer_index_temp_bytes.argtypes = [u32]; L.nvbio_hip_fm_build_dimer_index_temp_bytes.restype = u64
        L.nvbio_hip_fm_build_dimer_index.argtypes = [P(FMIndexStruct), vp, vp, u64, vp]
        L.nvbio_hip_fm_attach_dimer_index.argtypes = [P(FMIndexStruct), vp, vp]
        L.nvbio_hip_fm_trimer_index_bytes.argtypes = [u32]; L.nvbio_hip_fm_trimer_index_bytes.restype = u64
        L.nvbio_hip_fm_build_trimer_index_temp_bytes.argtypes = [u32]; L.nvbio_hip_fm_build_trimer_index_temp_bytes.restype = u64
        L.nvbio_hip_fm_build_trimer_index.argtypes = [P(FMIndexStruct), vp, vp, u64, vp]
        L.nvbio_hip_fm_attach_trimer_index.argtypes = [P(FMIndexStruct), vp, vp]
        L.nvbio_hip_map_exact.argtypes = [P(FMIndexStruct), P(StringSetStruct), vp, u32, P(MapParamsStruct), vp, vp, u32, vp, vp, vp]
        L.nvbio_hip_map.argtypes = [i32, u32, P(FMIndexStruct), P(FMIndexStruct), P(StringSetStruct), vp, u32, P(MapParamsStruct), vp, vp, u32, vp, vp, vp]
        L.nvbio_hip_alignment_invalid.argtypes = []
        L.nvbio_hip_alignment_invalid.restype = u64
        L.nvbio_hip_init_alignments.argtypes = [u32, vp, u32, vp, u32, vp, u32, vp]
        L.nvbio_hip_score_reduce.argtypes = [u32, vp, vp, vp, vp, vp, vp, u32, vp, u32, vp]
        L.nvbio_hip_sum_tree_node_count.argtypes = [u32]; L.nvbio_hip_sum_tree_node_count.restype = u32
        L.nvbio_hip_select_init.argtypes = [u32, vp, vp, vp, u32, vp, vp, u32, vp, vp, u32, i32, i32, vp]
        L.nvbio_hip_select_init_queued.argtypes = [u32, vp, vp, vp, vp, u32, vp, vp, u32, vp, vp, u32, i32, i32, vp]
        L.nvbio_hip_select_temp_bytes.argtypes = [u32, u32]; L.nvbio_hip_select_temp_bytes.restype = u64
        L.nvbio_hip_select.argtypes = [i32, u32, vp, u32, vp, u32, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, vp]
        L.nvbio_hip_hit_deque_replay.argtypes = [u32, vp, vp, vp, vp, vp, vp, u32, vp, vp]
        L.nvbio_hip_locate_hits.argtypes = [P(FMIndexStruct), P(FMIndexStruct), u32, vp, vp, vp]
        L.nvbio_hip_score_best_setup.argtypes = [u32, vp, vp, vp, vp, vp, u32, u64, u32, u32, vp, u32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.nvbio_hip_score_reduce_best_approx.argtypes = [u32, vp, vp, vp, vp, vp, vp, u32, vp, u32, i32, vp, vp, u32, u32, u32, u32, vp, vp, vp, vp]
        L.nvbio_hip_traceback_best_known.argtypes = [u32, vp, vp, vp, vp, vp, vp]
        L.nvbio_hip_anchor_score_setup.argtypes = [u32, vp, vp, vp, vp, vp, vp, u32, u32, u64, u32, u32, vp, vp, u32, i32, vp, i32, u32, vp, vp, vp, vp, vp, vp]
        L.nvbio_hip_anchor_score_finish.argtypes = [u32, vp, vp, vp, vp, i32, vp, vp, vp]
        L.nvbio_hip_opposite_score_setup.argtypes = [u32, vp, vp, vp, vp, i32, vp, vp, u32, u32, vp, vp, u32, i32, vp, i32, i32, P(PeParamsStruct), vp, vp, vp, vp, vp, vp, u64, vp, vp, vp, vp]
        L.nvbio_hip_scatter_rows.argtypes = [u32, vp, vp, vp, u32, vp]
        L.nvbio_hip_gather_rows.argtypes = [u32, vp, vp, vp, u32, vp]
        L.nvbio_hip_opposite_memo_lookup.argtypes = [u32, vp, vp, vp, vp, vp, vp, u32, vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.nvbio_hip_opposite_memo_update.argtypes = [u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, vp, vp]
        L.nvbio_hip_gather_ranges.argtypes = [u32, u32, vp, u32, vp, vp, vp]
        L.nvbio_hip_select_all.argtypes = [u64, u32, u32, u32, vp, u32, vp, vp, vp, vp, vp, vp]
        L.nvbio_hip_mark_straddling.argtypes = [u32, vp, u32, vp, vp, u32, vp, vp]
        L.nvbio_hip_score_all_setup.argtypes = [u32, vp, vp, vp, vp, vp, vp, u32, u64, u32, u32, vp, vp, vp, vp, vp]
        L.nvbio_hip_score_all_output.argtypes = [u32, vp, vp, vp, vp, vp, vp, vp, u32, vp, vp, vp, vp]
        L.nvbio_hip_traceback_all_setup.argtypes = [u32, vp, vp, vp, vp, u32, u64, u32, u32, vp, vp, vp, vp, vp]
        L.nvbio_hip_all_mapping_temp_bytes.argtypes = [u32]; L.nvbio_hip_all_mapping_temp_bytes.restype = u64
        L.nvbio_hip_inclusive_scan_u32.argtypes = [u32, vp, vp, vp, u64, vp]
        L.nvbio_hip_inclusive_scan_u64.argtypes = [u32, vp, vp, vp, u64, vp]
        L.nvbio_hip_sort_hi_bits.argtypes = [u32, vp, vp, vp, u64, vp]
        L.nvbio_hip_sort_hits.argtypes = [u32, vp, vp, vp, vp, vp, vp, u64, vp]
        L.nvbio_hip_sort_hits_pingpong.argtypes = [u32, vp, vp, vp, vp, vp, vp, vp, u64, vp]
        L.nvbio_hip_opposite_score_finish.argtypes = [u32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
        L.nvbio_hip_score_reduce_paired_best_approx.argtypes = [u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, i32, i32, i32, vp, vp, u32, vp, vp, u32, u32, u32, u32, vp]
        L.nvbio_hip_mark_discordant.argtypes = [u32, vp, vp, u32, vp]
        L.nvbio_hip_pack_read_queue.argtypes = [u32, vp, u32, vp, vp]
        L.nvbio_hip_mark_unaligned.argtypes = [u32, vp, vp, vp, vp]
        L.nvbio_hip_copy_flagged_temp_bytes.argtypes = [u32]; L.nvbio_hip_copy_flagged_temp_bytes.restype = u64
        L.nvbio_hip_copy_flagged.argtypes = [u32, vp, vp, vp, vp, vp, u64, vp]
        L.nvbio_hip_finish_alignment.argtypes = [u32, vp, P(StringSetStruct), vp, u64, P(StringSetStruct), vp, u32, vp, vp, i32, vp, i32, vp, vp, vp, vp, u32, vp, vp]
        L.nvbio_hip_traceback_best_setup.argtypes = [u32, vp, vp, u32, u32, vp, vp, u32, u64, u64, i32, vp, vp, vp, vp, vp, vp]
        L.nvbio_hip_traceback_best_setup_mates.argtypes = [u32, vp, vp, u32, u32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                                           u64, i32, vp, vp, vp, vp, vp, vp]
        L.nvbio_hip_score_reduce_paired.argtypes = [u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, i32, i32, i32, vp, vp, u32, vp]
        L.nvbio_hip_opposite_mate_windows.argtypes = [u32, vp, vp, vp, vp, vp, vp, u32, u32, vp, vp, u32, i32, vp, i32, i32, P(PeParamsStruct),
                                                      vp, vp, vp, vp, vp, vp]
        L.nvbio_hip_mapq.argtypes = [i32, i32, i32, vp, u32, vp, u32, vp, u32, vp, vp]
        L.nvbio_hip_mapq_paired.argtypes = [i32, i32, i32, vp, u32, vp, vp, u32, vp, vp, u32, u32, vp, vp]
        L.nvbio_hip_fm_locate.argtypes = [P(FMIndexStruct), vp, u32, vp, vp]
        L.nvbio_hip_fm_locate_ssa_iterator.argtypes = [P(FMIndexStruct), vp, u32, vp, vp]
        L.nvbio_hip_fm_lookup_ssa_iterator.argtypes = [P(FMIndexStruct), vp, u32, vp, vp]
        L.nvbio_hip_fm_filter_temp_bytes.argtypes = [u32]
        L.nvbio_hip_fm_filter_temp_bytes.restype = u64
        L.nvbio_hip_fm_filter_rank.argtypes = [P(FMIndexStruct), P(StringSetStruct), u32, vp, vp, vp, u64, vp]
        L.nvbio_hip_fm_filter_locate.argtypes = [P(FMIndexStruct), vp, vp, u32, u64, u64, vp, vp]
        L.nvbio_hip_build_bwt_occ_temp_bytes.argtypes = [u32]
        L.nvbio_hip_build_bwt_occ_temp_bytes.restype = u64
        L.nvbio_hip_build_bwt_occ.argtypes = [u32, vp, vp, vp, vp, u64, vp]
        L.nvbio_hip_comm_unique_id.argtypes = [vp]
        L.nvbio_hip_comm_init_rank.argtypes = [P(vp), C.c_int, C.c_int, vp]
        L.nvbio_hip_comm_init_all.argtypes = [vp, C.c_int, vp]
        L.nvbio_hip_comm_destroy.argtypes = [vp]
        L.nvbio_hip_comm_rank.argtypes = [vp, P(C.c_int), P(C.c_int)]
        L.nvbio_hip_gather_records.argtypes = [vp, vp, vp, u32, vp, C.c_int, vp]
        L.nvbio_hip_set_device.argtypes = [C.c_int]
        L.nvbio_hip_abi_version.restype = C.c_int
        L.nvbio_hip_set_test_switch.argtypes = [C.c_char_p, C.c_int]
        L.nvbio_hip_get_test_switch.argtypes = [C.c_char_p]
        L.nvbio_hip_fm_dense_ssa_entries.argtypes = [u32, u32]
        L.nvbio_hip_fm_dense_ssa_entries.restype = u64
        L.nvbio_hip_fm_build_dense_ssa.argtypes = [P(FMIndexStruct), u32, vp, vp]
        L.nvbio_hip_test_switch_name.argtypes = [C.c_int]
        L.nvbio_hip_test_switch_name.restype = C.c_char_p
        L.nvbio_hip_device_free_after.argtypes = [vp, vp]
        L.nvbio_hip_device_mem_info.argtypes = [P(u64), P(u64), P(u64)]
        L.nvbio_hip_arch.restype = C.c_char_p
        L.nvbio_hip_last_kernel.restype = C.c_char_p
        _lib = L
    return _lib


def check(err, what):
    if err != 0:
        raise RuntimeError("nvbio_amd: %s failed with hipError %d" % (what, err))


def test_switch_names():
    """the names of the library's test switches, as the library lists them (nvbio_hip_test_switch_name)"""
    out, k = [], 0
    while True:
        n = lib().nvbio_hip_test_switch_name(k)
        if not n:
            return out
        out.append(n.decode()); k += 1


def device_mem_info():
    """(free, total, idle bytes in the library's block cache) of the current device"""
    f, t, i = C.c_uint64(), C.c_uint64(), C.c_uint64()
    check(lib().nvbio_hip_device_mem_info(C.byref(f), C.byref(t), C.byref(i)), "nvbio_hip_device_mem_info")
    return f.value, t.value, i.value


def set_test_switch(name, value):
    """One of the library's test switches (include/nvbio_hip.h, 'Test switches'): an integer, 0 = the default execution.  The switches are seeded
    once from the environment and changed only through this call afterwards."""
    check(lib().nvbio_hip_set_test_switch(name.encode(), int(value)), "nvbio_hip_set_test_switch(%s)" % name)


class test_switch:
    """with test_switch("NVBIO_HIP_FORCE_32BIT", 1): ...   -- the switch is restored on exit"""

    def __init__(self, name, value):
        self.name, self.value = name, int(value)

    def __enter__(self):
        self.saved = lib().nvbio_hip_get_test_switch(self.name.encode())
        set_test_switch(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_test_switch(self.name, self.saved)
        return False


def current_stream_ptr():
    import torch
    if not torch.cuda.is_available():
        return C.c_void_p(0)          # host-only use of the transport seam (the CPU suite's gather over host memory)
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
