"""ctypes binding of the C ABI in include/lambda_ext.h (liblambda_ext.so).

This is the only way Python code reaches the HIP path: tests, bench.py and the smoke test call the same exported
symbols a C++ maintainer of the reference would link against.  If the shared library is missing
or no gfx950 device is usable, everything here raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

LIB_PATH = Path(os.environ.get("LX_LIB_PATH") or Path(__file__).resolve().parent / "csrc" / "liblambda_ext.so")  # (override: kernel variants during development)

LX_ALPH = 32
LX_OK = 0
LX_EINVAL = -1
LX_OPT_MAX_QLEN = 1
LX_OPT_QUERY_RUN = 2
LX_OPT_WORKSPACE_BYTES = 3

# every symbol include/lambda_ext.h declares (tests/test_abi.py checks that the library exports all of them)
EXPORTED_SYMBOLS = [
    "lx_abi_version", "lx_build_id", "lx_device_count", "lx_create", "lx_destroy", "lx_last_error", "lx_set_option", "lx_get_option", "lx_host_threads_info", "lx_plan_free_packing_bound", "lx_plan_free_packing_dev", "lx_set_band_centres", "lx_set_band_centres_dev",
    "lx_set_scoring", "lx_builtin_scoring", "lx_score_batch", "lx_score_batch_dev", "lx_align_batch",
    "lx_align_batch_dev", "lx_extend_batch_dev", "lx_prefilter_batch", "lx_synchronize", "lx_last_kernel_ms", "lx_last_kernel_name", "lx_last_trace_kernel_name", "lx_last_phase_ms",
    "lx_iterate_matches", "lx_set_queries", "lx_set_subject_seqs", "lx_iterate_matches_dev", "lx_widen_and_preprocess_dev", "lx_reserve", "lx_sort_words_dev", "lx_trim_result_cache",
    "lx_iterate_result_count", "lx_iterate_result_matches", "lx_iterate_result_ops", "lx_iterate_result_stats",
    "lx_iterate_result_free", "lx_karlin_params", "lx_length_adjustment", "lx_evalue", "lx_bitscore",
    "lx_widen_and_preprocess", "lx_postprocess_records", "lx_compute_lca", "lx_write_records", "lx_convert_ranks",
    "lx_set_subjects", "lx_extend_batch", "lx_extend_batch_rle", "lx_extend_batch_list", "lx_write_records_ex", "lx_check_output_options", "lx_write_footer", "lx_output_options_default", "lx_last_output_error", "lx_expand_ops", "lx_last_extend_stats", "lx_set_frames", "lx_untrue_qry_id", "lx_untrue_subj_id", "lx_translate_six_frames",
    "lx_plan_step",
]

LX_OPT_MAX_SLEN = 4
LX_OPT_TRACE_BYTES = 5
LX_OPT_BS_MATCH_RULE = 6
LX_OPT_PACKED_HALF = 7
LX_OPT_PASS2_MODE = 8
LX_OPT_BAND = 9
LX_OPT_EXTEND_CHUNK = 10
LX_OPT_MQ_SWEEP = 11
LX_OPT_ITERATE_RECORDS = 13
LX_OPT_HOST_THREADS = 14
LX_OPT_ADAPT_PERMILLE = 12


class Karlin(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("K", C.c_double), ("H", C.c_double), ("alpha", C.c_double),
                ("beta", C.c_double)]


class SearchParams(C.Structure):
    _fields_ = [("max_evalue", C.c_double), ("min_bitscore", C.c_int32), ("id_cutoff", C.c_int32),
                ("db_total_length", C.c_uint64), ("query_translated", C.c_int32), ("qry_num_frames", C.c_int32),
                ("sbj_num_frames", C.c_int32), ("bisulfite", C.c_int32), ("q_frame_mode", C.c_int32),
                ("s_frame_mode", C.c_int32), ("karlin", Karlin), ("band", C.c_int32), ("flags", C.c_int32)]


LX_ITERATE_NO_OPS = 1


class IterateStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("hits_duplicate", "failed_bitscore", "failed_evalue", "failed_identity",
                                          "num_ext_score", "num_ext_ali")]


BLAST_MATCH_DTYPE = np.dtype([
    ("qry_id", "<u8"), ("subj_id", "<u8"), ("n_qid", "<u8"), ("n_sid", "<u8"), ("q_start", "<u8"), ("q_end", "<u8"),
    ("s_start", "<u8"), ("s_end", "<u8"), ("score", "<i4"), ("alignment_length", "<i4"), ("num_matches", "<i4"),
    ("num_mismatches", "<i4"), ("num_positives", "<i4"), ("num_gap_opens", "<i4"), ("num_gap_extensions", "<i4"),
    ("identity", "<f4"), ("bit_score", "<f8"), ("e_value", "<f8"), ("ops_off", "<u8"), ("n_ops", "<u4"),
    ("q_frame", "<i2"), ("s_frame", "<i2")])


class RecordStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("qrys_with_hit", "hits_duplicate2", "hits_abundant", "hits_final", "pairs")]


class TaxTree(C.Structure):
    _fields_ = [("parents", C.c_void_p), ("heights", C.c_void_p), ("n_taxa", C.c_uint64), ("s_tax_off", C.c_void_p),
                ("s_tax_ids", C.c_void_p), ("n_s", C.c_uint64)]


class SeqNames(C.Structure):
    _fields_ = [("q_ids", C.POINTER(C.c_char_p)), ("q_lens", C.c_void_p), ("s_ids", C.POINTER(C.c_char_p)),
                ("s_lens", C.c_void_p), ("n_q", C.c_uint64), ("n_s", C.c_uint64)]


LX_OUT_BLAST_TAB, LX_OUT_BLAST_TAB_COMMENTS, LX_OUT_SAM = 0, 1, 2


class Scoring(C.Structure):
    _fields_ = [("alphabet_size", C.c_int32), ("gap_open", C.c_int32), ("gap_extend", C.c_int32),
                ("reserved", C.c_int32), ("matrix", C.c_int8 * (LX_ALPH * LX_ALPH))]

    def matrix_np(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.matrix).reshape(LX_ALPH, LX_ALPH).copy()


EXT_DTYPE = np.dtype([("q_off", "<u8"), ("s_off", "<u8"), ("q_len", "<u4"), ("s_len", "<u4")])
class SurvivorList(C.Structure):
    """lx_survivor_list (include/lambda_ext.h)."""
    _fields_ = [("count", C.c_uint64), ("index", C.c_void_p), ("hsp", C.c_void_p), ("codes_off", C.c_void_p), ("codes", C.c_void_p),
                ("codes_bytes", C.c_uint64)]


HSP_DTYPE = np.dtype([("score", "<i4"), ("q_begin", "<i4"), ("q_end", "<i4"), ("s_begin", "<i4"), ("s_end", "<i4"),
                      ("n_ops", "<i4"), ("num_matches", "<i4"), ("num_mismatches", "<i4"), ("num_positives", "<i4"),
                      ("num_gap_opens", "<i4"), ("num_gap_extensions", "<i4"), ("ops_shift", "<i4")])
MATCH_DTYPE = np.dtype([("qryId", "<u8"), ("subjId", "<u8"), ("qryStart", "<u8"), ("qryEnd", "<u8"),
                        ("subjStart", "<u8"), ("subjEnd", "<u8")])
SEED_DTYPE = np.dtype([("q_off", "<u8"), ("s_off", "<u8"), ("q_len", "<u4"), ("s_len", "<u4"),
                       ("qry_start", "<u4"), ("qry_end", "<u4"), ("subj_start", "<u4"), ("reserved", "<u4")])

_lib = None


class LambdaExtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"lambda_ext error {code}: {msg}")
        self.code = code


def load():
    """Loads liblambda_ext.so (raises if it has not been built: the product path must fail loudly)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(f"{LIB_PATH} not built -- run `python -m lambda_amd.build` (needs hipcc)")
    # PyTorch wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one process do not work
    # (the second one sees no GPU), so if torch is going to be used in this process it must be loaded first; our
    # library then binds to the runtime that is already resident (same SONAME).  A C++ host links /opt/rocm's.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing for tests/bench only, never a requirement of the C ABI
        pass
    lib = C.CDLL(str(LIB_PATH))
    vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
    lib.lx_abi_version.restype = i32
    lib.lx_build_id.restype = C.c_char_p
    lib.lx_device_count.restype = i32
    lib.lx_create.argtypes = [i32, C.POINTER(vp)]
    lib.lx_destroy.argtypes = [vp]
    lib.lx_destroy.restype = None
    lib.lx_last_error.argtypes = [vp]
    lib.lx_last_error.restype = C.c_char_p
    lib.lx_set_option.argtypes = [vp, i32, u64]
    lib.lx_get_option.argtypes = [vp, i32, C.POINTER(u64)]
    lib.lx_host_threads_info.argtypes = [C.POINTER(C.c_uint32)] * 3
    lib.lx_plan_free_packing_bound.argtypes = [u64, u64, C.c_uint32]
    lib.lx_plan_free_packing_bound.restype = u64
    lib.lx_plan_free_packing_dev.argtypes = [vp, vp, u64, u64, i32, C.c_uint32, C.POINTER(u64), vp, vp, vp, vp]
    lib.lx_set_band_centres.argtypes = [vp, vp, u64]
    lib.lx_set_band_centres_dev.argtypes = [vp, vp]
    lib.lx_set_scoring.argtypes = [vp, i32, C.POINTER(Scoring)]
    lib.lx_builtin_scoring.argtypes = [i32, i32, i32, i32, i32, C.POINTER(Scoring)]
    lib.lx_score_batch.argtypes = [vp, i32, vp, u64, vp, u64, vp, u64, vp]
    lib.lx_score_batch_dev.argtypes = [vp, i32, vp, vp, vp, u64, vp, vp]
    lib.lx_synchronize.argtypes = [vp]
    lib.lx_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.lx_last_kernel_name.argtypes = [vp]
    lib.lx_last_kernel_name.restype = C.c_char_p
    lib.lx_last_trace_kernel_name.argtypes = [vp]
    lib.lx_last_trace_kernel_name.restype = C.c_char_p
    lib.lx_last_phase_ms.argtypes = [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    for name, args in (("lx_align_batch", [vp, i32, vp, u64, vp, u64, vp, u64, vp, vp, vp, vp]),
                       ("lx_align_batch_dev", [vp, i32, vp, vp, vp, u64, vp, vp, vp, vp]),
                       ("lx_prefilter_batch", [vp, i32, vp, u64, vp, u64, vp, u64, C.c_uint32, C.c_int32, C.c_double, vp])):
        if hasattr(lib, name):
            getattr(lib, name).argtypes = args
    lib.lx_extend_batch_dev.argtypes = [vp, i32, vp, vp, vp, u64, vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    lib.lx_karlin_params.argtypes = [i32, i32, i32, i32, i32, C.POINTER(Karlin)]
    lib.lx_length_adjustment.argtypes = [u64, u64, C.POINTER(Karlin)]
    lib.lx_length_adjustment.restype = u64
    lib.lx_evalue.argtypes = [C.c_int32, u64, u64, C.POINTER(Karlin)]
    lib.lx_evalue.restype = C.c_double
    lib.lx_bitscore.argtypes = [C.c_int32, C.POINTER(Karlin)]
    lib.lx_bitscore.restype = C.c_double
    lib.lx_convert_ranks.argtypes = [i32, vp, u64, vp]
    lib.lx_set_subjects.argtypes = [vp, vp, u64]
    lib.lx_extend_batch.argtypes = [vp, i32, vp, u64, vp, u64, vp, u64, vp, i32, vp, vp, vp, C.POINTER(vp), C.POINTER(u64)]
    lib.lx_extend_batch_rle.argtypes = lib.lx_extend_batch.argtypes
    lib.lx_extend_batch_list.argtypes = [vp, i32, vp, u64, vp, u64, vp, u64, vp, i32, vp, C.POINTER(SurvivorList)]
    lib.lx_expand_ops.argtypes = [vp, i32, vp]
    lib.lx_last_extend_stats.argtypes = [vp, vp]
    lib.lx_set_frames.argtypes = [i32, i32, u64, u64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.lx_set_frames.restype = None
    lib.lx_untrue_qry_id.argtypes = [i32, u64, i32]
    lib.lx_untrue_qry_id.restype = u64
    lib.lx_untrue_subj_id.argtypes = [i32, u64, i32]
    lib.lx_untrue_subj_id.restype = u64
    lib.lx_translate_six_frames.argtypes = [vp, u64, i32, vp, u64, vp, vp]
    lib.lx_widen_and_preprocess.argtypes = [vp, u64, vp, vp]
    lib.lx_widen_and_preprocess.restype = u64
    lib.lx_iterate_matches.argtypes = [vp, i32, vp, u64, vp, vp, u64, vp, vp, u64, vp, vp, u64, vp, u64,
                                       C.POINTER(SearchParams), C.POINTER(vp)]
    lib.lx_set_queries.argtypes = [vp, vp, u64, vp, vp, u64, vp, i32]
    lib.lx_set_subject_seqs.argtypes = [vp, vp, vp, u64]
    lib.lx_iterate_matches_dev.argtypes = [vp, i32, vp, u64, C.POINTER(SearchParams), C.POINTER(vp)]
    lib.lx_trim_result_cache.restype = u64
    lib.lx_trim_result_cache.argtypes = []
    lib.lx_reserve.restype = i32
    lib.lx_reserve.argtypes = [vp, u64, u64, u64, u64]
    lib.lx_widen_and_preprocess_dev.argtypes = [vp, vp, u64, i32, vp, C.POINTER(u64)]
    lib.lx_iterate_result_count.argtypes = [vp]
    lib.lx_iterate_result_count.restype = u64
    lib.lx_iterate_result_matches.argtypes = [vp]
    lib.lx_iterate_result_matches.restype = vp
    lib.lx_iterate_result_ops.argtypes = [vp]
    lib.lx_iterate_result_ops.restype = vp
    lib.lx_iterate_result_stats.argtypes = [vp]
    lib.lx_iterate_result_stats.restype = IterateStats
    lib.lx_iterate_result_free.argtypes = [vp]
    lib.lx_iterate_result_free.restype = None
    lib.lx_postprocess_records.argtypes = [vp, u64, u64, C.POINTER(RecordStats)]
    lib.lx_postprocess_records.restype = u64
    lib.lx_write_records.argtypes = [C.c_char_p, i32, i32, C.c_char_p, vp, u64, vp, C.POINTER(SeqNames), vp, vp]
    lib.lx_write_records_ex.argtypes = [C.c_char_p, i32, i32, C.c_char_p, vp, u64, vp, C.POINTER(SeqNames), vp, vp, C.POINTER(OutputOptions)]
    lib.lx_write_footer.argtypes = [C.c_char_p, i32, u64]
    lib.lx_output_options_default.argtypes = [C.POINTER(OutputOptions)]
    lib.lx_output_options_default.restype = None
    lib.lx_last_output_error.restype = C.c_char_p
    lib.lx_compute_lca.argtypes = [vp, u64, C.POINTER(TaxTree), vp, vp, C.POINTER(u64)]
    _lib = lib
    return lib


def compute_lca(bms: np.ndarray, parents, heights, s_tax_off, s_tax_ids):
    """The LCA step of _writeRecord (src/search_algo.hpp:884-907) over a result list grouped by query: (n_qid, lcaTaxId) arrays."""
    m = np.ascontiguousarray(bms, dtype=BLAST_MATCH_DTYPE)
    par = np.ascontiguousarray(parents, dtype=np.uint32)
    hgt = np.ascontiguousarray(heights, dtype=np.uint32)
    off = np.ascontiguousarray(s_tax_off, dtype=np.uint64)
    ids = np.ascontiguousarray(s_tax_ids, dtype=np.uint32)
    tree = TaxTree(par.ctypes.data, hgt.ctypes.data, len(par), off.ctypes.data, ids.ctypes.data if len(ids) else None, len(off) - 1)
    qid = np.zeros(max(len(m), 1), dtype=np.uint64)
    lca = np.zeros(max(len(m), 1), dtype=np.uint32)
    cnt = C.c_uint64(0)
    rc = load().lx_compute_lca(_ptr(m) if len(m) else None, len(m), C.byref(tree), _ptr(qid), _ptr(lca), C.byref(cnt))
    if rc != 0:
        raise LambdaExtError(rc, "lx_compute_lca: ids outside the taxonomy, or a path that does not lead to the root")
    return qid[: cnt.value].copy(), lca[: cnt.value].copy()


def postprocess_records(bms: np.ndarray, max_matches: int = 25):
    """_writeRecord's sort / dedupe / top-N (src/search_algo.hpp:820-913) over a result list grouped by query."""
    m = np.ascontiguousarray(bms, dtype=BLAST_MATCH_DTYPE).copy()
    st = RecordStats()
    n = load().lx_postprocess_records(_ptr(m), len(m), max_matches, C.byref(st))
    return m[: int(n)], st


class OutputOptions(C.Structure):
    """lx_output_options (include/lambda_ext.h): the reference's output options, src/search_options.hpp:224-379."""
    _fields_ = [("columns", C.c_char_p), ("sam_tags", C.c_char_p), ("sam_seq", C.c_int32), ("sam_hard_clip", C.c_int32),
                ("sam_with_ref_header", C.c_int32), ("version_to_output", C.c_int32), ("version", C.c_char_p), ("command_line", C.c_char_p),
                ("db_name", C.c_char_p), ("genetic_code", C.c_int32), ("reserved", C.c_int32), ("tax", C.c_void_p), ("lca_qid", C.c_void_p),
                ("lca_tax", C.c_void_p), ("n_lca", C.c_uint64), ("tax_names", C.c_void_p)]


LX_SAM_SEQ_NEVER, LX_SAM_SEQ_UNIQ, LX_SAM_SEQ_ALWAYS = 0, 1, 2


def output_options(**kw) -> OutputOptions:
    """lx_output_options_default, then the given fields (str values are encoded)."""
    o = OutputOptions()
    load().lx_output_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v.encode() if isinstance(v, str) else v)
    return o


def last_output_error() -> str:
    return (load().lx_last_output_error() or b"").decode()


def write_footer(path, fmt: int, n_records: int):
    rc = load().lx_write_footer(str(path).encode(), fmt, C.c_uint64(n_records))
    if rc != LX_OK:
        raise LambdaExtError(rc, "lx_write_footer")


def write_records(path, fmt: int, bms: np.ndarray, ops: bytes, q_ids, q_lens, s_ids, s_lens, program="blastp",
                  write_header=True, q_ascii: bytes | None = None, q_ascii_off=None, options: OutputOptions | None = None):
    m = np.ascontiguousarray(bms, dtype=BLAST_MATCH_DTYPE)
    qa = (C.c_char_p * len(q_ids))(*[x.encode() for x in q_ids])
    sa = (C.c_char_p * len(s_ids))(*[x.encode() for x in s_ids])
    ql = np.ascontiguousarray(q_lens, dtype=np.uint64)
    sl = np.ascontiguousarray(s_lens, dtype=np.uint64)
    names = SeqNames(qa, ql.ctypes.data, sa, sl.ctypes.data, len(q_ids), len(s_ids))
    o = np.frombuffer(ops + b"\0", dtype=np.uint8)
    qasc = np.frombuffer(q_ascii, dtype=np.uint8) if q_ascii else None
    qoff = np.ascontiguousarray(q_ascii_off, dtype=np.uint64) if q_ascii_off is not None else None
    rc = load().lx_write_records_ex(str(path).encode(), fmt, 1 if write_header else 0, program.encode(), _ptr(m), len(m), _ptr(o),
                                    C.byref(names), _ptr(qasc) if qasc is not None else None, _ptr(qoff) if qoff is not None else None,
                                    C.byref(options) if options is not None else None)
    if rc != LX_OK:
        raise LambdaExtError(rc, "lx_write_records: " + last_output_error())


class StepPlan(C.Structure):
    _fields_ = [("family", C.c_int32), ("group_lanes", C.c_int32), ("strip_cols", C.c_int32), ("panels", C.c_int32),
                ("compact_codes", C.c_int32), ("queries_per_wavefront", C.c_int32), ("may_decline", C.c_int32), ("adapted", C.c_int32),
                ("slot_bytes", C.c_uint64), ("lds_bytes", C.c_uint64), ("score_bound", C.c_uint64), ("name", C.c_char * 160)]


PLAN_NO_SWEEP, PLAN_HALF, PLAN_I16_COMPACT_WIDE, PLAN_I16_PAIRS, PLAN_INT32, PLAN_MQ = range(6)


def plan_step(scoring, max_qlen: int, max_slen: int, query_run: int, n: int, pass2_mode: int = 2, mq_sweep: int = 1, packed_half: int = 1,
              trace_bytes: int = 64 << 30, survivor_share: float = -1.0, adapt_permille: int = 30) -> StepPlan:
    """lx_plan_step: how lx_extend_batch_dev would run such a batch (no device needed)."""
    lib = load()
    lib.lx_plan_step.argtypes = [C.c_void_p] + [C.c_uint64] * 8 + [C.c_double, C.c_uint64, C.c_void_p]
    lib.lx_plan_step.restype = C.c_int
    out = StepPlan()
    rc = lib.lx_plan_step(C.byref(scoring), max_qlen, max_slen, query_run, n, pass2_mode, mq_sweep, packed_half, trace_bytes, survivor_share,
                          adapt_permille, C.byref(out))
    if rc != LX_OK:
        raise LambdaExtError(rc, "lx_plan_step")
    return out


def karlin_params(method: int, match: int = 2, mismatch: int = -3, gap_open: int = -11, gap_extend: int = -1) -> Karlin:
    ka = Karlin()
    rc = load().lx_karlin_params(method, match, mismatch, gap_open, gap_extend, C.byref(ka))
    if rc != LX_OK:
        raise LambdaExtError(rc, "no Karlin-Altschul values for this scoring scheme")
    return ka


def widen_and_preprocess(matches: np.ndarray, qlens: np.ndarray, slens: np.ndarray) -> np.ndarray:
    m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE).copy()
    qlens = np.ascontiguousarray(qlens, dtype=np.uint64)
    slens = np.ascontiguousarray(slens, dtype=np.uint64)
    n = load().lx_widen_and_preprocess(_ptr(m), len(m), _ptr(qlens), _ptr(slens))
    return m[: int(n)]


LX_RANKS_AA27, LX_RANKS_DNA5_BS, LX_RANKS_SIMPLE = 0, 1, 2
LX_FRAMES_NONE, LX_FRAMES_REVCOMP, LX_FRAMES_TRANSLATED, LX_FRAMES_BISULFITE = 0, 1, 2, 3


def set_frames(q_mode: int, s_mode: int, qry_id: int, subj_id: int) -> tuple[int, int]:
    """_setFrames (src/search_algo.hpp:768-814): (qFrameShift, sFrameShift) of a frame-expanded id pair."""
    qf, sf = C.c_int32(), C.c_int32()
    load().lx_set_frames(q_mode, s_mode, qry_id, subj_id, C.byref(qf), C.byref(sf))
    return qf.value, sf.value


def translate_six_frames(dna5: np.ndarray, genetic_code: int = 1) -> list[np.ndarray]:
    """Six-frame translation of BioC++ dna5 ranks into SeqAn AminoAcid ranks, frames +1 +2 +3 -1 -2 -3."""
    src = np.ascontiguousarray(dna5, dtype=np.uint8)
    out = np.zeros(2 * src.size + 8, dtype=np.uint8)
    off, ln = np.zeros(6, dtype=np.uint64), np.zeros(6, dtype=np.uint64)
    rc = load().lx_translate_six_frames(_ptr(src), src.size, genetic_code, _ptr(out), out.size, _ptr(off), _ptr(ln))
    if rc != LX_OK:
        raise LambdaExtError(rc, "lx_translate_six_frames")
    return [out[int(o): int(o) + int(l)].copy() for o, l in zip(off, ln)]



def convert_ranks(kind: int, ranks: np.ndarray) -> np.ndarray:
    """BioC++ ranks -> the ranks the scoring tables use (src/seqan2_to_biocpp.hpp:352-395)."""
    src = np.ascontiguousarray(ranks, dtype=np.uint8)
    out = np.empty_like(src)
    rc = load().lx_convert_ranks(kind, _ptr(src), src.size, _ptr(out))
    if rc != LX_OK:
        raise LambdaExtError(rc, "lx_convert_ranks: unknown kind or rank outside the alphabet")
    return out


def builtin_scoring(method: int, match: int = 2, mismatch: int = -3, gap_open: int = -11, gap_extend: int = -1) -> Scoring:
    sc = Scoring()
    rc = load().lx_builtin_scoring(method, match, mismatch, gap_open, gap_extend, C.byref(sc))
    if rc != LX_OK:
        raise LambdaExtError(rc, "lx_builtin_scoring")
    return sc


def _sptr(a):
    return None if a is None else _ptr(a)


def _ssize(a) -> int:
    return 0 if a is None else int(a.size)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Handle:
    """One lx_handle: one device, one HIP stream (the analogue of the reference's per-thread LocalDataHolder)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.lx_create(device, C.byref(h))
        if rc != LX_OK:
            raise LambdaExtError(rc, self.lib.lx_last_error(None).decode())
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.lx_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int):
        if rc != LX_OK:
            raise LambdaExtError(rc, self.lib.lx_last_error(self.h).decode())

    def set_option(self, opt: int, value: int):
        self._check(self.lib.lx_set_option(self.h, opt, value))

    def set_band(self, band: int, centres=None, d_centres=None):
        """Band mode (LX_OPT_BAND): half width in diagonals (0 = off), optional per-extension centre diagonals for the next
        host-buffer call (numpy int32) / for the *_dev calls (torch int32 tensor on the device)."""
        self.set_option(LX_OPT_BAND, band)
        if centres is None:
            self._check(self.lib.lx_set_band_centres(self.h, None, 0))
        else:
            c = np.ascontiguousarray(centres, dtype=np.int32)
            self._check(self.lib.lx_set_band_centres(self.h, _ptr(c), len(c)))
        self._band_dev = d_centres  # keeps the tensor alive
        self._check(self.lib.lx_set_band_centres_dev(self.h, d_centres.data_ptr() if d_centres is not None else None))

    def get_option(self, opt: int) -> int:
        v = C.c_uint64()
        self._check(self.lib.lx_get_option(self.h, opt, C.byref(v)))
        return int(v.value)

    def set_scoring(self, sc: Scoring, slot: int = 0):
        self._check(self.lib.lx_set_scoring(self.h, slot, C.byref(sc)))

    # ---- host-buffer entry points -------------------------------------------------------------------
    def score_batch(self, q_res: np.ndarray, s_res: np.ndarray, ext: np.ndarray, slot: int = 0) -> np.ndarray:
        q_res = np.ascontiguousarray(q_res, dtype=np.uint8)
        s_res = None if s_res is None else np.ascontiguousarray(s_res, dtype=np.uint8)  # None: resident subjects
        ext = np.ascontiguousarray(ext, dtype=EXT_DTYPE)
        out = np.full(len(ext), -1, dtype=np.int32)
        self._check(self.lib.lx_score_batch(self.h, slot, _ptr(q_res), q_res.size, _sptr(s_res), _ssize(s_res), _ptr(ext),
                                            len(ext), _ptr(out)))
        return out

    def align_batch(self, q_res: np.ndarray, s_res: np.ndarray, ext: np.ndarray, slot: int = 0, known_score=None,
                    raw: bool = False):
        """lx_align_batch; known_score = the pass-1 scores of `ext` if the caller has them (saves the score pre-pass).
        raw=True returns (hsp, ops buffer, ops_off) without building the per-extension bytes objects."""
        q_res = np.ascontiguousarray(q_res, dtype=np.uint8)
        s_res = None if s_res is None else np.ascontiguousarray(s_res, dtype=np.uint8)  # None: resident subjects
        ext = np.ascontiguousarray(ext, dtype=EXT_DTYPE)
        n = len(ext)
        sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
        ops_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(sizes, out=ops_off[1:])
        hsp = np.zeros(n, dtype=HSP_DTYPE)
        ops = np.zeros(int(ops_off[-1]) + 1, dtype=np.uint8)
        ks = None if known_score is None else np.ascontiguousarray(known_score, dtype=np.int32)
        self._check(self.lib.lx_align_batch(self.h, slot, _ptr(q_res), q_res.size, _sptr(s_res), _ssize(s_res), _ptr(ext), n,
                                            None if ks is None else _ptr(ks), _ptr(hsp), _ptr(ops), _ptr(ops_off)))
        if raw:
            return hsp, ops, ops_off
        st = ops_off[:n].astype(np.int64) + hsp["ops_shift"].astype(np.int64)
        ops_list = [bytes(ops[int(st[i]):int(st[i]) + int(hsp["n_ops"][i])]) for i in range(n)]
        return hsp, ops_list

    def prefilter_batch(self, q_res, s_res, seeds, seed_length: int, pre_scoring: int, thresh: float, slot: int = 0):
        q_res = np.ascontiguousarray(q_res, dtype=np.uint8)
        s_res = None if s_res is None else np.ascontiguousarray(s_res, dtype=np.uint8)  # None: resident subjects
        seeds = np.ascontiguousarray(seeds, dtype=SEED_DTYPE)
        keep = np.zeros(len(seeds), dtype=np.uint8)
        self._check(self.lib.lx_prefilter_batch(self.h, slot, _ptr(q_res), q_res.size, _sptr(s_res), _ssize(s_res),
                                                _ptr(seeds), len(seeds), seed_length, pre_scoring, thresh, _ptr(keep)))
        return keep

    def extend_batch_rle(self, q_res, s_res, ext, min_score, slot: int = 0):
        """lx_extend_batch_rle: as extend_batch, the ops as run-length codes ((op << 6) | (len - 1), op 0/1/2 = M/D/I)."""
        return self.extend_batch(q_res, s_res, ext, min_score, slot=slot, rle=True)

    @staticmethod
    def expand_ops(codes: np.ndarray, n_ops: int) -> bytes:
        out = np.zeros(max(n_ops, 1), dtype=np.uint8)
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        rc = load().lx_expand_ops(_ptr(c), n_ops, _ptr(out))
        if rc != LX_OK:
            raise LambdaExtError(rc, "lx_expand_ops")
        return bytes(out[:n_ops])

    def extend_batch(self, q_res, s_res, ext, min_score, slot: int = 0, copy_ops: bool = True, rle: bool = False, out=None):
        """lx_extend_batch: both passes on host buffers.  min_score = int cut-off for all, or an int32 array per
        extension.  Returns (scores, hsp, ops_off, ops) -- ops is a copy of the handle-owned buffer (copy_ops=False: a
        view that the next call invalidates)."""
        q_res = np.ascontiguousarray(q_res, dtype=np.uint8)
        s_res = None if s_res is None else np.ascontiguousarray(s_res, dtype=np.uint8)
        ext = np.ascontiguousarray(ext, dtype=EXT_DTYPE)
        n = len(ext)
        per = None if np.isscalar(min_score) else np.ascontiguousarray(min_score, dtype=np.int32)
        # out = (score, hsp, off) arrays of an earlier call to write into (a caller that keeps its buffers pays no page faults)
        score, hsp, off = out if out is not None else (np.zeros(n, dtype=np.int32), np.zeros(n, dtype=HSP_DTYPE), np.zeros(n, dtype=np.uint64))
        p, nb = C.c_void_p(), C.c_uint64()
        fn = self.lib.lx_extend_batch_rle if rle else self.lib.lx_extend_batch
        self._check(fn(self.h, slot, _ptr(q_res), q_res.size, _sptr(s_res), _ssize(s_res), _ptr(ext), n,
                                             None if per is None else _ptr(per), 0 if per is not None else int(min_score),
                                             _ptr(score), _ptr(hsp), _ptr(off), C.byref(p), C.byref(nb)))
        if not nb.value:
            return score, hsp, off, np.zeros(1, np.uint8)
        ops = np.ctypeslib.as_array((C.c_uint8 * int(nb.value)).from_address(p.value))
        return score, hsp, off, ops.copy() if copy_ops else ops

    def extend_batch_list(self, q_res, s_res, ext, min_score, slot: int = 0, copy: bool = True, out_score=None):
        """lx_extend_batch_list: the scores of every extension and the survivors of the filter as a list.  Returns
        (scores, index, hsp, codes_off, codes): `index[k]` is survivor k's position in `ext`; copy=False gives views of the
        handle's buffers that the next extend_batch* call invalidates."""
        q_res = np.ascontiguousarray(q_res, dtype=np.uint8)
        s_res = None if s_res is None else np.ascontiguousarray(s_res, dtype=np.uint8)
        ext = np.ascontiguousarray(ext, dtype=EXT_DTYPE)
        n = len(ext)
        per = None if np.isscalar(min_score) else np.ascontiguousarray(min_score, dtype=np.int32)
        score = out_score if out_score is not None else np.zeros(n, dtype=np.int32)
        sl = SurvivorList()
        self._check(self.lib.lx_extend_batch_list(self.h, slot, _ptr(q_res), q_res.size, _sptr(s_res), _ssize(s_res), _ptr(ext), n,
                                                  None if per is None else _ptr(per), 0 if per is not None else int(min_score),
                                                  _ptr(score), C.byref(sl)))
        k = int(sl.count)
        if not k:
            return score, np.zeros(0, np.uint32), np.zeros(0, HSP_DTYPE), np.zeros(0, np.uint64), np.zeros(1, np.uint8)
        view = lambda ptr, dtype, count: np.frombuffer((C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype, count=count)
        index, hsp, off = view(sl.index, np.uint32, k), view(sl.hsp, HSP_DTYPE, k), view(sl.codes_off, np.uint64, k)
        codes = view(sl.codes, np.uint8, max(int(sl.codes_bytes), 1))
        if copy:
            index, hsp, off, codes = index.copy(), hsp.copy(), off.copy(), codes.copy()
        return score, index, hsp, off, codes

    def last_extend_stats(self):
        """(extensions, slots, cells, executed cells) of the last extend_batch call."""
        out = np.zeros(4, dtype=np.uint64)
        self._check(self.lib.lx_last_extend_stats(self.h, _ptr(out)))
        return tuple(int(x) for x in out)

    def set_subjects(self, s_res):
        """lx_set_subjects: keep the subject residues on the device; later host-buffer calls may pass s_res=None."""
        if s_res is None:
            self._check(self.lib.lx_set_subjects(self.h, None, 0))
            return
        s_res = np.ascontiguousarray(s_res, dtype=np.uint8)
        self._check(self.lib.lx_set_subjects(self.h, _sptr(s_res), _ssize(s_res)))

    # ---- device-resident entry points (torch tensors on this handle's device) --------------------------
    def score_batch_dev(self, d_q, d_s, d_ext, n: int, d_out, stream=None, slot: int = 0):
        self._check(self.lib.lx_score_batch_dev(self.h, slot, d_q.data_ptr(), d_s.data_ptr(), d_ext.data_ptr(), n,
                                                d_out.data_ptr(), stream))

    def align_batch_dev(self, d_q, d_s, d_ext, n: int, d_hsp, d_ops, d_ops_off, stream=None, slot: int = 0):
        self._check(self.lib.lx_align_batch_dev(self.h, slot, d_q.data_ptr(), d_s.data_ptr(), d_ext.data_ptr(), n,
                                                d_hsp.data_ptr(), d_ops.data_ptr(), d_ops_off.data_ptr(), stream))

    def iterate_matches(self, q_res, q_off, q_len, q_orig_len, s_res, s_off, s_len, matches, params: "SearchParams",
                        slot: int = 0):
        """iterateMatchesFullSimd: returns (blast_matches ndarray, list of ops bytes, IterateStats)."""
        q_res = np.ascontiguousarray(q_res, dtype=np.uint8)
        s_res = None if s_res is None else np.ascontiguousarray(s_res, dtype=np.uint8)  # None: resident subjects
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.uint64)
        s_off = np.ascontiguousarray(s_off, dtype=np.uint64)
        s_len = np.ascontiguousarray(s_len, dtype=np.uint64)
        q_orig = np.ascontiguousarray(q_orig_len, dtype=np.uint64)
        m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE).copy()
        res = C.c_void_p()
        self._check(self.lib.lx_iterate_matches(self.h, slot, _ptr(q_res), q_res.size, _ptr(q_off), _ptr(q_len), len(q_off),
                                                _ptr(q_orig), _sptr(s_res), _ssize(s_res), _ptr(s_off), _ptr(s_len),
                                                len(s_off), _ptr(m), len(m), C.byref(params), C.byref(res)))
        return self._take_iterate_result(res)

    def set_queries(self, q_res, q_off, q_len, q_orig_len=None, qry_num_frames: int = 1):
        """lx_set_queries: the (frame-expanded) query set becomes resident on the device (for iterate_matches_dev)."""
        q_res = np.ascontiguousarray(q_res, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.uint64)
        q_orig = None if q_orig_len is None else np.ascontiguousarray(q_orig_len, dtype=np.uint64)
        self._check(self.lib.lx_set_queries(self.h, _ptr(q_res), q_res.size, _ptr(q_off), _ptr(q_len), len(q_off),
                                            None if q_orig is None else _ptr(q_orig), qry_num_frames))

    def set_subject_seqs(self, s_off, s_len):
        s_off = np.ascontiguousarray(s_off, dtype=np.uint64)
        s_len = np.ascontiguousarray(s_len, dtype=np.uint64)
        self._check(self.lib.lx_set_subject_seqs(self.h, _ptr(s_off), _ptr(s_len), len(s_off)))

    def reserve(self, n_matches: int, n_windows: int, n_hsps: int, n_columns: int = 0):
        """lx_reserve: what the first iterate_matches_dev call would allocate inside the call, ahead of it."""
        self._check(self.lib.lx_reserve(self.h, n_matches, n_windows, n_hsps, n_columns))

    def iterate_matches_dev(self, d_matches, n: int, params: "SearchParams", slot: int = 0):
        """lx_iterate_matches_dev on a device tensor holding n lx_match records (48 bytes each); results as iterate_matches."""
        res = C.c_void_p()
        self._check(self.lib.lx_iterate_matches_dev(self.h, slot, d_matches.data_ptr() if n else None, n, C.byref(params), C.byref(res)))
        return self._take_iterate_result(res)

    def plan_free_packing_dev(self, d_ext, n: int, n_qseq: int, strip_cols: int = 19, cuts=None):
        """lx_plan_free_packing_dev: (plan [nwf, 16], wf_pan [nwf], wf_maxs [nwf], report [16]) for n lx_extension records in a device tensor."""
        cuts = np.ascontiguousarray([0, n] if cuts is None else cuts, dtype=np.uint64)
        nranges = len(cuts) - 1
        cap = int(self.lib.lx_plan_free_packing_bound(n, n_qseq, nranges))
        plan, pan, maxs, rep = np.zeros(cap * 16, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(16, np.uint32)
        self._check(self.lib.lx_plan_free_packing_dev(self.h, d_ext.data_ptr(), n, n_qseq, strip_cols, nranges, cuts.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                      _ptr(plan), _ptr(pan), _ptr(maxs), _ptr(rep)))
        nwf = int(rep[0])
        assert nwf <= cap, (nwf, cap)
        return plan[:nwf * 16].reshape(nwf, 16).copy(), pan[:nwf].copy(), maxs[:nwf].copy(), rep

    def widen_and_preprocess_dev(self, d_matches, n: int, bisulfite: bool = False):
        """lx_widen_and_preprocess_dev: the window list of a device match list (n lx_match records) as a MATCH_DTYPE array."""
        out = np.zeros(max(n, 1), dtype=MATCH_DTYPE)
        cnt = C.c_uint64(0)
        self._check(self.lib.lx_widen_and_preprocess_dev(self.h, d_matches.data_ptr() if n else None, n, 1 if bisulfite else 0, _ptr(out), C.byref(cnt)))
        return out[:cnt.value].copy()

    def _take_iterate_result(self, res):
        try:
            n = int(self.lib.lx_iterate_result_count(res))
            stats = self.lib.lx_iterate_result_stats(res)
            if n == 0:
                return np.zeros(0, dtype=BLAST_MATCH_DTYPE), [], stats
            buf = (C.c_char * (n * BLAST_MATCH_DTYPE.itemsize)).from_address(self.lib.lx_iterate_result_matches(res))
            bms = np.frombuffer(buf, dtype=BLAST_MATCH_DTYPE).copy()
            total = int((bms["ops_off"].astype(np.uint64) + bms["n_ops"].astype(np.uint64)).max())  # (not the last record's: bisulfite runs two passes)
            if not self.lib.lx_iterate_result_ops(res):  # LX_ITERATE_NO_OPS
                return bms, [], stats
            obuf = (C.c_char * max(total, 1)).from_address(self.lib.lx_iterate_result_ops(res))
            allops = bytes(obuf)
            ops = [allops[o:o + k] for o, k in zip(bms["ops_off"].tolist(), bms["n_ops"].tolist())]  # (record scalars one by one cost 0.7 ms per thousand)
            return bms, ops, stats
        finally:
            self.lib.lx_iterate_result_free(res)

    def extend_batch_dev(self, d_q, d_s, d_ext, n: int, min_score: int, d_score, d_hsp, d_ops, d_ops_off, d_count,
                         d_min_score=None, stream=None, slot: int = 0):
        self._check(self.lib.lx_extend_batch_dev(self.h, slot, d_q.data_ptr(), d_s.data_ptr(), d_ext.data_ptr(), n,
                                                 d_min_score.data_ptr() if d_min_score is not None else None, min_score,
                                                 d_score.data_ptr(), d_hsp.data_ptr(), d_ops.data_ptr(),
                                                 d_ops_off.data_ptr(), d_count.data_ptr(), stream))

    def synchronize(self):
        self._check(self.lib.lx_synchronize(self.h))

    def last_kernel_name(self) -> str:
        return self.lib.lx_last_kernel_name(self.h).decode()

    def last_trace_kernel_name(self) -> str:
        return self.lib.lx_last_trace_kernel_name(self.h).decode()

    def last_phase_ms(self, phase: int):
        ms, cnt = C.c_float(), C.c_int()
        self._check(self.lib.lx_last_phase_ms(self.h, phase, C.byref(ms), C.byref(cnt)))
        return float(ms.value), int(cnt.value)

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        self._check(self.lib.lx_last_kernel_ms(self.h, C.byref(ms)))
        return float(ms.value)
