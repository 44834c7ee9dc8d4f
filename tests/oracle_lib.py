"""ctypes binding of oracle/_build/liblx_oracle.so -- TEST INFRASTRUCTURE (the checker), never the product."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "oracle" / "_build" / "liblx_oracle.so"

LXO_ALPH = 32


class Scoring(C.Structure):
    _fields_ = [("alphabet_size", C.c_int32), ("gap_open", C.c_int32), ("gap_extend", C.c_int32),
                ("reserved", C.c_int32), ("matrix", C.c_int8 * (LXO_ALPH * LXO_ALPH))]


class Hsp(C.Structure):
    _fields_ = [("score", C.c_int32), ("q_begin", C.c_int32), ("q_end", C.c_int32), ("s_begin", C.c_int32),
                ("s_end", C.c_int32), ("n_ops", C.c_int32)]


class AlignStats(C.Structure):
    _fields_ = [("num_matches", C.c_int32), ("num_mismatches", C.c_int32), ("num_positives", C.c_int32),
                ("num_negatives", C.c_int32), ("num_gap_opens", C.c_int32), ("num_gap_extensions", C.c_int32),
                ("num_insertions", C.c_int32), ("num_deletions", C.c_int32), ("alignment_length", C.c_int32),
                ("alignment_score", C.c_int32), ("identity", C.c_float), ("similarity", C.c_float)]


class Karlin(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("K", C.c_double), ("H", C.c_double), ("alpha", C.c_double),
                ("beta", C.c_double)]


MATCH_DTYPE = np.dtype([("qryId", "<u8"), ("subjId", "<u8"), ("qryStart", "<u8"), ("qryEnd", "<u8"),
                        ("subjStart", "<u8"), ("subjEnd", "<u8")])


def scoring_from(sc_any) -> Scoring:
    """Copies a product-side capi.Scoring (same layout) into the oracle's own struct."""
    o = Scoring()
    C.memmove(C.byref(o), C.byref(sc_any), C.sizeof(Scoring))
    return o


def make_scoring(alphabet_size: int, matrix: np.ndarray, gap_open: int, gap_extend: int) -> Scoring:
    o = Scoring()
    o.alphabet_size, o.gap_open, o.gap_extend = alphabet_size, gap_open, gap_extend
    m = np.zeros((LXO_ALPH, LXO_ALPH), dtype=np.int8)
    m[: matrix.shape[0], : matrix.shape[1]] = matrix
    C.memmove(o.matrix, m.ctypes.data, m.nbytes)
    return o


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp, i32, u64, i64 = C.c_void_p, C.c_int32, C.c_uint64, C.c_int64
        lib.lxo_score.argtypes = [vp, i32, vp, i32, C.POINTER(Scoring), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        lib.lxo_score_banded.argtypes = [vp, i32, vp, i32, C.POINTER(Scoring), i32, i32, C.POINTER(i32)]
        lib.lxo_align.argtypes = [vp, i32, vp, i32, C.POINTER(Scoring), C.POINTER(Hsp), vp]
        lib.lxo_align_banded.argtypes = [vp, i32, vp, i32, C.POINTER(Scoring), i32, i32, C.POINTER(Hsp), vp]
        for f in (lib.lxo_score_batch, lib.lxo_score_batch_simd):
            f.argtypes = [vp, vp, vp, vp, vp, vp, u64, C.POINTER(Scoring), vp, vp, vp, i32]
        lib.lxo_band_size.argtypes = [u64]
        lib.lxo_band_size.restype = i64
        lib.lxo_widen_and_preprocess.argtypes = [vp, u64, vp, vp]
        lib.lxo_widen_and_preprocess.restype = u64
        lib.lxo_seed_looks_promising.argtypes = [vp, i64, vp, i64, i64, i64, i64, i64, i32, C.c_double, C.POINTER(Scoring)]
        lib.lxo_length_adjustment.argtypes = [u64, u64, C.POINTER(Karlin)]
        lib.lxo_length_adjustment.restype = u64
        lib.lxo_evalue.argtypes = [i32, u64, u64, C.POINTER(Karlin)]
        lib.lxo_evalue.restype = C.c_double
        lib.lxo_bitscore.argtypes = [i32, C.POINTER(Karlin)]
        lib.lxo_bitscore.restype = C.c_double
        lib.lxo_alignment_stats.argtypes = [vp, vp, C.POINTER(Hsp), vp, C.POINTER(Scoring), i32, C.POINTER(AlignStats)]
        lib.lxo_frame_of.argtypes = [C.c_int, u64, C.c_int]
        lib.lxo_frame_of.restype = i32
        lib.lxo_untrue_id.argtypes = [C.c_int, u64, i32, C.c_int]
        lib.lxo_untrue_id.restype = u64
        lib.lxo_translate_frame.argtypes = [vp, u64, C.c_int, vp]
        lib.lxo_translate_frame.restype = u64

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p)

    def score(self, q: np.ndarray, s: np.ndarray, sc: Scoring):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        s = np.ascontiguousarray(s, dtype=np.uint8)
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        rc = self.lib.lxo_score(self._p(q), len(q), self._p(s), len(s), C.byref(sc), C.byref(a), C.byref(b), C.byref(c))
        assert rc == 0
        return a.value, b.value, c.value

    def score_banded(self, q, s, sc, lo, hi):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        s = np.ascontiguousarray(s, dtype=np.uint8)
        a = C.c_int32()
        assert self.lib.lxo_score_banded(self._p(q), len(q), self._p(s), len(s), C.byref(sc), lo, hi, C.byref(a)) == 0
        return a.value

    def align(self, q: np.ndarray, s: np.ndarray, sc: Scoring):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        s = np.ascontiguousarray(s, dtype=np.uint8)
        hsp = Hsp()
        ops = np.zeros(len(q) + len(s) + 1, dtype=np.uint8)
        rc = self.lib.lxo_align(self._p(q), len(q), self._p(s), len(s), C.byref(sc), C.byref(hsp), self._p(ops))
        assert rc == 0
        return hsp, bytes(ops[: hsp.n_ops])

    def align_banded(self, q: np.ndarray, s: np.ndarray, sc: Scoring, lo: int, hi: int):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        s = np.ascontiguousarray(s, dtype=np.uint8)
        hsp = Hsp()
        ops = np.zeros(len(q) + len(s) + 1, dtype=np.uint8)
        rc = self.lib.lxo_align_banded(self._p(q), len(q), self._p(s), len(s), C.byref(sc), lo, hi, C.byref(hsp), self._p(ops))
        assert rc == 0
        return hsp, bytes(ops[: hsp.n_ops])

    def score_batch(self, q_res, s_res, ext, sc: Scoring, threads: int = 1, simd: bool = False, ends: bool = False):
        q_res = np.ascontiguousarray(q_res, dtype=np.uint8)
        s_res = np.ascontiguousarray(s_res, dtype=np.uint8)
        qo = np.ascontiguousarray(ext["q_off"], dtype=np.uint64)
        so = np.ascontiguousarray(ext["s_off"], dtype=np.uint64)
        ql = np.ascontiguousarray(ext["q_len"], dtype=np.uint32)
        sl = np.ascontiguousarray(ext["s_len"], dtype=np.uint32)
        n = len(ext)
        score = np.zeros(n, dtype=np.int32)
        qe = np.zeros(n, dtype=np.int32)
        se = np.zeros(n, dtype=np.int32)
        f = self.lib.lxo_score_batch_simd if simd else self.lib.lxo_score_batch
        rc = f(self._p(q_res), self._p(s_res), self._p(qo), self._p(ql), self._p(so), self._p(sl), n, C.byref(sc),
               self._p(score), self._p(qe), self._p(se), threads)
        assert rc == 0
        return (score, qe, se) if ends else score

    def align_batch(self, q_res, s_res, ext, sc: Scoring):
        out = []
        for x in ext:
            q = q_res[int(x["q_off"]): int(x["q_off"]) + int(x["q_len"])]
            s = s_res[int(x["s_off"]): int(x["s_off"]) + int(x["s_len"])]
            out.append(self.align(q, s, sc))
        return out

    def band_size(self, n: int) -> int:
        return int(self.lib.lxo_band_size(n))

    def widen_and_preprocess(self, matches: np.ndarray, qlens: np.ndarray, slens: np.ndarray) -> np.ndarray:
        m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE).copy()
        qlens = np.ascontiguousarray(qlens, dtype=np.uint64)
        slens = np.ascontiguousarray(slens, dtype=np.uint64)
        n = self.lib.lxo_widen_and_preprocess(self._p(m), len(m), self._p(qlens), self._p(slens))
        return m[: int(n)]

    def seed_looks_promising(self, q, s, qs, qe, ss, seed_length, pre_scoring, thresh, sc) -> bool:
        q = np.ascontiguousarray(q, dtype=np.uint8)
        s = np.ascontiguousarray(s, dtype=np.uint8)
        return bool(self.lib.lxo_seed_looks_promising(self._p(q), len(q), self._p(s), len(s), qs, qe, ss, seed_length,
                                                      pre_scoring, thresh, C.byref(sc)))

    def length_adjustment(self, db_len, q_len, ka: Karlin) -> int:
        return int(self.lib.lxo_length_adjustment(db_len, q_len, C.byref(ka)))

    def evalue(self, score, ql, dl, ka: Karlin) -> float:
        return float(self.lib.lxo_evalue(score, ql, dl, C.byref(ka)))

    def bitscore(self, score, ka: Karlin) -> float:
        return float(self.lib.lxo_bitscore(score, C.byref(ka)))

    def alignment_stats(self, q, s, hsp: Hsp, ops: bytes, sc: Scoring, bs_rule: int = 0) -> AlignStats:
        q = np.ascontiguousarray(q, dtype=np.uint8)
        s = np.ascontiguousarray(s, dtype=np.uint8)
        o = np.frombuffer(ops + b"\0", dtype=np.uint8)
        st = AlignStats()
        rc = self.lib.lxo_alignment_stats(self._p(q), self._p(s), C.byref(hsp), self._p(o), C.byref(sc), bs_rule, C.byref(st))
        assert rc == 0, rc
        return st


    def frame_of(self, mode: int, ident: int, is_subject: bool) -> int:
        return int(self.lib.lxo_frame_of(mode, ident, 1 if is_subject else 0))

    def untrue_id(self, mode: int, n_id: int, frame: int, is_subject: bool) -> int:
        return int(self.lib.lxo_untrue_id(mode, n_id, frame, 1 if is_subject else 0))

    def translate_frame(self, dna5: np.ndarray, frame: int) -> np.ndarray:
        d = np.ascontiguousarray(dna5, dtype=np.uint8)
        out = np.zeros(d.size // 3 + 1, dtype=np.uint8)
        n = int(self.lib.lxo_translate_frame(self._p(d), d.size, frame, self._p(out)))
        return out[:n]


_oracle = None


def load() -> Oracle:
    global _oracle
    if _oracle is None:
        srcs = [ROOT / "oracle" / n for n in ("lx_oracle.c", "lx_oracle_simd.cpp", "lx_oracle.h")]
        if not LIB.exists() or any(s.exists() and s.stat().st_mtime > LIB.stat().st_mtime for s in srcs):
            subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
        _oracle = Oracle(C.CDLL(str(LIB)))
    return _oracle
