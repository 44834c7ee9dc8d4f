"""Eight handles on eight host threads (the reference's model: one LocalDataHolder per OpenMP thread, nothing shared but statistics and
the writer -- /root/reference/src/search.cpp:379-385): every handle's lx_iterate_matches_dev on a protein seed list returns the records
of a handle that runs alone, and the calls share the GPU and the library's host threads instead of queueing behind a mutex."""
import threading
import time

import numpy as np
import pytest

from lambda_amd import capi, synth, workloads

pytestmark = pytest.mark.gpu


def _setup(q, qoff, qlen, qorig, s, soff, slen, w):
    h = capi.Handle(0)
    m_, ma, mi, go, ge = w.directions[0].scoring
    h.set_scoring(capi.builtin_scoring(m_, match=ma, mismatch=mi, gap_open=go, gap_extend=ge), 0)
    h.set_subjects(s)
    h.set_subject_seqs(soff, slen)
    h.set_queries(q, qoff, qlen, qorig, 1)
    return h


def test_eight_handles_on_eight_threads_share_gpu_and_host_threads():
    import torch

    w = workloads.WORKLOADS[1]
    q, qoff, qlen, qorig, s, soff, slen, m = synth.make_protein_seed_list_np(12_000, seed=77, lq=w.lq)
    ka = capi.karlin_params(*w.karlin)
    params = capi.SearchParams(w.max_evalue, -1, 0, w.db_length, 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka)
    d_m = torch.from_numpy(m.view(np.uint8).copy()).to("cuda:0")
    handles = [_setup(q, qoff, qlen, qorig, s, soff, slen, w) for _ in range(8)]
    try:
        calls = 4
        for h in handles:  # (every handle's buffers and kernels warm)
            want = h.iterate_matches_dev(d_m, len(m), params)
        assert len(want[0]) > 150_000
        import ctypes as C

        lib = capi.load()

        def raw_call(h):  # (the library's call alone: ctypes drops the interpreter lock for its duration)
            r = C.c_void_p()
            h._check(lib.lx_iterate_matches_dev(h.h, 0, d_m.data_ptr(), len(m), C.byref(params), C.byref(r)))
            lib.lx_iterate_result_free(r)

        t0 = time.perf_counter()
        for _ in range(calls):
            raw_call(handles[0])
        t_one = (time.perf_counter() - t0) / calls
        results, errors = [None] * 8, []

        def work(k):
            try:
                for _ in range(calls):
                    raw_call(handles[k])
            except Exception as e:  # noqa: BLE001
                errors.append((k, e))

        th = [threading.Thread(target=work, args=(k,)) for k in range(8)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        t_eight = (time.perf_counter() - t0) / calls
        assert not errors, errors
        th = [threading.Thread(target=lambda k=k: results.__setitem__(k, handles[k].iterate_matches_dev(d_m, len(m), params))) for k in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for k in range(8):
            bms, ops, st = results[k]
            assert bms.tobytes() == want[0].tobytes() and ops == want[1] and st.num_ext_ali == want[2].num_ext_ali, k
        print(f"one handle {t_one * 1e3:.2f} ms per call; eight handles on eight threads {t_eight * 1e3:.2f} ms per round of eight calls "
              f"({t_eight / (8 * t_one):.2f} x eight calls one after the other)")
        # eight calls at once take no longer than 1.3 x eight calls one after the other (the GPU is one: its share is an eighth each)
        assert t_eight <= 1.3 * 8 * t_one
    finally:
        for h in handles:
            h.close()
