"""Helper of tests/test_gpu_two_calls.py (run as a script, its environment set by the test): one ragged seed list through
lx_extend_batch_list and lx_extend_batch (arrays), the results written as an .npz; argv = out.npz ragged|strong [trace_bytes]."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

from lambda_amd import capi, synth


def case(name="ragged"):
    if name == "strong":
        # every window scores beyond the compact codes (11 x min(Lq, Ls) > 2046): the overflow area runs out, the chunk is run again
        q, s, ext = synth.make_ragged_lists_np(300, seed=15, lq_range=(500, 700), mean_windows=8.0)
        q[:] = 22  # tryptophan everywhere
        s[:] = 22
        return q, s, ext, np.full(len(ext), 91, dtype=np.int32)
    q, s, ext = synth.make_ragged_lists_np(2500, seed=4242, lq_range=(40, 520), mean_windows=6.0, merged_frac=0.2)
    ext = ext.copy()
    ext["s_len"][::29] = 0
    mins = np.where(np.arange(len(ext)) % 4 == 0, 95, 60).astype(np.int32)
    return q, s, ext, mins


if __name__ == "__main__":
    q, s, ext, mins = case(sys.argv[2])
    h = capi.Handle(0)
    h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)
    if len(sys.argv) > 3:
        h.set_option(capi.LX_OPT_TRACE_BYTES, int(sys.argv[3]))
    score, index, hsp, off, codes = h.extend_batch_list(q, s, ext, mins)
    order = np.argsort(index, kind="stable")
    # (the codes in caller order of the survivors: chunk boundaries move where a survivor's codes stand, not what they are)
    ends = np.append(off[1:], len(codes)) if len(off) else off
    per = []
    for k in order:
        n_ops, a, done = int(hsp["n_ops"][k]), int(off[k]), 0
        b = a
        while done < n_ops:
            done += (int(codes[b]) & 63) + 1
            b += 1
        per.append(bytes(codes[a:b]))
    r = h.extend_batch(q, s, ext, mins)  # the arrays entry: scores, rows and column bytes by caller index
    np.savez(sys.argv[1], score=score, index=index[order], hsp=hsp[order], codes=np.frombuffer(b"".join(per), dtype=np.uint8), score_rows=r[0], hsp_rows=r[1],
             kernel=np.array(h.last_trace_kernel_name()))
