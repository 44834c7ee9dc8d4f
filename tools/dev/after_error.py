"""Development aid: does a call that ended with a device-side error (broken LX_OPT_QUERY_RUN promise) leave state that changes
the next calls?  Three runs of the same uniform batch after the error must agree byte for byte."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import capi, synth
from tests.test_gpu_mq import run_fused, pack_free
from tests.test_gpu_score import SCHEMES

h = capi.Handle(0)
h.set_scoring(SCHEMES["blosum62"], 0)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
q2, s2, e2 = synth.make_batch_np(8, 120, 2, seed=2)
if which in ("five", "both"):
    try:
        run_fused(h, q2, s2, e2, 2, 50, mq=1)
    except capi.LambdaExtError as e:
        print("error 1:", e)
if which in ("pair", "both"):
    e3 = np.concatenate([e2[:4], e2[:4], e2[:4], e2[:4]])
    e3[1] = e2[5]
    try:
        run_fused(h, q2, s2, e3, 2, 50, mq=1)
    except capi.LambdaExtError as e:
        print("error 2:", e)
if which == "declined":
    q, s, ext = synth.make_ragged_lists_np(12, seed=11, lq_range=(230, 300), mean_windows=5.0, merged_frac=0.1)
    starts = np.unique(ext["q_off"])
    for k, qo in enumerate(starts):
        if k % 3 == 0:
            sel = np.nonzero(ext["q_off"] == qo)[0]
            L = int(ext["q_len"][sel[0]])
            q[qo:qo + L] = 22
            for i in sel[::2]:
                s[ext["s_off"][i]: ext["s_off"][i] + ext["s_len"][i]] = 22
    slots, src = pack_free(ext, np.random.default_rng(1))
    got = run_fused(h, q, s, slots, 2, 60, mq=1)
    print(got[5])
q, s, ext = synth.make_batch_np(12_000, 150, 16, seed=17, sub_rate=0.2, indel_rate=0.03)
h.set_option(capi.LX_OPT_PASS2_MODE, 2)
h.set_option(capi.LX_OPT_EXTEND_CHUNK, 50_000)
runs = []
for k in range(3):
    score, hsp, off, codes = h.extend_batch_rle(q, s, ext, 80)
    runs.append((score.copy(), hsp.copy(), off.copy(), codes.copy()))
    print(k, h.last_trace_kernel_name(), len(codes), int((hsp["n_ops"] > 0).sum()))
for k in (1, 2):
    same = all((runs[0][j] == runs[k][j]).all() if runs[0][j].shape == runs[k][j].shape else False for j in range(4))
    print("run 0 vs", k, "identical" if same else "DIFFERENT")
    if not same:
        for f in runs[0][1].dtype.names:
            d = np.nonzero(runs[0][1][f] != runs[k][1][f])[0]
            if len(d):
                print("  field", f, len(d), "rows differ, first", d[:5], runs[0][1][f][d[:5]], runs[k][1][f][d[:5]])
score_l, index, hsp_l, off_l, codes_l = h.extend_batch_list(q, s, ext, 80)
print("list:", h.last_trace_kernel_name(), len(codes_l), len(index))
score, hsp, off, codes = runs[0]
def codes_of(c, start, n_ops):
    k, done = int(start), 0
    while done < n_ops:
        done += (int(c[k]) & 63) + 1
        k += 1
    return bytes(c[int(start): k]), done
bad = 0
for k in range(len(index)):
    i, n_ops = int(index[k]), int(hsp_l["n_ops"][k])
    a, da = codes_of(codes_l, off_l[k], n_ops)
    b, db = codes_of(codes, off[i], n_ops)
    if a != b:
        bad += 1
        if bad <= 5:
            print("differs: survivor", k, "row", i, "n_ops", n_ops, "list", a.hex(), da, "rle", b.hex(), db, "chunk", i // 50000,
                  "expand equal:", capi.Handle.expand_ops(np.frombuffer(a, np.uint8), n_ops) == capi.Handle.expand_ops(np.frombuffer(b, np.uint8), n_ops))
print("survivors with different codes:", bad, "of", len(index))
