"""Python restatement of iterateMatchesFullSimd (/root/reference/src/search_algo.hpp:1177-1332) on top of the CPU
oracle -- TEST INFRASTRUCTURE, the checker for lx_iterate_matches."""
import numpy as np

from tests import oracle_lib


def iterate_matches(orc, sc, ka, q_res, q_off, q_len, q_orig_len, s_res, s_off, s_len, matches, max_evalue, min_bitscore,
                    id_cutoff, db_total_length, query_translated=False, q_frames=1, s_frames=1, bs_rule=0):
    stats = dict(hits_duplicate=0, failed_bitscore=0, failed_evalue=0, failed_identity=0)
    n0 = len(matches)
    m = orc.widen_and_preprocess(matches, q_len, s_len)  # :1198
    stats["hits_duplicate"] = n0 - len(m)
    items = []
    for i, x in enumerate(m):
        qs = int(q_off[x["qryId"]]) + int(x["qryStart"])
        ss = int(s_off[x["subjId"]]) + int(x["subjStart"])
        q = q_res[qs: qs + int(x["qryEnd"] - x["qryStart"])]
        s = s_res[ss: ss + int(x["subjEnd"] - x["subjStart"])]
        items.append((i, x, q, s))
    items.sort(key=lambda t: (len(t[2]), len(t[3])))  # :1229-1235, stable
    adj_cache = {}
    surv = []
    for i, x, q, s in items:
        score = orc.score(q, s, sc)[0]  # :1246
        nq = int(x["qryId"]) // q_frames
        ql = int(q_orig_len[nq])
        bit = ev = None
        if min_bitscore >= 0:  # :1254-1264
            bit = orc.bitscore(score, ka)
            if bit < min_bitscore:
                stats["failed_bitscore"] += 1
                continue
        if max_evalue >= 0:  # :1266-1276 + search_misc.hpp:56-80
            qq = ql // (3 if query_translated else 1)
            if qq not in adj_cache:
                adj_cache[qq] = orc.length_adjustment(db_total_length, qq, ka)
            a = adj_cache[qq]
            ev = orc.evalue(score, qq - a, db_total_length - a, ka)
            if ev > max_evalue:
                stats["failed_evalue"] += 1
                continue
        surv.append((x, q, s, score, bit, ev))
    surv.sort(key=lambda t: int(t[0]["qryId"]) // q_frames)  # :1299, stable
    out = []
    for x, q, s, score, bit, ev in surv:
        hsp, ops = orc.align(q, s, sc)  # :1296
        assert hsp.score == score
        st = orc.alignment_stats(q, s, hsp, ops, sc, bs_rule)  # :1308
        if st.identity < id_cutoff:  # :1310-1315
            stats["failed_identity"] += 1
            continue
        nq = int(x["qryId"]) // q_frames
        ql = int(q_orig_len[nq])
        if bit is None:
            bit = orc.bitscore(score, ka)
        if ev is None:
            qq = ql // (3 if query_translated else 1)
            if qq not in adj_cache:
                adj_cache[qq] = orc.length_adjustment(db_total_length, qq, ka)
            a = adj_cache[qq]
            ev = orc.evalue(score, qq - a, db_total_length - a, ka)
        out.append(dict(qry_id=int(x["qryId"]), subj_id=int(x["subjId"]), n_qid=nq, n_sid=int(x["subjId"]) // s_frames,
                        q_start=int(x["qryStart"]) + hsp.q_begin, q_end=int(x["qryStart"]) + hsp.q_end,
                        s_start=int(x["subjStart"]) + hsp.s_begin, s_end=int(x["subjStart"]) + hsp.s_end,
                        score=score, alignment_length=st.alignment_length, num_matches=st.num_matches,
                        num_mismatches=st.num_mismatches, num_positives=st.num_positives,
                        num_gap_opens=st.num_gap_opens, num_gap_extensions=st.num_gap_extensions,
                        identity=float(st.identity), bit_score=bit, e_value=ev, ops=ops))
    return out, stats
