"""Development aid: numpy model of lx_extend_batch's packing rules on the ragged list of bench.py (DESIGN.md sections 4, 8.5):
executed cells and padded share for sub-blocks of 4, naive streaming, pool + streamed pairs, at PANEL=152|104|88 columns."""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from lambda_amd import synth
q,s,ext = synth.make_ragged_lists_np(50000, seed=0x1A3BDA07)
lq = ext['q_len'].astype(np.int64); ls = ext['s_len'].astype(np.int64); qo = ext['q_off'].astype(np.int64)
cells = (lq*ls).sum()
import os
panel=int(os.environ.get("PANEL","152"))
pan = -(-lq//panel)
def greedy(order, maxq=4, width=16):
    ex=0; nw=0; slots=0
    cur_q=[]; cnt=0; wmax=0; wpan=0
    for i in order:
        qq=qo[i]
        if cnt==width or (qq not in cur_q and len(cur_q)==maxq):
            ex += 16*wpan*panel*(wmax+7); nw+=1
            cur_q=[]; cnt=0; wmax=0; wpan=0
        if qq not in cur_q: cur_q.append(qq)
        cnt+=1; wmax=max(wmax,ls[i]); wpan=max(wpan,pan[i])
    if cnt: ex += 16*wpan*panel*(wmax+7); nw+=1
    return ex, nw
# order A: (panels desc, s_len desc, q)
oA = np.lexsort((qo, -ls, -pan))
ex,nw = greedy(oA)
print('free packing, sort (panels, s_len): wavefronts',nw,'executed %.1f G padded %.1f %%'%(ex/1e9,100*(1-cells/ex)))
# order B: by query's typical length first: (panels desc, lq desc, q, s_len desc): keeps a query's windows together
oB = np.lexsort((-ls, qo, -lq, -pan))
ex,nw = greedy(oB)
print('free packing, sort (panels, lq, query): wavefronts',nw,'executed %.1f G padded %.1f %%'%(ex/1e9,100*(1-cells/ex)))
# order C: normal windows grouped by query (sorted by lq), merged windows (ls > lq + 2*sqrt+..) separately by length
b = (np.sqrt(lq).astype(np.int64)+1); normal = ls <= lq+2*b
oC = np.r_[np.lexsort((qo, -ls, -pan))[np.isin(np.lexsort((qo, -ls, -pan)), np.flatnonzero(~normal))], np.lexsort((-ls, qo, -lq, -pan))[np.isin(np.lexsort((-ls, qo, -lq, -pan)), np.flatnonzero(normal))]]
ex,nw = greedy(oC)
print('merged apart: wavefronts',nw,'executed %.1f G padded %.1f %%'%(ex/1e9,100*(1-cells/ex)))
for mq in (2,8,16):
    ex,nw = greedy(oA, maxq=mq); print('maxq',mq,'sortA: wavefronts',nw,'executed %.1f G padded %.1f %%'%(ex/1e9,100*(1-cells/ex)))
print('current: wavefronts', 676500//16, 'executed 73.7 G padded 39.9 %')

# hybrid: merged windows in sub-blocks of 4 (filled with the query's normal windows), the rest streamed query-major
starts = np.flatnonzero(np.r_[True, qo[1:]!=qo[:-1]]); ends = np.r_[starts[1:], len(ext)]
sb=[]  # (panels, maxlen)
stream=[] # window indices (normal leftovers), query-major, sorted by (panels desc, lq desc)
qorder = np.lexsort((-lq[starts], -pan[starts]))
for r in qorder:
    a,b_=starts[r],ends[r]
    idx=np.arange(a,b_)
    nm = idx[normal[idx]]; mg = idx[~normal[idx]]
    mg = mg[np.argsort(-ls[mg])]
    nm = list(nm)
    for k in range(0,len(mg),4):
        grp=list(mg[k:k+4])
        while len(grp)<4 and nm: grp.append(nm.pop())
        sb.append((pan[a], ls[grp].max(), len(grp)))
    stream.extend(nm)
sb.sort(key=lambda t:(-t[0],-t[1]))
ex=0; nw=0
for k in range(0,len(sb),4):
    g=sb[k:k+4]; ex+=16*g[0][0]*panel*(max(t[1] for t in g)+7); nw+=1
ex2,nw2 = greedy(np.array(stream))
print('hybrid: merged sub-blocks', len(sb), 'wavefronts', nw, '+ streamed wavefronts', nw2, 'executed %.1f G padded %.1f %%'%((ex+ex2)/1e9, 100*(1-cells/(ex+ex2))))

# pair granularity: a lane group's two windows share the query
def greedy_pairs(order, maxq=4):
    # order: list of window indices, query-major; form pairs per query run
    ex=0; nw=0
    cur_q=[]; cnt=0; wmax=0; wpan=0
    i=0; n=len(order)
    while i<n:
        qq=qo[order[i]]
        j=i
        while j<n and qo[order[j]]==qq: j+=1
        # windows i..j of this query -> pairs
        k=i
        while k<j:
            pair=order[k:k+2]
            if cnt==8 or (qq not in cur_q and len(cur_q)==maxq):
                ex += 16*wpan*panel*(wmax+7); nw+=1
                cur_q=[]; cnt=0; wmax=0; wpan=0
            if qq not in cur_q: cur_q.append(qq)
            cnt+=1; wmax=max(wmax,ls[pair].max()); wpan=max(wpan,pan[pair[0]])
            k+=2
        i=j
    if cnt: ex += 16*wpan*panel*(wmax+7); nw+=1
    return ex,nw
ex3,nw3 = greedy_pairs(np.array(stream))
print('hybrid, pairs: streamed wavefronts', nw3, 'executed %.1f G padded %.1f %%'%((ex+ex3)/1e9, 100*(1-cells/(ex+ex3))))
for mq in (4,8):
    ex3,nw3 = greedy_pairs(np.array(stream), maxq=mq)
    print('hybrid pairs maxq',mq,'executed %.1f G padded %.1f %%'%((ex+ex3)/1e9, 100*(1-cells/(ex+ex3))))
