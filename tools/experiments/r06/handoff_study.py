"""Round 6: the sort's hand-off studied on the CPU (profiles/r06_sort_staged.md, 3b): replay of libstdc++'s introsort loop on the key
arrays of tools/sort_study.py; per sort the entries at the hand-off level, dealt to the waves by the product's zigzag ("zig") / by
the chunk they start in ("own"); printed: mean and max single-wave levels on the critical wave, hand-off level, entries.
    python tools/experiments/r06/handoff_study.py 500 | 1280   (from the repo root, after tools/sort_study.py wrote the key files)"""
import numpy as np, sys

def replay(keys):
    """yield per level: list of (f,l) live sub-ranges; returns also depth-to-finish of each sub-range"""
    v = [(-int(k), i) for i, k in enumerate(keys)]
    lt = lambda a, b: a[0] < b[0]
    cur = [(0, len(v))]
    levels=[]
    while cur:
        levels.append(list(cur))
        nxt = []
        for f, l in cur:
            a, b, c = f + 1, f + (l - f) // 2, l - 1
            if lt(v[a], v[b]):
                m = b if lt(v[b], v[c]) else (c if lt(v[a], v[c]) else a)
            elif lt(v[a], v[c]): m = a
            elif lt(v[b], v[c]): m = c
            else: m = b
            v[f], v[m] = v[m], v[f]
            first, last, pv = f + 1, l, v[f]
            while True:
                while lt(v[first], pv): first += 1
                last -= 1
                while lt(pv, v[last]): last -= 1
                if not first < last: break
                v[first], v[last] = v[last], v[first]
                first += 1
            cut = first
            for ff, ll in ((f, cut), (cut, l)):
                if ll - ff > 16: nxt.append((ff, ll))
        cur = nxt
    return levels
def depth_below(levels, k, f, l):
    # number of further levels needed for sub-range (f,l) at level k (levels in which some descendant is live)
    d=0
    for j in range(k, len(levels)):
        if any(ff>=f and ll<=l for ff,ll in levels[j]): d+=1
        else: break
    return d
N=int(sys.argv[1]); nw=8
keys=np.fromfile(f'tools/microbench/keys_r{25 if N==500 else 64}.bin',np.uint8).reshape(-1,N)
kfm = 2 if N==500 else 4
res={'zig':[], 'own':[], 'lvl':[], 'n':[]}
for row in keys[::2]:
    lv=replay(row)
    # hand-off level: first level k>0 with n_alive<=kfm*nw and none >64
    k=None
    for j,c in enumerate(lv):
        if j>0 and len(c)<=kfm*nw and all(l-f<=64 for f,l in c): k=j;break
    if k is None: continue
    ents=lv[k]
    res['lvl'].append(k); res['n'].append(len(ents))
    dep={e:depth_below(lv,k,*e) for e in ents}
    def cost(assign):
        # assign: list of lists of entries per wave; pack greedily in order into passes of <=64 lanes; cost of a pass = max depth in it
        worst=0
        for es in assign:
            passes=[]; cur=[];used=0
            rest=list(es)
            while rest:
                cur=[];used=0;left=[]
                for e in rest:
                    if used+(e[1]-e[0])<=64: cur.append(e);used+=e[1]-e[0]
                    else: left.append(e)
                passes.append(cur);rest=left
            c=sum(max(dep[e] for e in p) for p in passes)
            worst=max(worst,c)
        return worst
    # zigzag (EPT=1) or list order round robin
    if N==500:
        order=sorted(range(len(ents)),key=lambda i:(-(ents[i][1]-ents[i][0]),i))
        assign=[[] for _ in range(nw)]
        for r,i in enumerate(order):
            blk,pos=divmod(r,nw); w= nw-1-pos if blk&1 else pos
            assign[w].append(ents[i])
    else:
        assign=[[] for _ in range(nw)]
        for j,e in enumerate(ents): assign[j%nw].append(e)
    res['zig'].append(cost(assign))
    own=[[] for _ in range(nw)]
    for e in ents:
        # wave of start position; with EPT>1 layout x=i*512+tid: chunk c=(f>>6); wave=c%8
        own[(e[0]>>6)%nw].append(e)
    res['own'].append(cost(own))
for k,v in res.items(): print(k, np.mean(v), np.max(v))
