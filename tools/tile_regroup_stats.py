import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import sps_oracle as O
from sps_amd import synthetic
which = sys.argv[1] if len(sys.argv)>1 else "c2"
if which=="c2": batch = synthetic.make_scene(scan_seed=1)["batch"]
else: batch = synthetic.make_nclt_scene(seed=5)["batch"]
vox, inv = O.unique_first(O.quantize(batch[:, :5], 0.1))
cm = O.CoordinateManager(vox)
for ts in (2,4,8,16): cm.ensure_stride(ts)
def block_order(c, ts):
    q = c.copy(); q[:,1:4] = np.floor_divide(c[:,1:4], 4*ts)
    _, bid = O.unique_first(q)
    bit = ((c[:,3]//ts)%4)*16 + ((c[:,2]//ts)%4)*4 + ((c[:,1]//ts)%4)
    return np.lexsort((bit, bid))
def union(M, order, T=16):
    V=len(order); Mo = M[order]; pad=(-V)%T
    if pad: Mo=np.vstack([Mo,np.zeros((pad,81),bool)])
    return Mo.reshape(-1,T,81).any(1).sum(1)
for l in range(2,5):
    ts=1<<l; c=cm.coords[ts]; V=len(c)
    M=np.zeros((V,81),bool)
    for k,(i,o) in enumerate(cm.k3(ts)): M[o,k]=True
    nat=block_order(c,ts)
    t = c[:,4]
    res={}
    res['nat']=union(M,nat).mean()
    def keyA(M): return np.packbits(np.concatenate([M[:,27:54],M[:,:27],M[:,54:]],1),axis=1)
    def keyB(M): return np.packbits(M,axis=1)
    for W in (256,512,1024,2048,4096):
        for kn,kf in (('A',keyA),):
            key=kf(M)
            o=np.concatenate([nat[s:s+W][np.lexsort(key[nat[s:s+W]].T[::-1])] for s in range(0,V,W)])
            res[f'w{W}{kn}']=union(M,o).mean()
    # greedy within window 1024: repeatedly pick seed (first unassigned in lex order) and 15 rows minimizing union growth
    W=1024; o=[]
    key=keyA(M)
    for s in range(0,V,W):
        idx=list(nat[s:s+W][np.lexsort(key[nat[s:s+W]].T[::-1])])
        Mi=M[idx]; alive=np.ones(len(idx),bool)
        while alive.any():
            seed=np.flatnonzero(alive)[0]; alive[seed]=False; cur=Mi[seed].copy(); grp=[seed]
            for _ in range(15):
                if not alive.any(): break
                cand=np.flatnonzero(alive)
                cost=(Mi[cand]&~cur).sum(1)
                j=cand[np.argmin(cost)]; alive[j]=False; grp.append(j); cur|=Mi[j]
            o+= [idx[g] for g in grp]
    res['greedy1024']=union(M,np.array(o)).mean()
    print(l, V, "pairs/row %.1f"%M.sum(1).mean(), {k:round(v,1) for k,v in res.items()})
print("small windows")
for l in range(2,5):
    ts=1<<l; c=cm.coords[ts]; V=len(c)
    M=np.zeros((V,81),bool)
    for k,(i,o) in enumerate(cm.k3(ts)): M[o,k]=True
    nat=block_order(c,ts)
    key=np.packbits(np.concatenate([M[:,27:54],M[:,:27],M[:,54:]],1),axis=1)
    out={}
    for W in (16,32,64,128,256):
        o=np.concatenate([nat[s:s+W][np.lexsort(key[nat[s:s+W]].T[::-1])] for s in range(0,V,W)])
        out[W]=round(union(M,o).mean(),1)
    # distinct neighbour rows gathered per tile (locality proxy): natural vs W=512
    def distinct(order):
        nb=np.full((V,81),-1)
        for k,(i,o) in enumerate(cm.k3(ts)): nb[o,k]=i
        T=16; tot=0; nt=0
        for s in range(0,V-T+1,T):
            rows=order[s:s+T]; x=nb[rows]; tot+=len(np.unique(x[x>=0])); nt+=1
        return tot/nt
    o512=np.concatenate([nat[s:s+512][np.lexsort(key[nat[s:s+512]].T[::-1])] for s in range(0,V,512)])
    o64=np.concatenate([nat[s:s+64][np.lexsort(key[nat[s:s+64]].T[::-1])] for s in range(0,V,64)])
    print(l, out, "distinct nbr rows/tile: natural %.0f  W64 %.0f  W512 %.0f  (pairs/tile %.0f)"%(distinct(nat),distinct(o64),distinct(o512),M.sum()/ (V/16)))
