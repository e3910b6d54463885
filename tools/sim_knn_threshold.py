"""How many candidates reach the lower bound of the k-th best ranking value that knn_mfma.hip derives from group maxima?
Schemes: A min over the 8 lanes of their 3rd largest tile maximum; A2 k-th of the 8 x 3 top tile maxima; E k-th of 8 x 5;
B k-th of all 64 tile maxima; C like A2 with 16 groups of 8 per lane (what the kernel does, with strided groups).  CPU only."""
import numpy as np
rng=np.random.default_rng(0)
N=1024;K=20
def run(pts,nq=256):
    res={k:[] for k in 'A A2 E B C'.split()}
    per_lane_max={k:[] for k in res}
    for q in rng.choice(N,nq,replace=False):
        d=-((pts-pts[q])**2).sum(1)   # larger = nearer
        # layout: tile t (32 cands) -> wave t%4 ; within tile row r -> half (r%8)//4
        idx=np.arange(N); tile=idx//32; row=idx%32; wave=tile%4; h=(row%8)//4
        lane=wave*2+h                                # 8 lanes
        tilemax=np.full((8,8),-np.inf)               # lane, local tile
        for l in range(8):
            m=lane==l
            v=d[m].reshape(8,16)                     # 8 tiles x 16 rows (index order)
            tilemax[l]=v.max(1)
        srt=-np.sort(-tilemax,axis=1)
        thrA=srt[:,2].min()
        thrA2=-np.sort(-srt[:,:3].ravel())[K-1]
        thrE=-np.sort(-srt[:,:5].ravel())[K-1]
        thrB=-np.sort(-tilemax.ravel())[K-1]
        # C: group maxima of 8 -> 16 per lane, top3
        gm=np.stack([d[lane==l].reshape(16,8).max(1) for l in range(8)])
        thrC=-np.sort(-(-np.sort(-gm,axis=1))[:,:3].ravel())[K-1]
        for k,t in zip('A A2 E B C'.split(),(thrA,thrA2,thrE,thrB,thrC)):
            sel=d>=t
            res[k].append(sel.sum())
            per_lane_max[k].append(max((sel&(lane==l)).sum() for l in range(8)))
    for k in res: print(k,'M mean %.1f p99 %.0f max %d | per-lane max: mean %.1f max %d'%(np.mean(res[k]),np.percentile(res[k],99),max(res[k]),np.mean(per_lane_max[k]),max(per_lane_max[k])))
print('uniform'); run(rng.random((N,3)))
# surface-like: points on sphere, sorted (spatially correlated order)
p=rng.normal(size=(N,3)); p/=np.linalg.norm(p,axis=1,keepdims=True)
print('sphere random order'); run(p)
ps=p[np.lexsort((p[:,2],p[:,1],p[:,0]))]
print('sphere sorted by x'); run(ps)
