"""Developer (CPU, numpy / scipy -- no GPU, not the product): what a seeded exact search has to look at in the early
ICP iterations of the 1M<->1M bench pair -- per iteration the seed radius, the rows of cells its ball meets on the level
the kernel would pick, and the OCCUPIED level-0 cells / target points inside the ball.  Written to test the premise of
"occupancy bits per brick of cells"  before building it: a traversal that only touches occupied
cells pays for ~19 occupied cells where the row walk pays for ~27 rows at iteration 1 (profiles/r05_experiments.md).
  python scripts/dev/dev_ball_statistics.py [uniform|rings]"""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np
from scipy.spatial import cKDTree
from libwave_amd import synth
pattern = sys.argv[1] if len(sys.argv)>1 else 'uniform'
n=1_000_000
ref,tgt,T_gt=synth.pair(n,seed=42,pattern=pattern) if pattern!='uniform' else synth.pair(n,seed=42)
tree=cKDTree(tgt)
h0=0.183 if pattern=='uniform' else 0.091
lo=tgt.min(0)-1e-3
cell=np.floor((tgt-lo)/h0).astype(np.int64)
dims=cell.max(0)+1
print('dims',dims, 'cells', dims.prod())
key=(cell[:,2]*dims[1]+cell[:,1])*dims[0]+cell[:,0]
occ=np.zeros(dims.prod(),dtype=np.uint8); occ[key]=1
cnt=np.bincount(key,minlength=dims.prod())
print('occupied cells',occ.sum(),'pts/occ',n/occ.sum())
rng=np.random.default_rng(0)
samp=rng.choice(n,4000,replace=False)
T=np.eye(4)
prev=None
def umeyama(p,q):
    mp,mq=p.mean(0),q.mean(0)
    H=(q-mq).T@(p-mp)/len(p)
    U,S,Vt=np.linalg.svd(H)
    D=np.eye(3); 
    if np.linalg.det(U)*np.linalg.det(Vt)<0: D[2,2]=-1
    R=U@D@Vt
    t=mq-R@mp
    M=np.eye(4);M[:3,:3]=R;M[:3,3]=t
    return M
for it in range(14):
    q=(ref@T[:3,:3].T+T[:3,3]).astype(np.float32)
    d,idx=tree.query(q,workers=16)
    ok=d<=3.0
    if prev is not None:
        # seed radius
        r=np.linalg.norm(q[samp]-tgt[prev[samp]],axis=1)
        dn=d[samp]
        R=r/h0
        # level selection: finest level with cell >= 0.2 r
        lvl=np.zeros(len(samp),int)
        for k in range(4):
            lvl+= (h0*2**k < 0.2*r)
        hl=h0*2.0**lvl
        Rl=r/hl
        rows=np.pi*(Rl+0.5)**2
        # occupied level-0 cells within ball (box distance <= r)
        occ_in=[];pts_in=[];cells_in=[]
        for j in range(0,len(samp),1):
            qq=q[samp[j]];rr=r[j]
            c0=np.floor((qq-rr-lo)/h0).astype(int);c1=np.floor((qq+rr-lo)/h0).astype(int)
            c0=np.maximum(c0,0);c1=np.minimum(c1,dims-1)
            if (c1<c0).any(): occ_in.append(0);pts_in.append(0);cells_in.append(0);continue
            xs=np.arange(c0[0],c1[0]+1);ys=np.arange(c0[1],c1[1]+1);zs=np.arange(c0[2],c1[2]+1)
            X,Y,Z=np.meshgrid(xs,ys,zs,indexing='ij')
            clo=lo+np.stack([X,Y,Z],-1)*h0
            dd=np.maximum(np.maximum(clo-qq,qq-(clo+h0)),0)
            inb=(dd**2).sum(-1)<=rr*rr
            kk=((Z*dims[1]+Y)*dims[0]+X)
            o=occ[kk]&inb
            occ_in.append(o.sum());pts_in.append(cnt[kk][o.astype(bool)].sum());cells_in.append(inb.sum())
        occ_in=np.array(occ_in);pts_in=np.array(pts_in);cells_in=np.array(cells_in)
        print(f'it {it}: seed r mean {r.mean():.3f} med {np.median(r):.3f} p90 {np.percentile(r,90):.3f}  nn d mean {dn.mean():.3f}; R0 mean {R.mean():.2f}; lvl mean {lvl.mean():.2f} rows(level) mean {rows.mean():.1f} p90 {np.percentile(rows,90):.1f}; L0 cells in ball mean {cells_in.mean():.1f}; occupied L0 cells in ball mean {occ_in.mean():.1f} p90 {np.percentile(occ_in,90):.0f}; pts mean {pts_in.mean():.1f} p90 {np.percentile(pts_in,90):.0f}; changed {np.mean(idx[samp]!=prev[samp]):.2f}')
    else:
        print(f'it {it}: nn d mean {d.mean():.3f} med {np.median(d):.3f} p90 {np.percentile(d,90):.3f} p99 {np.percentile(d,99):.3f}')
    prev=idx
    M=umeyama(q[ok].astype(np.float64),tgt[idx[ok]].astype(np.float64))
    T=M@T
