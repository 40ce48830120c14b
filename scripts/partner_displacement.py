"""Where the partners rep(beta) of a 256-row tile lie in the sorted representatives of a symmetric chain (oracle, CPU):
cumulative hit fraction of a window of +-h entries around the tile, and how the far partners cluster into 1024-entry blocks.
usage: partner_displacement.py L   (numbers quoted in DESIGN.md section 5, "One bound behind all three irregular kernels")"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as CO
from oracle import model as M
L = int(sys.argv[1]); tile=256
t=time.time()
o = CO.COracle(M.model_from_config(M.heisenberg_chain_config(L, symm=True)))
reps = o.enumerate(); n=len(reps)
print("L",L,"n",n,"enum s",round(time.time()-t,1))
rng=np.random.RandomState(1)
starts = rng.randint(0, max(1,n//tile-1), size=60)*tile
halos=[0,256,512,1024,2048,4096,16384,65536,1<<20]
hits={h:0 for h in halos}; total=0
blocks_per_tile=[]; hits_top=[]
for s in starts:
    betas,_,offs = o.apply_off_diag(reps[s:s+tile])
    rep,_,norms = o.state_info(betas)
    rep=rep[norms>0]
    j=np.searchsorted(reps,rep)
    total+=len(j)
    for h in halos: hits[h]+=int(((j>=s-h)&(j<s+tile+h)).sum())
    far=j[(j<s-512)|(j>=s+tile+512)]
    blk=far//1024
    u,c=np.unique(blk,return_counts=True)
    blocks_per_tile.append(len(u)); 
    c=np.sort(c)[::-1]
    hits_top.append((len(far), c[:8].sum() if len(c) else 0, c[:32].sum() if len(c) else 0))
print("per row", total/len(starts)/tile)
print({h: round(hits[h]/total,3) for h in halos})
print("far partners per tile, distinct 1024-blocks:", np.mean(blocks_per_tile))
ht=np.array(hits_top); print("far/tile", ht[:,0].mean(), "covered by top8 blocks", ht[:,1].sum()/ht[:,0].sum(), "top32", ht[:,2].sum()/ht[:,0].sum())
