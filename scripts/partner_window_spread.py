"""Per (tile, bond): the spread of the partner indices and the share of all partners ONE best window of width w per (tile, bond)
would catch (oracle, CPU).  usage: partner_window_spread.py L [tile rows]   (why more windows do not pay)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as CO
from oracle import model as M
L = int(sys.argv[1]); tile=int(sys.argv[2]) if len(sys.argv)>2 else 256
o = CO.COracle(M.model_from_config(M.heisenberg_chain_config(L, symm=True)))
t=time.time(); reps = o.enumerate(); n=len(reps)
print("L",L,"n",n,"tile",tile,"enum s",round(time.time()-t,1))
rng=np.random.RandomState(1)
starts = rng.randint(0, max(1,n//tile-1), size=40)*tile
spreads=[]; cnts=[]
cover={256:0,512:0,1024:0,2048:0,4096:0}; tot=0
for s in starts:
    a=reps[s:s+tile]
    betas,coefs,offs = o.apply_off_diag(a)
    # group id of each packet = flip mask
    row=np.repeat(np.arange(tile), np.diff(offs))
    flip = betas ^ a[row]
    rep,_,norms = o.state_info(betas)
    ok=norms>0
    j=np.searchsorted(reps,rep[ok]); fl=flip[ok]
    for f in np.unique(fl):
        jj=np.sort(j[fl==f])
        spreads.append(jj[-1]-jj[0]); cnts.append(len(jj))
        tot+=len(jj)
        for w in cover:
            # best window of width w: max number of points in any [x, x+w)
            hi=np.searchsorted(jj, jj+w, side='left')
            cover[w]+= int((hi-np.arange(len(jj))).max())
spreads=np.array(spreads); cnts=np.array(cnts)
print("groups x tiles", len(spreads), "median spread", np.median(spreads), "pcts", np.percentile(spreads,[25,50,75,90]))
print("fraction of packets covered by ONE best window per (tile, group):", {w: round(cover[w]/tot,3) for w in cover})
