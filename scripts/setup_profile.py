#!/usr/bin/env python3
"""One-time costs of a cached plan on heisenberg_chain_40_symm: enumeration, plan, ls_amd_plan_cache_slots (count pass + 88.5 GB of
hipMalloc), the resolving matvec, a gather-only matvec."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distributed_matvec_amd as D
from distributed_matvec_amd import config
def T():
    torch.cuda.synchronize(); return time.perf_counter()
t0=T()
basis,h=D.loadConfigFromDict(config.heisenberg_chain_config(40,symm=True),hamiltonian=True); t1=T()
reps,masks=D.enumerateStates(basis,1); t2=T()
pl=D.MatvecPlan(h,reps,torch.float64); t3=T()
rows=pl.cache_slots(100<<30); t4=T()
x=[D.fillRandom(reps[0],1,torch.float64)]; y=[torch.zeros_like(x[0])]; t5=T()
pl.matvec(x,y); t6=T()
pl.matvec(x,y); t7=T()
print(f"load {t1-t0:.2f} enumerate {t2-t1:.2f} plan {t3-t2:.2f} cache_slots {t4-t3:.2f} vectors {t5-t4:.2f} first matvec (resolve) {t6-t5:.2f} second {t7-t6:.3f}")
pl.destroy(); t8=T(); print(f"destroy {t8-t7:.2f}")
