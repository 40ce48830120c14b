export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
LS_AMD_CHAIN2=1 timeout 40 python - <<'PY'
import numpy as np, torch, time
import distributed_matvec_amd as D
from distributed_matvec_amd import config
from oracle import c_oracle as CO, model as M
for L in (16, 22):
    cfg = M.heisenberg_chain_config(L)
    o = CO.COracle(M.model_from_config(cfg)); reps_o = o.enumerate()
    basis, h = D.loadConfigFromDict(cfg, hamiltonian=True)
    reps, masks = D.enumerateStates(basis, 1)
    x = np.random.RandomState(L).rand(len(reps_o)) - 0.5
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros_like(xd)
    pl = D.matrixVectorProduct(h, [xd], [yd], reps, mode="pull")
    err = np.abs(yd.cpu().numpy() - o.local_matvec(reps_o, x)).max()
    print("L", L, pl.kernel, "max err", err, flush=True)
PY
LS_AMD_CHAIN2=1 timeout 40 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra | cut -c1-200
