# round 4, call 5f: amd.pcUpwindBlend sweep (NACA section / 16-layer wing / channel) + primal grid sequencing to 800 x 250 with the level policy
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
timeout 500 python tools/naca_adjoint_sweep.py --blend 0 0.2 0.35 0.5 --combos -1:additive:rcb:1 -1:deflated:rcb:1 > $O/sweep.log 2> $O/sweep.err
grep "SWEEP\||R|" $O/sweep.log; tail -3 $O/sweep.err
for b in 0 0.35; do timeout 200 python tools/adjoint_study.py --n 100 50 40 --restart 600 --maxit 600 --blend $b > $O/channel_blend_$b.log 2>&1; grep "iters" $O/channel_blend_$b.log; done
timeout 600 python - > $O/primal.log 2> $O/primal.err <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge; ge.build()
from dafoam_amd.workloads import naca_converged_primal
opts = {"solverName": "DASimpleFoam", "debug": True, "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}}
case, info = naca_converged_primal(800, 250, options=opts, verbose=True, max_steps=80)
np.savez_compressed("gpurun_out/r05f/naca_primal_800x250.npz", states=case.states, dims=np.array([800, 250]), first_cell=2e-5, res0=info[-1]["res0"], res=info[-1]["res"])
PY
cat $O/primal.log; tail -2 $O/primal.err
