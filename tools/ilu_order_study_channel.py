"""Dev tool (CPU only, oracle matrix): GMRES iterations behind the scalar twin of the node-block ILU(0) for different
ELIMINATION ORDERS of the nodes - bench channel: usage ilu_order_study_channel.py nx ny nz.  Results of round 3: profiles/r04l_ilu_ordering_study_cpu.log, DESIGN.md 6b."""
import sys, time, numpy as np, scipy.sparse as sp
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import linear as OL, jacobian as J
from oracle.foam_mesh import Geometry
from dafoam_amd.meshgen import bench_channel_case
from dafoam_amd.pyDASolvers import pyDASolvers
from common import options, norm_states
nx,ny,nz=[int(v) for v in sys.argv[1:4]]
case=bench_channel_case(nx,ny,nz); g=Geometry(case.mesh); N=g.nC
sc=J.state_scales(case,g,norm_states(case)); con=J.connectivity(case,g); col,_=J.greedy_coloring(con)
A=J.jacobian_colored(case,g,case.states,con,col,sc,mode="cs",lower_bound=0).tocsr(); n=A.shape[0]
rhs=np.zeros(n); rhs[0:3*N:3]=g.V; rhs*=sc
s=pyDASolvers(b"DASimpleFoam -python", options(case), case=case); S=s.pcStructure()
B=OL.NodeBlockILU.__new__(OL.NodeBlockILU); B.n,B.nu,B.bptr,B.bcol=n,S["nodeUnk"],S["bptr"].astype(np.int64),S["bcol"].astype(np.int64)
nu=S["nodeUnk"]
firstU=np.where((nu>=0)&(nu<3*N), nu, 10**9).min(axis=1); cell=np.where(firstU<10**9, firstU//3, -1)
i=cell%nx; j=(cell//nx)%ny; k=cell//(nx*ny); late=cell<0
print("nodes", nu.shape[0], "late", int(late.sum()), flush=True)
def order(keys, rev=()):
    ks=[(-q if t in rev else q) for t,q in enumerate(keys)]
    key=np.lexsort(ks[::-1]); return np.concatenate([key[~late[key]], key[late[key]]])
nat=np.argsort(S["natural"])
orders={"library default (rcm of the cell graph)": nat,
        "i fastest, j, k; late nodes last": order((k,j,i)),
        "k fastest, i, j slowest": order((j,i,k)),
        "j fastest, k, i slowest (planes of constant x)": order((i,k,j)),
        "same, x descending (against the flow)": order((i,k,j), rev=(0,)),
        "i fastest, x descending": order((k,j,i), rev=(2,))}
for nm,o in orders.items():
    twin=B.scalar_twin(A,node_order=o)
    x,info=OL.gmres(lambda v:A@v,rhs,twin,restart=1000,max_iters=600,rel_tol=1e-6)
    print(f"channel {nx}x{ny}x{nz}: ILU order {nm:50s} iterations {info['iters']}", flush=True)
