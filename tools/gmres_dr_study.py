"""Dev tool (CPU only): would DEFLATED RESTARTING keep the adjoint's iteration count with a fraction of the Krylov basis?  numpy
prototype of GMRES-DR(m, k) (Morgan 2002: k harmonic Ritz vectors carried across restarts) on the exact matrices of the host-emulated
kernel bodies at a converged NACA0012 section (optionally extruded), preconditioned by the host restatement of the node-block ILU(0)
with amd.pcUpwindBlend 0.5.  Round 4: profiles/r05_cpu_gmres_dr_study.log.
   python tools/gmres_dr_study.py <naca_primal_NXxNY.npz> <nz>"""
import sys, time, numpy as np, scipy.sparse as sp, scipy.linalg as sla, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests")); sys.path.insert(0, os.path.join(ROOT,"tools"))
from oracle import linear as OL
from dafoam_amd.meshgen import naca0012_case, extrude_naca_state
from dafoam_amd.pyDASolvers import pyDASolvers
from common import options
from naca_newton_cpu_twin import Twin

def gmres_dr(op, b, m, k, tol=1e-6, maxit=3000):
    """Right-preconditioned GMRES-DR(m,k) (Morgan 2002); op = A M^-1.  Returns (iterations, rel residual, history)."""
    n=b.size; beta=np.linalg.norm(b); r=b.copy(); x=np.zeros(n); its=0; hist=[1.0]
    V=np.zeros((n,m+1)); H=np.zeros((m+1,m)); V[:,0]=r/beta; c=np.zeros(m+1); c[0]=beta; j0=0
    while its<maxit:
        for j in range(j0,m):
            w=op(V[:,j]); h=V[:,:j+1].T@w; w-=V[:,:j+1]@h; h2=V[:,:j+1].T@w; w-=V[:,:j+1]@h2; h+=h2
            H[:j+1,j]=h; H[j+1,j]=np.linalg.norm(w); V[:,j+1]=w/H[j+1,j]; its+=1
            y,res,_,_=np.linalg.lstsq(H[:j+2,:j+1],c[:j+2],rcond=None)
            rn=np.linalg.norm(c[:j+2]-H[:j+2,:j+1]@y); hist.append(rn/beta)
            if rn<=tol*beta or its>=maxit: return its,rn/beta,hist
        y,_,_,_=np.linalg.lstsq(H,c,rcond=None)
        x+=V[:,:m]@y
        rvec=c-H@y                       # residual in the V_{m+1} basis
        if k==0:
            r=V@rvec; beta2=np.linalg.norm(r); V[:,0]=r/beta2; c[:]=0; c[0]=beta2; H[:]=0; j0=0; continue
        # harmonic Ritz vectors: (H_m + h^2 f e_m^T) g = theta g with f = H_m^-T e_m
        Hm=H[:m,:m]; em=np.zeros(m); em[-1]=1.0
        f=np.linalg.solve(Hm.T,em)
        G=Hm+H[m,m-1]**2*np.outer(f,em)
        ev,evec=np.linalg.eig(G)
        idx=np.argsort(np.abs(ev))[:k]
        # real basis of the selected (possibly complex-conjugate) eigenvectors
        cols=[]
        for i in idx:
            v=evec[:,i]
            cols.append(v.real)
            if np.abs(v.imag).max()>1e-12: cols.append(v.imag)
        Pk=np.array(cols).T[:,:k]
        Pk,_=np.linalg.qr(Pk)
        P1=np.zeros((m+1,k+1)); P1[:m,:k]=Pk
        rv=rvec-P1[:,:k]@(P1[:,:k].T@rvec); P1[:,k]=rv/np.linalg.norm(rv)
        Hnew=P1.T@H@Pk                    # (k+1) x k
        Vnew=V@P1
        V[:,:k+1]=Vnew; H[:]=0; H[:k+1,:k]=Hnew
        c=np.zeros(m+1); c[:k+1]=P1.T@rvec
        j0=k
    return its,hist[-1],hist

d=np.load(sys.argv[1]); nx,ny=[int(v) for v in d["dims"]]; fc=float(d["first_cell"]); nz=int(sys.argv[2])
c2=naca0012_case(nx,ny,1,first_cell=fc,perturb=0.0); c2.states=d["states"]; case=c2
if nz>1:
    case=naca0012_case(nx,ny,nz,span=0.1*nz,first_cell=fc,perturb=0.0,y_wall_section=c2.y_wall); case.states=extrude_naca_state(c2,c2.states,case,(nx,ny,nz))
T=Twin(case); N=T.N; n=T.n; W=case.states
A=T.jac(W).T.tocsr(); P0=T.jac(W,1).T.tocsr(); P=(P0+0.5*(A-P0)).tocsr()
rhs=np.zeros(n); rhs[0:3*N:3]=1.0/N
s=pyDASolvers(b"DASimpleFoam -python", options(case), case=case); S=s.pcStructure()
K=OL.OmpKrylov(8); K.set_operator(A); K.set_pc_bilu(P,S)
op=lambda v: A@K.pc_solve(v)
for (m,k) in ((600,0),(100,0),(60,20),(100,30),(100,50),(150,50)):
    t=time.time(); its,rel,h=gmres_dr(op,rhs,m,k,maxit=1500)
    print(f"{nx}x{ny}x{nz}: GMRES-DR(m={m}, k={k}): iterations {its} rel {rel:.1e}  basis vectors {m+1}  ({time.time()-t:.0f} s)",flush=True)
