"""Dev tool: time dRdW^T.psi (das_drdwt_mult_device) on the bench matrix."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from dafoam_amd.meshgen import bench_channel_case
from dafoam_amd.pyDAFoam import PYDAFOAM
from dafoam_amd import _capi
nx, ny, nz = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (100, 50, 40))]
case = bench_channel_case(nx, ny, nz)
D = PYDAFOAM(options={"solverName": "DASimpleFoam", "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}, "adjEqnOption": {"printInfo": 0}}, case=case)
D.solver.runColoring(); D.solverAD.initializedRdWTMatrixFree()
L = _capi.lib(); h = D.solver._h; n = D.getNLocalAdjointStates(); nnz = L.das_op_nnz(h)
x = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
for _ in range(5): L.das_drdwt_mult_device(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()))
torch.cuda.synchronize(); L.das_timer_reset(h); L.das_timer_enable(h, 1)
for _ in range(50): L.das_drdwt_mult_device(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()))
torch.cuda.synchronize()
ms = L.das_timer_avg_ms(h, b"spmv"); B = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n
print(f"spmv {ms:.4f} ms  {B/ms/1e6:.1f} GB/s  frac {B/ms/1e6/8000:.3f}  (n={n}, nnz={nnz})  checksum {float(y.sum()):.6e}")
