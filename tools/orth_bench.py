"""Dev tool (GPU): the two kernels of the delayed re-orthogonalisation (k_multidot2, k_dcgs2_update) on synthetic vectors -
variants of rows per thread / unroll, no mesh and no setup.  Prints ms and TB/s per variant (8 (K + 2) n bytes per kernel)."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16053000)
ap.add_argument("--K", type=int, nargs="+", default=[150])
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
import __graft_entry__ as ge
ge.build()
from dafoam_amd import _capi
L = _capi.lib()
for K in a.K:
    gb = 8.0 * (K + 2) * a.n / 1e9
    for rows in (4, 8, 16):
        d, u = C.c_double(-1), C.c_double(-1)
        _capi.check(L.das_debug_orth_bench(a.n, K, a.reps, rows, 0, 0, C.byref(d), C.byref(u)))
        print(f"K {K} k_multidot2<{rows:2d}>: {d.value:7.3f} ms  {gb / d.value:6.2f} TB/s", flush=True)
    for unroll, rpt in ((4, 1), (8, 1), (16, 1), (4, 2), (8, 2), (4, 4), (8, 4)):
        d, u = C.c_double(-1), C.c_double(-1)
        _capi.check(L.das_debug_orth_bench(a.n, K, a.reps, 0, unroll, rpt, C.byref(d), C.byref(u)))
        print(f"K {K} k_dcgs2_update<{unroll:2d},{rpt}>: {u.value:7.3f} ms  {gb / u.value:6.2f} TB/s", flush=True)
