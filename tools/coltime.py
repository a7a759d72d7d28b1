import time, sys
sys.path.insert(0,'.')
from dafoam_amd.meshgen import channel_case
from dafoam_amd.pyDASolvers import pyDASolvers
case = channel_case(100,50,40, perturb=0.0)
s = pyDASolvers(b"DASimpleFoam -python", {"debug": True}, case=case)
t=time.time(); s.runColoring(); print("total", time.time()-t, "cells", case.mesh.n_cells, "colors", s.getColoring()[1])
