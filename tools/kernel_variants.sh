#!/bin/bash
# Dev tool (GPU box): rebuild the library with a list of -D variants and time one kernel.
#   SpMV sweep (what chose SPMV_LANES=16, SPMV_UNROLL=4):
#     bash tools/kernel_variants.sh spmv "-DSPMV_LANES=32 -DSPMV_UNROLL=2" "-DSPMV_LANES=16 -DSPMV_UNROLL=4" ...
#   PC apply ablations / variants (PC_SEGREDUCE, PC_PF, PC_EXP_NOATOMIC, PC_EXP_NOBARRIER):
#     bash tools/kernel_variants.sh pc "-DPC_SEGREDUCE=0" "-DPC_SEGREDUCE=1" "-DPC_SEGREDUCE=1 -DPC_PF=4"
# The last build of the loop stays in dafoam_amd/lib - rebuild the default afterwards:
#     python -c "import __graft_entry__ as g; g.build(force=True)"
what=$1; shift
for cfg in "$@"; do
  DAS_HIPCC_FLAGS="$cfg" python -c "import __graft_entry__ as g; g.build(force=True)" >/dev/null 2>&1
  echo "== $cfg"
  if [ "$what" = spmv ]; then python tools/spmv_bench.py 2>&1 | grep "^spmv"
  else python tools/adjoint_study.py --n 40 30 24 --block 1024 --overlap 1 --fill 1 --maxit 40 --restart 40 2>&1 | grep "^block"; fi
done
