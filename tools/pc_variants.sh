for cfg in "512 4" "512 12" "256 16" "1024 8" "256 32"; do set -- $cfg
  DAS_HIPCC_FLAGS="-DPC_THREADS=$1 -DPC_PF=$2" python -c "import __graft_entry__ as g; g.build(force=True)" >/dev/null 2>&1
  echo "== threads $1 prefetch $2"
  python tools/adjoint_study.py --block 1024 --overlap 1 --fill 0 1 --maxit 100 --restart 100 2>&1 | grep "^block"
done
