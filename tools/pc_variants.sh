for cfg in "-DSPMV_LANES=64 -DSPMV_NT=0" "-DSPMV_LANES=64 -DSPMV_NT=1" "-DSPMV_LANES=32 -DSPMV_NT=0" "-DSPMV_LANES=32 -DSPMV_NT=1" "-DSPMV_LANES=16 -DSPMV_NT=1" "-DSPMV_LANES=8 -DSPMV_NT=1"; do
  DAS_HIPCC_FLAGS="$cfg" python -c "import __graft_entry__ as g; g.build(force=True)" >/dev/null 2>&1
  echo "== $cfg"; python tools/spmv_bench.py 2>&1 | grep "^spmv"
done
