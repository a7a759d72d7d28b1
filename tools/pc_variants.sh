for cfg in "-DSPMV_LANES=32 -DSPMV_UNROLL=2" "-DSPMV_LANES=32 -DSPMV_UNROLL=4" "-DSPMV_LANES=16 -DSPMV_UNROLL=2" "-DSPMV_LANES=16 -DSPMV_UNROLL=4" "-DSPMV_LANES=16 -DSPMV_UNROLL=8" "-DSPMV_LANES=8 -DSPMV_UNROLL=8"; do
  DAS_HIPCC_FLAGS="$cfg" python -c "import __graft_entry__ as g; g.build(force=True)" >/dev/null 2>&1
  echo "== $cfg"; python tools/spmv_bench.py 2>&1 | grep "^spmv"
done
