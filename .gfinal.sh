set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 700 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --no-cpu --no-solve > $O/prof_bench.log 2>&1
ls $O/prof/* | head
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o pmc -- python bench.py --steps 5 --warmup 102 --no-cpu --no-solve > $O/pmc_$ctr.log 2>&1
done
python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_per_kernel.json; head -c 1500 $O/pmc_per_kernel.json
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
