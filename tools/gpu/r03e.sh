# round 3, call e: new GPU tests (packed operator, at-size NACA0012), rocprofv3 kernel stats + PMC passes at 2 M cells, NACA adjoint study
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "packed or config1" > $O/pytest_new.log 2>&1; tail -4 $O/pytest_new.log | cut -c1-200
cd /tmp; R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu --no-solve --steps 100 --warmup 100 > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -o p -- python $R/bench.py --no-cpu --no-solve --steps 5 --warmup 102 > /dev/null 2> $R/$O/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write -o p -- python $R/bench.py --no-cpu --no-solve --steps 5 --warmup 102 > /dev/null 2> $R/$O/pmc_write.err
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_per_kernel_2M.json 2> $O/pmc_summary.err
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats_2M.csv
head -14 $O/bench_kernel_stats_2M.csv | cut -c1-160
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
timeout 900 python tools/adjoint_study.py --case naca --n 800 250 1 --restart 1000 --maxit 1000 --krylov-gb 100 > $O/study_naca200k.log 2>&1
grep -E "^pc |coloring|dRdWT|hist" $O/study_naca200k.log | cut -c1-300
