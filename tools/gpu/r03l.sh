# round 3, call l: tile launch order of the residual kernels (parity + assembly time + PMC bytes), kernel profile of the block path
export TMPDIR=/tmp
O=gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "residual_parity or drdwt or bench_size or rho or turbo or cyclic or naca" > $O/pytest_tile.log 2>&1; tail -5 $O/pytest_tile.log | cut -c1-220
cd /tmp; R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o blk -- python $R/tools/block_bench.py --n 250 100 80 --iters 30 --nrhs 4 > $R/$O/block_prof.log 2>&1
cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/block_kernel_stats_2M.csv
grep -E "^single|^block" $O/block_prof.log | cut -c1-300
head -16 $O/block_kernel_stats_2M.csv | cut -c1-200
rm -rf $O/prof
