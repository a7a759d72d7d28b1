# round 3, call m: block path after the TN rewrite (tests, per-RHS rate, kernel stats) and the run-based tile order
export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "block_gmres or residual_parity_simplefoam or renumbering" > $O/pytest_block.log 2>&1; tail -4 $O/pytest_block.log | cut -c1-220
timeout 900 python tools/tile_order_bench.py 250 100 80 0 2048 8192 > $O/tile_order.log 2>&1; grep tileLaunchOrder $O/tile_order.log
cd /tmp; R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o blk -- python $R/tools/block_bench.py --n 250 100 80 --iters 30 --nrhs 4 > $R/$O/block_prof.log 2>&1
cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/block_kernel_stats_2M.csv
grep -E "^single|^block" $O/block_prof.log | cut -c1-300
grep -E "tsgemm|spmm|sweep_m|k_cell|k_grad" $O/block_kernel_stats_2M.csv | cut -d, -f1-4 | cut -c1-60,200-260
rm -rf $O/prof
