# First GPU call of the next round (prepared at the end of round 4, when the GPU budget was spent): everything that could not be
# re-measured on the final tree.  ~25 GPU-minutes.
#   1. the full GPU tier on the final tree (round 4 verified only the new tests: profiles/r05n_*, and the bench line: r05j / r05l)
#   2. the default bench line WITH the CPU legs (cpu port at the bench size with 16 threads, psi parity leg)
#   2b. amd.gmresDeflation (GMRES-DR): the experimental GPU test, then the 2 M-cell wing with a 300- and a 200-vector basis
#   3. PMC passes on the wing workload (roofline.traffic is null for it) + the counter calibration on pure streams of 4 / 8 / 16 B per lane
export TMPDIR=/tmp
O=gpurun_out/r06a; mkdir -p $O
timeout 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log | cut -c1-200
DAS_TEST_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_gpu_zzz_edge_cases.py -q -s -k gmres_deflated > $O/pytest_gmres_dr.log 2>&1; tail -4 $O/pytest_gmres_dr.log | cut -c1-200
#   (then, if green: the deflated solver at size - 2 M-cell wing, basis 302 / 202 vectors instead of ~970)
timeout 500 python tools/naca_adjoint_sweep.py --nz 160 --dz 0.025 --blend 0.5 --combos a:deflated:rcb:1 --maxit 1500 --deflation 300:100 200:70 > $O/wing_gmres_dr.log 2> $O/wing_gmres_dr.err; grep SWEEP $O/wing_gmres_dr.log
DAS_BENCH_VERBOSE=1 timeout 700 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; grep "^\[bench" $O/bench.err | cut -c1-250
cd /tmp; R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -o p -- python $R/bench.py --no-cpu --no-parity --no-solve --window-at-warmup --steps 5 --warmup 440 > /dev/null 2> $R/$O/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write -o p -- python $R/bench.py --no-cpu --no-parity --no-solve --window-at-warmup --steps 5 --warmup 440 > /dev/null 2> $R/$O/pmc_write.err
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_per_kernel_wing2M.json 2> $O/pmc_summary.err
hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calib tools/gpu/pmc_calib.hip && /tmp/pmc_calib 8 > $O/pmc_calib_times.log
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/calib_fetch -o c -- /tmp/pmc_calib 8 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/calib_write -o c -- /tmp/pmc_calib 8 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/calib_fetch $O/calib_write > $O/pmc_calibration.json   # expected per launch: reads 8 GiB, gather 3 GiB (+ x reuse), write 8 GiB
rm -rf $O/pmc_fetch $O/pmc_write $O/calib_fetch $O/calib_write
