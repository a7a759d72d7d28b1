export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
timeout 18 python bench.py --nx 20 --ny 10 --nz 8 --steps 5 --warmup 5 --no-cpu --krylov-gb 1 > $O/bench_tiny.json 2> $O/bench_tiny.err; echo rc $?; tail -c 400 $O/bench_tiny.json; tail -c 300 $O/bench_tiny.err
