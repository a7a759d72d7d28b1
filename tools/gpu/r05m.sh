# round 4, call 5m: what CPU parallelism does the bench host really give the container? (cgroup quota, affinity, STREAM vs thread count)
O=gpurun_out/r05m; mkdir -p $O
{ echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) period $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"; echo "nproc: $(nproc)"; lscpu | grep -E "^CPU\(s\)|Thread|Core|Socket|NUMA node\(s\)|Model name"; } > $O/host_cpu.txt 2>&1
python - >> $O/host_cpu.txt 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
print("affinity", len(os.sched_getaffinity(0)))
from oracle import linear as OL
for nt in (8, 16, 32, 64, 96, 128, 192, 256):
    K = OL.OmpKrylov(nt)
    t = time.time(); g = K.stream_GBps(1 << 27, 3)
    print(f"threads {nt:4d}: STREAM triad {g:7.1f} GB/s  ({time.time() - t:.1f} s)", flush=True)
    del K
PY
cat $O/host_cpu.txt
