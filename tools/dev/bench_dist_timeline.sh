# kernel timeline of the world-1 rehearsal of bench.py's distributed legs at the wave sizes of an 8-GPU run; prints a stretch of the calibration
# in which consecutive tile kernels run on two streams
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/tlb; rm -rf $O; mkdir -p $O
SPL_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 rocprofv3 --kernel-trace -d $O/kt -o p -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-throughputs --no-c2-wide --regions 1 --c4-part-docs 15600 --no-c5 > $O/log.txt 2>&1
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/tlb/kt/**/*.db', recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
ev = list(db.execute(f"select d.start, d.end, s.kernel_name, d.queue_id, d.stream_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
pk = [i for i, e in enumerate(ev) if "k_pretokILi864" in e[2]]
# first index where three consecutive big tile kernels alternate streams
at = None
for a, b, c in zip(pk, pk[1:], pk[2:]):
    if ev[a][4] != ev[b][4] and ev[a][4] == ev[c][4]: at = a; break
print("events", len(ev), "first alternation at", at)
def show(ev):
    t0 = ev[0][0]
    for e in ev:
        print(f"{(e[0]-t0)/1e3:9.1f} {(e[1]-t0)/1e3:9.1f} {(e[1]-e[0])/1e3:7.1f}  {e[2][:34]:34s} q{e[3]} s{e[4]}")
if at is not None: show(ev[at + 40: at + 110])
PY
rm -rf $O/kt
