cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/tlw; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt -o p -- python tools/dev/wave_gather_ab.py > $O/log.txt 2>&1
tail -1 $O/log.txt
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/tlw/kt/**/*.db', recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
print(cols)
q = "queue_id" if "queue_id" in cols else None
st = "stream_id" if "stream_id" in cols else None
sel = f"select d.start, d.end, s.kernel_name{', d.'+q if q else ''}{', d.'+st if st else ''} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"
ev = list(db.execute(sel))
mc = [t for t in tabs if t.startswith("rocpd_memory_copy")]
# find the region of the one-stream timed steps: print 70 events starting 1/3 into the run and the last 70
def show(ev):
    t0 = ev[0][0]
    for e in ev:
        print(f"{(e[0]-t0)/1e3:9.1f} {(e[1]-t0)/1e3:9.1f} {(e[1]-e[0])/1e3:7.1f}  {e[2][:34]:34s} " + " ".join(str(x) for x in e[3:]))
n = len(ev)
print("events", n)
show(ev[n//4: n//4 + 50])
print("----- late (two streams)")
show(ev[n*5//8: n*5//8 + 50])
PY
rm -rf $O/kt
