cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/tl4; rm -rf $O; mkdir -p $O
PERTURB=1 GPU_MAX_HW_QUEUES=4 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt -o p -- python tools/dev/pipeline_ab.py c3 - d: sdma:sdma_d2h=1 nopick:pick_streams=0 d2: > $O/log.txt 2>&1
grep GB/s $O/log.txt
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/tl4/kt/**/*.db', recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
ev = [(s, e, n[:30], f"q{q} s{st}") for s, e, n, q, st in db.execute(f"select d.start, d.end, s.kernel_name, d.queue_id, d.stream_id from {kd} d join {ks} s on d.kernel_id=s.id")]
mc = [t for t in tabs if t.startswith("rocpd_memory_copy")]
cols = [r[1] for r in db.execute(f"pragma table_info({mc[0]})")]
print(cols)
ev += [(s, e, f"copy {sz} B", f"q{q} s{st}") for s, e, sz, q, st in db.execute(f"select start, end, size, queue_id, stream_id from {mc[0]}")]
ev.sort()
ev = ev[-70:]
t0 = ev[0][0]
for s, e, n, q in ev: print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:7.1f}  {n:30s} {q}")
PY
rm -rf $O/kt
