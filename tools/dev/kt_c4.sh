# On the GPU box: rocprofv3 kernel trace of the LARGE configurations (config 4 at 215 MB and 21 MB, config 5 at 210 MB; eight device-resident calls each):
# k_pretok / k_group_scan / k_tile_out / k_memo_fill by launch -> gpurun_out/kt_c4.txt (per-kernel statistics), gpurun_out/kt_c4_timeline.txt (the last launches in order)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
d=$R/gpurun_out/kt_c4; rm -rf $d; mkdir -p $d
cat > /tmp/c4_once.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, encode_device, reserve
dev = torch.device("cuda", 0)
for vocab, gen, n in (("llama3", "c4", 1000000), ("llama3", "c4", 100000), ("deepseek_v3", "c5", 100)):
    b = DeviceBatch(getattr(corpus, gen)(n), dev)
    tok = Tokenizer.from_pretrained(vocab); reserve(tok, b.n_bytes + (1 << 20), b.n_docs + 16)
    for _ in range(8): encode_device(tok, b)
    torch.cuda.synchronize()
    del tok, b
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $d -o p -- python /tmp/c4_once.py > $d/log.txt 2>&1
python tools/rocpd_timeline.py $(find $d -name "*.db" | head -1) 2>/dev/null | tail -40 > gpurun_out/kt_c4_timeline.txt
python tools/rocpd_summary.py $(find $d -name "*.db" | head -1) > gpurun_out/kt_c4.txt 2>&1
rm -rf $d
cat gpurun_out/kt_c4.txt
