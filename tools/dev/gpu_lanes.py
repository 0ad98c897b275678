"""Dev aid: C-ABI host->host rate with 1, 2, 3 pipelines on ONE GPU (spl_set_devices lists the ordinal several times)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import host_path_bench as h
for cfg in sys.argv[1:] or ["c3", "c4"]:
    for devs in ([0], [0, 0], [0, 0, 0]):
        r = h.measure(cfg, python_surface=False, devices=devs)
        print(cfg, "pipelines", len(devs), {k: r[k] for k in ("kernel_hbm", "c_abi_host", "c_abi_host_pageable", "decode_host")}, flush=True)
