#!/usr/bin/env python3
"""pmc.txt (tools/profile_round5.sh) + kernel_stats.txt -> <tag>_hbm_traffic.json and <tag>_pmc_sq.json, the two files bench.py
names as the source of `roofline.traffic` and `roofline_valu`.   usage: pmc_to_json.py pmc.txt kernel_stats.txt outdir [round tag, default r05]"""
import json
import re
import sys

pmc, stats, out = sys.argv[1:4]
TAG = sys.argv[4] if len(sys.argv) > 4 else "r05"
vals = {}
for ln in open(pmc):
    m = re.match(r"(\S+)\s+(\S+)\s+per launch\s+([\d.]+)\s+launches\s+(\d+)", ln)
    if m:
        vals.setdefault(m.group(1), {})[m.group(2)] = float(m.group(3))
        vals[m.group(1)]["_launches"] = int(m.group(4))
dur = {}
for ln in open(stats):
    for k in ("k_pretok", "k_tile_out"):
        if k in ln.split()[0] if ln.split() else False:
            f = ln.split()
            dur[k] = float(f[2])          # avg_us
hbm = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum (separate passes, --kernel-trace only; "
               "tools/profile_round5.sh), the bench command with --steps 20; per-dispatch averages. FETCH_SIZE / WRITE_SIZE are in KB. "
               "bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 is the guide's correction for wide coalesced streams (128-B requests "
               "tallied at 64 B) and therefore an UPPER bound here: most fetches of k_pretok are 32/48/64-byte table buckets "
               "(profiles/r02_fetch_calibration.txt); the lower bound is (FETCH_SIZE + WRITE_SIZE)*1024."}
for k, v in vals.items():
    if "FETCH_SIZE" in v:
        hit, miss = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
        hbm[k] = {"FETCH_SIZE_kb": v["FETCH_SIZE"], "WRITE_SIZE_kb": v.get("WRITE_SIZE", 0),
                  "bytes_per_launch": round((2 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0)) * 1024),
                  "bytes_per_launch_lower_bound": round((v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0)) * 1024),
                  "TCC_HIT_sum": round(hit), "TCC_MISS_sum": round(miss), "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss else None,
                  "launches": v["_launches"]}
json.dump(hbm, open(f"{out}/{TAG}_hbm_traffic.json", "w"), indent=1)
sq = {"_how": "rocprofv3 --pmc <one group per run> --kernel-trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-throughputs "
              "--no-c4 --no-c5 --no-c2-wide --regions 1 (tools/profile_round5.sh); per-dispatch averages over the rotating 1000 x ~1 KB cl100k bench batches; raw "
              f"numbers in {TAG}_pmc_passes.txt; kernel_us from the kernel trace of the bench command ({TAG}_kernel_stats.txt). kernel_cycles = "
              "kernel_us x 2400 MHz. valu_issue_frac = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x kernel_cycles); lane_utilisation = "
              "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU x 4) x 4."}
for k, v in vals.items():
    if "SQ_INSTS_VALU" in v and k in dur:
        cyc = round(dur[k] * 2400)
        e = {"kernel_us": dur[k], "kernel_cycles": cyc}
        for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                  "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in v:
                e[c] = round(v[c])
        if "SQ_ACTIVE_INST_VALU" in v and "SQ_THREAD_CYCLES_VALU" in v:
            e["lane_utilisation"] = round(v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"] * 4) * 4, 3)
            e["valu_issue_frac"] = round(v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), 4)
        sq[k] = e
json.dump(sq, open(f"{out}/{TAG}_pmc_sq.json", "w"), indent=1)
print(json.dumps({k: v for k, v in sq.items() if k != "_how"}, indent=0)[:600])
