#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc.sh into profiles/pmc_traffic.json.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly half the bytes of a wide
coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Both counters are in KiB.
Output: HBM bytes per launch for each kernel, keyed by the C-ABI entry point name used in bench.py, plus the hash of the
kernel sources and the shape the counters were collected on: bench.py reports `roofline.traffic` only when both match
the run it is doing (otherwise null), so a kernel change can never leave a stale number in the bench line.

    python tools/pmc_traffic.py <pmc dir> <out.json> [shape key, default "B4,H32,N4096,D64,causal,bf16"]"""
import collections, csv, glob, importlib.util, json, os, sys
src, dst = sys.argv[1], sys.argv[2]
shape = sys.argv[3] if len(sys.argv) > 3 else "B4,H32,N4096,D64,causal,bf16"
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(_root, "bench.py"))
_bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bench)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "fa::" in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[row["Kernel_Name"].split("fa::")[1].split("<")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = {"fa_fwd2_kernel": "fa_fwd", "fa_bwd_dq2_kernel": "fa_bwd_dq", "fa_bwd_dkv2_kernel": "fa_bwd_dkv",
         "fa_fwd3_kernel": "fa_fwd", "fa_bwd_dq3_kernel": "fa_bwd_dq", "fa_fwd4_kernel": "fa_fwd", "fa_bwd_dkv3_kernel": "fa_bwd_dkv",
         "fa_fwd_kernel": "fa_fwd", "fa_bwd_dq_kernel": "fa_bwd_dq", "fa_bwd_dkv_kernel": "fa_bwd_dkv",
         "fa_bwd_dq4_kernel": "fa_bwd_dq", "fa_bwd_dkv4_kernel": "fa_bwd_dkv"}
out = {"source_hash": _bench.kernel_source_hash(), "shape": shape}
for k, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        fetch = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) * 1024 * 2   # KiB -> B, x2 gfx950 correction
        write = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024
        out[names.get(k, k)] = {"bytes": round(fetch + write), "read_bytes": round(fetch), "write_bytes": round(write),
                                "kernel": k, "note": "FETCH_SIZE x2 (gfx950) + WRITE_SIZE, mean per launch"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
