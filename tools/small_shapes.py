#!/usr/bin/env python3
"""Which schedule wins on small grids?  Times v1 (128-row / 128-key workgroups) vs v2 (256-row pairs) per kernel
for the reference's benchmark shapes (B=4, H=8) via the FA_*_IMPL / *_V1 environment switches (separate processes)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for S in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "256,512,1024,2048".split(","))]:
    for env, tag in ((["--impl", "1,1,1"], "v1"), (["--impl", "2,2,2"], "v2")):
        e = dict(os.environ)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools/kbench.py"), "--batch", "4", "--heads", "8", "--seq", str(S),
                              "--rounds", "3", "--reps", "20", "--warm-ms", "100"] + env, env=e, capture_output=True, text=True).stdout
        print("S=%d %s: " % (S, tag) + " | ".join(l.split("median")[0].split()[0] + l.split("median")[1].split("(")[0] for l in out.strip().splitlines()), flush=True)
