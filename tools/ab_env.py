#!/usr/bin/env python3
"""A/B of schedule generations through the environment switches, one process per variant (NOT interleaved:
use for coarse comparisons only).  usage: ab_env.py [kbench args]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for env, tag in ((["--impl", "1,1,1"], "v1"), (["--impl", "2,2,2"], "v2"), (["--impl", "1,1,1"], "v1"), (["--impl", "2,2,2"], "v2")):
    e = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools/kbench.py")] + sys.argv[1:] + env, env=e, capture_output=True, text=True).stdout
    print("%s: " % tag + " | ".join(l.split("median")[0].split()[0] + l.split("median")[1].split("(")[0] for l in out.strip().splitlines()), flush=True)
