#!/usr/bin/env python3
"""Headline frame (cfg 2) for several alternative builds, alternating: ab_many.py reps name...  (names under cuda-pathtrace_amd/alt, or main)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = sys.argv[1]
for _ in range(2):
    for name in sys.argv[2:]:
        lib = os.path.join(root, "cuda-pathtrace_amd", "libptcore.so" if name == "main" else os.path.join("alt", name, "libptcore.so"))
        subprocess.call([sys.executable, os.path.join(root, "tools", "ab_raw.py"), "--child", lib, reps])
