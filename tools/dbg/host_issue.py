"""How long does the HOST need to issue one D+G step (no device sync inside)?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-fast-mode"]
# reuse bench's builder
import importlib
src = open("bench.py").read()
ns = {}
args = bench.parse_args() if hasattr(bench, "parse_args") else None
print("has parse_args", args is not None)
