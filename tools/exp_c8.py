import sys, runpy
sys.path.insert(0, "/root/repo")
import satt_amd
from satt_amd import ops
ops.ATTN_CLUSTER_SIZES = (8, 4, 2)
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-decode"]
runpy.run_path("/root/repo/bench.py", run_name="__main__")
