"""BASELINE configs[2] alone (bench.py's configs2 leg) — the command profiled for profiles/<tag>_antwall_kernel_stats.md."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.configs2_leg(0, steps=int(os.environ.get("STEPS", "2")), warmup=1)))
