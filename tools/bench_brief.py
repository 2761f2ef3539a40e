"""Condense one bench.py JSON line (file argument, or stdin) to: grid, it/s, per-kernel average ms."""
import json
import sys

text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads(text.strip().splitlines()[-1])
k = d.get("kernels", {})
brief = {n: (round(v.get("avg_ms", v.get("ms", float("nan"))), 4) if isinstance(v, dict) else v) for n, v in k.items()}
print(d["config"].get("grid"), round(d["value"], 1), brief)
