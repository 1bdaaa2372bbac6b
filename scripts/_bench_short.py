"""Condense one bench.py JSON line (stdin) to value + phase times: `python bench.py | tail -1 | python scripts/_bench_short.py label`."""
import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"]), {k: round(v, 4) for k, v in d["phases_ms"].items()})
