"""Prints one line per row of a tools/bench_gemm_split.py result file."""
import json
import sys

for l in open(sys.argv[1]):
    r = json.loads(l)
    if r.get('leg') == 'pack_a':
        print(f"{r['shape']:14s} pack_a {r['us']} us")
        continue
    cells = []
    for k, v in r.items():
        if k.startswith("v") and isinstance(v, dict):
            cells.append(f"{k}:{v['us']:7.1f}us {v['speedup']:5.2f}x err {v['err_max_ratio']:.2f}/{v['err_med_ratio']:.2f}")
    print(f"{r['shape']:14s} {r['leg']:3s} lib {r['lib_us']:6.1f}us ({r['lib_TFLOPs']:5.1f} TF) | " + " | ".join(cells))
