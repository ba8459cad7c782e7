"""`kernels` of two bench.py JSON lines side by side (round N-1 vs round N): python tools/compare_kernel_tables.py old.json new.json"""
import json
import sys

old, new = (json.load(open(p)) for p in sys.argv[1:3])
ko, kn = old.get("kernels", {}), new.get("kernels", {})
print(f"# {sys.argv[1]} -> {sys.argv[2]}")
print(f"# value: {old['value']:.2f} -> {new['value']:.2f} {new['unit']}; roofline.frac (K1 forward in the step): "
      f"{old['roofline']['frac']:.3f} -> {new['roofline']['frac']:.3f}")
for name, c in (("c2", "configs"), ("c3", "configs"), ("c4", "configs")):
    a, b = old.get(c, {}).get(name, {}), new.get(c, {}).get(name, {})
    if a and b:
        print(f"# {name}: {a['episodes_per_s']:.1f} -> {b['episodes_per_s']:.1f} episodes/s")
print(f"{'kernel':40s} {'bound':5s} {'old us':>9s} {'new us':>9s} {'old frac':>9s} {'new frac':>9s}  unit")
for name in list(kn):
    e = kn[name]
    if not isinstance(e, dict) or "frac" not in e:
        continue
    o = ko.get(name)
    ou = f"{o['us']:9.1f}" if isinstance(o, dict) and "us" in o else "        -"
    of = f"{o['frac']:9.3f}" if isinstance(o, dict) and "frac" in o else "        -"
    print(f"{name:40s} {e['bound']:5s} {ou} {e['us']:9.1f} {of} {e['frac']:9.3f}  {e['unit']}")
