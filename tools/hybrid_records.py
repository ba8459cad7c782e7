"""Builds a TunableOp records file from a base file with the entries of selected shapes taken from another file.

    python tools/hybrid_records.py <base.csv> <other.csv> <out.csv> [--min-ms 0.2] [--only substring,substring]

Entries of <other> replace those of <base> for shapes whose BASE time is at least --min-ms (and, with --only, whose key
contains one of the substrings).  Used for same-box A/B runs of kernel choices inside the step (tools/ab_records.sh): the
kernel TunableOp times fastest in isolation is not always the one under which the step runs fastest (the card's clock
under load depends on the kernels around it)."""
import argparse

ap = argparse.ArgumentParser()
ap.add_argument("base"); ap.add_argument("other"); ap.add_argument("out")
ap.add_argument("--min-ms", type=float, default=0.0)
ap.add_argument("--only", default="")
args = ap.parse_args()


def load(path):
    head, rows = [], {}
    for line in open(path):
        if line.startswith("Validator"):
            head.append(line)
            continue
        op, key, sol, t = line.rstrip("\n").split(",")
        rows[(op, key)] = (sol, t)
    return head, rows


head, base = load(args.base)
_, other = load(args.other)
only = [s for s in args.only.split(",") if s]
n = 0
for k, (sol, t) in list(base.items()):
    if k in other and float(t) >= args.min_ms and (not only or any(s in k[1] for s in only)) and other[k][0] != sol:
        base[k] = other[k]
        n += 1
with open(args.out, "w") as f:
    f.writelines(head)
    for (op, key), (sol, t) in base.items():
        f.write(f"{op},{key},{sol},{t}\n")
print(f"{n} entries replaced -> {args.out}")
