"""Like tools/pmc_families.py but for EVERY kernel of the run (library GEMMs included; the warm-up steps are
profiled too, so kernels of MIOpen's first-use search appear with inflated counts):
    python tools/pmc_all_kernels.py <fetch_dir> <write_dir> <episodes profiled>"""
import collections, csv, glob, re, sys
def load(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            n = re.sub(r"^void ", "", r["Kernel_Name"]); n = n.replace("fpsg::(anonymous namespace)::", "fpsg::")
            n = re.sub(r"at::native::(\(anonymous namespace\)::)?", "at::", n)[:90]
            a = agg[n]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg
f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
eps = float(sys.argv[3])
rows = sorted(set(f) | set(w), key=lambda k: -(2 * f[k][1] + w[k][1]))
tot = 0
print(f"# per episode; HBM MB = (2*FETCH_SIZE + WRITE_SIZE)*1024/1e6")
for k in rows:
    rd, wr = 2 * f[k][1] * 1024 / eps / 1e6, w[k][1] * 1024 / eps / 1e6
    tot += rd + wr
for k in rows[:40]:
    rd, wr = 2 * f[k][1] * 1024 / eps / 1e6, w[k][1] * 1024 / eps / 1e6
    print(f"{k:92s} {max(f[k][0], w[k][0]) / eps:7.1f} {rd:9.1f} {wr:9.1f} {rd + wr:9.1f}")
print(f"# total {tot:.0f} MB per episode")
