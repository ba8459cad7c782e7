"""Steady-state per-episode kernel table from a rocprofv3 kernel trace of bench.py.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --no-cpu-baseline ...
    python tools/steady_profile.py DIR [episodes_in_window] > profiles/rNN/steady_*.txt

A `--stats` summary of the whole process mixes the timed steps with the warm-up, where MIOpen's
first-use search runs every candidate solver (naive_conv, per-image GEMM convolutions ...).
The episode loop is periodic, so the window between the K1-forward launch number (last - E) and
the last K1-forward launch holds exactly E whole episodes of steady-state work whatever the
phase; this script aggregates the kernels that START inside that window.
"""
import collections
import csv
import glob
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("fpsg::(anonymous namespace)::", "fpsg::")
    name = re.sub(r"at::native::(\(anonymous namespace\)::)?", "at::", name)
    return name[:110]


def main():
    d = sys.argv[1]
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + d)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "chamfer_tile_kernel" in r[2] or "chamfer_fwd_kernel" in r[2]]
    per_episode = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # K1-forward launches per episode
    want = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    E = min(want, (len(marks) - 1) // per_episode)
    lo, hi = marks[-1 - E * per_episode], marks[-1]
    win = rows[lo:hi]
    wall = (rows[hi][0] - rows[lo][0]) / 1e6
    agg = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n in win:
        a = agg[short(n)]
        a[0] += 1
        a[1] += (e - s) / 1e3
    busy = sum(a[1] for a in agg.values()) / 1e3
    print(f"# steady-state window: {E} episodes, {len(win)} kernel launches, wall {wall:.1f} ms "
          f"({wall / E:.2f} ms/episode), kernel time {busy:.1f} ms ({busy / E:.2f} ms/episode, "
          f"{100 * busy / wall:.0f} % of wall)")
    print(f"# {'kernel':110s} {'calls/ep':>8s} {'avg us':>9s} {'ms/ep':>8s} {'%':>6s}")
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if us / 1e3 / E < 0.02:
            continue
        print(f"{n:112s} {c / E:8.1f} {us / c:9.1f} {us / 1e3 / E:8.3f} {100 * us / 1e3 / busy:6.2f}")
    groups = collections.OrderedDict([
        ("MIOpen / rocBLAS convolution + GEMM", r"miopen|igemm|Cijk|naive_conv|ck::|_ZN2ck|transpose|Im2d|Col2Im|SubTensor|gemm"),
        ("hand-written (libfpsg_hip.so)", r"fpsg::"),
        ("PyTorch elementwise / reduce / pool / optimizer", r"."),
    ])
    tot = collections.OrderedDict((g, 0.0) for g in groups)
    for n, (c, us) in agg.items():
        for g, pat in groups.items():
            if re.search(pat, n):
                tot[g] += us / 1e3
                break
    print("# groups (ms/episode):")
    for g, ms in tot.items():
        print(f"#   {g:50s} {ms / E:8.2f}")


if __name__ == "__main__":
    main()
