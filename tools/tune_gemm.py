"""Regenerates fpsg_amd/tuning/gemm_gfx950.csv: runs the BASELINE workloads (and the evaluation
forward) once with PyTorch TunableOp timing every library GEMM kernel for each new shape.

    python tools/tune_gemm.py gpurun_out/gemm_gfx950.csv      # on an MI355X; minutes
    cp gpurun_out/gemm_gfx950.csv fpsg_amd/tuning/gemm_gfx950.csv

Each workload runs in its own process (TunableOp writes its records at process exit and reads
the existing file at start, so the records accumulate)."""
import os
import subprocess
import sys

out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_gfx950.csv")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.dirname(out), exist_ok=True)
seed = os.path.join(root, "fpsg_amd", "tuning", "gemm_gfx950.csv")
if not os.path.exists(out) and os.path.exists(seed) and "--fresh" not in sys.argv:
    import shutil
    shutil.copyfile(seed, out)          # keep the existing records, time only shapes that are new
for wl, extra in (("c5", []), ("c4", []), ("c2", ["--no-graph"]), ("c3", [])):
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", wl, "--gemm-tuning", "tune", "--gemm-records", out,
           "--no-cpu-baseline", "--steps", "1", "--warmup", "1"] + extra
    print("[tune_gemm]", " ".join(cmd), flush=True)
    rc = subprocess.call(cmd, stdout=subprocess.DEVNULL)
    n = sum(1 for line in open(out) if not line.startswith("Validator")) if os.path.exists(out) else 0
    print(f"[tune_gemm] {wl}: rc={rc}, {n} records", flush=True)
    if rc:
        sys.exit(rc)
