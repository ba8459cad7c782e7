"""Library-GEMM kernel selection for the path's plain matrix products (decoder MLPs, PointNet
1x1 convolutions, the transform-domain products of K6).

The fp32 GEMMs are library calls (hipBLASLt / rocBLAS through ``torch.bmm`` / ``baddbmm`` /
``conv1d``); which of the libraries' kernels runs for a shape is left to a default heuristic
that is visibly off for the batched, odd-sized shapes of this path (36 x [512x512]x[512x1813],
[64x64] with a 116,032-long reduction ...: 44-99 TFLOP/s of a 157 TFLOP/s peak).  PyTorch's
TunableOp times every applicable library kernel per shape once and records the winner;
``fpsg_amd/tuning/gemm_gfx950.csv`` holds those records for the BASELINE workloads on MI355X
(ROCm 7.2 image; the file carries version validators and is ignored on a mismatch), produced by
``tools/tune_gemm.py``.  ``enable()`` only LOADS that file: shapes not in it run the default
kernel, nothing is timed at run time unless ``tune=True``.

``FPSG_GEMM_TUNING=0`` switches this off (A/B measurements), ``=tune`` times unknown shapes
online (minutes of warm-up) and appends them to the file named by ``FPSG_GEMM_TUNING_FILE``.
"""
from __future__ import annotations

import os

import torch

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "gemm_gfx950.csv")


def enable(path: str | None = None, tune: bool | None = None) -> dict:
    """Turns on recorded GEMM kernel selection; returns what was done (for logs / bench JSON)."""
    mode = os.environ.get("FPSG_GEMM_TUNING", "file")
    if mode in ("0", "off") or not torch.cuda.is_available():
        return {"gemm_tuning": "off"}
    if tune is None:
        tune = mode == "tune"
    path = path or os.environ.get("FPSG_GEMM_TUNING_FILE") or DEFAULT_FILE
    import torch.cuda.tunable as tunable
    if not tune and not os.path.exists(path):
        return {"gemm_tuning": "off (no records file)"}
    tunable.enable(True)
    records = path
    if not tune:
        # TunableOp may rewrite its file when the process ends; the ranks of a job share the
        # committed records, so each process works on a private copy
        import shutil
        import tempfile
        fd, records = tempfile.mkstemp(prefix="fpsg_gemm_records_", suffix=".csv")
        os.close(fd)
        shutil.copyfile(path, records)
        import atexit
        atexit.register(lambda p=records: os.path.exists(p) and os.remove(p))
    tunable.set_filename(records, insert_device_ordinal=False)
    tunable.tuning_enable(bool(tune))
    if tune:
        # FPSG_GEMM_TUNE_MS / _ITERS: time per candidate kernel; FPSG_GEMM_TUNE_ROTATE_MB: size of the operand copies the
        # timing loop rotates through (TunableOp's default only defeats the L2; the 256 MB Infinity Cache keeps the
        # operands of most of this path's products warm, which the step's products do not find)
        tunable.set_max_tuning_duration(int(os.environ.get("FPSG_GEMM_TUNE_MS", "30")))
        tunable.set_max_tuning_iterations(int(os.environ.get("FPSG_GEMM_TUNE_ITERS", "10")))
        if os.environ.get("FPSG_GEMM_TUNE_ROTATE_MB"):
            tunable.set_rotating_buffer_size(int(os.environ["FPSG_GEMM_TUNE_ROTATE_MB"]))
    loaded = bool(os.path.exists(records) and tunable.read_file(records))
    return {"gemm_tuning": "tune" if tune else "file", "gemm_records": os.path.basename(path),
            "gemm_records_loaded": loaded}


def disable() -> None:
    if torch.cuda.is_available():
        import torch.cuda.tunable as tunable
        tunable.enable(False)
