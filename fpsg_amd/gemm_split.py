"""K10: batched fp32 matrix products on the bf16 matrix pipe with three-way split operands (``csrc/gemm_split.hip``,
``fpsg_gemm_split``), opt-in with ``FPSG_GEMM_SPLIT=1``.

The library fp32 GEMMs behind the Winograd-domain products of the image trunk (``torch.bmm`` -> rocBLAS / hipBLASLt on
``v_mfma_f32_*_f32``, 1/16 of the bf16 MFMA rate) are 58 % of a c5 episode's kernel time.  Here every fp32 operand is
split exactly into three bf16 pieces on its way into LDS and the six leading cross products are accumulated in fp32 by
``v_mfma_f32_32x32x16_bf16``: fp32-grade results on a pipe with a 2.67x higher roof.  Not bit-identical to the library's
fp32 products, hence opt-in: the headline benchmark line stays on the fp32 MFMA (``dtype: f32``).
Reference layers: torchvision ``vgg16_bn.features`` (``src/models/image_net.py:14,21-24``)."""
from __future__ import annotations

import os

import torch

from . import _hip


def enabled() -> bool:
    return os.environ.get("FPSG_GEMM_SPLIT", "0") == "1"


def bmm_split(A: torch.Tensor, B: torch.Tensor, transB: bool = False, variant: int = -1, out: torch.Tensor | None = None):
    """``torch.bmm(A, B)`` (``transB``: ``torch.bmm(A, B.transpose(1, 2))``) for contiguous fp32 ``A [b, M, K]`` and
    ``B [b, K, N]`` (``[b, N, K]``) on a ROCm device.  Raises when the HIP library is missing: no fallback."""
    lib = _hip.load()
    _hip.dev_tensor(A, torch.float32, "A")
    _hip.dev_tensor(B, torch.float32, "B")
    b, M, K = A.shape
    N = B.shape[1] if transB else B.shape[2]
    if B.shape[0] != b or (B.shape[2] if transB else B.shape[1]) != K:
        raise ValueError(f"bmm_split: shapes {tuple(A.shape)} x {tuple(B.shape)} (transB={transB}) do not match")
    C = out if out is not None else torch.empty((b, M, N), dtype=torch.float32, device=A.device)
    nws = lib.fpsg_gemm_split_workspace_floats(b, M, N, K, 1 if transB else 0, variant)
    ws = torch.empty((nws,), dtype=torch.float32, device=A.device) if nws else None
    with torch.cuda.device(A.device):
        _hip.check(lib.fpsg_gemm_split(_hip.ptr(A), _hip.ptr(B), _hip.ptr(C), b, M, N, K, K, K if transB else N, N,
                                       M * K, B.shape[1] * B.shape[2], M * N, 1 if transB else 0, variant,
                                       _hip.ptr(ws) if ws is not None else None, nws, _hip.stream_of(A)),
                   "fpsg_gemm_split")
    return C


def pack_a(A: torch.Tensor, variant: int = -1) -> torch.Tensor:
    """The A operand ``[b, M, K]`` of ``bmm_packed`` split once into three bf16 planes in the layout of the GEMM's LDS
    image (``fpsg_gemm_split_pack_a``): for operands that stay constant over many products -- the transformed filters
    of an optimizer step.  Returns an opaque uint8 tensor (about 1.5x the bytes of ``A``)."""
    lib = _hip.load()
    _hip.dev_tensor(A, torch.float32, "A")
    b, M, K = A.shape
    nbytes = lib.fpsg_gemm_split_packed_a_bytes(b, M, K, variant)
    if nbytes == 0:
        raise ValueError(f"pack_a: variant {variant} unknown")
    Ap = torch.empty((nbytes,), dtype=torch.uint8, device=A.device)
    with torch.cuda.device(A.device):
        _hip.check(lib.fpsg_gemm_split_pack_a(_hip.ptr(A), b, M, K, K, M * K, variant, _hip.ptr(Ap), _hip.stream_of(A)),
                   "fpsg_gemm_split_pack_a")
    return Ap


def bmm_packed(Ap: torch.Tensor, shape_a, B: torch.Tensor, variant: int = -1, out: torch.Tensor | None = None):
    """``torch.bmm(A, B)`` for ``Ap = pack_a(A, variant)``, ``shape_a = A.shape`` and contiguous fp32 ``B [b, K, N]``:
    the same values as ``bmm_split(A, B)``; A comes in by LDS-DMA, only B is split inside the kernel."""
    lib = _hip.load()
    _hip.dev_tensor(B, torch.float32, "B")
    b, M, K = shape_a
    if B.shape[0] != b or B.shape[1] != K:
        raise ValueError(f"bmm_packed: shapes {tuple(shape_a)} x {tuple(B.shape)} do not match")
    N = B.shape[2]
    C = out if out is not None else torch.empty((b, M, N), dtype=torch.float32, device=B.device)
    with torch.cuda.device(B.device):
        _hip.check(lib.fpsg_gemm_split_nn_packed(_hip.ptr(Ap), _hip.ptr(B), _hip.ptr(C), b, M, N, K, N, N, K * N, M * N,
                                                 variant, _hip.stream_of(B)), "fpsg_gemm_split_nn_packed")
    return C


def bmm_persistent(Ap: torch.Tensor, shape_a, B: torch.Tensor, variant: int = -1, out: torch.Tensor | None = None):
    """``bmm_packed`` as one persistent launch (``fpsg_gemm_split_nn_persistent``); ``Ap = pack_a(A, 0)``."""
    lib = _hip.load()
    _hip.dev_tensor(B, torch.float32, "B")
    b, M, K = shape_a
    if B.shape[0] != b or B.shape[1] != K:
        raise ValueError(f"bmm_persistent: shapes {tuple(shape_a)} x {tuple(B.shape)} do not match")
    N = B.shape[2]
    C = out if out is not None else torch.empty((b, M, N), dtype=torch.float32, device=B.device)
    with torch.cuda.device(B.device):
        _hip.check(lib.fpsg_gemm_split_nn_persistent(_hip.ptr(Ap), _hip.ptr(B), _hip.ptr(C), b, M, N, K, N, N, K * N,
                                                     M * N, variant, _hip.stream_of(B)), "fpsg_gemm_split_nn_persistent")
    return C


def bmm_f32(A: torch.Tensor, B: torch.Tensor, variant: int = -1, out: torch.Tensor | None = None):
    """``torch.bmm(A, B)`` for contiguous fp32 ``A [b, M, K]``, ``B [b, K, N]`` by K11 (``fpsg_gemm_f32_nn``: fp32 MFMA,
    hand-written persistent kernel).  Raises ``FpsgHipError`` where the entry point's alignment rules do not hold."""
    lib = _hip.load()
    _hip.dev_tensor(A, torch.float32, "A")
    _hip.dev_tensor(B, torch.float32, "B")
    b, M, K = A.shape
    if B.shape[0] != b or B.shape[1] != K:
        raise ValueError(f"bmm_f32: shapes {tuple(A.shape)} x {tuple(B.shape)} do not match")
    N = B.shape[2]
    C = out if out is not None else torch.empty((b, M, N), dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        _hip.check(lib.fpsg_gemm_f32_nn(_hip.ptr(A), _hip.ptr(B), _hip.ptr(C), b, M, N, K, K, N, N, M * K, K * N, M * N,
                                        variant, _hip.stream_of(A)), "fpsg_gemm_f32_nn")
    return C


def f32_eligible(b: int, M: int, N: int, K: int) -> bool:
    """Shapes ``fpsg_gemm_f32_nn`` serves for contiguous operands (16-byte DMA pieces)."""
    return K % 4 == 0 and N % 4 == 0 and (M * K) % 4 == 0 and (K * N) % 4 == 0
