"""K11 ``fpsg_gemm_f32_nn`` (batched fp32 GEMM on the fp32 matrix pipe, hand-written persistent kernel with consumer and
producer waves) through the C ABI against a float64 product and the library fp32 GEMM of the same operands; exact on small
integers (any lane / register / k-order map error shows); ragged rows, columns and reductions; ranges that start inside a
batch entry.  The products it stands in for: torchvision ``vgg16_bn.features`` in the Winograd domain (reference
``src/models/image_net.py:14,21-24``), the decoder's wide layers (``src/models/point_cloud_net.py:66-79``)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from fpsg_amd.gemm_split import bmm_f32  # noqa: E402

SHAPES = [(2, 256, 300, 32), (3, 100, 72, 48), (1, 512, 260, 128), (2, 37, 1000, 72), (36, 256, 592, 256),
          (5, 512, 1812, 64), (1, 256, 40000, 16), (7, 300, 36, 20), (9, 256, 7252, 48), (16, 769, 4736, 1540)]


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("b,M,N,K", SHAPES)
def test_matches_float64_like_the_library(gpu, variant, b, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(b * 999 + M + N + K)
    A = torch.randn(b, M, K, generator=g).to(gpu)
    B = torch.randn(b, K, N, generator=g).to(gpu)
    C = torch.full((b, M, N), float("nan"), device=gpu)
    bmm_f32(A, B, variant, out=C)
    nb = min(b, 3)
    ref = torch.bmm(A[:nb].double(), B[:nb].double())
    scale = float(ref.pow(2).mean().sqrt())
    err = float((C[:nb].double() - ref).abs().max()) / scale
    lib_err = float((torch.bmm(A[:nb], B[:nb]).double() - ref).abs().max()) / scale
    assert err <= max(2.5 * lib_err, 4e-7 * max(1.0, (K / 256) ** 0.5)), (err, lib_err)
    if b > nb:      # the other batch entries against the library's fp32 product
        lib = torch.bmm(A[nb:], B[nb:])
        assert float((C[nb:] - lib).abs().max()) / scale < 2e-5


def test_exact_on_integers(gpu):
    g = torch.Generator(device="cpu").manual_seed(5)
    A = torch.randint(-8, 9, (2, 300, 80), generator=g).float().to(gpu)
    B = torch.randint(-8, 9, (2, 80, 500), generator=g).float().to(gpu)
    for v in (0, 1, 2, 3, -1):
        assert torch.equal(bmm_f32(A, B, v), torch.bmm(A.double(), B.double()).float())


def test_deterministic_and_loud_on_misaligned_rows(gpu):
    from fpsg_amd._hip import FpsgHipError
    g = torch.Generator(device="cpu").manual_seed(6)
    A = torch.randn(4, 256, 64, generator=g).to(gpu)
    B = torch.randn(4, 64, 1812, generator=g).to(gpu)
    assert torch.equal(bmm_f32(A, B), bmm_f32(A, B))
    with pytest.raises(FpsgHipError, match="multiples of 4"):
        bmm_f32(A, B[:, :, :1811].contiguous())
