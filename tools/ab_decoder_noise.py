import sys, copy, os, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_decoder_gpu as t
from _gradcheck import deviations, summarize
gpu = torch.device('cuda:0')
for B in (5, 32):
    dec, hidden, grids = t._make(B, 7); dec.train()
    ref = copy.deepcopy(dec).double(); ref.batched = False
    h64 = hidden.double().requires_grad_()
    out64 = t._literal(ref, h64, [[g.double() for g in c] for c in grids])
    w = torch.randn(out64.shape, generator=torch.Generator().manual_seed(3))
    (out64 * w.double()).sum().backward()
    truth = {"latent.x.grad": h64.grad, **{n: q.grad for n, q in ref.named_parameters()}}
    for label, env, batched in (("hip batched", {}, True), ("library BN batched", {"FPSG_FUSED_BN": "0"}, True),
                                ("hip looped", {}, False), ("library BN looped", {"FPSG_FUSED_BN": "0"}, False),
                                ("batched, default gemm", {"PYTORCH_TUNABLEOP_ENABLED": "0"}, True)):
        os.environ.pop("FPSG_FUSED_BN", None)
        os.environ.update(env)
        if "PYTORCH_TUNABLEOP_ENABLED" in env:
            torch.cuda.tunable.enable(False)
        dev = copy.deepcopy(dec).to(gpu); dev.batched = batched
        hg = hidden.to(gpu).requires_grad_()
        out = dev(hg, grid=[[g.to(gpu) for g in c] for c in grids])
        (out * w.to(gpu)).sum().backward()
        got = {"latent.x.grad": hg.grad, **{n: q.grad for n, q in dev.named_parameters()}}
        print(B, label, "fwd err", float((out.detach().cpu().double() - out64.detach()).abs().max()), summarize(deviations(got, truth)[0]))
