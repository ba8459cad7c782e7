"""K7 (fpsg_adam_step through fpsg_amd.optim.FlatAdam) against torch.optim.Adam -- the optimizer the
reference constructs at src/trainNetwork.py:118-123 -- and the float64 oracle: parameters and moments
over several steps with a StepLR decay in between, state-dict round trip, flat-buffer plumbing."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _net():
    torch.manual_seed(0)
    # no BatchNorm behind a biased convolution: such a bias has a round-off gradient, which Adam's
    # normalisation turns into +-lr steps that no two runs share
    return nn.Sequential(nn.Conv1d(3, 37, 1), nn.Tanh(), nn.Conv1d(37, 5, 1), nn.Flatten(), nn.Linear(5 * 31, 7))


def _loss(net, x):
    return net(x).square().mean()


def test_matches_torch_adam_over_steps(gpu):
    from fpsg_amd.optim import FlatAdam
    a, b = _net().to(gpu), None
    b = copy.deepcopy(a)
    opt_a = FlatAdam(a.parameters(), lr=3e-3, betas=(0.9, 0.999))
    opt_b = torch.optim.Adam(b.parameters(), lr=3e-3, betas=(0.9, 0.999))
    sched_a = torch.optim.lr_scheduler.StepLR(opt_a, step_size=3, gamma=0.5)
    sched_b = torch.optim.lr_scheduler.StepLR(opt_b, step_size=3, gamma=0.5)
    # parameters now live in one flat buffer, values untouched
    flat = opt_a.flat_param
    for p in a.parameters():
        assert flat.data_ptr() <= p.data_ptr() < flat.data_ptr() + flat.numel() * 4
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    torch.manual_seed(1)
    for it in range(8):
        x = torch.randn(16, 3, 31, device=gpu)
        for net, opt, sched in ((a, opt_a, sched_a), (b, opt_b, sched_b)):
            opt.zero_grad(set_to_none=True)
            _loss(net, x).backward()
            opt.step()
            sched.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-5, atol=2e-7), (it, float((pa - pb).abs().max()))
    assert opt_a.param_groups[0]["lr"] == opt_b.param_groups[0]["lr"] == 3e-3 * 0.25
    for pa, pb in zip(a.parameters(), b.parameters()):
        sa, sb = opt_a.state[pa], opt_b.state[pb]
        assert float(sa["step"].detach()) == float(sb["step"].detach()) == 8
        # eight steps of fp32 recurrences on gradients that already differ in the last digits
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=2e-4, atol=1e-9)
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-4, atol=1e-12)


def test_kernel_against_float64_oracle(gpu, oracle):
    from fpsg_amd import _hip
    lib = _hip.load()
    rng = np.random.default_rng(3)
    n = 100003                                               # not a multiple of 4: the tail path
    p = rng.standard_normal(n).astype(np.float32); g = (rng.standard_normal(n) * 0.1).astype(np.float32)
    m = (rng.standard_normal(n) * 0.05).astype(np.float32); v = (rng.random(n) * 0.01).astype(np.float32)
    tp, tg, tm, tv = (torch.from_numpy(a.copy()).to(gpu) for a in (p, g, m, v))
    for t, scale in ((1, 1.0), (7, 0.125), (1000, 1.0)):
        rp, rm, rv = oracle.adam_step(p, g * np.float32(scale), m, v, t, lr=2e-3)
        qp, qm, qv = tp.clone(), tm.clone(), tv.clone()
        rc = lib.fpsg_adam_step(_hip.ptr(qp), _hip.ptr(tg), _hip.ptr(qm), _hip.ptr(qv), n, 2e-3, 0.9, 0.999, 1e-8, t,
                                scale, None)
        assert rc == 0
        for got, want in ((qp, rp), (qm, rm), (qv, rv)):
            err = np.abs(got.cpu().numpy().astype(np.float64) - want)
            assert float(err.max()) <= 1e-6 * float(np.abs(want).max()) + 1e-9
    assert lib.fpsg_adam_step(_hip.ptr(tp), _hip.ptr(tg), _hip.ptr(tm), _hip.ptr(tv), n, 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, None) != 0
    assert lib.fpsg_adam_step(_hip.ptr(tp), None, _hip.ptr(tm), _hip.ptr(tv), n, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None) != 0


def test_state_dict_round_trip_and_resume(gpu):
    from fpsg_amd.optim import FlatAdam
    a = _net().to(gpu)
    opt = FlatAdam(a.parameters(), lr=1e-3)
    x = torch.randn(8, 3, 31, device=gpu)
    for _ in range(3):
        opt.zero_grad(set_to_none=True); _loss(a, x).backward(); opt.step()
    sd = copy.deepcopy(opt.state_dict())
    weights = copy.deepcopy(a.state_dict())
    opt.zero_grad(set_to_none=True); _loss(a, x).backward(); opt.step()
    after = [p.detach().clone() for p in a.parameters()]
    # a fresh model + optimizer resumed from the saved state takes the same 4th step
    b = _net().to(gpu)
    b.load_state_dict(weights)
    opt_b = FlatAdam(b.parameters(), lr=1e-3)
    opt_b.load_state_dict(sd)
    assert opt_b._t == 3
    opt_b.zero_grad(set_to_none=True); _loss(b, x).backward(); opt_b.step()
    for pa, pb in zip(after, b.parameters()):       # (the library's conv backward is not bit-reproducible run to run)
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-8)
    # a torch.optim.Adam state dict loads too (same per-parameter entries)
    c = _net().to(gpu); c.load_state_dict(weights)
    ref = torch.optim.Adam(c.parameters(), lr=1e-3)
    ref.load_state_dict(sd)
    assert float(ref.state[next(iter(c.parameters()))]["step"]) == 3


def test_train_step_uses_the_flat_gradient_buffer_in_place(gpu):
    from fpsg_amd.engine import TrainStep, build_model, build_optimizer, default_options
    from fpsg_amd.episodes import synthetic_episode
    from fpsg_amd.optim import FlatAdam
    torch.manual_seed(0)
    opt = default_options(device="cuda", intra_recon=True)
    m = build_model(opt).to(gpu).train()
    optimizer, _ = build_optimizer(m, opt)
    assert isinstance(optimizer, FlatAdam)
    step = TrainStep(m, optimizer)
    assert optimizer._bound is step.buckets.flat
    before = optimizer.flat_param.clone()
    ep = synthetic_episode(2, 1, n_pts=2048, img_size=64, seed=3, device=gpu)
    step([ep, ep])
    assert optimizer._flat_gradient() is step.buckets.flat and optimizer._gather is None
    assert not torch.equal(before, optimizer.flat_param) and bool(torch.isfinite(optimizer.flat_param).all())


def test_mean_over_episodes_inside_the_adam_step_equals_the_separate_pass(gpu):
    """K7's ``grad_scale`` (1/E applied while the step reads the flat gradient buffer) against scaling the buffer first
    and stepping with scale 1: the same fp32 product per element, so parameters and both moments are equal bit for
    bit; ``TrainStep`` uses the folded form (the buffer then keeps the SUM) and resets the scale afterwards."""
    from fpsg_amd.optim import FlatAdam
    a, b = _net().to(gpu), _net().to(gpu)
    b.load_state_dict(a.state_dict())
    opt_a, opt_b = FlatAdam(a.parameters(), lr=2e-3), FlatAdam(b.parameters(), lr=2e-3)
    g = torch.Generator(device=gpu).manual_seed(11)
    for _ in range(3):
        flat = torch.randn(opt_a.flat_param.numel(), device=gpu, generator=g) * 3.0
        fa, fb = flat.clone(), flat.clone()
        for opt, f in ((opt_a, fa), (opt_b, fb)):
            for p, off, n in opt._layout:
                p.grad = f[off:off + n].view(p.shape)
            opt.bind_gradients(f)
        opt_a.grad_scale = 1.0 / 7
        opt_a.step()
        fb.mul_(1.0 / 7)
        opt_b.step()
        assert torch.equal(fa, flat)                                       # the folded form leaves the sum in place
    for x, y in ((opt_a.flat_param, opt_b.flat_param), (opt_a.flat_exp_avg, opt_b.flat_exp_avg),
                 (opt_a.flat_exp_avg_sq, opt_b.flat_exp_avg_sq)):
        assert torch.equal(x, y)


def test_pointer_table_step_equals_flat_step(gpu):
    """fpsg_adam_step_segments (gradients read where autograd left them) == fpsg_adam_step on a flat
    copy of the same gradients, bit for bit; a parameter without gradient counts as zero; the
    train step of one episode on one rank uses it and leaves the flat buffer untouched."""
    from fpsg_amd.optim import FlatAdam, flat_layout
    a, b = _net().to(gpu), _net().to(gpu)
    opt_a, opt_b = FlatAdam(a.parameters(), lr=2e-3), FlatAdam(b.parameters(), lr=2e-3)
    layout, total = flat_layout([p for p in b.parameters()])
    flat = torch.zeros(total, device=gpu)
    opt_b.bind_gradients(flat)
    for p, off, n in layout:
        p.grad = flat[off:off + n].view(p.shape)
    torch.manual_seed(4)
    for it in range(3):
        x = torch.randn(8, 3, 31, device=gpu)
        opt_a.zero_grad(set_to_none=True)
        _loss(a, x).backward()
        skipped = list(a.parameters())[1]
        if it == 1:
            skipped.grad = None                                   # a parameter without gradient
        with torch.no_grad():                                     # the same gradients, flat, for b
            for (pb, off, n), pa in zip(layout, reversed(list(a.parameters()))):
                flat[off:off + n].copy_((pa.grad if pa.grad is not None else torch.zeros_like(pa)).reshape(-1))
        assert opt_a._bound_gradient() is None and opt_a._pointer_table() is not None
        assert opt_b._bound_gradient() is flat
        opt_a.step(); opt_b.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.equal(pa, pb), it
    assert opt_a._gather is None


def test_single_episode_step_reads_gradients_in_place(gpu):
    from fpsg_amd.engine import TrainStep, build_model, build_optimizer, default_options
    from fpsg_amd.episodes import synthetic_episode
    torch.manual_seed(0)
    opt = default_options(device="cuda", intra_recon=True)
    m = build_model(opt).to(gpu).train()
    optimizer, _ = build_optimizer(m, opt)
    step = TrainStep(m, optimizer)
    ep = synthetic_episode(2, 1, n_pts=2048, img_size=64, seed=3, device=gpu)
    before = optimizer.flat_param.clone()
    step.buckets.flat.fill_(7.0)
    step([ep])
    assert bool((step.buckets.flat == 7.0).all())                 # no gather happened
    assert optimizer._gather is None and optimizer._gtab_host is not None
    assert not torch.equal(before, optimizer.flat_param) and bool(torch.isfinite(optimizer.flat_param).all())


def test_pointer_table_step_with_unaligned_gradient_views(gpu):
    """A gradient that autograd took over from a VIEW (e.g. a slice of a stacked tensor's gradient) may start at
    any 4-byte offset: the pointer-table kernel must not assume 16-byte alignment of the tensor start."""
    from fpsg_amd.optim import FlatAdam
    torch.manual_seed(3)
    params = [nn.Parameter(torch.randn(n, device=gpu)) for n in (1539, 8, 769, 5)]
    ref = [nn.Parameter(p.detach().clone()) for p in params]
    opt = FlatAdam(params, lr=1e-2)
    opt_ref = torch.optim.Adam(ref, lr=1e-2)
    big = torch.randn(4000, device=gpu)
    off = 1                                           # odd float offset: 4-byte aligned only
    misaligned = 0
    for p, r in zip(params, ref):
        g = big[off:off + p.numel()]
        misaligned += g.data_ptr() % 16 != 0
        p.grad = g                                    # a view with a storage offset, read in place
        r.grad = g.clone()
        off += p.numel() + 1
    assert misaligned >= 3 and opt._pointer_table() is not None
    opt.step()
    opt_ref.step()
    for p, r in zip(params, ref):
        assert torch.allclose(p, r, rtol=1e-6, atol=1e-7)


def test_gradient_accumulation_over_episodes_in_one_launch(gpu, monkeypatch):
    """FlatGradBuckets.absorb through fpsg_flat_accumulate_segments (pointer table, one launch) against the
    multi-tensor path it replaces: first episode copies (zeros for a parameter without gradient), later episodes
    add -- bit for bit (the same fp32 adds); odd sizes, a tail that is not a multiple of 4, an unaligned gradient
    view and a channels_last parameter (falls back) included."""
    from fpsg_amd.dist import FlatGradBuckets
    torch.manual_seed(3)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Parameter(torch.randn(37, 5))
            self.b = nn.Parameter(torch.randn(3))
            self.c = nn.Parameter(torch.randn(130, 7, 3))
            self.unused = nn.Parameter(torch.randn(11))
            self.d = nn.Parameter(torch.randn(2, 2))

    def run(flag, channels_last=False, lazy=False, episodes=3):
        monkeypatch.setenv("FPSG_ABSORB_SEGMENTS", flag)
        torch.manual_seed(7)
        net = Net().to(gpu)
        if channels_last:
            net.e = nn.Parameter(torch.randn(4, 6, 3, 3, device=gpu).contiguous(memory_format=torch.channels_last))
        fb = FlatGradBuckets(net, bucket_mb=0.001)
        fb.lazy = lazy
        snaps = []
        for ep in range(episodes):
            fb.detach()
            big = torch.randn(5000, device=gpu)
            for n_, p in net.named_parameters():
                if n_ == "unused" or (n_ == "d" and ep == 1):
                    continue
                if n_ == "b":
                    p.grad = big[1:4]                          # a view that starts 4 bytes past an aligned address
                else:
                    p.grad = torch.randn_like(p)
            fb.absorb(first=(ep == 0))
            if not lazy:
                snaps.append(fb.flat.clone())
        if lazy:
            assert len(fb._stash) == episodes % 8
            fb.flush()
            assert not fb._stash
            snaps.append(fb.flat.clone())
        return snaps

    for cl in (False, True):
        seg, ref = run("1", cl), run("0", cl)
        for s_, r_ in zip(seg, ref):
            assert torch.equal(s_, r_)
    # the deferred form (FlatGradBuckets.lazy, the default): the episodes' gradient tensors are kept and added by ONE
    # launch of fpsg_flat_accumulate_tables per 8 episodes -- the same sums in the same order, bit for bit
    for episodes, cl in ((3, False), (8, False), (11, False), (3, True)):
        lazy, ref = run("1", cl, lazy=True, episodes=episodes), run("0", cl, episodes=episodes)
        assert torch.equal(lazy[-1], ref[-1]), (episodes, cl)
    # the one-launch form is really taken for the plain model
    calls = {"n": 0}
    orig = FlatGradBuckets._absorb_segments

    def counted(self, first):
        ok = orig(self, first)
        calls["n"] += 1 if ok else 0
        return ok

    monkeypatch.setattr(FlatGradBuckets, "_absorb_segments", counted)
    run("1")
    assert calls["n"] == 3
