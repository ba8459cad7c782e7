"""Host-side selection logic that needs no GPU: Winograd eligibility / tile-size rules, the
transformed-filter cache's lifetime, GEMM-records plumbing, fused BatchNorm dispatch on CPU tensors."""
import os

import pytest
import torch
import torch.nn as nn


def test_winograd_tile_and_width_rules(monkeypatch):
    from fpsg_amd import winograd as wg
    monkeypatch.delenv("FPSG_WINOGRAD_M", raising=False)
    monkeypatch.delenv("FPSG_WINOGRAD_RAGGED", raising=False)
    # 14 and 30 are even but not multiples of 4: F(4x4) with half-empty edge tiles (winograd.ragged_enabled)
    assert [wg.tile_size(s, s) for s in (224, 112, 56, 28, 14, 24, 30, 10, 6)] == [4, 4, 4, 4, 4, 2, 4, 2, 2]
    monkeypatch.setenv("FPSG_WINOGRAD_RAGGED", "0")
    assert [wg.tile_size(s, s) for s in (224, 28, 14, 30)] == [4, 4, 2, 2]
    monkeypatch.delenv("FPSG_WINOGRAD_RAGGED")
    monkeypatch.setenv("FPSG_WINOGRAD_M", "2")
    assert wg.tile_size(224, 224) == 2
    monkeypatch.delenv("FPSG_WINOGRAD_M")
    # VGG16 at 224x224: every layer but conv1_1 (3 input channels) qualifies
    plan = [(3, 64, 224), (64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56),
            (256, 512, 28), (512, 512, 28), (512, 512, 14)]
    got = [wg._wide_enough(c, k, wg.tile_size(h, h)) for c, k, h in plan]
    assert got == [False] + [True] * 8
    assert not wg._wide_enough(64, 128, 2) and wg._wide_enough(128, 256, 2)      # 2x2 tiles need wider layers


def test_winograd_eligibility_needs_gpu_fp32_3x3(monkeypatch):
    from fpsg_amd import winograd as wg
    conv = nn.Conv2d(64, 64, 3, padding=1)
    assert not wg.eligible(torch.randn(1, 64, 8, 8), conv)                        # CPU tensor
    monkeypatch.setenv("FPSG_WINOGRAD", "0")
    assert not wg.enabled()
    monkeypatch.delenv("FPSG_WINOGRAD")
    assert wg.enabled()
    with pytest.raises(ValueError):
        wg.conv3x3(torch.randn(1, 4, 8, 8), torch.randn(4, 5, 3, 3))
    with pytest.raises(ValueError):
        wg.conv3x3(torch.randn(1, 4, 6, 8), torch.randn(4, 4, 3, 3), 4)
    monkeypatch.setenv("FPSG_WINOGRAD_FUSED", "0")
    px = 37 * 224 * 224
    assert not wg._can_fuse(4, 64, 64, px)
    monkeypatch.delenv("FPSG_WINOGRAD_FUSED")
    assert wg._can_fuse(4, 64, 128, px) and not wg._can_fuse(4, 128, 64, px) and not wg._can_fuse(2, 64, 64, px)
    # K6f addresses its input with 32-bit byte offsets: 4 GiB and above takes the three-kernel form
    assert wg._can_fuse(4, 64, 64, (1 << 24) - 1) and not wg._can_fuse(4, 64, 64, 1 << 24)
    # K6f tiles whole 4x4 blocks: F(4x4) with ragged edge tiles (14x14, 30x18) keeps the three-kernel form
    assert wg._can_fuse(4, 64, 64, 3 * 16 * 16, 16, 16) and not wg._can_fuse(4, 64, 64, 3 * 14 * 14, 14, 14)
    assert not wg._can_fuse(4, 64, 64, 2 * 30 * 16, 30, 16) and not wg._can_fuse(4, 64, 64, 2 * 16 * 18, 16, 18)


def test_filter_cache_lives_only_inside_weights_frozen():
    from fpsg_amd import winograd as wg
    assert wg._frozen_cache is None
    with wg.weights_frozen():
        assert wg._frozen_cache == {}
        wg._frozen_cache["k"] = 1
        with wg.weights_frozen():                       # nested: same cache
            assert wg._frozen_cache == {"k": 1}
        assert wg._frozen_cache == {"k": 1}
    assert wg._frozen_cache is None


def test_gemm_tuning_is_off_without_gpu_and_records_are_well_formed():
    from fpsg_amd import gemm_tuning
    if not torch.cuda.is_available():
        assert gemm_tuning.enable() == {"gemm_tuning": "off"}
    lines = open(gemm_tuning.DEFAULT_FILE).read().strip().splitlines()
    validators = [l for l in lines if l.startswith("Validator,")]
    records = [l for l in lines if not l.startswith("Validator,")]
    assert {v.split(",")[1] for v in validators} >= {"PT_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION", "GCN_ARCH_NAME"}
    assert any("gfx950" in v for v in validators)
    assert len(records) > 100
    for r in records:
        op, shape, solution = r.split(",")[:3]
        assert op.startswith("Gemm") and "_" in shape
        assert solution.startswith(("Gemm_Rocblas_", "Gemm_Hipblaslt_", "Default"))


def test_fused_bn_helpers_run_the_plain_modules_on_cpu():
    """CPU tensors (the CPU port used by the parity tests) take PyTorch's own modules: same values as
    the module chain, including the max-over-points and pooled forms."""
    from fpsg_amd.fused_bn import conv_bn_act, conv_bn_act_max, conv_bn_act_pool
    torch.manual_seed(0)
    conv1, bn1 = nn.Conv1d(8, 16, 1), nn.BatchNorm1d(16)
    x = torch.randn(3, 8, 128)
    import copy
    c2, b2 = copy.deepcopy(conv1), copy.deepcopy(bn1)
    assert torch.equal(conv_bn_act_max(conv1, bn1, x, "relu"), torch.relu(b2(c2(x))).max(dim=2)[0])
    conv2, bn2, pool = nn.Conv2d(4, 6, 3, padding=1), nn.BatchNorm2d(6), nn.MaxPool2d(2, 2)
    c3, b3 = copy.deepcopy(conv2), copy.deepcopy(bn2)
    img = torch.randn(2, 4, 8, 8)
    assert torch.equal(conv_bn_act_pool(conv2, bn2, pool, img, "relu"), pool(torch.relu(b3(c3(img)))))
    c4, b4 = copy.deepcopy(c3), copy.deepcopy(b3)
    assert torch.equal(conv_bn_act(c3, b3, img, ("leaky", 0.2)), torch.nn.functional.leaky_relu(b4(c4(img)), 0.2))


def test_deferred_batch_counters_match_immediate_increments():
    """fpsg_amd.bn_counters: inside deferred() the num_batches_tracked increments of the fused BatchNorm paths are
    collected and applied as multi-tensor adds at the end of the block -- same values as incrementing on the spot,
    also for a module that runs twice and for nested blocks."""
    import torch
    from fpsg_amd import bn_counters
    a, b, c = (torch.nn.BatchNorm1d(4) for _ in range(3))
    c.num_batches_tracked = None                       # track_running_stats=False style module
    b.num_batches_tracked += 5
    bn_counters.count_batch(a)
    assert int(a.num_batches_tracked) == 1
    with bn_counters.deferred():
        bn_counters.count_batch(a)
        bn_counters.count_batch(b)
        with bn_counters.deferred():
            bn_counters.count_batch(a)
        bn_counters.count_batch(c)
        assert int(a.num_batches_tracked) == 1 and int(b.num_batches_tracked) == 5     # not yet
    assert int(a.num_batches_tracked) == 3 and int(b.num_batches_tracked) == 6
    assert bn_counters._tls.pending is None


def test_deferred_running_statistics_keep_order_and_values():
    """bn_counters.update_running inside deferred(): a tensor updated twice in the block gets both updates in order
    (rounds), tensors of different calls share the multi-tensor ops; values are those of immediate updates bit for bit."""
    import torch
    from fpsg_amd import bn_counters
    torch.manual_seed(0)
    a0, b0, c0 = torch.randn(5), torch.randn(5), torch.randn(3)
    adds = [torch.randn(5), torch.randn(5), torch.randn(3), torch.randn(5)]

    def run(deferred):
        a, b, c = a0.clone(), b0.clone(), c0.clone()
        ctx = bn_counters.deferred() if deferred else None
        if ctx:
            ctx.__enter__()
        bn_counters.update_running([a, b], 0.9, [adds[0], adds[1]])
        bn_counters.update_running([c], 0.81, [adds[2]])
        bn_counters.update_running([a], 0.9, [adds[3]])            # second update of `a`
        if ctx:
            assert torch.equal(a, a0)                               # nothing applied yet
            ctx.__exit__(None, None, None)
        return a, b, c

    for x, y in zip(run(True), run(False)):
        assert torch.equal(x, y)
    a, _, _ = run(True)
    assert torch.equal(a, (a0 * 0.9 + adds[0]) * 0.9 + adds[3])
    assert bn_counters._tls.stats is None


def test_flat_layout_keeps_the_decoder_stack_groups_together():
    """fpsg_amd.optim.layout_order: reverse registration order, but the same tensor of the 16 patch MLPs (a stack
    group tagged by PCDecoder) sits in one run, in patch order -- so that densely packed it is a contiguous
    [16, ...] block; every parameter appears once; an incomplete group falls back to the plain order."""
    import copy
    from fpsg_amd.engine import default_options
    from fpsg_amd.optim import flat_layout, layout_order
    from fpsg_amd.point_cloud_net import PCDecoder
    dec = PCDecoder(conf=default_options(device="cpu"))
    ps = list(dec.parameters())
    order = layout_order(ps)
    assert len(order) == len(ps) and {id(p) for p in order} == {id(p) for p in ps}
    layout, total = flat_layout(ps)
    assert total == sum(p.numel() for p in ps) and [id(p) for p, _, _ in layout] == [id(p) for p in order]
    off_of = {id(p): off for p, off, _ in layout}
    nodes = [n for c in dec.cluster_pool for n in c.node_pool]
    defs = [c.deformer for c in dec.cluster_pool]
    for mods in (nodes, defs):
        for name, p0 in mods[0].named_parameters():
            offs = [off_of[id(dict(m.named_parameters())[name])] for m in mods]
            assert offs == [offs[0] + i * p0.numel() for i in range(len(mods))], name
    # state-dict keys are the reference's
    assert "cluster_pool.0.node_pool.1.conv1.weight" in dec.state_dict()
    # a group with a member missing (a frozen parameter) is laid out in the plain order
    some = [p for p in ps if p is not nodes[3].conv2.weight]
    plain = layout_order(some)
    idx = [i for i, p in enumerate(plain)
           if hasattr(p, "_fpsg_stack") and p._fpsg_stack[0][1:] == ("node", "conv2.weight")]
    assert len(idx) == len(nodes) - 1 and idx != list(range(idx[0], idx[0] + len(idx)))
    # copies lose the tags (torch's Parameter.__deepcopy__): they are simply laid out in the plain order
    clone = copy.deepcopy(dec)
    assert [id(p) for p in layout_order(list(clone.parameters()))] == [id(p) for p in reversed(list(clone.parameters()))]


def test_deferred_bookkeeping_is_per_thread():
    """a second thread's training-mode forward while a deferred() block is open neither joins the block
    nor loses its own update."""
    import threading
    import torch
    from fpsg_amd import bn_counters
    a, b = torch.nn.BatchNorm1d(3), torch.nn.BatchNorm1d(3)
    with bn_counters.deferred():
        bn_counters.count_batch(a)
        t = threading.Thread(target=lambda: bn_counters.count_batch(b, 2))
        t.start()
        t.join()
        assert int(b.num_batches_tracked) == 2          # applied at once: the other thread has no open block
        assert int(a.num_batches_tracked) == 0          # still pending in this thread's block
    assert int(a.num_batches_tracked) == 1 and int(b.num_batches_tracked) == 2
