"""Episode forward / loss of the image-conditioned few-shot point-cloud generator.

Mirrors reference ``src/models/few_shot.py``: ``ImgPCProtoNet.__init__ :23-61``,
``.loss :63-73`` / ``._loss_single_class :75-129`` (training loss dict),
``._return_reconstruction :131-176`` (evaluation: Chamfer + EMD) and
``.draw_reconstruction :179-213`` -- same constructor, same ``sample`` dict contract
(``xs,xq,xad [1,S|Q|S,3,H,W]``, ``pcs,pcq,pcad [1,S|Q|S,N,3]``), same result keys.

The point-set distances are this package's HIP kernels (``fpsg_amd.metrics``) instead of
Kaolin / neuralnet_pytorch.  Deliberate fixes of reference defects:
no module-import-time ``.cuda()`` (the zero placeholder is created on the inputs' device),
and ``metric='emd'`` works instead of raising AttributeError.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from .metrics import chamfer_distance, emd_loss, episode_chamfer_losses
from .utils import emd_wrapper

_AGGREGATOR = ["single", "multi", "mask_single", "mask_multi"]


def _fused_losses_enabled() -> bool:
    """``FPSG_FUSED_LOSSES=0``: the loss sums as separate PyTorch operations (A/B measurements)."""
    return os.environ.get("FPSG_FUSED_LOSSES", "1") != "0"


class _SplitRows(torch.autograd.Function):
    """``(t[:n], t[n:])`` as views; the backward is ONE concatenation.  Autograd's own slicing gives each slice a
    zero-filled full-size gradient, copies the slice's gradient in and adds the two: five launches per split."""

    @staticmethod
    def forward(ctx, t, n):
        ctx.set_materialize_grads(False)
        ctx.n, ctx.rest = n, (t.shape[0] - n,) + tuple(t.shape[1:])
        return t[:n], t[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        ref = ga if ga is not None else gb
        if ga is None:
            ga = ref.new_zeros((ctx.n,) + tuple(ref.shape[1:]))
        if gb is None:
            gb = ref.new_zeros(ctx.rest)
        return torch.cat([ga, gb]), None


def _split_rows(t, n):
    if t.is_cuda and t.requires_grad and torch.is_grad_enabled():
        return _SplitRows.apply(t, n)
    return t[:n], t[n:]


class ImgPCProtoNet(nn.Module):
    def __init__(self, img_encoder, pc_encoder, pc_decoder, mask_learner=None, query_factor=1.0,
                 support_factor=1.0, metric="cd", intra_support=False, aggregate="single"):
        super().__init__()
        self.img_encoder = img_encoder
        self.pc_encoder = pc_encoder
        self.pc_decoder = pc_decoder
        self.mask_allocater = mask_learner
        self.query_factor = query_factor
        self.support_factor = support_factor
        self.intra_flag = intra_support
        if aggregate not in _AGGREGATOR:
            raise NotImplementedError(f"Found unsupported prototype aggragation: {aggregate}")
        self.aggregate = aggregate
        if metric == "cd":
            self.metric_module = None
            self.pc_metric = chamfer_distance
        elif metric == "emd":
            # training needs gradients: the differentiable approximate-assignment solver (K2)
            self.pc_metric = lambda a, b: emd_loss(a, b, reduce="sum", sinkhorn=False).reshape(1)
        else:
            raise NotImplementedError(
                f"Found unsupported point cloud reconstruction metrics: {metric}")
        # evaluation-only distance (reference few_shot.py:168); an attribute so that a test
        # can drive the module on CPU with the oracle's implementations
        self.emd_metric = emd_wrapper
        self.overlap_encoders = False      # see _encode; switched on by bench.py / the trainer
        self._side_stream = None

    # ------------------------------------------------------------------ shared forward
    def _encode(self, img_s, img_q, img_ad, pc_s, pc_ad):
        """Image features of (ad ++ query) and point features of (support ++ ad), each in one
        encoder call (reference :84-100)."""
        n_support, n_query = img_s.size(1), img_q.size(1)
        img_corpus = torch.cat([img_ad.reshape(n_support, *img_ad.shape[2:]),
                                img_q.reshape(n_query, *img_q.shape[2:])], dim=0)
        pc_corpus = torch.cat([pc_s.reshape(n_support, *pc_s.shape[2:]),
                               pc_ad.reshape(n_support, *pc_ad.shape[2:])], dim=0).transpose(2, 1)
        if self.overlap_encoders and img_corpus.is_cuda:
            # the two encoders are independent: the point encoder (memory-bound BN / max
            # passes) runs on a second HIP stream beside the compute-bound image trunk;
            # autograd replays each backward on its forward stream, so they overlap there too
            cur = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=img_corpus.device)
            side = self._side_stream
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                pc_z = self.pc_encoder(pc_corpus)
            img_z = self.img_encoder(img_corpus)
            cur.wait_stream(side)
            pc_z.record_stream(cur)
            pc_corpus.record_stream(side)
        else:
            img_z = self.img_encoder(img_corpus)
            pc_z = self.pc_encoder(pc_corpus)
        return (*_split_rows(img_z, n_support), *_split_rows(pc_z, n_support))

    def _encode_for_eval(self, img_q, pc_s):
        """``(img_zq, pc_z_proto)`` for the evaluation outputs.  The reference's ``_return_reconstruction``
        (``few_shot.py:131-176``) runs the training forward -- it also encodes the S "ad" images and the S "ad" clouds --
        and then drops ``img_zad`` / ``pc_z_ad`` unused (``:146,161``: only ``img_zq`` and ``pc_z_proto`` reach the
        decoder).  In evaluation mode (``model.eval()``, ``evaluate_Network.py:99``) BatchNorm normalises with its running
        statistics, so a sample's features do not depend on what else is in the batch: encoding only the Q query images
        and the S support clouds gives the same features (to the last bits of a GEMM's summation order: measured
        <= 1e-7 on ``cd_loss`` / ``emd_loss``) for 5/37 of the image trunk's and half of the point encoder's work.  Only
        when every module is in evaluation mode; ``FPSG_EVAL_PRUNE=0``: the full forward (A/B, tests)."""
        n_support, n_query = pc_s.size(1), img_q.size(1)
        img_zq = self.img_encoder(img_q.reshape(n_query, *img_q.shape[2:]))
        pc_z_proto = self.pc_encoder(pc_s.reshape(n_support, *pc_s.shape[2:]).transpose(2, 1))
        return img_zq, pc_z_proto

    def _eval_prune(self) -> bool:
        return (os.environ.get("FPSG_EVAL_PRUNE", "1") != "0" and not self.training
                and not any(m.training for m in self.modules()))

    def _decode_queries(self, img_zq, pc_z_proto, pack=None):
        """Class prototype = mean of the support features, broadcast to every query
        (reference :104-107)."""
        proto = pc_z_proto.mean(0, keepdim=True).expand(img_zq.size(0), -1)
        return self._decode(torch.cat([img_zq, proto], dim=1), pack)

    def _decode(self, hidden, pack=None):
        if pack is not None:
            return self.pc_decoder(hidden, pack=pack)
        return self.pc_decoder(hidden)

    def _decoder_pack(self):
        """Stacked decoder parameters shared by the episode's two decodes (PCDecoder.pack_parameters);
        None for a decoder without the batched form."""
        fn = getattr(self.pc_decoder, "pack_parameters", None)
        return fn() if fn is not None else None

    # ------------------------------------------------------------------------ training
    def loss(self, sample):
        return self._loss_single_class(sample["xs"], sample["xq"], sample["xad"], sample["pcs"],
                                       sample["pcq"], sample["pcad"])

    def _loss_single_class(self, img_s, img_q, img_ad, pc_s, pc_q, pc_ad):
        img_zad, img_zq, pc_z_proto, pc_z_ad = self._encode(img_s, img_q, img_ad, pc_s, pc_ad)
        pack = self._decoder_pack() if self.intra_flag else None
        ref_q = pc_q.squeeze(0).contiguous()
        if self.intra_flag:
            ref_s = pc_ad.squeeze(0).contiguous()
            n_q = img_zq.size(0)
            pair = getattr(self.pc_decoder, "forward_pair", None)
            if pair is not None and pack is not None:
                # the two decodes (queries, then supports) side by side: every GEMM of the decoder once (PCDecoder.forward_pair)
                proto = pc_z_proto.mean(0, keepdim=True).expand(n_q, -1)
                syn = pair(torch.cat([img_zq, proto], dim=1), torch.cat([img_zad, pc_z_ad], dim=1), pack=pack)
                syn_q, syn_s = syn[:n_q], syn[n_q:]
            else:
                syn_q = self._decode_queries(img_zq, pc_z_proto, pack)
                syn_s = self._decode(torch.cat([img_zad, pc_z_ad], dim=1), pack)
                syn = None
            if self.pc_metric is chamfer_distance and syn_q.shape[1:] == syn_s.shape[1:] \
                    and ref_q.shape[1:] == ref_s.shape[1:]:
                # the two Chamfer calls of the reference (few_shot.py:110,117) as ONE launch over
                # Q + S cloud pairs: per-pair results are independent, and a larger batch fills
                # the chip better (K1 is 6 us + 0.9 us per pair)
                if syn is None:
                    syn = torch.cat([syn_q, syn_s])
                ref = torch.cat([ref_q, ref_s])
                if syn.is_cuda and _fused_losses_enabled():
                    # K1l: the two sums and the weighted total in one launch behind K1 (and one in the backward)
                    loss_rec_q, loss_rec_s, loss_recon = episode_chamfer_losses(syn, ref, n_q, self.query_factor,
                                                                                self.support_factor)
                    return {"ttl_loss": loss_recon, "recon_loss": loss_recon, "query_rec_loss": loss_rec_q,
                            "support_rec_loss": loss_rec_s}
                cd = self.pc_metric(syn, ref)
                loss_rec_q, loss_rec_s = cd[:n_q].sum(), cd[n_q:].sum()
            else:
                loss_rec_q = self.pc_metric(syn_q, ref_q).sum()
                loss_rec_s = self.pc_metric(syn_s, ref_s).sum()
        else:
            syn_q = self._decode_queries(img_zq, pc_z_proto, pack)
            loss_rec_q = self.pc_metric(syn_q, ref_q).sum()
            loss_rec_s = torch.zeros(1, dtype=loss_rec_q.dtype, device=loss_rec_q.device)
        loss_recon = self.query_factor * loss_rec_q + self.support_factor * loss_rec_s
        return {"ttl_loss": loss_recon, "recon_loss": loss_recon, "query_rec_loss": loss_rec_q,
                "support_rec_loss": loss_rec_s}

    # ---------------------------------------------------------------------- evaluation
    def _reconstruct_for_eval(self, sample):
        """The part of ``_return_reconstruction`` in front of the EMD: ``(syn_pc, ref_pc_q, cd_loss, diameter)`` with no host
        read in it (``engine.EvalItem`` replays it as a hipGraph).  ``diameter`` (0-dim device tensor) is what the Sinkhorn
        form's annealing schedule starts from -- ``metrics.sinkhorn_divergence`` would compute the same value itself."""
        if self._eval_prune():
            img_zq, pc_z_proto = self._encode_for_eval(sample["xq"], sample["pcs"])
        else:
            _, img_zq, pc_z_proto, _ = self._encode(sample["xs"], sample["xq"], sample["xad"], sample["pcs"], sample["pcad"])
        syn_pc = self._decode_queries(img_zq, pc_z_proto)
        ref_pc_q = sample["pcq"].squeeze(0).contiguous()
        loss_rec_q = self.pc_metric(syn_pc, ref_pc_q).sum()
        pts = torch.cat([syn_pc.detach().reshape(-1, 3), ref_pc_q.reshape(-1, 3)])
        diameter = (pts.amax(0) - pts.amin(0)).norm()
        return syn_pc, ref_pc_q, self.query_factor * loss_rec_q, diameter

    def _return_reconstruction(self, sample):
        if self._eval_prune():
            img_zq, pc_z_proto = self._encode_for_eval(sample["xq"], sample["pcs"])
        else:
            _, img_zq, pc_z_proto, _ = self._encode(sample["xs"], sample["xq"], sample["xad"], sample["pcs"], sample["pcad"])
        syn_pc = self._decode_queries(img_zq, pc_z_proto)
        ref_pc_q = sample["pcq"].squeeze(0).contiguous()
        loss_rec_q = self.pc_metric(syn_pc, ref_pc_q).sum()
        emd_loss = self.emd_metric(syn_pc, ref_pc_q).sum()
        return {"cd_loss": self.query_factor * loss_rec_q, "emd_loss": emd_loss}

    @torch.no_grad()
    def reconstruct(self, sample):
        """Generated query clouds ``[Q, N, 3]`` from the support clouds and query images only
        (the forward used by ``draw_reconstruction``, reference :186-202)."""
        xs, xq, pcs = sample["xs"], sample["xq"], sample["pcs"]
        img_z = self.img_encoder(xq.reshape(-1, *xq.shape[2:]))
        pc_z = self.pc_encoder(pcs.reshape(-1, *pcs.shape[2:]).transpose(2, 1))
        return self._decode_queries(img_z, pc_z)

    def draw_reconstruction(self, sample, img_path):
        """Writes ``<dir>/<stem>.png`` (generated vs ground truth, one column per query) and
        the clouds as ``<stem>_<code>.npy`` / ``<stem>_<code>_gt.npy``.

        ``img_path`` is ``[stem, directory]`` as ``evaluate_Network.py:111`` passes it; a plain
        path string (what ``trainNetwork.py:183,205`` passes, which makes the reference write
        garbage names -- SURVEY.md F10) is split into directory and stem instead."""
        from .visualization import visualize_point_clouds, write_png
        if isinstance(img_path, (str, os.PathLike)):
            directory, stem = os.path.split(os.fspath(img_path))
            stem = os.path.splitext(stem)[0]
        else:
            stem, directory = img_path[0], img_path[1]
        code = sample["tmp"]
        code = int(code.reshape(-1)[0]) if torch.is_tensor(code) else int(np.ravel(code)[0])
        syn_pc = self.reconstruct(sample)
        pcq = sample["pcq"].squeeze(0)
        panels = [visualize_point_clouds(g, pcq[i], i) for i, g in enumerate(syn_pc)]
        write_png(os.path.join(directory, f"{stem}.png"),
                  np.concatenate(panels, axis=1).transpose(1, 2, 0))
        np.save(os.path.join(directory, f"{stem}_{code}.npy"), syn_pc.squeeze(0).cpu().numpy())
        np.save(os.path.join(directory, f"{stem}_{code}_gt.npy"), pcq.squeeze(0).cpu().numpy())
