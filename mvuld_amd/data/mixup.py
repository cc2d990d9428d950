"""Mixup / CutMix of the Swin fine-tune job (``mixup_fn(samples, targets)``: main.py:268-269; built by data/build.py:86-95 from
``config.AUG.{MIXUP, CUTMIX, CUTMIX_MINMAX, MIXUP_PROB, MIXUP_SWITCH_PROB, MIXUP_MODE}`` + ``MODEL.LABEL_SMOOTHING``).

The reference takes the class from timm (third party, absent from the reference tree; requirements pin timm 0.4.12): this is a
restatement of its published modes -- "batch": one (lam, box) per batch; "elem": one per sample; "pair": one per pair (b, B-1-b);
partner = the batch reversed in all three -- with the parameters drawn on
the host from numpy's global generator with the SAME calls in the SAME order as timm's ``_params_per_batch`` / ``cutmix_bbox_and_lam`` /
``rand_bbox`` (so a run seeded like the reference's, main.py: ``np.random.seed(seed)``, draws the same sequence), and the mixing itself on
the device (``mvuld_mixup_batch``: images and label-smoothed soft targets).  Parity with timm itself is unpinned (library absent); the
arithmetic is checked against a torch restatement of the same formulas in the tests.

Attribution: ``rand_bbox``, ``rand_bbox_minmax``, ``cutmix_bbox_and_lam`` and the per-batch parameter draw follow timm 0.4.12
``timm/data/mixup.py`` (Copyright 2020 Ross Wightman, Apache License 2.0, https://github.com/rwightman/pytorch-image-models) closely on
purpose -- the numpy draws have to come in timm's order for a seeded run to reproduce the reference's batches.
"""
import numpy as np
import torch

from ..hip import call, dt, ptr, require_gpu


def rand_bbox(img_shape, lam, margin=0.0):
    """timm.data.mixup.rand_bbox: box of area ratio (1 - lam), centre uniform over the image, clipped."""
    ratio = np.sqrt(1 - lam)
    img_h, img_w = img_shape[-2:]
    cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
    margin_y, margin_x = int(margin * cut_h), int(margin * cut_w)
    cy = np.random.randint(0 + margin_y, img_h - margin_y)
    cx = np.random.randint(0 + margin_x, img_w - margin_x)
    yl = np.clip(cy - cut_h // 2, 0, img_h)
    yh = np.clip(cy + cut_h // 2, 0, img_h)
    xl = np.clip(cx - cut_w // 2, 0, img_w)
    xh = np.clip(cx + cut_w // 2, 0, img_w)
    return int(yl), int(yh), int(xl), int(xh)


def rand_bbox_minmax(img_shape, minmax):
    """timm.data.mixup.rand_bbox_minmax: box sides uniform in [minmax[0], minmax[1]] x image side."""
    assert len(minmax) == 2
    img_h, img_w = img_shape[-2:]
    cut_h = np.random.randint(int(img_h * minmax[0]), int(img_h * minmax[1]))
    cut_w = np.random.randint(int(img_w * minmax[0]), int(img_w * minmax[1]))
    yl = np.random.randint(0, img_h - cut_h)
    xl = np.random.randint(0, img_w - cut_w)
    return int(yl), int(yl + cut_h), int(xl), int(xl + cut_w)


class Mixup:
    def __init__(self, mixup_alpha=1.0, cutmix_alpha=0.0, cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode="batch", correct_lam=True,
                 label_smoothing=0.1, num_classes=1000):
        if mode not in ("batch", "pair", "elem"):
            raise ValueError(f"Mixup mode {mode!r}: batch | pair | elem")
        self.mode = mode
        self.mixup_alpha, self.cutmix_alpha, self.cutmix_minmax = mixup_alpha, cutmix_alpha, cutmix_minmax
        if self.cutmix_minmax is not None:
            assert len(self.cutmix_minmax) == 2
            self.cutmix_alpha = 1.0              # force cutmix alpha == 1.0 when minmax active to keep logic simple & safe
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.label_smoothing, self.num_classes = label_smoothing, num_classes
        self.correct_lam = correct_lam           # correct lambda based on clipped area for cutmix
        self.mixup_enabled = True

    def _params_per_batch(self):
        lam, use_cutmix = 1.0, False
        if self.mixup_enabled and np.random.rand() < self.mix_prob:
            if self.mixup_alpha > 0.0 and self.cutmix_alpha > 0.0:
                use_cutmix = np.random.rand() < self.switch_prob
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha) if use_cutmix else np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.mixup_alpha > 0.0:
                lam_mix = np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.cutmix_alpha > 0.0:
                use_cutmix = True
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha)
            else:
                assert False, "One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true."
            lam = float(lam_mix)
        return lam, use_cutmix

    def draw(self, shape):
        """-> (lam, use_cutmix, (yl, yh, xl, xh)): the batch's parameters (host randomness only)."""
        lam, use_cutmix = self._params_per_batch()
        box = (0, 0, 0, 0)
        if lam == 1.0:
            return 1.0, False, box
        if use_cutmix:
            box = rand_bbox_minmax(shape, self.cutmix_minmax) if self.cutmix_minmax is not None else rand_bbox(shape, lam)
            if self.correct_lam or self.cutmix_minmax is not None:
                lam = 1.0 - (box[1] - box[0]) * (box[3] - box[2]) / float(shape[-2] * shape[-1])
        return lam, use_cutmix, box

    def _params_per_elem(self, batch_size):
        """timm's per-element draw: vectors of lam / use_cutmix, numpy calls in timm's order."""
        lam = np.ones(batch_size, dtype=np.float32)
        use_cutmix = np.zeros(batch_size, dtype=bool)
        if self.mixup_enabled:
            if self.mixup_alpha > 0.0 and self.cutmix_alpha > 0.0:
                use_cutmix = np.random.rand(batch_size) < self.switch_prob
                lam_mix = np.where(use_cutmix, np.random.beta(self.cutmix_alpha, self.cutmix_alpha, size=batch_size),
                                   np.random.beta(self.mixup_alpha, self.mixup_alpha, size=batch_size))
            elif self.mixup_alpha > 0.0:
                lam_mix = np.random.beta(self.mixup_alpha, self.mixup_alpha, size=batch_size)
            elif self.cutmix_alpha > 0.0:
                use_cutmix = np.ones(batch_size, dtype=bool)
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha, size=batch_size)
            else:
                assert False, "One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true."
            lam = np.where(np.random.rand(batch_size) < self.mix_prob, lam_mix.astype(np.float32), lam)
        return lam, use_cutmix

    def draw_rows(self, shape):
        """-> params [B, 6] float32 {lam, cutmix, yl, yh, xl, xh} of the "elem" / "pair" modes (host randomness only; the boxes are drawn in
        sample order inside the loop, as timm's _mix_elem / _mix_pair draw them)."""
        B = shape[0]
        n = B if self.mode == "elem" else B // 2
        lam_batch, use_cutmix = self._params_per_elem(n)
        rows = np.zeros((B, 6), dtype=np.float32)
        rows[:, 0] = 1.0
        for i in range(n):
            lam = float(lam_batch[i])
            box, cut = (0, 0, 0, 0), False
            if lam != 1.0 and use_cutmix[i]:
                cut = True
                box = rand_bbox_minmax(shape, self.cutmix_minmax) if self.cutmix_minmax is not None else rand_bbox(shape, lam)
                if self.correct_lam or self.cutmix_minmax is not None:
                    lam = 1.0 - (box[1] - box[0]) * (box[3] - box[2]) / float(shape[-2] * shape[-1])
            rows[i] = (lam, float(cut), *box)
            if self.mode == "pair":
                rows[B - 1 - i] = rows[i]
        return rows

    def __call__(self, x, target):
        """x [B, C, H, W] (device), target [B] int64 (device) -> (mixed x, soft targets [B, num_classes] fp32)."""
        assert x.shape[0] % 2 == 0, "Batch size should be even when using this"
        require_gpu(x, target)
        B, C, H, W = x.shape
        if self.mode != "batch":
            rows = torch.from_numpy(self.draw_rows(x.shape)).to(x.device, non_blocking=True)
            x = x.contiguous()
            y = torch.empty_like(x)
            soft = torch.empty((B, self.num_classes), dtype=torch.float32, device=x.device)
            call("mixup_rows", ptr(x), ptr(y), ptr(target.contiguous()), ptr(soft), B, C, H, W, self.num_classes, ptr(rows),
                 float(self.label_smoothing), dt(x))
            return y, soft
        lam, use_cutmix, (yl, yh, xl, xh) = self.draw(x.shape)
        x = x.contiguous()
        y = torch.empty_like(x)
        soft = torch.empty((B, self.num_classes), dtype=torch.float32, device=x.device)
        call("mixup_batch", ptr(x), ptr(y), ptr(target.contiguous()), ptr(soft), B, C, H, W, self.num_classes, float(lam), int(use_cutmix), yl, yh,
             xl, xh, float(self.label_smoothing), dt(x))
        return y, soft


def build_mixup(config):
    """data/build.py:86-95."""
    active = config.AUG.MIXUP > 0 or config.AUG.CUTMIX > 0.0 or config.AUG.CUTMIX_MINMAX is not None
    if not active:
        return None
    return Mixup(mixup_alpha=config.AUG.MIXUP, cutmix_alpha=config.AUG.CUTMIX, cutmix_minmax=config.AUG.CUTMIX_MINMAX, prob=config.AUG.MIXUP_PROB,
                 switch_prob=config.AUG.MIXUP_SWITCH_PROB, mode=config.AUG.MIXUP_MODE, label_smoothing=config.MODEL.LABEL_SMOOTHING,
                 num_classes=config.MODEL.NUM_CLASSES)
