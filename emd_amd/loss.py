"""The image-loss tail of the training step on the HIP path.

`image_loss(...)` = S3Gaussian/train.py:226-363 with utils/loss_utils.py:21-98:
    l1_loss(image, gt) + lambda_depth * compute_depth("l2", depth * mask, gt_depth * mask)
    + lambda_dssim * (1 - ssim(image, gt)) + lambda_sky * sky BCE(weight)
computed, together with dL/dimage, dL/ddepth and dL/dweight, by four HIP launches (emd_image_loss) instead of ~100 torch
launches with a dozen image-sized temporaries.  No CPU path."""
import ctypes as C

import torch

from . import _lib as L


class _ImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, depth, gt_depth, mask, weight, sky_mask, lam_dssim, lam_depth, lam_sky, max_depth):
        if image.device.type != "cuda":
            raise L.EmdError("image_loss needs tensors on a ROCm device; there is no CPU path")
        lib = L.load()
        dev = image.device
        H, W = image.shape[-2:]
        f = lambda t: None if t is None else t.detach().contiguous().float()
        image_c, gt_c, depth_c, gtd_c, weight_c = f(image), f(gt), f(depth), f(gt_depth), f(weight)
        mask_c = None if mask is None else mask.detach().reshape(-1, H, W)[0].contiguous().float()
        sky_c = None if sky_mask is None else sky_mask.detach().reshape(H, W).contiguous().to(torch.uint8)
        a = L.EmdLossArgs()
        a.height, a.width = H, W
        a.image, a.gt = image_c.data_ptr(), gt_c.data_ptr()
        a.depth, a.gt_depth, a.mask = L.ptr(depth_c), L.ptr(gtd_c), L.ptr(mask_c)
        if sky_c is None or weight_c is None:      # the sky term needs both (train.py:360); without it dL/dweight is zero
            weight_c = sky_c = None
        a.weight, a.sky_mask = L.ptr(weight_c), L.ptr(sky_c)
        a.lambda_dssim, a.max_depth = lam_dssim, max_depth
        a.lambda_depth = lam_depth if depth is not None else 0.0
        a.lambda_sky = lam_sky if weight_c is not None else 0.0
        losses = torch.empty(5, device=dev, dtype=torch.float32)
        a.losses = losses.data_ptr()
        need = ctx.needs_input_grad
        g_img = torch.empty_like(image_c) if need[0] else None
        g_dep = torch.empty_like(depth_c) if (depth is not None and need[2]) else None
        g_wgt = (torch.empty_like(weight_c) if weight_c is not None else torch.zeros_like(f(weight))) if (weight is not None and need[5]) else None
        a.dL_dimage, a.dL_ddepth, a.dL_dweight = L.ptr(g_img), L.ptr(g_dep), L.ptr(g_wgt)
        nbytes = lib.emd_image_loss_workspace(H, W)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        L.check(lib.emd_image_loss(C.byref(a), ws.data_ptr(), nbytes, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                "emd_image_loss")
        ctx.save_for_backward(g_img, g_dep, g_wgt)
        ctx.shapes = (image.shape, None if depth is None else depth.shape, None if weight is None else weight.shape)
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), losses

    @staticmethod
    def backward(ctx, g, _g_terms):
        g_img, g_dep, g_wgt = ctx.saved_tensors
        s = ctx.shapes
        have = [t for t in (g_img, g_dep, g_wgt) if t is not None]
        # the three gradients were formed by the forward's launches; the upstream scalar scales them in ONE multi-tensor launch (three image-sized
        # element-wise launches before round 6)
        scaled = iter(torch._foreach_mul(have, g) if len(have) > 1 else [t * g for t in have])
        pick = lambda t, shape: None if t is None else next(scaled).view(shape)
        return (pick(g_img, s[0]), None, pick(g_dep, s[1]), None, None, pick(g_wgt, s[2]), None, None, None, None, None)


def image_loss(image, gt, depth=None, gt_depth=None, mask=None, weight=None, sky_mask=None, lambda_dssim=0.2, lambda_depth=0.5,
               lambda_sky=0.05, max_depth=80.0):
    """-> (loss, {"l1", "ssim", "depth", "sky"}).  image, gt [3,H,W]; depth, gt_depth, weight [1,H,W] or [H,W]; mask (float or
    bool, [1,H,W] / [3,H,W] / [H,W]) multiplies both depths as train.py does; sky_mask bool [1,H,W].  A term whose inputs are
    None is skipped, as in the reference."""
    total, terms = _ImageLoss.apply(image, gt, depth, gt_depth, mask, weight, sky_mask, float(lambda_dssim), float(lambda_depth),
                                    float(lambda_sky), float(max_depth))
    return total, {"l1": terms[1], "ssim": terms[2], "depth": terms[3], "sky": terms[4]}
