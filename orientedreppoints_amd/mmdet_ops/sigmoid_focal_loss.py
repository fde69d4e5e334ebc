"""Mirror of mmdet/ops/sigmoid_focal_loss/sigmoid_focal_loss.py:9-56 and its extension module
(sigmoid_focal_loss.cpp:14-47: forward / backward, CPU input -> error)."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


class _FocalExt(object):
    @staticmethod
    def forward(logits, targets, num_classes, gamma, alpha):
        if not logits.is_cuda:
            raise RuntimeError("SigmoidFocalLoss is not implemented on the CPU")
        assert logits.dim() == 2, "logits should be NxClass"
        x = logits.contiguous()
        t = targets.contiguous()
        if x.dtype not in (torch.float32, torch.float64):        # AT_DISPATCH_FLOATING_TYPES (sigmoid_focal_loss_cuda.cu:121)
            raise TypeError("sigmoid_focal_loss: float32 / float64 logits only")
        if t.dtype != torch.long:
            t = t.long()
        losses = torch.empty_like(x)
        fn = _lib.lib().orp_sigmoid_focal_loss_forward if x.dtype == torch.float32 else _lib.lib().orp_sigmoid_focal_loss_forward_f64
        with torch.cuda.device(x.device):
            rc = fn(_lib.ptr(x), _lib.ptr(t), x.size(0), x.size(1), float(gamma), float(alpha), _lib.ptr(losses),
                    _lib.stream_of(x))
        _lib.check(rc, "orp_sigmoid_focal_loss_forward")
        return losses

    @staticmethod
    def backward(logits, targets, d_losses, num_classes, gamma, alpha):
        if not logits.is_cuda:
            raise RuntimeError("SigmoidFocalLoss is not implemented on the CPU")
        assert logits.size(1) == num_classes, "logits.size(1) should be num_classes"
        x = logits.contiguous()
        t = targets.contiguous()
        if t.dtype != torch.long:
            t = t.long()
        if x.dtype not in (torch.float32, torch.float64):
            raise TypeError("sigmoid_focal_loss: float32 / float64 logits only")
        g = d_losses.to(x.dtype).contiguous()
        d_logits = torch.zeros_like(x)
        fn = _lib.lib().orp_sigmoid_focal_loss_backward if x.dtype == torch.float32 else _lib.lib().orp_sigmoid_focal_loss_backward_f64
        with torch.cuda.device(x.device):
            rc = fn(_lib.ptr(x), _lib.ptr(t), _lib.ptr(g), x.size(0), x.size(1), float(gamma), float(alpha),
                    _lib.ptr(d_logits), _lib.stream_of(x))
        _lib.check(rc, "orp_sigmoid_focal_loss_backward")
        return d_logits


sigmoid_focal_loss_cuda = _FocalExt()


class SigmoidFocalLossFunction(Function):

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)   # under autocast: fp32 inputs, autocast off inside
    def forward(ctx, input, target, gamma=2.0, alpha=0.25):
        ctx.save_for_backward(input, target)
        num_classes = input.shape[1]
        ctx.num_classes = num_classes
        ctx.gamma = gamma
        ctx.alpha = alpha
        loss = sigmoid_focal_loss_cuda.forward(input, target, num_classes, gamma, alpha)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loss):
        input, target = ctx.saved_tensors
        d_loss = d_loss.contiguous()
        d_input = sigmoid_focal_loss_cuda.backward(input, target, d_loss, ctx.num_classes, ctx.gamma, ctx.alpha)
        return d_input, None, None, None, None


sigmoid_focal_loss = SigmoidFocalLossFunction.apply


class SigmoidFocalLoss(nn.Module):

    def __init__(self, gamma, alpha):
        super(SigmoidFocalLoss, self).__init__()
        self.gamma = gamma
        self.alpha = alpha

    def forward(self, logits, targets):
        assert logits.is_cuda
        loss = sigmoid_focal_loss(logits, targets, self.gamma, self.alpha)
        return loss.sum()

    def __repr__(self):
        tmpstr = self.__class__.__name__ + '(gamma={}, alpha={})'.format(self.gamma, self.alpha)
        return tmpstr
