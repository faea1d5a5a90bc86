"""Gradient regularisers (host code): path length (generator), R1 and WGAN-GP (discriminator).

Registry names (``generator_regs['ppl']``, ``discriminator_regs['r1' | 'grad_pen']``), constructor arguments, ``calc_reg``
call signatures, reported statistics and profiler scope names are those of the reference's
``train_parts/regularizations.py`` (``PPLreg`` :12-37, ``R1reg`` :41-56, ``Grad_pen`` :60-85) -- that is the interface the
loss orchestration and the configs bind to.  The arithmetic of each penalty is restated in the method that cites its lines.

All three are the same computation with different plugs, and are written that way here: *probe* a network for an
(output, input) pair, take the gradient of the summed output with respect to that input **with the graph kept**
(``InputGradientPenalty.input_gradient``), reduce the gradient to one number per sample (``penalty``), and back-propagate
``mean(penalty) * gain`` (``backprop``).  The second differentiation is what makes arbitrary-order autograd a requirement of
every op on the hot path.

Data-parallel note: ``calc_reg(..., sync=...)`` is forwarded to the forward passes as in the reference; with
``parallel.GradReducer`` the exchange itself is completed by ``finish()`` at the end of the phase whatever the flags were.
"""
import math

import torch

from .. import utils
from ..torch_utils import training_stats
from ..torch_utils.ops import conv2d_gradfix

generator_regs = utils.ClassRegistry()
discriminator_regs = utils.ClassRegistry()

_scope = torch.autograd.profiler.record_function


class InputGradientPenalty:
    """Skeleton shared by the regularisers.  Subclasses set the profiler scope names and provide ``penalty``."""
    forward_scope = backward_scope = gradient_scope = None

    def input_gradient(self, output, wrt, grad_output=None):
        """d<output, grad_output> / d wrt as a differentiable tensor (``grad_output`` defaults to ones, i.e. the sum of a scalar-per-sample
        output).  Parameter gradients are suppressed for this inner pass: only the outer backward may write to ``.grad``."""
        if grad_output is None:
            output = output.sum()
        with _scope(self.gradient_scope), conv2d_gradfix.no_weight_gradients():
            (g,) = torch.autograd.grad(outputs=[output], inputs=[wrt], grad_outputs=None if grad_output is None else [grad_output],
                                       create_graph=True)
        return g

    def backprop(self, per_sample, gain, tie=None):
        """backward of mean(per_sample) * gain.  ``tie`` (a network output of the same pass) enters with weight zero: the value is
        unchanged, but the first-order graph of that output takes part in the backward as it does in the reference (:37, :56)."""
        with _scope(self.backward_scope):
            loss = per_sample if tie is None else tie * 0 + per_sample
            loss.mean().mul(gain).backward()


@generator_regs.add_to_registry("ppl")
class PPLreg(InputGradientPenalty):
    """Path-length regulariser: the image's response to a random direction, differentiated w.r.t. the per-layer latents
    ``ws``, should have the same length for every sample -- an exponential moving average of that length (reference :12-37)."""
    forward_scope, gradient_scope, backward_scope = 'Gpl_forward', 'pl_grads', 'Gpl_backward'

    def __init__(self, pl_batch_shrink=2., pl_decay=0.01, pl_weight=2.):
        self.pl_batch_shrink = pl_batch_shrink
        self.pl_decay = pl_decay
        self.pl_weight = pl_weight
        self.pl_mean = torch.zeros([])      # running mean of the path length; lives on the model's device after the first call

    def penalty(self, pl_grads):
        """:29-33.  The moving average is advanced *before* the penalty is taken and stays inside the graph, so the penalty's gradient
        also flows through ``mean(lengths) * pl_decay``."""
        lengths = pl_grads.square().sum(2).mean(1).sqrt()
        target = torch.lerp(self.pl_mean, lengths.mean(), self.pl_decay)
        self.pl_mean.copy_(target.detach())
        return (lengths - target).square()

    def calc_reg(self, model, real_img, real_c, gen_z, gen_c, sync, gain):
        if self.pl_weight == 0 or not hasattr(model.G, 'G_mapping'):    # needs the split generator (reference :20)
            return
        if self.pl_mean.device != torch.device(model.device):
            self.pl_mean = self.pl_mean.to(model.device)
        n = int(gen_z.shape[0] // self.pl_batch_shrink)
        with _scope(self.forward_scope):
            img, ws = model.run_Gws(gen_z[:n], gen_c[:n], sync=sync)
            direction = torch.randn_like(img) / math.sqrt(img.shape[2] * img.shape[3])
            pl_penalty = self.penalty(self.input_gradient(img, ws, grad_output=direction))
            training_stats.report('Loss/pl_penalty', pl_penalty)
            loss_Gpl = pl_penalty * self.pl_weight
            training_stats.report('Loss/G/reg', loss_Gpl)
        self.backprop(loss_Gpl, gain, tie=img[:, 0, 0, 0])


@discriminator_regs.add_to_registry("r1")
class R1reg(InputGradientPenalty):
    """R1: squared norm of d D(real) / d real (reference :41-56).  The discriminator's forward on the reals (with the reals requiring
    grad) is done by the caller and handed in as ``real_logits`` / ``real_img_tmp``."""
    forward_scope, gradient_scope, backward_scope = 'Dr1_forward', 'r1_grads', 'Dr1_backward'

    def __init__(self, r1_gamma=10.):
        self.r1_gamma = r1_gamma

    def penalty(self, r1_grads):
        return r1_grads.square().sum([1, 2, 3])

    def calc_reg(self, model, real_img, real_c, gen_z, gen_c, real_logits, real_img_tmp, sync, gain):
        if self.r1_gamma == 0:
            return
        with _scope(self.forward_scope):
            r1_penalty = self.penalty(self.input_gradient(real_logits, real_img_tmp))
            loss_Dr1 = r1_penalty * (self.r1_gamma / 2)
            training_stats.report('Loss/r1_penalty', r1_penalty)
            training_stats.report('Loss/D/r1reg', loss_Dr1)
        self.backprop(loss_Dr1, gain, tie=real_logits)


@discriminator_regs.add_to_registry("grad_pen")
class Grad_pen(InputGradientPenalty):
    """WGAN-GP: (|d D(x~) / d x~| - 1)^2 at random interpolates x~ of reals and fakes (reference :60-85).  Runs its own discriminator
    forward; unlike the two above, parameter gradients of the inner pass are NOT suppressed in the reference (:74-77), nor here."""
    forward_scope, gradient_scope, backward_scope = 'Dgrad_pen_forward', 'grad_pen_grads', 'Dgrad_pen_backward'

    def __init__(self, alpha=10.):
        self.alpha = alpha

    def penalty(self, grad):
        return self.alpha * (grad.flatten(1).norm(dim=1) - 1) ** 2

    def calc_reg(self, model, real_img, real_c, gen_z, gen_c, real_logits, real_img_tmp, sync, gain):
        with _scope(self.forward_scope):
            real = real_img.to(model.device)
            with torch.no_grad():
                fake = model.run_G(gen_z, gen_c, sync=False)
            t = torch.rand(real.shape[0], 1, 1, 1).to(real.device)       # drawn on the host, as the reference does (:69)
            mix = torch.lerp(fake, real, t).requires_grad_(True)
            logits = model.run_D(mix, gen_c, sync=sync)
            with _scope(self.gradient_scope):
                (grad,) = torch.autograd.grad(outputs=logits, inputs=mix, grad_outputs=torch.ones_like(logits), create_graph=True)
            loss_gp = self.penalty(grad)
            training_stats.report('Loss/D/grad_pen', loss_gp)
        self.backprop(loss_gp, gain)
