"""Scalar GAN losses (host code).  Registry names and formulas of the reference's ``train_parts/losses.py:9-56``:
``calc_loss(pred_real, pred_fake=None)`` returns the discriminator loss when both logits are given and the generator loss
(on ``pred_real`` = logits of generated images) otherwise."""
import torch
import torch.nn.functional as F

from .. import utils

losses = utils.ClassRegistry()


@losses.add_to_registry("bcew")
class BCEWithLogits(torch.nn.Module):
    def calc_loss(self, pred_real, pred_fake=None):
        real_term = F.binary_cross_entropy_with_logits(pred_real, torch.ones_like(pred_real))
        if pred_fake is None:
            return real_term
        return real_term + F.binary_cross_entropy_with_logits(pred_fake, torch.zeros_like(pred_fake))


@losses.add_to_registry("hinge")
class Hinge(torch.nn.Module):
    def calc_loss(self, pred_real, pred_fake=None):
        if pred_fake is None:
            return -pred_real.mean()
        return F.relu(1 - pred_real).mean() + F.relu(1 + pred_fake).mean()


@losses.add_to_registry("wasserstein")
class Wasserstein(torch.nn.Module):
    def calc_loss(self, pred_real, pred_fake=None):
        if pred_fake is None:
            return -pred_real.mean()
        return pred_fake.mean() - pred_real.mean()


@losses.add_to_registry("softplus")
class Softplus(torch.nn.Module):
    def calc_loss(self, pred_real, pred_fake=None):
        if pred_fake is None:
            return F.softplus(-pred_real).mean()
        return F.softplus(-pred_real).mean() + F.softplus(pred_fake).mean()
