"""style_big_gan_amd -- MI355X (gfx950) native hot path for the Style-Big-GAN custom-op layer.

Layout
  csrc/                 hand-written HIP kernels + the C ABI (include/sbg_hip.h) -> libsbg_hip.so
  _lib.py               ctypes binding (fails loudly when the library is missing; no CPU fallback)
  torch_utils/ops/      host-side mirror of the reference op API (bias_act, upfirdn2d, conv2d_resample,
                        conv2d_gradfix, fma) as autograd Functions over the HIP kernels
  train_parts/, biggan/ host-side mirror of the model / loss / trainer surface that calls the ops
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def install_reference_aliases():
    """Register this package's modules under the reference's import paths
    (``stylegan2ada.torch_utils.ops.*``, ``biggan.layers``, ``train_parts.*``, ``utils``) so code written
    against the reference imports the MI355X implementation unchanged.  See INTEGRATION.md."""
    from ._aliases import install
    install()
