"""CPU oracle for the DAE training hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (NumPy, fp32 or fp64), the arithmetic that the
reference (louislung/DAE_RNN_News_Recommendation, 100 % Python on TensorFlow 1.12)
executes per mini-batch.  It exists so that every HIP kernel and the end-to-end
``fit()`` of ``dae_rnn_news_recommendation_amd`` can be checked against the
reference's algorithm.

Rules (enforced by tests/test_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
    leg may import anything from here -- as the checker, never as the product path;
  * the product package never imports ``oracle`` and never falls back to it.

Pinning status (see DESIGN.md "Oracle"):
  * miners / masks / weighted_loss / noise / batching / CSR feed: PINNED -- golden
    vectors in tests/golden/ were produced by importing the reference's own
    ``autoencoder/triplet_loss_utils.py`` and ``autoencoder/utils.py`` (executed over
    a NumPy stand-in for the ~30 TF ops they call; tests/golden/make_golden.py) and
    the loop bodies of the reference's own pytest files are restated in tests/.
  * encode / decode formulae, tied-weight gradients, optimizers, the ``fit`` loop:
    PARITY UNPINNED against TensorFlow itself -- TF 1.12 cannot be installed in this
    image (no wheel for py3.10, no network).  They are restated from the reference
    call sites and cross-checked against torch-CPU autograd and fp64 finite
    differences only.
"""
from .dae_oracle import *  # noqa: F401,F403
