"""The four update rules of the oracle (oracle.opt_apply, restated from the TF 1.12 optimizers the reference constructs at
autoencoder/autoencoder.py:452-472) against an independent implementation of the same published algorithms: torch.optim.

  GradientDescentOptimizer(lr)            == torch.optim.SGD(lr)
  MomentumOptimizer(lr, m) (no Nesterov)  == torch.optim.SGD(lr, momentum=m, dampening=0): acc = m*acc + g; p -= lr*acc
  AdagradOptimizer(lr), accumulator 0.1   == torch.optim.Adagrad(lr, initial_accumulator_value=0.1, eps=0): p -= lr*g/sqrt(acc)
  AdamOptimizer(lr), beta .9/.999, eps 1e-8, TF's "epsilon hat" form  p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
        == torch.optim.Adam whose eps is rescaled every step to eps / sqrt(1 - b2^t)  (torch adds its eps AFTER dividing sqrt(v) by
           sqrt(1 - b2^t); multiplying numerator and denominator through gives TF's form with eps_hat = eps_torch * sqrt(1 - b2^t)).
"""
import numpy as np
import pytest
import torch

import oracle as O


def _run_oracle(opt, p0, grads, lr, mom):
    p = [p0.copy()]
    st = O.OptState(opt, [p0.shape], np.float64)
    for g in grads:
        O.opt_apply(st, p, [g], lr, mom, np.float64)
    return p[0]


def _run_torch(opt, p0, grads, lr, mom):
    p = torch.tensor(p0, dtype=torch.float64, requires_grad=True)
    if opt == "gradient_descent":
        o = torch.optim.SGD([p], lr=lr)
    elif opt == "momentum":
        o = torch.optim.SGD([p], lr=lr, momentum=mom, dampening=0.0, nesterov=False)
    elif opt == "ada_grad":
        o = torch.optim.Adagrad([p], lr=lr, initial_accumulator_value=0.1, eps=0.0, lr_decay=0.0)
    else:
        o = torch.optim.Adam([p], lr=lr, betas=(0.9, 0.999), eps=1e-8)
    for t, g in enumerate(grads, start=1):
        if opt == "adam":
            for grp in o.param_groups:
                grp["eps"] = 1e-8 / np.sqrt(1.0 - 0.999 ** t)
        p.grad = torch.tensor(g, dtype=torch.float64)
        o.step()
    return p.detach().numpy()


@pytest.mark.parametrize("opt", ["gradient_descent", "momentum", "ada_grad", "adam"])
def test_update_rules_match_torch_optim(opt):
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal((7, 5))
    grads = [rng.standard_normal((7, 5)) * s for s in (1.0, 0.1, 3.0, 0.01, 1.0, 0.5)]
    a = _run_oracle(opt, p0, grads, 0.05, 0.5)
    b = _run_torch(opt, p0, grads, 0.05, 0.5)
    assert np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(b)), (opt, np.max(np.abs(a - b)))
