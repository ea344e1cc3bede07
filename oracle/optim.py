"""Optimisers of the reference path, restated for the float64 oracle.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

  * ``adabelief``  - optax.adabelief as driven by STARRED's Optimizer(method='adabelief')
                     (reference call sites: star_photometry.py:113-122, roi_modelling.py:326-334).
  * ``lbfgsb``     - scipy L-BFGS-B on the same loss, as Optimizer(method='l-bfgs-b') does
                     (roi_modelling.py:278-280, starred_utilities.py:33-34).
"""
import numpy as np
import torch
from scipy.optimize import minimize


def learning_rate(t, lr0, schedule, decay_rate=0.99, transition_steps=10):
    """optax.exponential_decay(init, transition_steps, decay_rate), continuous (no staircase)."""
    if not schedule:
        return lr0
    return lr0 * decay_rate ** (t / transition_steps)


def adabelief(loss_fn, params, free, lr0, n_iter, schedule=True, decay_rate=0.99, transition_steps=10,
              b1=0.9, b2=0.999, eps=1e-16, eps_root=1e-16):
    """Run exactly n_iter AdaBelief steps on the entries of ``params`` named in ``free``.

    Returns (params_final, loss_history, loss_initial) with loss_history[t] = loss(theta_{t+1}),
    i.e. the loss after the t-th update (len == n_iter, reference test_starred_calls.py:58).
    """
    p = {k: v.clone() for k, v in params.items()}
    m = {k: torch.zeros_like(p[k]) for k in free}
    s = {k: torch.zeros_like(p[k]) for k in free}
    losses = []
    for t in range(n_iter):
        for k in free:
            p[k] = p[k].detach().requires_grad_(True)
        L = loss_fn(p)
        grads = torch.autograd.grad(L, [p[k] for k in free])
        losses.append(float(L.detach()))
        lr = learning_rate(t, lr0, schedule, decay_rate, transition_steps)
        with torch.no_grad():
            for k, g in zip(free, grads):
                m[k] = b1 * m[k] + (1 - b1) * g
                s[k] = b2 * s[k] + (1 - b2) * (g - m[k]) ** 2 + eps_root
                mh = m[k] / (1 - b1 ** (t + 1))
                sh = s[k] / (1 - b2 ** (t + 1))
                p[k] = p[k].detach() - lr * mh / (torch.sqrt(sh) + eps)
    with torch.no_grad():
        p = {k: v.detach() for k, v in p.items()}
        losses.append(float(loss_fn(p)))
    return p, losses[1:], losses[0]


def value_and_grad(loss_fn, params, free):
    p = {k: v.clone() for k, v in params.items()}
    for k in free:
        p[k] = p[k].detach().requires_grad_(True)
    L = loss_fn(p)
    grads = torch.autograd.grad(L, [p[k] for k in free])
    return float(L.detach()), {k: g for k, g in zip(free, grads)}


def lbfgsb(loss_fn, params, free, maxiter, bounds=None):
    """scipy L-BFGS-B over the flattened free entries.  bounds: dict name -> (lo, hi) arrays."""
    shapes = [tuple(params[k].shape) for k in free]
    sizes = [int(np.prod(sh)) for sh in shapes]

    def unflat(x):
        p = {k: v.clone() for k, v in params.items()}
        o = 0
        for k, sh, sz in zip(free, shapes, sizes):
            p[k] = torch.as_tensor(x[o:o + sz].reshape(sh), dtype=params[k].dtype)
            o += sz
        return p

    def fun(x):
        L, g = value_and_grad(loss_fn, unflat(x), free)
        return L, np.concatenate([g[k].numpy().ravel() for k in free]).astype(np.float64)

    x0 = np.concatenate([params[k].numpy().ravel() for k in free]).astype(np.float64)
    bnds = None
    if bounds is not None:
        lo = np.concatenate([np.broadcast_to(bounds[k][0], sh).ravel() for k, sh in zip(free, shapes)])
        hi = np.concatenate([np.broadcast_to(bounds[k][1], sh).ravel() for k, sh in zip(free, shapes)])
        bnds = list(zip(lo, hi))
    hist = []
    res = minimize(fun, x0, jac=True, method='L-BFGS-B', bounds=bnds,
                   options={'maxiter': maxiter}, callback=lambda xk: hist.append(fun(xk)[0]))
    return unflat(res.x), hist, res
