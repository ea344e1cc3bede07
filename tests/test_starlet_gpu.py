"""The starlet l1 regulariser alone (value + sub-gradient through the exact edge-replicating adjoint),
isolated from any chi2 term: zero-weight stamps make the PSF-fit evaluation return exactly
lam_hf * sum W_0 |w_0| + lam * sum_j W_j |w_j| of the pixel grid and its gradient.  Smooth images put
energy in the coarse scales and at the image borders, which is where the adjoint is delicate."""
import numpy as np
import pytest
import torch

from oracle import model as om, optim as oo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,ss', [(16, 1), (16, 2), (24, 2), (32, 2), (64, 2)])
@pytest.mark.parametrize('kind', ['smooth', 'noise'])
def test_starlet_value_and_gradient(ctx, n, ss, kind):
    from lightcurver_amd.psf_batch import PsfBatch
    N = n * ss
    J = om.n_scales(N)
    rng = np.random.default_rng(N + len(kind))
    u = np.arange(N)
    if kind == 'smooth':
        img = 5 * np.exp(-0.5 * (((u[None] - 0.3 * N) / (0.2 * N)) ** 2 + ((u[:, None] - 0.6 * N) / (0.15 * N)) ** 2))
        img = img + 0.3 + 0.01 * rng.standard_normal((N, N))
    else:
        img = rng.standard_normal((N, N))
    W = rng.uniform(0.5, 2.0, (J + 1, N, N))
    b = PsfBatch(np.zeros((1, 1, n, n), np.float32), np.zeros((1, 1, n, n), np.float32), ss, ctx)
    b.set_moffat(np.array([[3.0, 3.0, 0.0, 2.5]]))
    b.set_stars(np.zeros((1, 1, 4)))
    b.set_grid(img[None])
    b.set_regularization(W[None, :J], 1.5, 0.8)
    out = b.evaluate()
    h = om.T(img).requires_grad_(True)
    L = om.l1_starlet(h, om.T(W), 1.5, 0.8, J)
    (g,) = torch.autograd.grad(L, h)
    assert abs(out['loss'][0] - float(L)) / float(L) < 1e-5
    err = np.abs(out['grad_grid'][0] - g.numpy()).max() / np.abs(g.numpy()).max()
    assert err < 2e-5, err
