import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import model as om, optim as oo
from lightcurver_amd import _lib
from tests.test_joint_gpu import _setup
ctx = _lib.default_context()
E, M, n, ss, alpha = 3, 3, 24, 2, 0.5
for alpha in (0.0, 0.5):
  ds, j, po, data, sig2, psf = _setup(ctx, E, M, n, ss, 100 + n + M, alpha_sigma=alpha)
  N = n * ss
  W = om.propagate_noise_deconv(sig2, psf, ss)
  for name, kw, okw in [('chi2 only', {}, {}), ('starlet', dict(W=W.numpy(), lam_scales=1.5, lam_hf=0.8), dict(W=W, lam_scales=1.5, lam_hf=0.8)),
                        ('pos', dict(lam_positivity=20.0), dict(lam_pos=20.0))]:
    j.set_loss(**kw)
    j.set_free(['h'])
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, **okw)
    L, g = oo.value_and_grad(fn, po, ['h'])
    loss, grads = j.loss_grad(['h'])
    d = (grads['h'] - g['h'].numpy()).reshape(N, N)
    go = g['h'].numpy().reshape(N, N)
    bad = np.argwhere(np.abs(d) > 1e-4 * np.abs(go).max())
    print(alpha, name, 'loss rel', abs(loss - L) / abs(L), 'max err', np.abs(d).max() / np.abs(go).max(), 'n bad', len(bad), bad[:12].tolist())
print(np.round(d[:7, :7] / np.abs(go).max(), 4))
# same image through the PSF-fit kernel's starlet (shared device function)
from lightcurver_amd.psf_batch import PsfBatch
hh = po['h'].numpy().reshape(1, N, N)
b = PsfBatch(np.zeros((1, 1, n, n), np.float32), np.zeros((1, 1, n, n), np.float32), ss, ctx)
b.set_moffat(np.array([[3.0, 3.0, 0.0, 2.5]])); b.set_stars(np.zeros((1, 1, 4))); b.set_grid(hh)
b.set_regularization(W[:5].numpy()[None], 1.5, 0.8)
out = b.evaluate()
fn = lambda q: om.l1_starlet(q['h'].reshape(N, N), W, 1.5, 0.8, 5)
L, g = oo.value_and_grad(fn, po, ['h'])
d2 = out['grad_grid'][0] - g['h'].numpy().reshape(N, N)
print('psf-kernel starlet: loss rel', abs(out['loss'][0] - L) / abs(L), 'max err', np.abs(d2).max() / np.abs(g['h'].numpy()).max())
print(np.round(d2[:7, :7] / np.abs(g['h'].numpy()).max(), 4))
