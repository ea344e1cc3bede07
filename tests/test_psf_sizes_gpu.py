"""Stamp sizes without a kernel of their own: ``build_psf`` fits them embedded in the next instantiated size
(lightcurver_amd/starred/procedures/psf_routines.py, ``_fit_size``).  The reference's ``stamp_size_stars`` is a free integer
(lightcurver/pipeline/example_config_file/config.yaml:205).

Two statements: (i) the embedded call is exactly the native kernel on the explicitly padded problem (bit for bit), and
(ii) against the ORACLE's fit at the caller's own size (its own stage A by scipy L-BFGS-B, its own noise propagation, the
same AdaBelief iterations) the products the pipeline keeps - narrow PSF, reduced chi2, star positions - agree at the
per-cent level.  They cannot agree better: the embedded model is a slightly different model at the stamp's edge (Moffat
wings normalised over a larger frame, coarse starlet scales that see the ring) - the oracle fitted at n = 20 and the oracle
fitted on the padded 24 x 24 problem differ by 1.8 % of the PSF peak at a corner pixel, 2.5e-3 px in position and 0.7 % in
reduced chi2; a natively instantiated size (n = 24 below) sits at 0.9 % / 7e-4 px / 0.2 % against the
same oracle procedure.  An embedded fit is a working fall-back for sizes without a kernel, not the native-size model.
PARITY UNPINNED (oracle/__init__.py)."""
import math

import numpy as np
import pytest

from lightcurver_amd import _lib
from lightcurver_amd.starred.procedures.psf_routines import _fit_size, build_psf_batch
from lightcurver_amd.synthetic import make_psf_dataset
from oracle import model as om, optim as oo

pytestmark = pytest.mark.gpu
N_ANALYTIC, N_PIXELS = 60, 200


def _oracle_fit(ds, f, ss, fwhm_guess):
    """The same procedure as build_psf_batch, by the oracle at the stamp's own size."""
    data, sig2 = om.T(ds['data'][f]), om.T(ds['noisemap'][f]) ** 2
    mask = om.T(ds['masks'][f].astype(np.float64))
    S, n, _ = ds['data'][f].shape
    N = n * ss
    norm = float((data * mask).max())
    data, sig2 = data / norm, sig2 / norm ** 2
    f0 = math.sqrt(max(fwhm_guess ** 2 - (2.0 / ss) ** 2, (1.0 / ss) ** 2))
    p = dict(fwhm_x=f0, fwhm_y=f0, phi=0.0, beta=2.5, B=np.zeros(N * N), a=np.clip((data * mask).sum((-1, -2)).numpy(), 1e-6, None),
             x0=np.zeros(S), y0=np.zeros(S), sky=np.zeros(S))
    p = {k: om.T(v) for k, v in p.items()}
    bounds = dict(fwhm_x=(0.5 / ss, n / 2), fwhm_y=(0.5 / ss, n / 2), phi=(-math.pi, math.pi), beta=(1.1, 50.),
                  a=(0, np.inf), x0=(-n / 4, n / 4), y0=(-n / 4, n / 4))
    fn = lambda q: om.psf_loss(q, data, sig2, mask, ss)
    p, _, _ = oo.lbfgsb(fn, p, ['fwhm_x', 'fwhm_y', 'phi', 'beta', 'a', 'x0', 'y0'], N_ANALYTIC, bounds)
    W = om.propagate_noise_psf(p, sig2, mask, ss)
    fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, W=W, lam_scales=1.0, lam_hf=1.0)
    p, _, _ = oo.adabelief(fn, p, ['B', 'a', 'x0', 'y0'], 1e-4, N_PIXELS, schedule=True)
    narrow = om.psf_outputs(p, ss, n)[0].numpy()
    chi2 = om.reduced_chi2(data, om.psf_model(p, ss, n), sig2, mask)
    return narrow, float(chi2), p['x0'].numpy(), p['y0'].numpy()


def _compare(n, seed):
    ss, F, S = 2, 1, 4
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=seed)
    masks = [ds['masks'][f] for f in range(F)]
    res = build_psf_batch([ds['data'][f] for f in range(F)], [ds['noisemap'][f] for f in range(F)], ss, masks=masks,
                          n_iter_analytic=N_ANALYTIC, n_iter_adabelief=N_PIXELS, guess_method_star_position='center',
                          guess_fwhm_pixels=ds['fwhm_guess'])
    worst = dict(psf=0.0, l2=0.0, chi2=0.0, pos=0.0)
    for f in range(F):
        r = res[f]
        N = n * ss
        assert r['narrow_psf'].shape == (N, N) and r['full_psf'].shape == (N, N) and r['residuals'].shape == (S, n, n)
        assert abs(r['narrow_psf'].sum() - 1.0) < 1e-5 and abs(r['full_psf'].sum() - 1.0) < 1e-5
        assert r['kwargs_psf']['kwargs_background']['background'].shape == (N * N,)
        assert len(r['adabelief_extra_fields']['loss_history']) == N_PIXELS
        narrow, chi2, x0, y0 = _oracle_fit(ds, f, ss, float(ds['fwhm_guess'][f]))
        worst['psf'] = max(worst['psf'], float(np.abs(r['narrow_psf'] - narrow).max() / narrow.max()))
        worst['l2'] = max(worst['l2'], float(np.linalg.norm(r['narrow_psf'] - narrow) / np.linalg.norm(narrow)))
        worst['chi2'] = max(worst['chi2'], abs(r['chi2'] - chi2) / chi2)
        worst['pos'] = max(worst['pos'], float(np.abs(r['kwargs_psf']['kwargs_gaussian']['x0'] - x0).max()),
                           float(np.abs(r['kwargs_psf']['kwargs_gaussian']['y0'] - y0).max()))
    return worst


def test_embedded_sizes_against_the_oracle_at_the_callers_size(ctx):
    assert _fit_size(24, 2) == 24 and _fit_size(20, 2) == 24 and _fit_size(28, 2) == 32 and _fit_size(40, 2) == 64
    native = _compare(24, 11)
    print('native 24:', native)
    assert native['psf'] < 2e-2 and native['chi2'] < 5e-3 and native['pos'] < 2e-3
    for n in (20, 28):     # (40 -> 64 measured once: 0.7 % / 1.3 % / 1e-3 / 1e-3 px; the oracle takes minutes there)
        w = _compare(n, 11 + n)
        print(f'embedded {n} -> {_fit_size(n, 2)}:', w)
        assert w['psf'] < 8e-2 and w['l2'] < 3e-2, (n, w)      # worst pixel (at the edge) / whole PSF in the l2 norm
        assert w['chi2'] < 2e-2 and w['pos'] < 1e-2, (n, w)


def test_embedded_call_is_the_native_kernel_on_the_padded_problem(ctx):
    ss, F, S, n = 2, 3, 5, 20
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=5)
    kw = dict(n_iter_analytic=30, n_iter_adabelief=100, guess_method_star_position='center', guess_fwhm_pixels=ds['fwhm_guess'])
    res = build_psf_batch(list(ds['data']), list(ds['noisemap']), ss, masks=list(ds['masks']), **kw)
    m, pad = 24, 2
    big = lambda a, fill: np.pad(np.asarray(a, np.float64), ((0, 0), (0, 0), (pad, pad), (pad, pad)), constant_values=fill)
    ref = build_psf_batch(list(big(ds['data'], 0.0)), list(big(ds['noisemap'], 1.0)), ss, masks=list(big(ds['masks'], 0.0)), **kw)
    P, N = pad * ss, n * ss
    for f in range(F):
        cut = ref[f]['narrow_psf'][P:P + N, P:P + N]
        assert np.array_equal(res[f]['narrow_psf'], cut / cut.sum())
        assert np.array_equal(res[f]['residuals'], ref[f]['residuals'][:, pad:pad + n, pad:pad + n])
        assert res[f]['chi2'] == ref[f]['chi2']
        assert np.array_equal(res[f]['kwargs_psf']['kwargs_gaussian']['x0'], ref[f]['kwargs_psf']['kwargs_gaussian']['x0'])
        assert res[f]['adabelief_extra_fields']['loss_history'] == ref[f]['adabelief_extra_fields']['loss_history']


def test_sizes_beyond_the_largest_kernel_are_refused(ctx):
    with pytest.raises(_lib.LcError):
        _fit_size(130, 2)
    with pytest.raises(_lib.LcError):
        _fit_size(20, 3)
