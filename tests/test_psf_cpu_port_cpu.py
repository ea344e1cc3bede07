"""The C / OpenMP fp32 restatement of the PSF pixel-grid stage (oracle/psf_cpu.c: bench.py's cpu_baseline, and a
second checker of the HIP path) pinned to the float64 torch oracle: single evaluations (loss, chi2, model, every
gradient) to fp32 accuracy, and an AdaBelief trajectory.  CPU only."""
import numpy as np
import pytest

from oracle import model as om, optim as oo, psf_cpu
from lightcurver_amd.synthetic import make_psf_dataset
from tests import helpers as H


def _problem(n, ss, S, F, seed, jitter=0.2):
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=seed)
    rng = np.random.default_rng(seed + 1)
    plist = [H.psf_initial_params(ds, f, ss, rng, jitter) for f in range(F)]
    N = n * ss
    J = om.n_scales(N)
    Tm = np.stack([om.moffat(N, ss, p['fwhm_x'], p['fwhm_y'], p['phi'], p['beta']).numpy() for p in plist])
    Ws = []
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        Ws.append(om.propagate_noise_psf(plist[f], sig2, mask, ss))
    st = psf_cpu.PsfCpuState(ds['data'], H.weights_from(ds), ss, Tm, np.stack([w[:J].numpy() for w in Ws]),
                             np.stack([p['B'].numpy() for p in plist]), H.stars_array(plist))
    return ds, plist, Ws, st


@pytest.mark.parametrize('n,ss,S', [(16, 1, 3), (16, 2, 4), (32, 2, 8)])
def test_c_port_evaluation_matches_the_float64_oracle(n, ss, S):
    F = 2
    ds, plist, Ws, st = _problem(n, ss, S, F, 50 + n + ss)
    N = n * ss
    out = st.evaluate(1.3, 0.7, model=True)
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, W=Ws[f], lam_scales=1.3, lam_hf=0.7)
        L, g = oo.value_and_grad(fn, plist[f], ['a', 'x0', 'y0', 'B'])
        assert abs(out['loss'][f] - L) / abs(L) < 2e-5
        assert H.rel_err(out['model'][f], om.psf_model(plist[f], ss, n).numpy()) < 2e-5
        assert H.rel_err(out['grad_grid'][f], g['B'].numpy().reshape(N, N)) < 5e-5
        gs = np.stack([g['a'].numpy(), g['x0'].numpy(), g['y0'].numpy()], axis=-1)
        for q in range(3):
            assert H.rel_err(out['grad_stars'][f][:, q], gs[:, q]) < 5e-5, q


def test_c_port_trajectory_matches_the_float64_oracle():
    n, ss, S, F, T = 16, 2, 4, 2, 25
    ds, plist, Ws, st = _problem(n, ss, S, F, 5 + n, jitter=0.1)
    hist = st.run_adabelief(T, lr0=1e-4, schedule=True, threads=2)
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, W=Ws[f], lam_scales=1.0, lam_hf=1.0)
        pf, lh, l0 = oo.adabelief(fn, plist[f], ['B', 'a', 'x0', 'y0'], 1e-4, T, schedule=True)
        ref = np.array([l0] + lh)
        assert np.abs(hist[f] - ref).max() / np.abs(ref).max() < 1e-4
        assert H.rel_err(st.stars[f][:, 0], pf['a'].numpy()) < 1e-5
        assert np.median(np.abs(st.B[f] - pf['B'].numpy())) < 1e-7
