"""CPU oracle for the lightcurver PSF-fit / joint forward-model hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lightcurver_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

PARITY UNPINNED.  The arithmetic of this path lives in the third-party package
``starred-astro >= 1.4.7`` (reference ``pyproject.toml:26``), which is neither
vendored under /root/reference nor installable here, and the reference's own
tests (``tests/test_starred_calls/test_starred_calls.py``) pin structure only
(keys, shapes, ``len(loss_curve) == n_iter``) - no flux, position, PSF pixel or
loss value.  This oracle is therefore a float64 restatement of the published
STARRED algorithm (Michalewicz et al. 2023; Millon et al. 2024) as frozen in
``DESIGN.md`` section "SPEC", anchored on the reference's call sites, and
pinned only by library known-answer checks (scipy ``fftconvolve``/
``map_coordinates``, starlet exact reconstruction, finite differences).

Contents: ``model.py`` / ``optim.py`` (torch float64 + autograd: the oracle), ``prep.py`` (stamp pre-processing),
``psf_cpu.c``, ``joint_ps_cpu.c`` and ``joint_cpu.c`` (plain-C restatements of the PSF pixel-grid stage, of the
point-source-only joint fit in the direct separable form, and of the joint fit with the pixelated background - radix-2 FFT
convolution, hand-derived adjoints; their float64 builds are further, algorithmically independent checkers, their float32
builds the ``cpu_baseline`` "port" figures of ``bench.py``).
"""
