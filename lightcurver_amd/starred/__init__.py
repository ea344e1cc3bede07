"""Mirror of the STARRED module tree for the entry points lightcurver imports.

``from starred.procedures.psf_routines import build_psf`` (reference
lightcurver/processes/psf_modelling.py:7) becomes
``from lightcurver_amd.starred.procedures.psf_routines import build_psf`` and likewise for
``deconvolution.deconvolution.setup_model``, ``deconvolution.loss.Loss``,
``deconvolution.parameters.ParametersDeconv``, ``optim.optimization.Optimizer``,
``optim.inference_base.FisherCovariance``, ``utils.noise_utils.propagate_noise`` and
``psf.psf.apply_distortion`` (star_photometry.py:7-12, roi_modelling.py:19-23,
starred_utilities.py:4-7).  Every compute call lands in liblcmi.so (HIP, gfx950).
"""
