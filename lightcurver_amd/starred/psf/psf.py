"""``apply_distortion`` (reference call sites: lightcurver/processes/star_photometry.py:303-304,
roi_file_preparation.py:179-180).  Field distortion is not fitted by this build (build_psf returns an
empty kwargs_distortion), so the only distortion it can apply is the identity."""
import numpy as np


def apply_distortion(narrow_psf, kwargs_distortion, star_xy_coordinates):
    if kwargs_distortion:
        raise NotImplementedError('field distortion is not built (DESIGN.md, out of scope)')
    return np.asarray(narrow_psf)
