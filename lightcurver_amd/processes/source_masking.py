"""Detection, de-blending and segmentation of the sources in one small stamp: what the reference gets from
``sep.extract(data, thresh=3., err=noisemap, minarea=15, segmentation_map=True, deblend_cont=0.001)`` inside
``mask_surrounding_stars`` (lightcurver/processes/psf_modelling.py:35-61).  ``sep`` (a C library, SExtractor's core) is
not a dependency of this package; this is a host-side NumPy / SciPy restatement of the parts that call exercises, with
sep's defaults for everything the call leaves unset:

  1. matched filter with the per-pixel noise (sep filter_type='matched', default 3 x 3 kernel [[1,2,1],[2,4,2],[1,2,1]]):
     S/N image  sum_i k_i d_i / s_i^2  /  sqrt(sum_i k_i^2 / s_i^2);  a pixel is detected above ``thresh``;
  2. 8-connected groups of at least ``minarea`` detected pixels;
  3. multi-threshold de-blending (SExtractor): ``deblend_nthresh`` = 32 levels spaced exponentially between the detection
     threshold and the peak of the group; descending the levels, an island of pixels above a level that carries at least
     ``deblend_cont`` of the group's total flux becomes a branch, and a group with two or more branches at some level is
     split there; the remaining pixels join the branch whose peak is closest in (distance / branch size) units.
  4. cleaning (sep ``clean=True``, ``clean_param=1.0``: SExtractor's CLEAN step): every detection is compared with its
     brighter neighbours; a neighbour is modelled as a Moffat profile of index ``clean_param`` with the neighbour's
     central amplitude, stretched to its second-moment ellipse and scaled so that it crosses the detection threshold
     at the neighbour's isophotal area.  A detection that would not have reached ``minarea`` pixels above the
     threshold without the wing of such a neighbour under it (wing at its position > excess of its ``minarea``-th
     brightest pixel over the threshold) is merged into the neighbour that contributes most.
Positions are the barycentres of the filtered image over the object's pixels (SExtractor's first moments; sep's ``x``,
``y``).  Not restated: the exact weighting of sep's pixel re-attribution after a split.  Unpinned against sep itself
(absent here); tests/test_host_logic_cpu.py checks the behaviour on synthetic blends."""
import numpy as np
from scipy import ndimage

DEFAULT_KERNEL = np.array([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]])
EIGHT = np.ones((3, 3), dtype=int)


def matched_filter_snr(data, noisemap, kernel=DEFAULT_KERNEL):
    w = 1.0 / np.square(noisemap)
    num = ndimage.correlate(data * w, kernel, mode='constant', cval=0.0)
    den = np.sqrt(ndimage.correlate(w, kernel ** 2, mode='constant', cval=0.0))
    with np.errstate(invalid='ignore', divide='ignore'):
        return np.where(den > 0, num / den, 0.0)


def _deblend(snr, group, thresh, nthresh, cont, minarea):
    """Split one connected group.  Returns a list of boolean masks (one per object), all inside ``group``."""
    vals = np.where(group, snr - thresh, 0.0)
    total = vals.sum()
    peak = snr[group].max()
    if not (peak > thresh) or total <= 0:
        return [group]
    levels = thresh * (peak / thresh) ** (np.arange(1, nthresh) / float(nthresh))
    branches = None
    for lev in levels:
        lab, k = ndimage.label(group & (snr > lev), structure=EIGHT)
        if k < 2:
            continue
        idx = np.arange(1, k + 1)
        flux = ndimage.sum(vals, lab, idx)
        area = ndimage.sum(group, lab, idx)
        good = idx[(flux >= cont * total) & (area >= 1)]
        if good.size >= 2:
            branches = [(lab == g) for g in good]   # keep descending: a finer split of the same group replaces this one
            # SExtractor splits at the first (lowest) level where the contrast criterion holds for >= 2 islands and then
            # recurses into each island; the recursion is the same procedure on the island
            out = []
            for b in branches:
                out.extend(_deblend(snr, b, lev, max(nthresh // 2, 4), cont * total / max(vals[b].sum(), 1e-300), 1))
            branches = out
            break
    if not branches:
        return [group]
    # pixels of the group outside every branch: to the branch whose peak is nearest in units of the branch's size
    yy, xx = np.nonzero(group)
    peaks, sizes = [], []
    for b in branches:
        py, px = np.unravel_index(np.argmax(np.where(b, snr, -np.inf)), snr.shape)
        peaks.append((py, px))
        sizes.append(max(np.sqrt(b.sum() / np.pi), 1.0))
    peaks, sizes = np.array(peaks, dtype=float), np.array(sizes)
    d = np.hypot(yy[:, None] - peaks[None, :, 0], xx[:, None] - peaks[None, :, 1]) / sizes[None, :]
    owner = np.argmin(d, axis=1)
    for i, b in enumerate(branches):
        inside = b[yy, xx]
        owner[inside] = i
    out = []
    for i in range(len(branches)):
        m = np.zeros_like(group)
        m[yy[owner == i], xx[owner == i]] = True
        out.append(m)
    return out


CLEAN_ZONE = 10.0   # SExtractor: neighbours further than 10 (a_1 + a_2) are not compared


def _shape(m, snr, thresh, minarea):
    """Moments of one object on the filtered image (SExtractor's pre-analysis): centre, ellipse coefficients, isophotal
    area, central amplitude of the equivalent Gaussian and the largest background the object would survive."""
    yy, xx = np.nonzero(m)
    v = snr[yy, xx]
    tot = v.sum()
    mx, my = (v * xx).sum() / tot, (v * yy).sum() / tot
    x2 = max((v * xx * xx).sum() / tot - mx * mx, 1.0 / 12.0)      # (a pixel's own variance: SExtractor's floor)
    y2 = max((v * yy * yy).sum() / tot - my * my, 1.0 / 12.0)
    xy = (v * xx * yy).sum() / tot - mx * my
    det = x2 * y2 - xy * xy
    if det < 1.0 / 144.0:                                           # singular: a round object of that floor
        xy, det = 0.0, x2 * y2
    half, root = 0.5 * (x2 + y2), np.sqrt(max(0.25 * (x2 - y2) ** 2 + xy * xy, 0.0))
    a, b = np.sqrt(half + root), np.sqrt(max(half - root, 1.0 / 12.0))
    srt = np.sort(v)[::-1]
    mthresh = float(srt[minarea - 1] - thresh) if srt.size >= minarea else 0.0
    area = np.pi * a * b
    return dict(mx=mx, my=my, a=a, cxx=y2 / det, cyy=x2 / det, cxy=-2.0 * xy / det, fdflux=float(tot), npix=int(v.size),
                unitarea=area, amp=float(tot / (2.0 * area)), mthresh=max(mthresh, 0.0))


def clean(masks, snr, thresh, minarea, clean_param=1.0):
    """Merges every detection that only exists on the wing of a brighter neighbour into that neighbour.  masks: list of
    boolean images; returns the surviving list (merged pixels added to their neighbour's mask)."""
    if len(masks) < 2:
        return masks
    beta = float(clean_param)
    sh = [_shape(m, snr, thresh, minarea) for m in masks]
    alive = [True] * len(masks)
    masks = [m.copy() for m in masks]
    for i in np.argsort([q['fdflux'] for q in sh]):                 # faintest first
        best, into = 0.0, -1
        for j in range(len(masks)):
            if j == i or not alive[j] or sh[j]['fdflux'] <= sh[i]['fdflux']:
                continue
            dx, dy = sh[i]['mx'] - sh[j]['mx'], sh[i]['my'] - sh[j]['my']
            if dx * dx + dy * dy >= (CLEAN_ZONE * (sh[i]['a'] + sh[j]['a'])) ** 2:
                continue
            q = sh[j]
            ratio = q['amp'] / thresh
            if ratio <= 1.0:
                continue
            alpha = (ratio ** (1.0 / beta) - 1.0) * q['unitarea'] / q['npix']
            val = 1.0 + alpha * (q['cxx'] * dx * dx + q['cyy'] * dy * dy + q['cxy'] * dx * dy)
            wing = q['amp'] * val ** (-beta) if 1.0 < val < 1e10 else 0.0
            if wing > sh[i]['mthresh'] and wing > best:
                best, into = wing, j
        if into >= 0:
            masks[into] |= masks[i]
            alive[i] = False
    return [m for m, ok in zip(masks, alive) if ok]


def extract(data, noisemap, thresh=3.0, minarea=15, deblend_nthresh=32, deblend_cont=0.001, clean_detections=True,
            clean_param=1.0):
    """-> (objects, segmentation map).  objects: structured array with 'x', 'y' (barycentres, pixel units, x = column),
    'npix', 'flux'; segmentation map: int array, pixels of object i carry i + 1, background 0 (sep's convention)."""
    data = np.asarray(data, dtype=np.float64)
    noisemap = np.asarray(noisemap, dtype=np.float64)
    ok = np.isfinite(data) & np.isfinite(noisemap) & (noisemap > 0)
    snr = matched_filter_snr(np.where(ok, data, 0.0), np.where(ok, noisemap, np.inf))
    det = snr > thresh
    lab, k = ndimage.label(det, structure=EIGHT)
    seg = np.zeros(data.shape, dtype=np.int32)
    found = []
    for g in range(1, k + 1):
        group = lab == g
        if group.sum() < minarea:
            continue
        found.extend(m for m in _deblend(snr, group, thresh, deblend_nthresh, deblend_cont, minarea)
                     if np.where(m, snr, 0.0).sum() > 0)
    if clean_detections:
        found = clean(found, snr, thresh, minarea, clean_param)
    objs = []
    ys, xs = np.indices(data.shape)
    for m in found:
        wgt = np.where(m, snr, 0.0)
        tot = wgt.sum()
        objs.append((float((wgt * xs).sum() / tot), float((wgt * ys).sum() / tot), int(m.sum()), float(np.where(m, data, 0.0).sum())))
        seg[m] = len(objs)
    objects = np.array(objs, dtype=[('x', 'f8'), ('y', 'f8'), ('npix', 'i4'), ('flux', 'f8')])
    return objects, seg
