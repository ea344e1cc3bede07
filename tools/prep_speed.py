"""Ad-hoc: device time and achieved HBM bandwidth of the fused stamp pre-processing at C3 scale."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd import _lib
from lightcurver_amd.processes.preprocessing import prepare_stamps
K, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
data = rng.normal(0, 5, (K, n, n)).astype(np.float32)
noise = np.full((K, n, n), 5.0, np.float32)
bad = rng.random((K, n, n)) < 0.02
ctx = _lib.Context(0)
for _ in range(3):
    out = prepare_stamps(data, noisemap=noise, bad=bad, ctx=ctx)
px = K * n * n
# algorithmic bytes: read data 4 + noise 4 + bad 1, write data 4 + noise 4 + weight 4 = 21 B / pixel
print(f'K={K} n={n}: kernel {out["kernel_ms"]*1e3:.1f} us, {px * 21 / (out["kernel_ms"]*1e-3) / 1e9:.0f} GB/s algorithmic '
      f'({px * 21 / (out["kernel_ms"]*1e-3) / 8e12 * 100:.1f} % of 8 TB/s)')
