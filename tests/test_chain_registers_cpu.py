"""Register budget of the kernels that run WHILE the fused reduction + update waits for the regulariser chain in its kernel
(csrc/joint_reg_fused.h, csrc/joint_gm.h): up to two update blocks are resident per CU, one wave each per SIMD; what they leave
of the 512 registers per lane and SIMD must hold a wave of every kernel of the four-launch chain, or a chain that runs late
cannot be scheduled at all and every update block waits out its bound ("the regulariser of an iteration did not complete in
time": seen on an MI355X in round 4 with a 130-register a1' kernel, and again with a 123-register f1 kernel).

hipcc cross-compiles without a GPU; the figures come from -Rpass-analysis=kernel-resource-usage of csrc/joint_fit.hip."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, '..', 'lightcurver_amd', 'csrc')


def _alloc(vgpr, agpr):
    # unified register file of gfx90a and later: accumulation registers start at the next multiple of 4, the total is
    # allocated in granules of 8
    return (((vgpr + 3) // 4) * 4 + agpr + 7) // 8 * 8


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='no hipcc')
def test_chain_kernels_fit_beside_two_waiting_update_blocks(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    out = subprocess.run([hipcc, '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Rpass-analysis=kernel-resource-usage',
                          '-c', 'joint_fit.hip', '-o', str(tmp_path / 'jf.o')], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    usage, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r'remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]): (\d+)', line)
        if m and name:
            usage[name][m.group(1).split()[0]] = int(m.group(2))
    upd = [v for k, v in usage.items() if 'joint_reduce_update_kernel' in k]
    assert len(upd) == 1
    r_upd = _alloc(upd[0]['VGPRs'], upd[0].get('AGPRs', 0))
    # the flag-checking form only serves the 128 x 128 grid (lc_joint_step_update: N^2 / 32 <= 2 CUs blocks)
    chain = {k: v for k, v in usage.items() if 'mreg_mmx_kernelILi128E' in k or 'mreg_finish3_kernel' in k}
    assert len(chain) == 5, sorted(chain)
    for k, v in chain.items():
        r = _alloc(v['VGPRs'], v.get('AGPRs', 0))
        assert 2 * r_upd + r <= 512, (k, v, r_upd)
        assert v.get('ScratchSize', 0) == 0, (k, v)
