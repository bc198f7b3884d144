"""No kernel of the library may spill registers, except the two that are known to and are never dispatched.

Round 6 met this the hard way: three RGB-side options added to the epilogue of the row-streaming conv as RUN-TIME switches of one
instantiation put 34 more live registers into the kernel every masked 8 -> 8 conv of the step uses; hipcc kept the launch bound (three
waves per SIMD = 168 VGPRs) and spilled 62 of them (masked conv alone 54.8 -> 57.7 us, the fused form 68 -> 117 us, a bench box "slower
than the others") -- parity tests cannot see that.  hipcc cross-compiles without a GPU, so the resource report is a CPU-tier check."""
import glob
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# (mangled-name fragments) the general-epilogue forms of the row-streaming Winograd kernel: kept for tests / A-B runs, not dispatched
# (conv_wino_strip.hip: "costs half the resident waves and measured 0.8-0.9x the tile kernel")
KNOWN = ('22conv_wino_strip_kernelILi8ELi2ELi0E', '22conv_wino_strip_kernelILi16ELi2ELi0E')


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_no_kernel_spills_registers():
    import __graft_entry__ as ge
    srcs = sorted(glob.glob(os.path.join(ROOT, 'pggan-pytorch_amd', 'csrc', '*.hip')))
    assert srcs
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for src in srcs:
            base = os.path.basename(src)
            cmd = [HIPCC] + ge.FLAGS + ge.FILE_FLAGS.get(base, []) + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', os.path.join(tmp, base + '.o')]
            procs.append((base, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)))
        spills, kernels = [], 0
        for base, p in procs:
            out = p.communicate()[0]
            assert p.returncode == 0, (base, out[-2000:])
            name = None
            for line in out.splitlines():
                m = re.search(r'Function Name: (\S+)', line)
                if m:
                    name = m.group(1)
                    kernels += 1
                m = re.search(r'(VGPRs|SGPRs) Spill: (\d+)', line)
                if m and m.group(1) == 'VGPRs' and int(m.group(2)) > 0 and not any(k in name for k in KNOWN):
                    spills.append((base, name, int(m.group(2))))
    assert kernels > 100, kernels                      # (the report was really produced)
    assert not spills, spills
