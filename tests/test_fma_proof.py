"""The fused multiply-adds the HIP kernels use in the FIRST 1-D pass of each transform are bit-identical to the reference's
separate multiply and add -- checked against the reference's own object code (oracle/_ref/fma_proof, source oracle/fma_proof.c:
DCT_block / CDCT_block / IDCT_block / CIDCT_block on a million random blocks).  The same program shows that fusing the second
passes, or anything with the decoder's double-literal table, is NOT bit-safe, which is why those stay un-fused."""
import os
import re
import subprocess

import pytest

from oracle import pyoracle as po

PROOF = os.path.join(po.REF_DIR, "fma_proof")


@pytest.mark.ref
@pytest.mark.skipif(not os.path.exists(PROOF), reason="oracle/_ref/fma_proof not built (needs /root/reference)")
def test_first_pass_fma_is_bit_identical_to_reference_objects():
    r = subprocess.run([PROOF, "1000000", "20261003"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    quant = re.findall(r"(\d+) levels differ \(must be 0\)", r.stdout)      # round 5: the power-of-two quantiser as fma + truncating conversion
    assert len(quant) == 2 and all(int(x) == 0 for x in quant), r.stdout
    must = re.findall(r"(\d+) blocks differ \(must be 0\)", r.stdout)
    assert len(must) == 6 and all(int(x) == 0 for x in must), r.stdout      # fused x4, folded forward pass 1 x2
    broken = re.findall(r"(\d+) blocks differ \(expected > 0", r.stdout)
    assert len(broken) == 3 and all(int(x) > 0 for x in broken), r.stdout
