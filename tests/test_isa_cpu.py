"""The compiler must not serialise the loads of the element-wise / gather kernels.

hipcc turns a bounds branch around a load (`if (i < n) v = p[i];`, `v = ok ? p[i] : 0.f;`) into branch + load +
`s_waitcnt vmcnt(0)`: the loads of a thread then run as a chain of dependent memory latencies.  tools/isa_scan.py finds the pattern
in the disassembly; this test keeps the kernels that were rewritten in round 4 free of it (hipcc cross-compiles without a GPU)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

FILES = ["norm.hip", "comer.hip", "train_ops.hip", "msdeform.hip", "gradcam.hip", "affinity.hip", "losses.hip", "convstem.hip"]
# kernels that may keep the pattern, and why
ALLOWED = {
    "mrfp_dwconv_bwd_w_kernelIf": "fp32-gradient instantiation: the engine passes fp16 gradients",
    "msda_fwd_kernel": "one-thread-per-channel fallback for head widths that are not a multiple of 4",
    "matvec_cols_kernel": "fallback of the affinity sweeps for hw % 4 != 0",
    "matvec_rows_kernel": "fallback of the affinity sweeps for hw % 4 != 0",
    "seg_loss_bwd_y_kernel": "eight int64 label loads of a row group (COCO path); the class loads are batched",
}


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_rewritten_kernels_issue_their_loads_together():
    import isa_scan
    src = os.path.join(ROOT, "weclip-vit-comer_amd", "csrc")
    with ThreadPoolExecutor(4) as ex:
        rows = [r for part in ex.map(lambda f: isa_scan.report([os.path.join(src, f)], threshold=4), FILES) for r in part]
    bad = [(alone, loads, f, name) for alone, loads, f, name, _ in rows if not any(a in name for a in ALLOWED)]
    assert not bad, "loads waited for one at a time (see tools/isa_scan.py): %s" % bad
