"""Run-time switches of the HIP path."""
import os

# GEMM operand precision on the MFMA path:
#   "fast"  : activations and weights rounded once to fp16 (fp32 accumulate) -- one MFMA pass.
#   "exact" : activations (and non-fp16 weights) carried as fp16 hi+lo pairs, 2-3 MFMA passes
#             accumulated in the same tile; products match the reference's fp32 GEMMs to ~1e-6.
# The reference's forced-fp16 out-projection (clip/myAtt.py:321) is a single fp16 pass in both.
precision = os.environ.get("WECLIP_PRECISION", "fast")


def exact():
    return precision == "exact"
