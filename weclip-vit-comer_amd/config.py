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


# `fast` precision only: which operands of the adapter -> fuse chain in front of attn_pred = sigmoid(F^T F) carry their fp16
# remainder (hi+lo) in the FORWARD GEMMs.  Bit mask: 1 = cat (fuse input), 2 = linear_fuse weight, 4 = t1 (proj output),
# 8 = proj_2 weights, 16 = proj weights, 32 = the encoder's block outputs (the adapters' input).  attn_pred squares F's rounding error (256-long Gram product) and feeds
# get_aff_loss and the seg-trans affinity; measured at 512^2 against the reference fixture: see DESIGN.md §3.
head_lo = 0            # (an attribute, not an environment switch: tools/head_lo_probe.py sets it per measurement)

# `fast` precision only: PAR affinities kept as 16-bit fixed point between the sweeps (csrc/par.hip wc_par_forward_h); False =
# fp32 affinities like the reference (WeCLIP_model/PAR.py:64-92) with the fast GEMMs unchanged.
par_q16 = os.environ.get("WECLIP_PAR_F16", "1") != "0"
