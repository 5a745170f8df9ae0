#!/usr/bin/env python3
"""Reduces a rocprofv3 --pmc SQ pass of bench.py (counter_collection.csv) to per-kernel-family sums and ratios:
where the waves' cycles go (parked on s_waitcnt/barrier, issue-stalled, issuing) and how busy the MFMA pipe is.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md).

  python tools/pmc_sq.py gpurun_out/pmc_sq profiles/r01_pmc_sq_c2_bf16.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

# kernel-name fragments, matched LONGEST FIRST against the demangled or mangled name (r04: "igemm_kernel" is not a substring of
# "igemm_pipe_kernel" and "wgemm_kernel" is not one of "wgemm_pipe_kernel": the round's two new kernels fell out of the summaries)
FAMILIES = ["brig_kernel", "igemm_pipe_kernel", "igemm_kernel", "wgemm_pipe_kernel", "wgemm_kernel", "wgrad_small_kernel", "conv_strip_kernel",
            "conv_fewin_kernel", "conv_fewout_kernel", "norm_act_fwd_vec", "norm_act_bwd_vec", "norm_act_fwd_small", "norm_act_bwd_small", "norm_act_fwd_reg", "norm_act_bwd_reg",
            "adam_flat_dev_kernel", "weight_prep_batched_kernel", "act_bwd_vec_kernel", "pack_pair_kernel", "pack_pair_idx_kernel",
            "ws_slab_sum_kernel", "slab_sum_kernel", "rgbuv_hist_fwd_kernel", "rgbuv_hist_bwd_kernel", "softmax256_kernel",
            "rgbuv_hist_fwd3_kernel", "rgbuv_hist_bwd3_kernel", "rgbuv_hist_fold_kernel", "hist_grad_prep_kernel", "rgbuv_points_kernel",
            "head_softmax_kernel", "head_dgrad_kernel", "adam_prep_batched_kernel", "bottleneck_kernel", "view_colsum_px8", "colsum_batched_kernel",
            "tanh_l1_fwd_pair_kernel", "tanh_l1_bwd_kernel", "bce_logits_kernel", "dropout_mask_kernel"]
FAMILIES.sort(key=len, reverse=True)


def family_of(name):
    return next((k for k in FAMILIES if k in name), None)


def main():
    folder, out = sys.argv[1:3]
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for f in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                fam = family_of(name)
                if fam is None:
                    continue
                acc[fam][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[fam].add(row.get("Dispatch_Id"))
    res = {}
    for fam, c in acc.items():
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        r = {"launches": len(launches[fam]), **{k: v for k, v in c.items()}}
        if wc:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                      "SQ_WAIT_INST_LDS"):
                if k in c:
                    r["frac_" + k] = round(c[k] / wc, 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_conflict_per_active"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if c.get("SQ_BUSY_CYCLES"):
            r["mfma_busy_per_sq_busy"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / c["SQ_BUSY_CYCLES"], 4)
        res[fam] = r
    json.dump({"source": "rocprofv3 --pmc (SQ block, one pass) of bench.py --config c2; quad-cycle units for SQ_WAVE_CYCLES / "
                         "SQ_WAIT_* / SQ_ACTIVE_INST_*", "kernels": res}, open(out, "w"), indent=1)
    for fam, r in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        print(fam, {k: v for k, v in r.items() if k.startswith("frac_") or k.startswith("mfma") or k.startswith("lds_") or k == "launches"})


if __name__ == "__main__":
    main()
