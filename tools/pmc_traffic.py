#!/usr/bin/env python3
"""Reduces two rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE, separate passes as the TCC slots require) of
`bench.py` to HBM bytes per launch per kernel family.  Units and gfx950 correction follow MI355X_MICROARCH.md (HBM):
both counters are in KiB; FETCH_SIZE reports half of the bytes of a wide coalesced stream, so it is doubled.

  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

# kernel-name fragment -> (C-ABI entry point, counts as a launch of it?).  Helper kernels of an entry point (slab sums, folds, prep)
# add their bytes to the entry's total but not to its launch count, so "per launch" means per C-ABI call as in bench.py.
FAMILIES = {"igemm_pipe_kernel": ("p2p_igemm", True), "igemm_kernel": ("p2p_igemm", True), "brig_kernel": ("p2p_igemm", True),
            "wgemm_pipe_kernel": ("p2p_wgemm", True), "wgemm_kernel": ("p2p_wgemm", True),
            "ws_slab_sum_kernel": ("p2p_wgrad_small", False), "slab_sum_kernel": ("p2p_wgemm", False),
            "norm_act_fwd_vec": ("p2p_norm_act_fwd", True), "norm_act_fwd_small": ("p2p_norm_act_fwd", True),
            "norm_act_bwd_vec": ("p2p_norm_act_bwd", True), "norm_act_bwd_small": ("p2p_norm_act_bwd", True),
            "norm_act_fwd_reg": ("p2p_norm_act_fwd", True), "norm_act_bwd_reg": ("p2p_norm_act_bwd", True),
            "adam_flat_dev_kernel": ("p2p_adam_flat_dev", True),
            "weight_prep_kernel": ("p2p_weight_prep_pad", True), "rgbuv_hist_fwd_kernel": ("p2p_rgbuv_hist_fwd", True),
            "rgbuv_hist_bwd_kernel": ("p2p_rgbuv_hist_hellinger_bwd", True), "rgbuv_hist_fwd3_kernel": ("p2p_rgbuv_hist_fwd3", True),
            "rgbuv_hist_fold_kernel": ("p2p_rgbuv_hist_fwd3", False),
            "rgbuv_hist_bwd3_kernel": ("p2p_rgbuv_hist_hellinger_bwd3", True), "hist_grad_prep_kernel": ("p2p_rgbuv_hist_hellinger_bwd3", False),
            "head_softmax_kernel": ("p2p_head_softmax_cce", True),
            "head_dgrad_kernel": ("p2p_head_dgrad", True), "wgrad_small_kernel": ("p2p_wgrad_small", True),
            "conv_strip_kernel": ("p2p_conv_strip", True),
            "conv_fewin_kernel": ("p2p_conv_fewin", True), "conv_fewout_kernel": ("p2p_conv_fewout", True)}


def load(folder, counter):
    per = defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {folder}")
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "")
                hit = next((v for k, v in FAMILIES.items() if k in name), None)       # first match: the longer fragments come first
                if hit is None:
                    continue
                fam, primary = hit
                # GEN (edge) instantiations of igemm_kernel are a different entry point
                if fam == "p2p_igemm" and "igemm_kernel" in name and ("Lb1ELb" in name or ", true," in name):
                    fam = "p2p_igemm_edge"
                per[fam][0] += float(row["Counter_Value"])
                per[fam][1] += 1 if primary else 0
    return per


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    config = sys.argv[4] if len(sys.argv) > 4 else "c2"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from palette_and_histo_gan_amd.build import source_fingerprint
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    res = {}
    for fam in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(fam, [0.0, 0])
        w, nw = write.get(fam, [0.0, 0])
        res[fam] = {"launches_fetch_pass": nf, "launches_write_pass": nw,
                    "fetch_bytes_per_launch": (f / nf) * 1024 * 2 if nf else None,      # gfx950: FETCH_SIZE counts 64 B per 128 B request
                    "write_bytes_per_launch": (w / nw) * 1024 if nw else None}
        if nf and nw:
            res[fam]["hbm_bytes_per_launch"] = res[fam]["fetch_bytes_per_launch"] + res[fam]["write_bytes_per_launch"]
    commit = os.popen("git rev-parse --short HEAD 2>/dev/null").read().strip()
    json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --config {config}, "
                         "KiB units, FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)",
               "fingerprint": source_fingerprint(), "commit": commit, "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
