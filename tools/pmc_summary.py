#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc passes (counter_collection CSVs) into one JSON per kernel family.

    python tools/pmc_summary.py <dir with pmc*/ sub-directories> <out.json>

Per family: launches, MFMA pipe busy fraction, wait fractions, VALU+SALU per MFMA, LDS bank-conflict fraction
and HBM bytes per launch.  FETCH_SIZE / WRITE_SIZE are KiB (x1024).  On gfx950 FETCH_SIZE reports HALF of the bytes read,
for EVERY access width: round 4 calibrated it against kernels that touch exactly 1 GiB once (tools/clock/fetch_calib.hip,
profiles/r4_fetch_size_calibration.txt) -- 16 / 8 / 4 bytes per lane, 8 bytes at a 16-byte pitch and 16-byte LDS-DMA all
report 524 300 KiB = TCC_EA0_RDREQ (8.39 M requests of 128 B) x 64 B, TCC_BUBBLE (the counter meant to tally the 128-byte
requests) reads 0; WRITE_SIZE reports the bytes written exactly (1 048 576 KiB), partial 64-byte lines as whole ones.  So
`hbm_bytes_per_launch` = 2 * FETCH + WRITE for every family (round 3 doubled only the 16-byte streams and left the
dgrad number "between" two readings).  It counts L2 <-> fabric traffic: Infinity-Cache hits are in it.
"""
import csv
import glob
import json
import os
import re
import sys

FAMILIES = [  # (family, regex on the kernel name, wide 16-byte reads?)
    ("conv_pl_kernel_fwd", r"conv_pl_kernel<0,", True),
    ("conv_pl_kernel_dgrad", r"conv_pl_kernel<1,", True),
    ("conv_pl9_kernel_fwd", r"conv_pl9_kernel<0,", True),
    ("conv_pl9_kernel_dgrad", r"conv_pl9_kernel<1,", True),
    ("wgrad_group9_kernel_rows14", r"wgrad_group9_kernel<6,", True),
    ("wgrad_group9_kernel_rows28", r"wgrad_group9_kernel<8,", True),
    ("wgrad_group9_kernel_rows56", r"wgrad_group9_kernel<12,", True),
    ("wgrad_group1_kernel", r"wgrad_group1_kernel", True),
    ("wgrad_group_stem_kernel", r"wgrad_group_stem_kernel", True),
    ("wgrad_reduce_multi_kernel", r"wgrad_reduce_multi_kernel", True),
    ("wgrad_pl9_kernel", r"wgrad_pl9_kernel", True),
    ("wgrad_pl_kernel", r"wgrad_pl_kernel", True),
    ("pl_maxpool_fwd_kernel", r"pl_maxpool_fwd_kernel", True),
    ("pl_maxpool_bwd_kernel", r"pl_maxpool_bwd", True),
    ("pl_avgpool_affine_kernel", r"pl_avgpool_affine_kernel", True),
    ("conv_x6_kernel_fwd", r"conv_x6_kernel<\d+,\s*\d+,\s*\d+,\s*0,", True),
    ("conv_x6_kernel_dgrad", r"conv_x6_kernel<\d+,\s*\d+,\s*\d+,\s*1,", True),
    ("wgrad_x6_kernel", r"wgrad_x6_kernel", True),
    ("conv_igemm_kernel_fwd", r"conv_igemm_kernel<\d+,\s*\d+,\s*0,", False),
    ("conv_igemm_kernel_dgrad", r"conv_igemm_kernel<\d+,\s*\d+,\s*[12],", False),
    ("conv_wgrad_kernel", r"conv_wgrad_kernel", False),
    ("wgrad_reduce_kernel", r"wgrad_reduce_kernel", True),
    ("pool3_vec_kernel", r"pool3_vec_kernel", True),
    ("pool_max2_bwd_vec_kernel", r"pool_max2_bwd_vec_kernel", True),
    ("pool_fwd_kernel", r"pool_fwd_kernel", False),
    ("pool_bwd_kernel", r"pool_bwd_kernel", False),
    ("relu_bn_bwd_kernel", r"relu_bn_bwd_kernel", True),
    ("sgd_multi_kernel", r"sgd_multi_kernel", True),
]


def family_of(name):
    for fam, rx, _ in FAMILIES:
        if re.search(rx, name):
            return fam
    return None


def main():
    root, out = sys.argv[1], sys.argv[2]
    sums = {}      # family -> counter -> [sum, n]
    clock = {}     # family -> [sum of GRBM_GUI_ACTIVE / 8, sum of dispatch durations in ns]
    for path in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                fam = family_of(row["Kernel_Name"])
                if fam is None:
                    continue
                c = sums.setdefault(fam, {}).setdefault(row["Counter_Name"], [0.0, 0])
                c[0] += float(row["Counter_Value"])
                c[1] += 1
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    k = clock.setdefault(fam, [0.0, 0.0])
                    k[0] += float(row["Counter_Value"]) / 8.0
                    k[1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    wide = {fam: w for fam, _, w in FAMILIES}
    res = {}
    for fam, cs in sums.items():
        def tot(name):
            return cs[name][0] if name in cs else None

        def per(name):
            return cs[name][0] / cs[name][1] if name in cs and cs[name][1] else None
        n = max(v[1] for v in cs.values())
        r = {"launches_sampled": n, "wide_reads": wide[fam]}
        gui, busy = tot("GRBM_GUI_ACTIVE"), tot("SQ_VALU_MFMA_BUSY_CYCLES")
        wc = tot("SQ_WAVE_CYCLES")
        if busy is not None and gui:
            # busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (separate passes of the
            # same command, hence per-launch averages)
            r["mfma_pipe_busy_frac"] = round(per("SQ_VALU_MFMA_BUSY_CYCLES") / (per("GRBM_GUI_ACTIVE") / 8 * 1024), 4)
        if wc:
            for key, cname in (("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_any_frac", "SQ_WAIT_INST_ANY"),
                               ("active_inst_any_frac", "SQ_ACTIVE_INST_ANY")):
                if tot(cname) is not None:
                    r[key] = round(tot(cname) / wc, 4)
        if gui and per("GRBM_GUI_ACTIVE"):
            r["gpu_active_cycles_per_launch"] = round(per("GRBM_GUI_ACTIVE"))
        if fam in clock and clock[fam][1] > 0:
            # shader clock the kernels of this family actually ran at (GRBM_GUI_ACTIVE counts per XCD; the chip clocks to its
            # power budget: MI355X_MICROARCH.md, "DVFS give-back")
            r["effective_clock_ghz"] = round(clock[fam][0] / clock[fam][1], 3)
        mf = per("SQ_INSTS_MFMA")
        if mf:
            other = sum(per(cn) or 0.0 for cn in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"))
            valu = per("SQ_INSTS_VALU")
            # (counted like the round-2 review did: SQ_INSTS_VALU taken as the non-matrix vector instructions)
            if valu is not None:
                r["non_mfma_insts_per_mfma"] = round(other / mf, 2)
                r["valu_per_mfma"] = round(valu / mf, 2)
            if per("SQ_INSTS_SALU") is not None:
                r["salu_per_mfma"] = round(per("SQ_INSTS_SALU") / mf, 2)
        lds_act, lds_conf = tot("SQ_LDS_IDX_ACTIVE"), tot("SQ_LDS_BANK_CONFLICT")
        if lds_act:
            r["lds_bank_conflict_frac"] = round(lds_conf / lds_act, 4)
        f, w = per("FETCH_SIZE"), per("WRITE_SIZE")
        if f is not None:
            r["fetch_bytes_per_launch_raw"] = round(f * 1024)
        if w is not None:
            r["write_bytes_per_launch"] = round(w * 1024)
        if f is not None and w is not None:
            r["hbm_bytes_per_launch"] = round((2 * f + w) * 1024)
        for cname in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
                      "SQ_INSTS_MFMA"):
            if per(cname) is not None:
                r[cname.lower() + "_per_launch"] = round(per(cname))
        res[fam] = r
    res["_note"] = ("effective_clock_ghz = GRBM_GUI_ACTIVE / dispatch duration; NOT what the waves see under matrix load: the cycle "
                    "counter calibrated against the real-time counter inside the kernels (tools/ablate_conv_pl.py, "
                    "tools/trace_wgrad_pl.py) reads 1.7-1.9 GHz in the MFMA-dense launches")
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True)[:4000])


if __name__ == "__main__":
    main()
