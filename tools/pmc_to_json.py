#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the rocprofv3 --pmc passes of tools/pmc_passes.sh (what bench.py reports as
`traffic` and `pmc`).  usage: python tools/pmc_to_json.py <dir with p*/p*_counter_collection.csv> <source label>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

# (name pattern, pattern that must NOT occur): the filter-bank variants of the packed STFT kernel are instantiations
# <0, 400, true, 1|2> of the same template as the spectrum-out kernel <0, 400, true, 0>
KERNELS = {"mcep_mfma_fwd": ("mcep_mfma_fwd_kernel_hILi8ELb0ELb0E", None), "stft512_fwd": ("stft512_fwd_pk_kernel<0, 400, true, 0,", None),
           "stft512_fbank_fwd": ("stft512_fwd_pk_kernel<0, 400, true, 1,", None),
           "mcep_mfma_bwd": ("mcep_mfma_bwd2_kernel_h", None), "stft512_bwd": ("stft512_bwd_pk_kernel<400, 80, false, false>", None),
           "stft512_istft": ("stft512_bwd_pk_kernel<400, 80, true, false>", None),
           "frame_window_lpc24_fwd": ("frame_window_lpc24_kernel", None),
           "frame_window_lpc24_mfma_fwd": ("frame_window_lpc24_mfma_kernel", None),
           "frame_window_lpc24_bwd_mfma": ("frame_window_lpc24_bwd_mfma_kernel", None), "stft512_mcep_fused_fwd": ("mcep_mfma_fwd_kernel_hILi8ELb1ELb0E", None)}   # (these two arrive mangled)
WAVES_PER_SIMD = {"mcep_mfma_fwd": 2, "stft512_fwd": 4, "stft512_fbank_fwd": 4, "mcep_mfma_bwd": 2, "stft512_bwd": 4, "stft512_istft": 4,
                  "frame_window_lpc24_fwd": 2, "frame_window_lpc24_mfma_fwd": 3, "frame_window_lpc24_bwd_mfma": 2, "stft512_mcep_fused_fwd": 2}
FRAMES = 204800

acc = defaultdict(lambda: [0.0, 0])
dirs = sys.argv[1].split(",")   # several pass directories (forward command, backward command, ...)
# A kernel's record comes from ONE command: the first directory in which it was dispatched.  Dispatches of the same kernel by
# a later command (e.g. the with-history launch of mcep_mfma_fwd in the backward command) are kept apart under "<key>@<dir>".
first_dir = {}
for di, d in enumerate(dirs):
    for f in sorted(glob.glob(d + "/p*/p*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            for key, (pat, _) in KERNELS.items():
                if pat in row["Kernel_Name"].replace("(int)", "").replace("(bool)", ""):
                    k2 = key if first_dir.setdefault(key, di) == di else key + "@" + os.path.basename(d.rstrip("/"))
                    a = acc[(k2, row["Counter_Name"])]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
out = {"source": sys.argv[2] if len(sys.argv) > 2 else sys.argv[1],
       "note": "KB per launch. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts wide coalesced reads at half "
               "their bytes: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; WRITE_SIZE is used as reported.",
       "kernels": {}}
for key in sorted({k for k, _ in acc}, key=lambda k: (k.split("@")[0] not in KERNELS, k)):
    c = {n: s / k for (kk, n), (s, k) in acc.items() if kk == key}
    if "FETCH_SIZE" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    out["kernels"][key] = {
        "FETCH_SIZE_KB": round(c["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(c["WRITE_SIZE"], 1),
        "hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024, "frames_per_launch": FRAMES,
        "derived": {
            "cycles": round(cyc, 1),
            "valu_busy": round(4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), 3),
            "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cyc), 3),
            "lds_busy": round(c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 3),
            "lds_conflict_frac": round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), 3),
            "waves_per_simd": WAVES_PER_SIMD[key.split("@")[0]],
            "valu_insts_per_frame": round(c["SQ_INSTS_VALU"] / FRAMES, 1),
            "mfma_insts_per_frame": round(c.get("SQ_INSTS_MFMA", 0.0) / FRAMES, 1),
        },
    }
out["derived_note"] = ("Derived from the same --pmc passes (gfx950: 256 CUs, 1024 SIMDs, 8 XCDs). cycles = GRBM_GUI_ACTIVE / 8; "
                       "valu_busy = 4 * SQ_ACTIVE_INST_VALU / (1024 * cycles); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * cycles); "
                       "lds_busy = SQ_LDS_IDX_ACTIVE / (256 * cycles); lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
                       "waves_per_simd = resident waves per SIMD of the persistent launch; valu_insts_per_frame = SQ_INSTS_VALU / "
                       "frames (wave instructions).  Counter runs serialise the kernels and lower the clock: cycles here are longer "
                       "than the un-instrumented launch.")
json.dump(out, sys.stdout, indent=1)
print()
