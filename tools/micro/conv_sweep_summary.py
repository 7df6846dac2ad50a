"""Condense the counter passes of tools/prof_conv_sweep.sh: per case the timed result of conv_sweep.py --one and, for the
convolution kernel of that case, MFMA-busy cycles over (4 SIMDs x CU-busy cycles) and the bf16 matrix operations counted."""
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = []
for jf in sorted(glob.glob(os.path.join(src, "c*.json")), key=lambda p: int(os.path.basename(p)[1:-5])):
    txt = open(jf).read().strip()
    if not txt:
        continue
    case = json.loads(txt.splitlines()[-1])
    d = jf[:-5]
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    acc = {}
    if f:
        for r in csv.DictReader(open(f[0])):
            name = r["Kernel_Name"]
            if "conv_" not in name or "wgrad" in name and case["what"] != "wgrad":
                continue
            if case["what"] == "wgrad" and "wgrad" not in name:
                continue
            a = acc.setdefault(r["Counter_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    per = {k: v[1] / max(1, v[0]) for k, v in acc.items()}
    busy, cu, mops, gui = (per.get(k, 0.0) for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "GRBM_GUI_ACTIVE"))
    case.update({"launches_counted": acc.get("SQ_BUSY_CU_CYCLES", [0])[0], "mfma_busy_cycles": busy, "cu_busy_cycles": cu,
                 "mfma_busy_frac_of_cu_busy": busy / (4.0 * cu) if cu else None, "mfma_busy_frac_of_chip": busy / (gui * 256 * 4) if gui else None,
                 "gui_active_cycles_sum_over_8_xcds": gui, "bf16_mops": mops, "counted_gflop": mops * 512 * 1e-9})
    rows.append(case)
json.dump({"what": "one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE pass per case "
                   "(tools/prof_conv_sweep.sh); us / tflops are the timings of that same (profiled) run", "cases": rows}, open(dst, "w"), indent=1)
for r in rows:
    f = r["mfma_busy_frac_of_cu_busy"]
    print(f"{r['cin']}->{r['cout']} {r['image']} N={r['batch']} {r['what']} {r['kernel']}: {r['us']:.0f} us {r['tflops']:.0f} TFLOP/s  MFMA busy / CU busy = {f if f is None else round(f, 3)}")
