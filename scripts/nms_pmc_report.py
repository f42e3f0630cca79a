#!/usr/bin/env python3
"""occupancy / issue / LDS report of the rotated-NMS kernels from the rocprofv3 PMC passes of scripts/pmc_cmd.sh
(gpurun_out/pmc_<tag>): python scripts/nms_pmc_report.py gpurun_out/pmc_nms200k k_nms_cull k_nms_heavy ... > profiles/...
Derivations (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
GRBM_GUI_ACTIVE is summed over the 8 XCDs -> kernel cycles = GRBM_GUI_ACTIVE / 8; clock = cycles / duration."""
import collections, csv, glob, sys
import json
args = sys.argv[1:]
json_out = None
if "--json" in args:
    k = args.index("--json"); json_out = args[k + 1]; del args[k:k + 2]
root, kernels = args[0], args[1:]
summary = {}
trace = (glob.glob(root + "/p1/*/*kernel_trace.csv") + glob.glob(root + "/p1/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(trace)))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/p*/*/*_counter_collection.csv") + glob.glob(root + "/p*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        for k in kernels:
            if k in r["Kernel_Name"]:
                cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
CUS, MAXW = 256, 32
for k in kernels:
    d = [r for r in rows if k in r["Kernel_Name"]]
    if not d:
        continue
    us = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in d) / len(d)
    r0 = d[0]
    c = {n: sum(v) / len(v) for n, v in cnt[k].items()}
    cycles = c["GRBM_GUI_ACTIVE"] / 8
    # threads per workgroup and workgroups of the WHOLE grid (round 5 counted the x dimension only: k_iou_cull_lanes' 2-D grid
    # came out as 4 "theoretical" waves per CU beside 9.3 achieved)
    wg = int(r0["Workgroup_Size_X"]) * int(r0.get("Workgroup_Size_Y", 1) or 1) * int(r0.get("Workgroup_Size_Z", 1) or 1)
    grid = (int(r0["Grid_Size_X"]) * int(r0.get("Grid_Size_Y", 1) or 1) * int(r0.get("Grid_Size_Z", 1) or 1)) // wg
    lds = int(r0["LDS_Block_Size"]); vg = int(r0["VGPR_Count"]) + int(r0["Accum_VGPR_Count"])
    waves_in_flight = c["SQ_WAVE_CYCLES"] * 4 / cycles
    by_vgpr = min(8, 512 // max(vg, 1)) * 4
    by_lds = (160 * 1024 // lds) * (wg // 64) if lds else MAXW
    limit = min(MAXW, by_vgpr, by_lds, -(-grid // CUS) * (wg // 64))
    print(f"== {k}  ({len(d)} dispatches in the kernel-trace pass; PMC passes slow a kernel by a few %)")
    print(f"  duration {us:9.1f} us   grid {grid} workgroups x {wg} threads   VGPR {vg}  SGPR {r0['SGPR_Count']}  LDS {lds} B/workgroup  scratch {r0['Scratch_Size']} B")
    print(f"  clock {cycles / us / 1e3:5.2f} GHz   waves launched {c['SQ_WAVES']:.0f}")
    print(f"  theoretical occupancy: {limit} waves/CU of {MAXW} (limits: VGPR {by_vgpr}, LDS {by_lds}, grid {-(-grid // CUS) * (wg // 64)})")
    print(f"  ACHIEVED occupancy: {waves_in_flight / CUS:5.2f} waves/CU = {waves_in_flight / CUS / MAXW * 100:4.1f} % of the 32-wave maximum "
          f"(mean resident waves chip-wide {waves_in_flight:.0f} = SQ_WAVE_CYCLES*4 / kernel cycles)")
    wc = c["SQ_WAVE_CYCLES"]
    print(f"  wave time: issuing {c['SQ_ACTIVE_INST_ANY'] / wc * 100:4.1f} % (VALU {c['SQ_ACTIVE_INST_VALU'] / wc * 100:4.1f} %, LDS {c.get('SQ_ACTIVE_INST_LDS', 0) / wc * 100:4.1f} %, "
          f"scalar {c.get('SQ_ACTIVE_INST_SCA', 0) / wc * 100:4.1f} %), waiting on s_waitcnt/barrier {c['SQ_WAIT_ANY'] / wc * 100:4.1f} %, "
          f"issue-stalled {c['SQ_WAIT_INST_ANY'] / wc * 100:4.1f} %")
    simd_cycles = cycles * CUS * 4
    print(f"  VALU utilisation: {c['SQ_ACTIVE_INST_VALU'] * 4 / simd_cycles * 100:4.1f} % of SIMD issue cycles "
          f"(SQ_ACTIVE_INST_VALU*4 / (cycles x 1024 SIMDs)); lane efficiency {c['SQ_THREAD_CYCLES_VALU'] / max(c['SQ_ACTIVE_INST_VALU'] * 64, 1) * 100:4.1f} % "
          f"(SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64))")
    print(f"  instructions: VALU {c['SQ_INSTS_VALU']:.3g}  SALU {c['SQ_INSTS_SALU']:.3g}  LDS {c['SQ_INSTS_LDS']:.3g}  VMEM rd {c['SQ_INSTS_VMEM_RD']:.3g} wr {c['SQ_INSTS_VMEM_WR']:.3g}  SMEM {c.get('SQ_INSTS_SMEM', 0):.3g}")
    if c.get("SQ_LDS_IDX_ACTIVE"):
        print(f"  LDS: bank-conflict cycles {c['SQ_LDS_BANK_CONFLICT']:.3g} of {c['SQ_LDS_IDX_ACTIVE']:.3g} LDS-array cycles = {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'] * 100:4.1f} %; "
              f"LDS array busy {c['SQ_LDS_IDX_ACTIVE'] / (cycles * CUS) * 100:4.1f} % of CU cycles")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        print(f"  HBM: FETCH_SIZE {c['FETCH_SIZE']:.0f} KiB (x2 on gfx950 for wide reads), WRITE_SIZE {c['WRITE_SIZE']:.0f} KiB per dispatch")
    summary[k] = {"us": round(us, 1), "waves_per_cu": round(waves_in_flight / CUS, 2), "occupancy_pct": round(waves_in_flight / CUS / MAXW * 100, 1),
                  "valu_util_pct": round(c['SQ_ACTIVE_INST_VALU'] * 4 / simd_cycles * 100, 1), "theoretical_waves_per_cu": limit}
if json_out:
    json.dump(summary, open(json_out, "w"), indent=1)
