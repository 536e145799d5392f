"""profiles/rNN_pmc_summary.json from the three PMC csv files written by scripts/prof_collect.sh."""
import csv, json, sys, collections, os
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
acc = collections.defaultdict(list); kern = None; dur = []
for name in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES"):
    for r in csv.DictReader(open(os.path.join(P, f"{tag}_pmc_{name}_round0.csv"))):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"])); kern = r["Kernel_Name"]
        if name == "SQ_WAVE_CYCLES" and r["Counter_Name"] == "SQ_INSTS_MFMA":
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
m = {k: sum(v) / len(v) for k, v in acc.items()}
M, S, D, C = 2000, 1000, 5, 10
us = sum(dur) / len(dur)
simd_cycles = 1024 * us * 1e-6 * 2.4e9
out = {
    "command": "rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline (three separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*; scripts/prof_collect.sh)",
    "kernel": kern + " dense round 0 (2000 galaxies x 1000 draws)",
    "launches_averaged": len(acc["SQ_INSTS_MFMA"]),
    "FETCH_SIZE_KB_raw": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
    "gfx950_correction": "FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section): doubled",
    "hbm_bytes_per_launch": (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024,
    "algorithmic_bytes_per_launch": 4.0 * D * M * S + 4.0 * C * M,
    "kernel_us_under_pmc": us,
    "SQ_INSTS_MFMA": m["SQ_INSTS_MFMA"], "SQ_INSTS_VALU": m["SQ_INSTS_VALU"],
    "SQ_VALU_MFMA_BUSY_CYCLES": m["SQ_VALU_MFMA_BUSY_CYCLES"],
    "SQ_VALU_MFMA_COEXEC_CYCLES": m["SQ_VALU_MFMA_COEXEC_CYCLES"],
    "SQ_ACTIVE_INST_VALU_quad": m["SQ_ACTIVE_INST_VALU"], "SQ_WAVE_CYCLES_quad": m["SQ_WAVE_CYCLES"],
    "SQ_WAIT_INST_ANY_quad": m["SQ_WAIT_INST_ANY"], "SQ_BUSY_CYCLES": m["SQ_BUSY_CYCLES"],
    "mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
    "valu_busy_frac": 4 * m["SQ_ACTIVE_INST_VALU"] / simd_cycles,
    "note": "busy fractions = counter / (1024 SIMDs x launch time x 2.4 GHz); ACTIVE_INST_VALU counts quad-cycles; "
            "COEXEC = 0 means fp32 MFMA and VALU never overlap on this kernel, so mfma_busy + valu_busy <= 1 is the ceiling",
}
json.dump(out, open(os.path.join(P, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
