"""profiles/r02_pmc_summary.json from the PMC csv files written by scripts/prof_collect_r02.sh (three separate
rocprofv3 --pmc passes of the default bench command: FETCH_SIZE | WRITE_SIZE | SQ_*), for the persistent sampler
kernel (one launch = one bench step: 2000 galaxies x 1000 draws, first attempts + retries) and the training kernel."""
import collections, csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def section(kernel_tag, alg_bytes):
    acc = collections.defaultdict(list); kern = None; dur = []
    for name in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES"):
        seen = set()
        for r in csv.DictReader(open(os.path.join(P, f"{tag}_pmc_{name}_{kernel_tag}.csv"))):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); kern = r["Kernel_Name"]
            if name == "SQ_WAVE_CYCLES" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    us = sum(dur) / len(dur)
    # SQ_WAVE_CYCLES counts quad-cycles summed over waves; 3 waves/SIMD x 1024 SIMDs resident in the persistent sampler:
    # the clock under the profiler follows from it; busy fractions use the nominal 2.4 GHz like round 1
    simd_cycles = 1024 * us * 1e-6 * 2.4e9
    return {
        "kernel": kern, "launches_averaged": len(dur), "kernel_us_under_pmc": us,
        "FETCH_SIZE_KB_raw": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
        "gfx950_correction": "FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section): doubled",
        "hbm_bytes_per_launch": (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024,
        "algorithmic_bytes_per_launch": alg_bytes,
        "SQ_INSTS_MFMA": m["SQ_INSTS_MFMA"], "SQ_INSTS_VALU": m["SQ_INSTS_VALU"],
        "SQ_VALU_MFMA_BUSY_CYCLES": m["SQ_VALU_MFMA_BUSY_CYCLES"],
        "SQ_VALU_MFMA_COEXEC_CYCLES": m["SQ_VALU_MFMA_COEXEC_CYCLES"],
        "SQ_ACTIVE_INST_VALU_quad": m["SQ_ACTIVE_INST_VALU"], "SQ_WAVE_CYCLES_quad": m["SQ_WAVE_CYCLES"],
        "SQ_WAIT_ANY_quad": m["SQ_WAIT_ANY"], "SQ_BUSY_CYCLES": m["SQ_BUSY_CYCLES"],
        "mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
        "valu_busy_frac": 4 * m["SQ_ACTIVE_INST_VALU"] / simd_cycles,
        "wait_frac_of_wave_cycles": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
    }


M, S, D, C, P_ = 2000, 1000, 5, 10, 32300
out = section("sampler", 4.0 * D * M * S + 4.0 * C * M)
out["command"] = ("rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
                  "--no-cpu-baseline (three separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*; scripts/prof_collect_r02.sh)")
out["note"] = ("busy fractions = counter / (1024 SIMDs x launch time x 2.4 GHz); ACTIVE_INST_VALU and WAVE_CYCLES count "
               "quad-cycles.  One launch = one whole bench step (about 3.2e6 flow evaluations for 2.0e6 accepted draws).")
out["train"] = section("train16384", 4.0 * (D + C) * 16384 + 4.0 * 2 * P_)
json.dump(out, open(os.path.join(P, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
