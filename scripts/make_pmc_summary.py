"""profiles/rNN_pmc_summary.json (argument: the tag, default r04) from the per-kernel counter rows written by
scripts/prof_collect.sh (profiles/rNN_pmc_maf.csv: default bench; rNN_pmc_nsf.csv: bench.py --workload nsf_cfg3; four separate
rocprofv3 --pmc passes each: FETCH_SIZE | WRITE_SIZE | two SQ sets).  Keys of the top level / "train" are the ones
bench.py reads (hbm_bytes_per_launch, *_busy_frac); "nsf" holds the sampler / log_prob / training kernels of cfg3."""
import collections, csv, json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def section(rows, pat, grid=None, alg_bytes=None, note=None):
    """rows: the aggregated table of prof_collect.sh (one row per kernel x launch shape x counter)."""
    acc = collections.defaultdict(list); dur = {}; kern = None; meta = None
    # of several launch shapes of one kernel (warm-up fit, bench batch ...) take `grid`, else the one with the longest launches
    cand = [r for r in rows if re.search(pat, r["Kernel_Name"]) and (grid is None or int(r["Grid_Size"]) == grid)]
    if cand and grid is None:
        best = max(cand, key=lambda r: float(r["Mean_us"]))
        cand = [r for r in cand if r["Grid_Size"] == best["Grid_Size"] and r["Kernel_Name"] == best["Kernel_Name"]]
    for r in cand:
        acc[r["Counter_Name"]].append(float(r["Counter_Mean"])); kern = r["Kernel_Name"]
        dur[(r["Counter_Name"], "all")] = float(r["Mean_us"])
        meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
        meta["launches_per_counter_pass"] = int(r["Launches"])
    if not acc:
        return None
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    d_sq = [v for (c, _), v in dur.items() if c == "SQ_WAVE_CYCLES"] or list(dur.values())
    us = sum(d_sq) / len(d_sq)
    simd_cycles = 1024 * us * 1e-6 * 2.4e9
    out = {"kernel": kern, "launch": meta, "launches_averaged": len(d_sq), "kernel_us_under_pmc": us,
           "FETCH_SIZE_KB_raw": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
           "gfx950_correction": "FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section): doubled",
           "hbm_bytes_per_launch": (2 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024,
           "algorithmic_bytes_per_launch": alg_bytes}
    for k in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_VALU_TRANS_F32",
              "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if k in m:
            out[k] = m[k]
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in m:
            out[k + "_quad"] = m[k]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        out["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
        out["valu_busy_frac"] = 4 * m["SQ_ACTIVE_INST_VALU"] / simd_cycles
        out["wait_frac_of_wave_cycles"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    if all(k in m for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_TRANS_F32")):
        # what the vector issue port of a SIMD spends: 4 cycles per plain VALU instruction, 8 per transcendental, 8 per MFMA
        # (MI355X_MICROARCH.md, per-instruction constants) -- an estimate from instruction counts, not a counter
        port = 4 * (m["SQ_INSTS_VALU"] - m["SQ_INSTS_VALU_TRANS_F32"]) + 8 * m["SQ_INSTS_VALU_TRANS_F32"] + 8 * m["SQ_INSTS_MFMA"]
        out["vector_issue_port_frac_est"] = port / simd_cycles
    if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
        out["issue_stall_frac_of_wave_cycles"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
        out["issuing_frac_of_wave_cycles"] = m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    if note:
        out["note"] = note
    return out


maf = list(csv.DictReader(open(os.path.join(P, f"{tag}_pmc_maf.csv"))))
M, S, D, C, P_ = 2000, 1000, 5, 10, 32300
# the default sampler (round 5): every product in fp32 -- template argument PREC = 1 is the last one of the kernel name
out = section(maf, r"k_maf_samp16<.*, [12]>", alg_bytes=4.0 * D * M * S + 4.0 * C * M)
if out is None:
    out = section(maf, "k_maf_samp16", alg_bytes=4.0 * D * M * S + 4.0 * C * M)
out["command"] = ("rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
                  "--no-cpu-baseline (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_* | SQ_*; scripts/prof_collect.sh)")
out["note"] = ("busy fractions = counter / (1024 SIMDs x launch time x 2.4 GHz); *_quad counters count quad-cycles.  One sampler "
               "launch = one whole bench step (2000 galaxies x 1000 accepted draws, first attempts + retries).")
# training kernel at the bench batch: 16 384 rows = 256 workgroups of 512 threads
out["train"] = section(maf, "k_maf_trainc", grid=256 * 512, alg_bytes=4.0 * (D + C) * 16384 + 4.0 * 2 * P_,
                       note="cooperative 16-row kernel, batch 16 384: per-workgroup gradient partials are written with plain stores "
                            "(256 x 145 KB) and summed by k_gather_c")
out["sampler_split_bf16_leg"] = section(maf, r"k_maf_samp16<.*, 0>", alg_bytes=4.0 * D * M * S + 4.0 * C * M,
                                        note="the opt-in fast mode (sf_set_sampler_fp32(0)): hidden H x H blocks as split-bf16 x3 products")
out["train_gather"] = section(maf, "k_gather_c")
out["train_prep"] = section(maf, "k_train_prep")
nsf_path = os.path.join(P, f"{tag}_pmc_nsf.csv")
if os.path.exists(nsf_path):
    nsf = list(csv.DictReader(open(nsf_path)))
    Mn, Dn, Cn = 20000, 8, 20
    out["nsf"] = {
        "command": "the same passes over  bench.py --workload nsf_cfg3 --steps 2 --warmup 1  (BASELINE configs[2])",
        "sampler": section(nsf, r"k_sample_persist<NsfOps<\d+, \d+, \d+, true, 2(, \w+)?>", alg_bytes=4.0 * Dn * Mn * 1000 + 4.0 * Cn * Mn),
        "sampler_fp32_leg": section(nsf, r"k_sample_persist<NsfOps<\d+, \d+, \d+, true, 0(, \w+)?>", alg_bytes=4.0 * Dn * Mn * 1000 + 4.0 * Cn * Mn),
        "log_prob": section(nsf, "k_logprob"),
        "train16384": section(nsf, "k_nsf_trainc", grid=512 * 256, alg_bytes=4.0 * (Dn + Cn) * 16384 + 4.0 * 2 * 91570,
                              note="cooperative 16-row kernel, batch 16 384, one chunk per workgroup: per-workgroup gradient partials "
                                   "(512 x 0.6 MB, plain stores) summed by k_gather_c2 -- the fastest form (sf_nsfc.hip, sf_nsfc_acc_mode)"),
        "train16384_gather": section(nsf, "k_gather_c2"),
    }
    ap = os.path.join(P, f"{tag}_pmc_nsfatomic.csv")
    if os.path.exists(ap):
        na = list(csv.DictReader(open(ap)))
        out["nsf"]["train16384_xcd_replicas"] = section(
            na, "k_nsf_trainc", grid=512 * 256, alg_bytes=4.0 * (Dn + Cn) * 16384 + 4.0 * 2 * 91570,
            note="the same launch with SF_GRAD_ACC=atomic: f32 atomics into one gradient replica per XCD, nothing but inputs and the "
                 "u stash crosses the HBM")
ar_path = os.path.join(P, f"{tag}_pmc_nsfar.csv")
if os.path.exists(ar_path):   # the lampe backend's flow (bench.py --workload nsfar_cfg2): 16-candidate register-tile sampler, training kernel
    ar = list(csv.DictReader(open(ar_path)))
    out["nsfar"] = {
        "command": "the same passes over  bench.py --workload nsfar_cfg2 --steps 3 --warmup 1",
        "sampler": section(ar, "k_ar_samp16", alg_bytes=4.0 * D * M * S + 4.0 * C * M,
                           note="persistent launch of the catalogue call: sixteen candidates per wave in register tiles, weight blocks straight "
                                "from L2; survivors of 32 attempts go to k_ar_find16 / k_ar_resolve16"),
        "find": section(ar, "k_ar_find16"),
        "train": section(ar, "k_ar_train"),
    }
json.dump(out, open(os.path.join(P, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else "...") for k, v in out.items()}, indent=1)[:3000])
