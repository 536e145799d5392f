import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c, r = d["config"], d["roofline"]
print({k: round(d[k], 3) for k in ("value", "ms_per_step")}, "first-try acc", round(c["first_attempt_acceptance"], 4),
      "launches/step", c["launches_per_step"], "unfilled", c["unfilled_slots"], "evals/step", round(c["flow_evaluations_per_step"]),
      "fit loss", round(c["fit_final_loss"], 3))
print("sampler kernel ms", round(r["launch_ms"], 3), "frac", round(r["frac"], 4), "| train", round(d["train"]["value"] / 1e6, 1),
      "Mpairs/s, kernel ms", round(d["roofline_train"]["launch_ms"], 4), "frac", round(d["roofline_train"]["frac"], 4),
      "| log_prob", round(d["log_prob"]["value"] / 1e6, 1), "Mrows/s")
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"]["cores"], "core(s)")
