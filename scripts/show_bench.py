import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:round(d[k],3) for k in ("value","ms_per_step")}, "acc",round(d["config"]["acceptance"],4), "rounds",d["config"]["rounds_per_step"], "unfilled",d["config"]["unfilled_slots"], "loss", round(d["config"]["fit_final_loss"],3), "k_ms", round(d["roofline"]["launch_ms"],3), "train", round(d["train"]["value"]/1e6,1))
