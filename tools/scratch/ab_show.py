import json, sys
for f in sys.argv[1:] or ("ab_new", "ab_prev"):
    d = json.loads(open("/root/repo/gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    ph = d["solve_stats"]["phase_ms_per_qp"]; tr = d["roofline"].get("traffic")
    print(f, round(d["value"], 1), "frac", round(d["roofline"]["frac"], 3), "ldl", round(d["ldl_solve"]["achieved"]), "solved", d["solve_stats"]["all_solved"],
          {k: (round(v, 1) if not isinstance(v, list) else [round(x, 1) for x in v]) for k, v in ph.items()})
