import json, sys, os
d = sys.argv[1]
for name in sys.argv[2:]:
    vals, h, ph = [], None, None
    for rep in (1, 2):
        try:
            j = json.load(open(os.path.join(d, "bench_%s_%d.json" % (name, rep))))
            vals.append(round(j["value"])); h = j["solve_stats"]["solution_sha256_16"]; ph = j["solve_stats"]["phase_ms_per_qp"]
        except Exception as e:
            vals.append("FAILED")
    line = "%-12s QP/s %s hash %s" % (name, vals, h)
    if ph:
        line += "  ms/QP total %.1f update %.1f factor %.1f solve %.1f ls %.1f resid %.1f" % (ph["total"], ph["update"], ph["factor"], ph["solve"], ph["linesearch"], ph["residuals"])
    print(line)
    try:
        t = json.load(open(os.path.join(d, "pt_" + name, "phase_traffic.json")))
        print("             traffic MB/QP (read, write): " + "  ".join("%s %.2f %.2f" % (k, t[k]["read"], t[k]["write"]) for k in ("factor", "sweep16", "solve") if k in t)
              + "  | factor %.2f ms sweep16 %.0f us solve %.3f ms" % (t["run"]["factor"]["ms"], t["sweep16"]["us_per_sweep"], t["run"]["solve"]["ms_per_rep"]))
    except Exception as e:
        pass
