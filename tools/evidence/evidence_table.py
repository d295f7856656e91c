#!/usr/bin/env python3
"""Prints the figures of an evidence set (profiles/<round>/final/, written by tools/evidence/round_artifacts.sh) that DESIGN.md section 4 quotes,
so that the table there is copied from the files and not retyped.  usage: evidence_table.py [profiles/r04/final]"""
import csv
import json
import os
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles", "r06", "final")
J = lambda f: json.load(open(os.path.join(d, f)))
b = J("bench_default.json")
r, st = b["roofline"], b["solve_stats"]
runs = [J(f) for f in ("bench_under_kernel_trace.json", "pmc_fetch_bench.json", "pmc_write_bench.json", "pmc_sq_bench.json", "bench_default.json")]
print("five default runs: %s QP/s" % ", ".join("%.0f" % x["value"] for x in runs))
print("reported: %.0f QP/s, %.1f ms per launch; setup+solve %.0f (device half %.0f), set_problem %.1f s, k_setup %.1f s" % (
    b["value"], r["kernel_ms"], b["value_setup_plus_solve"], b["setup"]["value_device_setup_plus_solve"], b["setup"]["set_problem_s"], b["setup"]["batch_setup_s"]))
rows = list(csv.DictReader(open(os.path.join(d, "rocprofv3_kernel_stats_bench_default.csv"))))
k = [x for x in rows if "k_solve" in x["Name"]][0]
print("kernel trace: %s calls %s average %.1f ms; bench's own events in that run %.1f ms" % (k["Name"], k["Calls"], float(k["AverageNs"]) * 1e-6, runs[0]["roofline"]["kernel_ms"]))
print("needed bytes %.3f TB + fused away %.3f TB; achieved %.2f TB/s frac %.3f; round-2 model frac %.3f" % (
    r["algorithmic_bytes_per_launch"] * 1e-12, r["fused_away_bytes_per_launch"] * 1e-12, r["achieved"] * 1e-3, r["frac"], r["frac_survey_model"]))
print("traffic %.2f TB = %.3f x needed (%.3f x round-2 model) = %.2f TB/s = %.2f of copy %.2f TB/s (read-only stream %.2f)" % (
    r["traffic"] * 1e-12, r["traffic_over_algorithmic"], r["traffic"] / r["survey_model_bytes_per_launch"], r["traffic_GBps"] * 1e-3,
    r["traffic_frac_of_measured_copy"], r["measured_copy_GBps"] * 1e-3, r["measured_read_GBps"] * 1e-3))
p = st["phase_ms_per_qp"]
print("per QP ms: total %.1f update %.1f (panel wave %.1f, sweeps %.1f, ranks %.0f) factor %.1f solve %.1f line search %.1f residuals %.1f" % (
    p["total"], p["update"], p["dbg"][1], st["per_qp_mean"]["n_sweeps"], st["per_qp_mean"]["n_rank1"], p["factor"], p["solve"], p["linesearch"], p["residuals"]))
t = J("k_solve_pmc_traffic.json")["sq_shares"]
print("PMC shares: wait %.0f %% issue-stalled %.0f %% active %.0f %% (VALU %.1f %%, LDS %.1f %%), bank conflicts %.1f %%" % (
    100 * t["SQ_WAIT_ANY"], 100 * t["SQ_WAIT_INST_ANY"], 100 * t["SQ_ACTIVE_INST_ANY"], 100 * t["SQ_ACTIVE_INST_VALU"], 100 * t["SQ_ACTIVE_INST_LDS"], 100 * t["SQ_LDS_BANK_CONFLICT"]))
ls = b["ldl_solve"]
print("k_ldlsolve_all: %.2f TB/s = %.2f of 8 = %.2f of read stream" % (ls["achieved"] * 1e-3, ls["frac"], ls["frac_of_measured_read"]))
m, mk, b5 = J("bench_mpc160.json"), J("bench_mpc160_kkt.json"), J("bench_b512.json")
print("mpc-160 %.0f QP/s (%.1f ms per step); kkt %.0f QP/s (%.1f ms); B=512 %.0f QP/s" % (m["value"], m["ms_per_step"], mk["value"], mk["ms_per_step"], b5["value"]))
c, cm = b["cpu_baseline"], m["cpu_baseline"]
print("cpu n=1000: %.0f QP/s on %d threads, tried %s, alone %.3f s, loaded %.2f s, wall %.0f s; mpc-160: %.0f QP/s on %d threads, tried %s" % (
    c["value"], c["cores"], c["threads_tried_qps"], c["single_qp_alone_s"], c["setup_plus_solve_s_per_qp"], c["wall_s"], cm["value"], cm["cores"], cm["threads_tried_qps"]))
def opt(name):
    pth = os.path.join(d, name)
    return open(pth).read().strip() if os.path.exists(pth) else "(%s: not in this evidence set)" % name
print("\n".join(l for l in opt("coop_timing.txt").splitlines() if "amdgpu" not in l))
if os.path.exists(os.path.join(d, "coop_config5.txt")):
    print("\n".join(l for l in open(os.path.join(d, "coop_config5.txt")).read().splitlines() if "amdgpu" not in l))
if os.path.exists(os.path.join(d, "rocprofv3_kernel_stats_coop_n2500.csv")):
    for x in csv.DictReader(open(os.path.join(d, "rocprofv3_kernel_stats_coop_n2500.csv"))):
        if "k_co_" in x["Name"]:
            print("   %-28s %6s calls, %.1f us average" % (re.sub(r"\(.*", "", x["Name"]).replace("qp512::", ""), x["Calls"], float(x["AverageNs"]) * 1e-3))
print(opt("sweep_probe.txt"))
if "mpc160" in b:
    for k in ("schur", "kkt"):
        mm = b["mpc160"].get(k, {})
        if "value" in mm:
            print("default line, mpc160.%s: %.0f QP/s (%.2f ms per step, kernel %.2f)" % (k, mm["value"], mm["ms_per_step"], mm["kernel_ms_per_step"]))
if "mpc160" in b:   # round 6: the roofline object of the config-3 lines
    for k in ("schur", "kkt"):
        rr = b["mpc160"].get(k, {}).get("roofline")
        if rr:
            print("default line, mpc160.%s roofline: needed %.3f GB per launch in %.2f ms = %.0f GB/s = frac %.3f; traffic %s = %s x needed; per QP KB: %s" % (
                k, rr["algorithmic_bytes_per_launch"] * 1e-9, rr["kernel_ms"], rr["achieved"], rr["frac"],
                ("%.3f GB" % (rr["traffic"] * 1e-9)) if rr["traffic"] else "null", ("%.2f" % rr["traffic_over_algorithmic"]) if rr["traffic"] else "null",
                {kk: round(v / 1024, 1) for kk, v in rr["bytes_per_qp"].items()}))
for v, what in ((-1, "guarded prefix tree (default)"), (0, "unguarded prefix tree"), (1, "running pivot in every column")):
    f = "bench_rank_sums_%d.json" % v
    if os.path.exists(os.path.join(d, f)):
        x = J(f)
        print("pivot sums, %-32s %.0f QP/s, frac %.4f, update %.2f ms per QP, hash %s, guard %s" % (what + ":", x["value"], x["roofline"]["frac"], x["solve_stats"]["phase_ms_per_qp"]["update"],
              x["solve_stats"]["solution_sha256_16"], x["solve_stats"].get("pivot_guard")))
print("pivot guard of the reported line: %s" % st.get("pivot_guard"))
if os.path.exists(os.path.join(d, "bench_sweep_ranks_32.json")):
    s32 = J("bench_sweep_ranks_32.json")
    print("sweep_ranks 32: %.0f QP/s, frac %.3f, needed %.3f TB, sweeps per QP %.1f, hash %s (default: %s)" % (
        s32["value"], s32["roofline"]["frac"], s32["roofline"]["algorithmic_bytes_per_launch"] * 1e-12, s32["solve_stats"]["per_qp_mean"]["n_sweeps"],
        s32["solve_stats"]["solution_sha256_16"], st["solution_sha256_16"]))
print("\n".join(l for l in opt("setup_timing.txt").splitlines() if "amdgpu" not in l))
if os.path.exists(os.path.join(d, "phase_traffic", "phase_traffic.json")):
    pt = json.load(open(os.path.join(d, "phase_traffic", "phase_traffic.json")))
    for k in ("factor", "sweep16", "sweep8", "sweep64", "solve"):
        if k in pt:
            print("phase traffic %-8s read %.2f write %.2f MB per QP, moved / needed %.3f" % (k, pt[k]["read"], pt[k]["write"], pt[k]["moved_over_needed"]))
print(opt("sload_coherence.txt"))
log = open(os.path.join(d, "pytest_gpu.log")).read().strip().splitlines()
print("\n".join(l for l in log if l.startswith(("fuzz campaigns", "  [hip]", "  fuzz total"))))
print(log[-1])
