"""Summarises the rocprofv3 PMC passes of tools/evidence/round_artifacts.sh for k_solve: HBM bytes per launch as
MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts half of the
bytes of wide coalesced reads -> doubled), the SQ wait/active shares, stamped with the hash of the kernel sources."""
import csv
import glob
import json
import os
import sys

out, repo = sys.argv[1], sys.argv[2]
# optional third argument: a prefix for the pass directories (round 6: "mpc_" / "mpck_" = the passes over `bench.py --workload mpc-160 [--kkt]`,
# whose FIRST k_solve launch is the cold warm-up solve: the figure is the mean of the warm-started launches after it)
prefix = sys.argv[3] if len(sys.argv) > 3 else ""
skip_first = 1 if prefix else 0
sys.path.insert(0, repo)
from bench import lib_sha256, source_sha256  # noqa: E402


def per_launch(d, counter):
    vals = []
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection*.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_solve" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                vals.append((int(r.get("Dispatch_Id", len(vals)) or len(vals)), float(r["Counter_Value"])))
    vals = [v for _, v in sorted(vals)][skip_first:]
    return (sum(vals) / len(vals) if vals else None), len(vals)


fe, nf = per_launch("pmc_%sfetch" % prefix, "FETCH_SIZE")
wr, nw = per_launch("pmc_%swrite" % prefix, "WRITE_SIZE")
sq = {}
for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
          "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
    v, n = per_launch("pmc_%ssq" % prefix, c)
    if v is not None:
        sq[c] = v
batch, n, m, kernel = 8192, 1000, 2000, "k_solve"
try:  # the bench line of the FETCH pass names the workload it ran
    with open(os.path.join(out, "pmc_%sfetch_bench.json" % prefix)) as f:
        cfg = json.load(f)["config"]
    batch, n, m, kernel = int(cfg["batch_per_gpu"]), int(cfg.get("n", n)), int(cfg.get("m", m)), cfg.get("kernel", kernel)
except Exception:
    pass
res = {"kernel": kernel, "command": ("python bench.py --workload mpc-160" + (" --kkt" if prefix == "mpck_" else "") + " --steps 5 --no-cpu (batch %d, n=%d, m=%d): mean of the warm-started launches" % (batch, n, m))
       if prefix else "python bench.py --steps 1 --warmup 0 --no-cpu (batch %d, n=%d, m=%d)" % (batch, n, m),
       "batch": batch, "n": n, "m": m, "source_sha256": source_sha256(), "lib_sha256": lib_sha256(),
       "FETCH_SIZE_KB_per_launch": fe, "WRITE_SIZE_KB_per_launch": wr, "launches_seen": [nf, nw],
       "gfx950_correction": "FETCH_SIZE x 2 (128-byte requests tallied at 64 B), WRITE_SIZE as reported",
       "traffic_bytes_per_launch": (2 * fe + wr) * 1024 if fe is not None and wr is not None else None, "sq": sq}
if sq.get("SQ_WAVE_CYCLES"):
    wc = sq["SQ_WAVE_CYCLES"]
    res["sq_shares"] = {k: sq[k] / wc for k in sq if k != "SQ_WAVE_CYCLES"}
print(json.dumps(res, indent=1))
