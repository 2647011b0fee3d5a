#!/usr/bin/env python3
"""PROJECTED strong scaling of a Newton step over the GPUs of one node (no multi-GPU box is available to the build: everything here is a
projection, and says so): rank 0's LOCAL work at N = 1, 2, 4, 8 measured on one MI355X (tools/rank_local_timing.py: its blocks' subtrees + the
replicated root front, no reductions) + a model of the reductions the step holds -- one of the root panel, `solves` of the root right-hand
side -- as the library performs them by default (reduce-scatter + all-gather over peer copies, tlpk_api.cpp: multi_allreduce_rs): every shard
sends N - 1 slices of 1/N of the buffer over N - 1 DIFFERENT xGMI links at once, the owner adds them, and sends its slice back the same way:
    t_reduce(bytes, N) = 2 * [ t_lat + (bytes / N) / (eff * BW_link) ] + t_sum,    BW_link = 153 GB/s per direction (MI355X_MICROARCH.md),
eff = 0.7, t_lat = 15 us per phase (copy enqueue + event hand-over across devices), t_sum = 5 us.
    python tools/scale_projection.py gpurun_out/rank_local_c4.txt gpurun_out/rank_local_headline.txt
JSON=path also writes the table as JSON (profiles/scale_projection.json: bench.py copies it into its line as `"projected"`, labelled as a projection)."""
import json, os, re, sys

BW, EFF, TLAT, TSUM = 153e9, 0.7, 15e-6, 5e-6


def t_reduce(nbytes, n):
    return 0.0 if n == 1 else 2 * (TLAT + (nbytes / n) / (EFF * BW)) + TSUM


def parse(path):
    rows = {}
    for line in open(path):
        mt = re.match(r"nranks=(\d+): root panel (\d+) doubles, root rhs (\d+) doubles; rank 0 owns (\d+) blocks, ([\d.]+) ms/step", line)
        if mt:
            rows[int(mt.group(1))] = (float(mt.group(5)), int(mt.group(2)), int(mt.group(3)), int(mt.group(4)))
    return rows


doc = {"kind": "PROJECTION, not a measurement: rank 0's local work at N = 1, 2, 4, 8 measured on ONE MI355X (tools/rank_local_timing.py) + modelled reductions "
               "(reduce-scatter + all-gather over peer copies: 153 GB/s per xGMI link and direction, efficiency 0.7, 15 us per phase, 5 us per sum); no collective has crossed a peer link",
       "unit": "ms per Newton step (1 update + 4 right-hand sides)", "workloads": {}}
for path in sys.argv[1:]:
    rows = parse(path)
    if 1 not in rows:
        print(path, ": no N = 1 line"); continue
    base = rows[1][0]
    print(f"# {path}: PROJECTED (rank-local work measured on one GPU + modelled reductions; no collective has crossed a peer link)")
    print("#  N  blocks/rank  rank-local ms  reductions ms (1 panel + 4 rhs)  projected ms/step  speed-up  efficiency")
    for n in sorted(rows):
        loc, cp, cr, nb = rows[n]
        red = 1e3 * (t_reduce(8 * cp, n) + 4 * t_reduce(8 * cr, n))
        tot = loc + red
        print(f"  {n:2d}  {nb:11d}  {loc:13.1f}  {red:32.3f}  {tot:17.1f}  {base / tot:8.2f}  {base / tot / n:10.2f}")
        name = "north_star" if "headline" in os.path.basename(path) else "c4"
        w = doc["workloads"].setdefault(name, {"source": os.path.basename(path), "n_gpus": [], "blocks_per_rank": [], "rank_local_ms": [], "reductions_ms": [], "step_ms": [], "speed_up": [], "efficiency": []})
        for k, v in (("n_gpus", n), ("blocks_per_rank", nb), ("rank_local_ms", loc), ("reductions_ms", round(red, 3)), ("step_ms", round(tot, 2)), ("speed_up", round(base / tot, 3)),
                     ("efficiency", round(base / tot / n, 3))):
            w[k].append(v)
if os.environ.get("JSON"):
    json.dump(doc, open(os.environ["JSON"], "w"), indent=1)
