#!/usr/bin/env python3
"""profiles/<tag>_pmc_counters.txt (written by tools/pmc_dump.py) -> profiles/traffic.json: HBM bytes per launch of
each matinv kernel = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
Both counters are KiB. Per /opt/skills/guides/MI355X_MICROARCH.md (section HBM) FETCH_SIZE on gfx950 reports exactly half
of the bytes of a coalesced streaming read (TCC_EA0_RDREQ tallied at 64 B for 128-B requests) and WRITE_SIZE is exact;
both factors were re-checked here on a known byte count: matinv_gj_tile_f64 reads every input element exactly once, and
2 * FETCH_SIZE * 1024 / (batch * n * n * 8) = 1.0002.
usage: pmc_to_traffic.py profiles/r01_pmc_counters.txt batch"""
import json
import os
import re
import sys

src, batch = sys.argv[1], int(sys.argv[2])
vals = {}
for ln in open(src):
    if ln.startswith("#") or ln.startswith("pass"):
        continue
    f = ln.split(None, 4)
    if len(f) < 5 or f[1] not in ("FETCH_SIZE", "WRITE_SIZE"):
        continue
    # a kernel may show up in several workloads' passes (the natural-order kernel also runs once, rejecting everything, in
    # the GENERAL workloads): keep, per kernel, the workload in which it moved the most bytes -- its full job
    wl, kern = f[0].rsplit("_", 2)[0], f[4].strip()
    # a natural-order kernel seen in a GENERAL workload's pass rejected every matrix there (it read them and wrote nothing):
    # not its job -- only the pivoting kernels are taken from those passes
    if wl.endswith("g") and "tilep" not in kern and "tileq" not in kern:
        continue
    vals.setdefault((wl, kern), {})[f[1]] = float(f[2])
path = os.path.join(os.path.dirname(src), "traffic.json")
table = {}
best = {}
for (wl, kern), v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        tot = 2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]
        if kern not in best or tot > best[kern][0]:
            best[kern] = (tot, v, wl)
for kern, (_, v, wl) in best.items():
    if "worklist" in kern:
        continue
    m = re.search(r"(\d+)", wl)  # the workload names its n: gj64, chol144, gj192g ...
    if not m:
        continue
    n, b = int(m.group(1)), batch  # (the FETCH_SIZE / WRITE_SIZE passes all run the default batch)
    rd, wr = 2.0 * v["FETCH_SIZE"] * 1024.0, v["WRITE_SIZE"] * 1024.0
    if (rd + wr) < 0.1 * b * 2 * n * n * 8:
        continue  # a launch that moved (almost) nothing: an empty work-list pass, not this kernel's job
    table[f"{kern}|n={n}"] = rd + wr
    table[f"{kern}|n={n}|detail"] = {"batch": b, "read_bytes": rd, "write_bytes": wr, "workload": wl,
                                     "algorithmic_bytes": b * 2 * n * n * 8,
                                     "traffic_over_algorithmic": (rd + wr) / (b * 2 * n * n * 8), "source": os.path.basename(src)}
json.dump(table, open(path, "w"), indent=1, sort_keys=True)
for k, v in table.items():
    if not k.endswith("detail"):
        print(k, v, round(table[k + "|detail"]["traffic_over_algorithmic"], 4))
