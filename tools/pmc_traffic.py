#!/usr/bin/env python3
"""Merge rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/traffic.json (HBM bytes per launch).

usage: pmc_traffic.py <fetch.db> <write.db> <kernel-substring> <key> [note]
Correction per /opt/skills/guides/MI355X_MICROARCH.md section HBM: both counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a coalesced streaming read (TCC_EA0_RDREQ tallied at 64 B for 128-B requests), so it is
doubled; WRITE_SIZE is exact. The factor was checked here against this kernel's own known byte count (every input
element is read exactly once): 2*FETCH_SIZE*1024 / (batch*n*n*8) = 1.000.
"""
import json
import os
import sqlite3
import sys


def avg(db, counter, like):
    cur = sqlite3.connect(db).cursor()
    row = cur.execute("select avg(value), count(*) from counters_collection where counter_name=? and kernel_name like ?",
                      (counter, f"%{like}%")).fetchone()
    return row


def main():
    fetch_db, write_db, like, key = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    f, nf = avg(fetch_db, "FETCH_SIZE", like)
    w, nw = avg(write_db, "WRITE_SIZE", like)
    rd, wr = 2.0 * f * 1024.0, w * 1024.0
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    table[key] = rd + wr
    table[key + "|detail"] = {"read_bytes": rd, "write_bytes": wr, "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
                              "dispatches": [nf, nw], "note": note}
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    print(key, "read", rd, "write", wr, "total", rd + wr)


if __name__ == "__main__":
    main()
