#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc results (.db files): per kernel name, mean counter value per dispatch."""
import re
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    try:
        rows = db.execute("select k.name, p.name, e.value from rocpd_pmc_event e "
                          "join rocpd_info_pmc p on e.pmc_id = p.id "
                          "join rocpd_kernel_dispatch d on e.event_id = d.event_id "
                          "join rocpd_info_kernel_symbol ks on d.kernel_id = ks.id "
                          "join (select id, kernel_name as name from rocpd_info_kernel_symbol) k on k.id = ks.id").fetchall()
    except sqlite3.OperationalError as ex:
        print(path, "query failed:", ex)
        for r in db.execute("select name, sql from sqlite_master where name like 'rocpd_pmc_event%' or name like 'rocpd_kernel_dispatch%' or name like 'rocpd_info_pmc%' or name like 'rocpd_info_kernel_symbol%'"):
            print(r)
        continue
    agg = {}
    for kname, cname, val in rows:
        kname = re.sub(r"\(.*$", "", kname)
        a = agg.setdefault((kname, cname), [0, 0.0])
        a[0] += 1; a[1] += val
    for (kname, cname), (n, tot) in sorted(agg.items()):
        if "gemm" in kname or len(sys.argv) > 5:
            print(f"{kname[-70:]:70s} {cname:32s} n={n:3d} mean={tot / n:16.1f}")
