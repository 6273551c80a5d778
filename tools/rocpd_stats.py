"""Per-kernel statistics from a rocprofv3 rocpd (.db) result: python tools/rocpd_stats.py file.db [n_rows]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
q = ("select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start), min(d.end-d.start) from rocpd_kernel_dispatch d "
     "join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name order by 4 desc")
rows = list(c.execute(q))
tot = sum(r[3] for r in rows)
print("name,calls,avg_us,min_us,total_us,percent")
for n, k, a, s, m in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    n = n.split("(")[0].replace("void ", "")
    print(f"{n},{k},{a/1e3:.1f},{m/1e3:.1f},{s/1e3:.1f},{100*s/tot:.1f}")
