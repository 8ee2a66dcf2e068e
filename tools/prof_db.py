"""Per-kernel duration summary from rocprofv3's sqlite output (rocprofv3 --kernel-trace -d DIR -o NAME)."""
import glob
import sqlite3
import sys


def summarize(db, like="%"):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d "
         f"join {ks} s on d.kernel_id=s.id where s.kernel_name like ? group by s.kernel_name order by 3 desc")
    return con.execute(q, (like,)).fetchall()


if __name__ == "__main__":
    like = sys.argv[2] if len(sys.argv) > 2 else "%mi355%"
    for db in sorted(glob.glob(sys.argv[1])):
        print(db)
        for name, n, avg, mn, mx in summarize(db, like):
            print(f"   {name[:60]:60s} n={n:4d} avg={avg/1e3:8.1f}us min={mn/1e3:8.1f} max={mx/1e3:8.1f}")
