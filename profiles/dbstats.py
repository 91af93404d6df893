import sqlite3, glob, sys
f=sorted(glob.glob(sys.argv[1]+'/*/*_results.db'))[-1]
db=sqlite3.connect(f)
rows=list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
print("%-110s %8s %12s %10s %6s"%("kernel","calls","total_us","avg_us","%"))
for r in rows[:24]: print("%-110s %8d %12.1f %10.2f %6.2f"%(r[0][:110],r[1],r[2],r[3],r[4]))
