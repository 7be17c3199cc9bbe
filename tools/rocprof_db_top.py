import sqlite3, sys
db=sqlite3.connect(sys.argv[1])
cur=db.cursor()
rows=cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc limit 16").fetchall()
for r in rows: print("%-80s n=%6d total=%9.1f ms avg=%9.1f us min=%8.1f max=%9.1f"%(r[0][:80].replace("pgv::(anonymous namespace)::",""),r[1],r[2],r[3],r[4],r[5]))
