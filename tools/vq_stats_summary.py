import csv
rows=list(csv.DictReader(open("gpurun_out/r05_vq_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total busy ms per decode of 96 images:", round(tot/4/1e6,2))
for r in rows[:16]: print(r["Name"][:100], r["Calls"], round(float(r["TotalDurationNs"])/4/1e6,2), r["Percentage"])
