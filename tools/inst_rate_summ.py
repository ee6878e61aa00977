import re,sys
pref=sys.argv[1:] or [""]
rows={}
for l in open("gpurun_out/inst_rate_k.txt"):
    m=re.match(r"(\S+)\s+W=(\d)\s+([\d.]+) ms.*clock ([\d.]+)",l)
    if not m: continue
    n,w,ms,clk=m.group(1),int(m.group(2)),float(m.group(3)),float(m.group(4))
    rows.setdefault(n,{})[w]=(ms*1e-3*clk*1e9/(w*128*24), ms)
for n,r in rows.items(): print(f"{n:20s}", " ".join(f"W={w}:{v[0]:6.1f} ({v[1]:.3f}ms)" for w,v in sorted(r.items())))
