import re, collections, sys
rows=[]
for l in open(sys.argv[1] if len(sys.argv)>1 else '/root/repo/gpurun_out/bench8.err'):
    m=re.match(r'(\w+) (\d+)->(\d+) k(\d) @(\d+) n(\d+)(?: \S.*?)?\s+calls/step\s+([\d.]+)\s+ms/step\s+([\d.]+)\s+TFLOP/s\s+([\d.]+)',l)
    if m: rows.append((m.group(1),int(m.group(2)),int(m.group(3)),int(m.group(4)),int(m.group(5)),int(m.group(6)),float(m.group(7)),float(m.group(8)),float(m.group(9))))
tot=sum(r[7] for r in rows); print('rows',len(rows),'total conv ms',tot)
agg=collections.defaultdict(lambda:[0,0.0])
for f,ci,co,k,h,n,calls,ms,tf in rows:
    reg = 'ks4' if k==4 else ('H<=32' if h<=32 else ('H64-128' if h<=128 else 'H>=256'))
    a=agg[(f,reg)]; a[0]+=ms; a[1]+=ms*tf
for k,(ms,fl) in sorted(agg.items(), key=lambda kv:-kv[1][0]): print('%-8s %-8s %7.2f ms  avg %6.1f TF'%(k[0],k[1],ms,fl/ms))
