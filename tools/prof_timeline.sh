# kernel timeline of the graph-replayed bf16 4-step inference WITH launch lanes: wall time per
# inference, sum of kernel durations, busy union, and a weighted estimate of CU occupancy
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_tl
rocprofv3 --kernel-trace --output-format csv -d $O/prof_tl -o p -- python3 $R/bench.py ${BARGS:---workload infer4 --gemm bf16} --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-fast-mode > /dev/null 2>&1
python3 - <<'PY' > $O/prof_tl.txt
import csv, os, re
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/prof_tl/"
rows=list(csv.DictReader(open(O+"p_kernel_trace.csv")))
ev=[]
for r in rows:
    n=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); n=re.sub(r"^void ","",n)[:40]
    wg=int(r["Workgroup_Size_X"])*int(r["Workgroup_Size_Y"])*int(r["Workgroup_Size_Z"])
    grid=int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])*int(r["Grid_Size_Z"])
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n,grid//max(wg,1)))
ev.sort()
# the last inference: from the last 'reflect_pad' burst ... find big idle gaps (> 200 us) as separators
GAP=int(os.environ.get("GAPNS","150000")); gaps=[i for i in range(1,len(ev)) if ev[i][0]-max(e[1] for e in ev[max(0,i-40):i])>GAP]
seg=ev[gaps[-1]:] if gaps else ev
t0=seg[0][0]; t1=max(e[1] for e in seg)
tot=sum(e[1]-e[0] for e in seg)
# busy union
u=0; cur_s,cur_e=seg[0][0],seg[0][1]
for s,e,_,_ in seg[1:]:
    if s>cur_e: u+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
u+=cur_e-cur_s
print(f"last inference: {len(seg)} kernels, wall {(t1-t0)/1e3:.1f} us, sum of kernel durations {tot/1e3:.1f} us, busy union {u/1e3:.1f} us, mean concurrency {tot/(t1-t0):.2f}")
# block-weighted occupancy: sum(min(blocks,256)*dur)/(256*wall) -- crude (ignores blocks/CU)
occ=sum(min(b,256)*(e-s) for s,e,_,b in seg)/(256.0*(t1-t0))
print(f"block-weighted CU occupancy (1 block per CU, capped at 256): {occ:.2f}")
import collections
agg=collections.defaultdict(lambda:[0,0])
for s,e,n,b in seg: agg[n][0]+=1; agg[n][1]+=e-s
for n,(c,d) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:14]:
    print(f"  {n:42s} {c:5d} calls {d/1e3:9.1f} us  avg {d/c/1e3:7.1f}")
print("first 120 kernels of the segment: start(us) dur(us) blocks name")
for s,e,n,b in seg[:120]:
    print(f"  {(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} {b:6d} {n}")
PY
python3 - <<'PY' >> $O/prof_tl.txt
# idle analysis over the last third of the trace: gaps of the busy union
import csv, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/prof_tl/"
rows=list(csv.DictReader(open(O+"p_kernel_trace.csv")))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in rows)
n=len(ev); ev=ev[2*n//3:]
t0=ev[0][0]; t1=max(e for _,e in ev)
idle=0; cur=ev[0][1]; big=[]
for s,e in ev[1:]:
    if s>cur:
        idle+=s-cur
        if s-cur>20000: big.append((s-cur)/1e3)
    cur=max(cur,e)
print(f"last third of the trace: wall {(t1-t0)/1e6:.2f} ms, GPU idle (no kernel running) {idle/1e6:.2f} ms = {100*idle/(t1-t0):.1f} %, gaps > 20 us: {len(big)} (sum {sum(big)/1e3:.2f} ms)")
PY
