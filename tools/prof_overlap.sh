# kernel trace of the laned stage-2 step with MRD and the mel term knocked out (generator + MPD): which big kernels
# overlap, by hardware queue  ->  gpurun_out/prof_overlap.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_ov
KO=${KO-mrd,mel} rocprofv3 --kernel-trace --output-format csv -d $O/prof_ov -o p -- python3 $R/tools/knockout.py > $O/prof_ov_run.txt 2>&1
python3 - <<'PY' > $O/prof_overlap.txt
import csv, os, re, collections, glob
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/prof_ov/"
f=glob.glob(O+"**/p_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
ev=[]
for r in rows:
    n=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); n=re.sub(r"^void ","",n)[:36]
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n,r.get("Queue_Id","?"),r.get("Stream_Id","?")))
ev.sort()
T0=ev[0][0]; T1=max(e[1] for e in ev)
lo=T0+(T1-T0)*0.80
seg=[e for e in ev if e[0]>=lo]
t0=seg[0][0]; t1=max(e[1] for e in seg)
pts=[]
for s,e,_,_,_ in seg: pts.append((s,1)); pts.append((e,-1))
pts.sort()
hist=collections.Counter(); c=0; last=t0
for t,d in pts:
    hist[c]+=t-last; last=t; c+=d
wall=t1-t0
print(f"window: {wall/1e6:.2f} ms, {len(seg)} kernels, sum of durations {sum(e[1]-e[0] for e in seg)/1e6:.2f} ms")
for k in sorted(hist): print(f"  {k} kernels in flight: {hist[k]/1e6:8.2f} ms  {100*hist[k]/wall:5.1f} %")
big=[e for e in seg if e[1]-e[0]>300000]
pts=[]
for s,e,_,_,_ in big: pts.append((s,1)); pts.append((e,-1))
pts.sort(); hist=collections.Counter(); c=0; last=t0
for t,d in pts:
    hist[c]+=t-last; last=t; c+=d
print("kernels longer than 300 us only:")
for k in sorted(hist): print(f"  {k} in flight: {hist[k]/1e6:8.2f} ms  {100*hist[k]/wall:5.1f} %")
q=collections.Counter()
for s,e,n,qq,ss in seg: q[qq]+=e-s
print("busy time by queue id:", {k: round(v/1e6,2) for k,v in q.items()})
print("kernels of the last 230 ms of the trace (start us, dur us, queue, stream, blocks, kernel):")
tl=[e for e in ev if e[0]>=T1-230e6]
import json
json.dump([[ (s-tl[0][0])/1e3, (e-s)/1e3, qq, ss, n] for s,e,n,qq,ss in tl], open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/prof_overlap_tl.json","w"))
PY
rm -rf $O/prof_ov
head -60 $O/prof_overlap.txt
