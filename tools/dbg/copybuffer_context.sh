# who launches __amd_rocclr_copyBuffer in the train step?  kernel trace (lanes off: one stream, in order) of a short
# bench run; prints, for every copyBuffer, the kernel before and after it, aggregated
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_cb
F2G_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d $O/prof_cb -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode ${BARGS} > /dev/null 2>&1
python3 - <<'PY' > $O/copybuffer_context.txt
import csv, os, re, collections, glob
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/prof_cb/"
f=glob.glob(O+"**/p_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ev=[]
for r in rows:
    n=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); n=re.sub(r"^void ","",n)[:48]
    grid=int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])*int(r["Grid_Size_Z"])
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n,grid))
ev.sort()
half=ev[len(ev)//2:]     # the timed step (second of two)
print("kernels in the step:", len(half))
c=collections.Counter(e[2] for e in half)
for n,k in c.most_common(70): print(f"{k:6d} {n}")
ctx=collections.Counter()
for i,e in enumerate(half):
    if "copyBuffer" in e[2]:
        prev=half[i-1][2] if i else "-"; nxt=half[i+1][2] if i+1<len(half) else "-"
        ctx[(prev,nxt,e[3])]+=1
print("\ncopyBuffer contexts (previous kernel, next kernel, grid size):")
for (p,n,g),k in ctx.most_common(40): print(f"{k:5d}  after {p:48s} before {n:48s} grid {g}")
PY
rm -rf $O/prof_cb
