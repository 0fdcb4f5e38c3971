cd $GRAFT_REPO_ROOT
R=$(pwd)
mkdir -p gpurun_out/r3g
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r3g/trace -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/r3g/trace.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3g/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
i=idx[-3]
t0=int(rows[i]['Start_Timestamp'])
for r in rows[i-6:i+8]:
    print('%9.1f %8.1f q%s grid=%s wg=%s %s' % ((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Queue_Id'),r.get('Grid_Size'),r.get('Workgroup_Size'),r['Kernel_Name'][:60]))
mf=glob.glob('gpurun_out/r3g/trace/**/*memory_copy_trace.csv',recursive=True)
if mf:
    m=list(csv.DictReader(open(mf[0])))
    print(len(m),'memcopies', list(m[0].keys()) if m else '')
    for r in m:
        s=int(r['Start_Timestamp'])
        if t0-200000 < s < t0+600000: print('copy %9.1f %8.1f %s %s' % ((s-t0)/1e3,(int(r['End_Timestamp'])-s)/1e3,r.get('Direction'),r.get('Bytes','')))
PY
rm -rf gpurun_out/r3g/trace
