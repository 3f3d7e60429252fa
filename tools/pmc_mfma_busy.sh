#!/bin/bash
# usage (GPU box): tools/pmc_mfma_busy.sh <out.txt> <python script and args...>
# MFMA-busy fraction per kernel: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
OUT=$1; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ppm
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/ppm -o p -- python "$@" > /tmp/ppm.log 2>&1
python - > $OUT <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/ppm/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
rows=[]
for k,v in agg.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in v or 'GRBM_GUI_ACTIVE' not in v: continue
    n=len(v['GRBM_GUI_ACTIVE']); gui=sum(v['GRBM_GUI_ACTIVE'])/n; mf=sum(v['SQ_VALU_MFMA_BUSY_CYCLES'])/n
    wait=sum(v.get('SQ_WAIT_ANY',[0]))/max(1,len(v.get('SQ_WAIT_ANY',[0]))); wc=sum(v.get('SQ_WAVE_CYCLES',[1]))/max(1,len(v.get('SQ_WAVE_CYCLES',[1])))
    rows.append((gui*n, k.replace('lasso::','').replace('(anonymous namespace)::','')[:70], n, gui/8/2.4e3, mf/(gui/8*1024) if gui else 0, wait/wc if wc else 0))
for tot,k,n,us,busy,w in sorted(rows,reverse=True)[:16]:
    print('%-70s n=%-5d ~%8.1f us  mfma_busy %.2f  wait %.2f'%(k,n,us,busy,w))
PY
cat $OUT
