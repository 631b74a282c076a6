#!/bin/bash
# usage: tools/kernel_resources.sh csrc-file.hip  -> one line per kernel: name VGPRs SGPR-spill VGPR-spill scratch LDS
f=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r"remark: +(.*?): +(\S+)", l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k=="Function Name":
        cur={"name":v}; rows.append(cur)
    elif cur is not None: cur[k]=v
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip().replace("(anonymous namespace)::","")
    print("%-90s VGPR %3s AGPR %3s sgprspill %3s vgprspill %3s scratch %4s lds %6s occ %s"%(n[:90],r.get("VGPRs"),r.get("AGPRs"),r.get("SGPRs Spill"),r.get("VGPRs Spill"),r.get("ScratchSize [bytes/lane]"),r.get("LDS Size [bytes/block]"),r.get("Occupancy [waves/SIMD]")))
'
