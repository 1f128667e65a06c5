#!/bin/bash
# usage: res.sh out.s [extra flags]  -- compile ptmi.hip device-only to asm and list per-kernel resources
out=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-unused-command-line-argument "$@" --cuda-device-only -S -o $out /root/repo/webgpu-path-tracer_amd/csrc/ptmi.hip 2>&1 | grep -v warning | head
python3 - $out <<'PY'
import re,sys,subprocess
txt=open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name=subprocess.run(["c++filt",m.group(1)],stdout=subprocess.PIPE,text=True).stdout.strip().split("(")[0].replace("void ptmi::","")
    body=m.group(2)
    g=lambda k: re.search(r"\.amdhsa_"+k+r"\s+(\d+)",body).group(1)
    # count instructions
    s=txt.find(m.group(1)+":")
    e=txt.find(".Lfunc_end",s)
    code=[l for l in txt[s:e].splitlines() if re.match(r"^\s+[a-z]",l) and not l.strip().startswith(".")]
    nv=sum(1 for l in code if l.strip().startswith("v_"))
    nmov=sum(1 for l in code if re.match(r"\s+v_(mov|cndmask|readlane|writelane|pk_mov|accvgpr)",l))
    npk=sum(1 for l in code if re.match(r"\s+v_pk_",l))
    print("%-42s vgpr %3s sgpr %3s scratch %4s  instr %5d valu %5d mov/cnd %4d pk %3d"%(name[:42],g("next_free_vgpr"),g("next_free_sgpr"),g("private_segment_fixed_size"),len(code),nv,nmov,npk))
PY
