import re,sys,subprocess
src=sys.argv[1]
out=subprocess.run(["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-std=c++17","-fPIC","-c",src,"-o","/tmp/_res.o","-I",".","-Rpass-analysis=kernel-resource-usage"]+sys.argv[2:],capture_output=True,text=True).stderr
cur=None
rows={}
for l in out.splitlines():
    m=re.search(r"remark:\s+(.*?) \[-Rpass",l)
    if not m: 
        if "error" in l: print(l)
        continue
    s=m.group(1).strip()
    if s.startswith("Function Name:"):
        cur=s.split(":",1)[1].strip(); rows[cur]={}
    elif cur and ":" in s:
        k,v=s.split(":",1); rows[cur][k.strip()]=v.strip()
for f,r in rows.items():
    name=subprocess.run(["c++filt",f],capture_output=True,text=True).stdout.strip()[:100]
    print(f"{name:90s} V={r.get('VGPRs')} A={r.get('AGPRs')} S={r.get('TotalSGPRs')} spillV={r.get('VGPRs Spill')} spillS={r.get('SGPRs Spill')} scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')}")
