#!/bin/bash
# compile bert.hip alone and report k_qa's registers / scratch / wait placement (CPU-side check of the generated ISA)
cd /root/repo/ragmeup_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -c csrc/bert.hip -o lib/obj/bert.o -save-temps=obj 2>&1 | grep -E "error|warning" | head
cd /root/repo/ragmeup_amd/lib/obj && python - <<'PY'
import re
s=open('bert-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for name in re.findall(r'\.amdhsa_kernel (\S*k_qa\w*)', s):
    if 'items' in name or 'pack' in name: continue
    m=re.search(r'\.amdhsa_kernel '+re.escape(name)+r'\n(.*?)\.end_amdhsa_kernel', s, re.S)
    print(name[:40], {key: re.search(r'\.amdhsa_'+key+r'\s+(\S+)',m.group(1)).group(1) for key in ('next_free_vgpr','accum_offset','private_segment_fixed_size')})
    start=s.index('\n'+name+':'); end=s.index('s_endpgm',start)
    open('/tmp/k_qa.s','w').write(s[start:end])
PY
echo "scratch ops: $(grep -c scratch_ /tmp/k_qa.s)"; grep -n "s_waitcnt vmcnt" /tmp/k_qa.s | tr '\n' ' '
