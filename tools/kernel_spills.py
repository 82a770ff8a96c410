import os
import sys as _s
_s.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib,subprocess,re,sys
b=importlib.import_module("diff-mining_amd.build")
src=sys.argv[1]
subprocess.run([b._hipcc()]+b.FLAGS+["-S","--cuda-device-only","-o","/tmp/k.s",src]+sys.argv[2:],check=True,stderr=subprocess.DEVNULL)
text=open('/tmp/k.s').read()
for m in re.finditer(r"\.name:\s+(\S+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", text, re.S):
    print(m.group(1)[-60:], 'vgpr',m.group(3),'vspill', m.group(4),'sspill',m.group(2))
