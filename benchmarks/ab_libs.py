"""Interleaved A/B of several library builds (one process each per round): python ab_libs.py B rounds lib1 lib2 ..."""
import subprocess, sys, json, os
B=sys.argv[1]; rounds=int(sys.argv[2]); libs=sys.argv[3:]
best={}
for r in range(rounds):
    for lib in libs:
        env=dict(os.environ); 
        if lib!='default': env['VRWKV_HIP_LIB']=lib
        out=subprocess.run([sys.executable,'benchmarks/wkv7_ab.py','--B',B,'--fwd','-1','--bwd','-1','--rounds','3'],env=env,capture_output=True,text=True).stdout
        for l in out.splitlines():
            if l.startswith('{'):
                d=json.loads(l)
                for k,v in d.items():
                    if k!='B': best[(lib,k)]=min(best.get((lib,k),1e9),v)
for k,v in sorted(best.items()): print(k,v)
