"""tools/show_loop.py KERNEL_SYMBOL [N] -- instruction mix of every inner loop of a kernel in a hipcc -S listing (LISTING=path,
default csrc/nbody_fast.s from `make asm`), and the text of loop N."""
import re,sys
import os
FILE = os.environ.get('LISTING', '/root/repo/cuda-nbody_amd/csrc/nbody_fast.s')
src=open(FILE).read().split("\n")
KERNEL = sys.argv[1] if len(sys.argv)>1 else "_ZN2nb12_GLOBAL__N_121integrate_bodies_fastIfLi2ELi8ELi2EEEvNS_5ShardIT_EE"
which = int(sys.argv[2]) if len(sys.argv)>2 else 1
start = next(i for i, l in enumerate(src) if l.startswith(KERNEL + ":"))
n=0
for i in range(start, len(src)):
    if src[i].startswith(".Lfunc_end"): break
    if "Inner Loop Header" in src[i]:
        n+=1
        label = src[i - 1].split(":")[0].strip()
        end = next((k for k in range(i, len(src)) if ("s_cbranch" in src[k] or "s_branch" in src[k]) and label in src[k]), None)
        if end is None:
            continue
        body=src[i-1:end+1]
        cnt=lambda p: sum(1 for l in body if l.strip().startswith(p))
        print("==== loop", n, label, src[i].strip(), "lines", end-i, "v_pk", cnt("v_pk_"), "rsq", cnt("v_rsq"), "v_mov", cnt("v_mov"), "s_load", cnt("s_load"), "ds", cnt("ds_"), "s_nop", cnt("s_nop"), "trans", cnt("v_rcp")+cnt("v_sqrt")+cnt("v_rsq"), "lane_ops", cnt("v_readlane")+cnt("v_writelane"), "s_mov", cnt("s_mov"), "v_other", sum(1 for l in body if l.strip().startswith("v_") and not l.strip().startswith(("v_pk_","v_rsq"))))
        if n==which:
            print("\n".join(body))
for i in range(start, len(src)):
    if src[i].startswith(".Lfunc_end"): 
        for l in src[i:i+40]:
            if "NumSgprs" in l or "NumVgprs" in l or "ScratchSize" in l or "Occupancy" in l: print(l)
        break
