import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # first: its HIP runtime is the process's
import __graft_entry__ as entry
pkg = entry.load_package(); pkg.check(pkg.lib().nb_set_device(0))
uid = pkg.comm_unique_id()
maps = open("/proc/self/maps").read()
print("unique id bytes:", len(uid))
print(sorted({l.split()[-1] for l in maps.splitlines() if "rccl" in l or "amdhip" in l}))
