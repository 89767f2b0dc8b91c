import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyaudioanalysis_amd import _ffi, distributed as D
_ffi.lib(); _ffi.init(0)
t=time.time(); c = D.RcclGather(1, 0, lambda p: p); print("init %.1f s" % (time.time()-t), os.environ.get("NCCL_IB_DISABLE"), os.environ.get("NCCL_SOCKET_IFNAME")); c.close()
