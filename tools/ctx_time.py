"""Time the library load and rc_ctx_create (HIP runtime start, streams, tables): tools/ctx_time.py"""
import sys, time, os
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
t=time.time()
from rnacode_amd import api
api.lib()
t1=time.time()
c=api.Context(0)
t2=time.time()
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "load %.3f s, context %.3f s" % (t1-t, t2-t1))
c.close()
