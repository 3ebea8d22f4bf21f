import sys, time
sys.path.insert(0, ".")
import numpy as np
from tests import test_gpu_dist as t
t0=time.time()
shards = [t.keys_of("uniform", 3000017, 1000), t.keys_of("uniform", 2600001, 1001)]
t1=time.time()
res = t.run_ranks(shards, 4)
t2=time.time()
t.check_sorted_ranges(shards, res)
t3=time.time()
print("gen %.3f run %.3f check %.3f" % (t1-t0, t2-t1, t3-t2))
for outs, st in res:
    print([ (rc, o.size, int(o[0]), int(o[-1])) for rc,o in outs], st)
