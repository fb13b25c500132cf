"""Which cloud seeds take the same discontinuous decisions (max-pool argmax, ranking signs) in fp32 on the GPU as in fp64 on the CPU?
Runs tests/test_gpu_m2ae.py's oracle comparison over a range of seeds.   python tools/m2ae_seed_scan.py 200 226 240"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_m2ae as T

epoch, lo, hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for seed in range(lo, hi):
    try:
        T.test_m2ae_forward_backward_against_oracle(epoch, seed)
        print("seed %d: ok" % seed, flush=True)
    except AssertionError as ex:
        print("seed %d: %s" % (seed, str(ex)[:160].replace("\n", " ")), flush=True)
