"""Point-M2AE gradient parity, parameter by parameter in network order, for given seeds (the decision-injected comparison of
tests/test_gpu_m2ae.py): python tools/m2ae_grad_diag.py SEED [SEED ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_m2ae import _compare_with_oracle  # noqa: E402

for seed in [int(a) for a in sys.argv[1:]] or [30]:
    rows = _compare_with_oracle(seed)
    print("seed %d: %d parameters; those with e_prod > 3e-5 (e_prod, e_cpu, zero-gradient):" % (seed, len(rows)))
    for name, e_prod, e_cpu, zero in rows:
        if e_prod > 3e-5:
            print("   %-52s %.2e  %.2e  %s" % (name, e_prod, e_cpu, "zero" if zero else ""))
