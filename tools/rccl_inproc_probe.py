"""Diagnostic: ark355_comm_init / ark355_prove_sharded inside ONE python process that also has torch loaded (what
tests/o3_cases.check_instance(sharded=True) does), at a tiny size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (the pytest process has it loaded)
import snark_amd
import o3_cases as O
from oracle import synthetic as S
from oracle.fields import BLS12_381 as C
lib = snark_amd.lib()
ctx = lib.ctx_create(0)
O.check_instance(lib, ctx, C, S.mulchain_csr(C.r, 300), [(5, 7)], sharded=True)
lib.ctx_destroy(ctx)
print("probe ok")
