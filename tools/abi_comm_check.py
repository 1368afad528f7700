"""1-rank check of the C-ABI RCCL communicator (dftk_mi_comm_* / dftk_mi_allreduce_sum_f64)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftk_jl_amd as dftk  # noqa: E402
from dftk_jl_amd._lib import check, load  # noqa: E402

lib = load()
buf = C.create_string_buffer(128)
check(lib.dftk_mi_comm_get_unique_id(buf))
h = C.c_void_p()
check(lib.dftk_mi_comm_init_rank(buf.raw, 1, 0, 0, C.byref(h)))
t = torch.arange(10, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
check(lib.dftk_mi_allreduce_sum_f64(h, t.data_ptr(), t.numel(), None))
torch.cuda.synchronize()
assert t.sum().item() == 45.0
lib.dftk_mi_comm_destroy(h)
print("abi allreduce ok")
