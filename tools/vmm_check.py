import numpy as np, torch
from zpc_amd import lib
from zpc_amd.containers import Allocator
L = lib()
va = Allocator(virtual_reserve=1 << 30)
v = L.container__v_int_virtual(va._h, 1000)
p0 = L.get_handle_container__v_int_virtual(v)
L.resize_container__v_int_virtual(v, 50_000_000)
p1 = L.get_handle_container__v_int_virtual(v)
print("VMM pointer stable:", p0 == p1, hex(p0), "cap", L.container_capacity__v_int_virtual(v))
