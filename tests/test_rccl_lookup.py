"""RCCL is looked up at run time (dlopen).  When every name fails -- here RTCORE_RCCL_LIB names a file that does not
exist -- the library must say so through its error string and go on, not crash (CPU only: nothing here touches a
GPU).  Own process: the look-up happens once per process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_missing_librccl_is_reported_not_fatal():
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
import __graft_entry__ as ge
lib = ge.load_package().load()
assert lib.rt_comm_available() == 0
ident = (C.c_ubyte * 128)()
rc = lib.rt_comm_unique_id(ident)
msg = lib.rt_last_error().decode()
assert rc != 0 and "librccl not found" in msg and "/no/such/librccl.so" in msg, (rc, msg)
comm = C.c_void_p()
assert lib.rt_comm_init_rank(ident, 0, 1, C.byref(comm)) != 0 and not comm.value
assert lib.rt_comm_destroy(None) == 0 and lib.rt_multi_gpu_destroy(None) == 0
print("ok")
''' % ROOT
    env = dict(os.environ, RTCORE_RCCL_LIB="/no/such/librccl.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
