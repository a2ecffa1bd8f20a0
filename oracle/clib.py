"""ORACLE — test infrastructure only.  ctypes loader for oracle/libpd_oracle.so
(the plain-C restatements msda_ref.c / lsa_ref.c), built by oracle/Makefile."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libpd_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("msda_ref.c", "lsa_ref.c")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpd_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB
