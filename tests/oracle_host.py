"""TEST INFRASTRUCTURE: loads the checker build of the host driver -- tests/_build/libpangene_oraclehost.so, the same host sources linked against
the plain-C oracle (oracle/liboracle.so) instead of the HIP backend.  Only tests/, __graft_entry__.smoke() and nothing of the product use it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tests", "_build", "libpangene_oraclehost.so")


def load():
    from pangene_amd import capi
    return capi.load(LIB)
